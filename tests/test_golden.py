"""Committed fixtures (tests/golden/replay_digests.json, made by tools/make_golden.py FROM THE REFERENCE'S OWN CODE,
oracle/_ref): the oracle must produce them (CPU), the translated reference must still produce them wherever it can be
built (CPU), and the HIP path must produce them with neither in the loop (GPU)."""
import json
import os

import pytest

from tools import make_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "replay_digests.json")))["cases"]


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_reproduces_committed_digests(name):
    from tests import oracle_lib
    c = GOLDEN[name]
    got = make_golden.replay(lambda g, p, s, v: oracle_lib.OracleTable(g, p, s, v), c["number"], c["groups"], c["rounds"])
    assert got == {k: c[k] for k in ("inputs", "outcomes", "state")}


SLOW_FOR_THE_REFERENCE = {"config3_bench_launch"}          # 4.2 M rows = 45 s of the translated reference: RG_RUN_SLOW=1 (the oracle and the GPU reproduce it in every run)


@pytest.mark.parametrize("name", [pytest.param(n, marks=pytest.mark.slow) if n in SLOW_FOR_THE_REFERENCE else n for n in sorted(GOLDEN)])
def test_reference_code_reproduces_committed_digests(name):           # (the full-size cases too: about 1 M rows each, a few seconds of the translated reference)
    from tests import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref/libref.so needs the reference checkout to be built")
    c = GOLDEN[name]
    got = make_golden.replay(lambda g, p, s, v: ref_lib.RefTable(g, p, s, v), c["number"], c["groups"], c["rounds"])
    assert got == {k: c[k] for k in ("inputs", "outcomes", "state")}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_gpu_reproduces_committed_digests(name):
    from rafting_amd import engine
    c = GOLDEN[name]
    got = make_golden.replay(lambda g, p, s, v: engine.Table(g, p, s, v), c["number"], c["groups"], c["rounds"])
    assert got == {k: c[k] for k in ("inputs", "outcomes", "state")}
