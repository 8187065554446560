"""Committed fixtures (tests/golden/replay_digests.json, made by tools/make_golden.py FROM THE REFERENCE'S OWN CODE,
oracle/_ref): the oracle must produce them (CPU), the translated reference must still produce them wherever it can be
built (CPU), and the HIP path must produce them with neither in the loop (GPU)."""
import json
import os

import pytest

from tools import make_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "replay_digests.json")))["cases"]


# Round 5: the 16 shard launches of `bench.py --gpus N` (131 072 groups x 64 rounds each, 8.4 M rows; tools/make_golden.py SHARD_CASES). Every rank of a
# multi-GPU bench run checks its own; the suites re-derive a sample so that they stay within minutes: the oracle two of them in every run, the GPU four
# plus two on the compact kernel. All sixteen were MADE by the translated reference (tools/make_golden.py: 1.5 to 3 minutes of that fidelity build apiece);
# re-deriving them with it is behind RG_RUN_SLOW=1 (two of them — config 4's shard 5, config 5's shard 2 — were re-derived that way in this round's suite runs
# before the suite was trimmed to stay under ten minutes: profiles/r05_cpu_suite.txt), like round 4's 10.5 M-row replay.
SHARDS = {n for n, c in GOLDEN.items() if "shard" in c}
TINY = {"config4_shard0_emulation_launch", "config4_shard1_emulation_launch"}      # blocks 0 and 1 of 256 groups x 4 rounds: the emulated two-rank bench run's cases
ORACLE_SAMPLE = {"config4_shard7_bench_launch", "config5_shard3_bench_launch"} | TINY
REFERENCE_SAMPLE = set(TINY)
GPU_SAMPLE = {"config4_shard0_bench_launch", "config4_shard6_bench_launch", "config5_shard1_bench_launch", "config5_shard7_bench_launch"} | TINY


def _replay(mk, c):
    return make_golden.replay(mk, c["number"], c["groups"], c["rounds"], c.get("shard"))


def _cases(sample):
    return [pytest.param(n, marks=pytest.mark.slow) if (n in SHARDS and n not in sample) else n for n in sorted(GOLDEN)]


@pytest.mark.parametrize("name", _cases(ORACLE_SAMPLE))
def test_oracle_reproduces_committed_digests(name):
    from tests import oracle_lib
    c = GOLDEN[name]
    got = _replay(lambda g, p, s, v: oracle_lib.OracleTable(g, p, s, v), c)
    assert got == {k: c[k] for k in ("inputs", "outcomes", "state")}


SLOW_FOR_THE_REFERENCE = {"config3_bench_launch"} | (SHARDS - REFERENCE_SAMPLE)          # 4.2 M rows = 45 s of the translated reference: RG_RUN_SLOW=1 (the oracle and the GPU reproduce it in every run)


@pytest.mark.parametrize("name", [pytest.param(n, marks=pytest.mark.slow) if n in SLOW_FOR_THE_REFERENCE else n for n in sorted(GOLDEN)])
def test_reference_code_reproduces_committed_digests(name):           # (the full-size cases too: about 1 M rows each, a few seconds of the translated reference)
    from tests import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref/libref.so needs the reference checkout to be built")
    c = GOLDEN[name]
    got = _replay(lambda g, p, s, v: ref_lib.RefTable(g, p, s, v), c)
    assert got == {k: c[k] for k in ("inputs", "outcomes", "state")}


@pytest.mark.gpu
@pytest.mark.parametrize("name", _cases(GPU_SAMPLE))
def test_gpu_reproduces_committed_digests(name):
    from rafting_amd import engine
    c = GOLDEN[name]
    got = _replay(lambda g, p, s, v: engine.Table(g, p, s, v), c)
    assert got == {k: c[k] for k in ("inputs", "outcomes", "state")}


@pytest.mark.gpu
@pytest.mark.parametrize("name,compact_outcomes", [("config4_shard6_bench_launch", True), ("config5_shard1_bench_launch", False)])
def test_the_bench_launch_shape_on_the_compact_kernel_reproduces_the_reference_digest(name, compact_outcomes):
    """the shape `bench.py --gpus 8` gives every GPU — 131 072 groups x 64 rounds in ONE launch of step32_kernel's 128-VGPR variant
    (`WAVES = 4`), HBM-resident compact rows, with compact outcome rows (rg_submit32c) and with the wide columns (rg_submit32) — against the digest the
    reference's own code produced for that shard (neither oracle nor reference in the loop)."""
    from rafting_amd import engine, workload
    c = GOLDEN[name]
    cfg = workload.CONFIGS[c["number"]]
    gen = workload.ReplayGenerator(cfg, first_gid=c["shard"] * c["groups"], count=c["groups"])
    st0 = gen.initial_state()
    t = engine.Table(c["groups"], cfg.cluster, cfg.self_slot, cfg.pre_vote)
    t.load_state(st0)
    db = engine.DeviceBatch32(t, gen.next_batch(c["rounds"]), compact=compact_outcomes, wide=False)
    t.submit_device(db)
    t.sync()
    out = db.outcome(st0.role_epoch) if compact_outcomes else db.outcome()
    assert make_golden.canonical_outcome_digest(out) == c["outcomes"]
    assert make_golden.state_digest(t.read_state()) == c["state"]
    assert t.wide_body_workgroups() == 0
    db.free()
    t.close()
