"""BASELINE.json's configurations at FULL size on one MI355X, bit-exact against the CPU oracle
(the C oracle replays tens of millions of rows per second, so full size is affordable), plus the
size-independent properties the sharded deployment relies on. `pytest -m gpu`."""
import dataclasses

import numpy as np
import pytest

from rafting_amd import abi, engine, workload
from tests import oracle_lib
from tests.helpers import compare_outcomes, compare_states

pytestmark = pytest.mark.gpu

ROWS = {"format": "wide"}


@pytest.fixture(autouse=True, params=["wide", "compact"])
def rows_format(request):
    """every replay of this module runs on wide rows (rg_batch_t: step_split_kernel / step_kernel by size) and on compact rows
    (rg_batch32_t packed by rg_batch32_pack: step32_kernel)"""
    ROWS["format"] = request.param
    yield request.param
    ROWS["format"] = "wide"


def _replay(cfg, rounds, batches=2, first=0, count=None):
    gen = workload.ReplayGenerator(cfg, first, count)
    st0 = gen.initial_state()
    gpu = engine.Table(gen.n, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc = oracle_lib.OracleTable(gen.n, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    gpu.load_state(st0)
    orc.load_state(st0)
    hist = np.zeros(256, dtype=np.int64)
    rows = 0
    for i in range(batches):
        b = gen.next_batch(rounds)
        db = engine.DeviceBatch(gpu, b) if ROWS["format"] == "wide" else engine.DeviceBatch32(gpu, b)
        gpu.submit_device(db)
        gpu.sync()
        ref = orc.submit(b)
        compare_outcomes(ref, db.outcome(), "%s batch %d (%s rows)" % (cfg.name, i, ROWS["format"]))
        db.free()
        hist += np.bincount(ref.status, minlength=256)
        rows += b.rounds * b.count
    fin_o, fin_g = orc.read_state(), gpu.read_state()
    compare_states(fin_o, fin_g, cfg.name)
    return gen, fin_g, hist, rows


def test_config3_with_conflicting_append_entries():
    """the stream VERDICT r1 asked about: 0.5 % of the rows are AppendEntries of a new leader that overwrite the follower's last
    uncommitted entries (conflict -> truncate -> append, none of it in tier 1). Outcomes and state equal the oracle's, the model's
    independent bookkeeping equals the final state, and the rows really truncate."""
    import dataclasses
    cfg = dataclasses.replace(workload.config(3, 16384), p_conflict=0.005, name="config3 + 0.5 % conflicts")
    gen, fin, hist, rows = _replay(cfg, 48, batches=2)
    assert hist[abi.OK] == rows
    assert np.array_equal(fin.current_term, gen.term) and np.array_equal(fin.last_index, gen.last) and np.array_equal(fin.commit_index, gen.commit)


def test_kernel_choice_follows_batch_size(monkeypatch, rows_format):
    """up to one wavefront of groups per SIMD (65 536 rows on MI355X) a batch is decided by the two-wavefront kernel,
    beyond by the single-wavefront one; RG_SPLIT overrides"""
    if rows_format == "compact":
        pytest.skip("compact rows always go to rg::step32_kernel")
    monkeypatch.delenv("RG_SPLIT", raising=False)
    t = engine.Table(131072, 5, 0, True)
    assert t.step_kernel(64) == "rg::step_split_kernel" and t.step_kernel(65536) == "rg::step_split_kernel"
    assert t.step_kernel(65537) == "rg::step_kernel" and t.step_kernel() == "rg::step_kernel"
    monkeypatch.setenv("RG_SPLIT", "1")
    assert engine.Table(131072, 5, 0, True).step_kernel() == "rg::step_split_kernel"
    monkeypatch.setenv("RG_SPLIT", "0")
    assert engine.Table(64, 5, 0, True).step_kernel() == "rg::step_kernel"


@pytest.mark.parametrize("split", ["0", "1"])
def test_both_step_kernels_at_full_size(monkeypatch, split):
    """config 3 (65 536 groups) and config 5's churn mix at 131 072 groups, with the kernel choice forced either way."""
    monkeypatch.setenv("RG_SPLIT", split)
    for cfg, rounds in ((workload.config(3), 24), (workload.config(5, 131072), 12)):
        _, _, hist, rows = _replay(cfg, rounds)
        assert hist[abi.OK] == rows


@pytest.mark.parametrize("number,rounds", [(2, 64), ("2f", 64), (3, 48), (4, 6), (5, 6)])
def test_baseline_config_full_size(number, rounds):
    """configs[1..4] of BASELINE.json at their full group counts (4 096 / 65 536 / 1 M / 1 M churn)."""
    cfg = workload.config(number)
    gen, fin, hist, rows = _replay(cfg, rounds)
    assert hist[abi.OK] == rows, "the replay model emitted ill-formed / stale / asserting rows: %s" % (
        {i: int(c) for i, c in enumerate(hist) if c})
    # the workload model tracks the protocol state on its own: it must agree with the engine's final state
    assert np.array_equal(fin.current_term, gen.term)
    assert np.array_equal(fin.role_epoch.astype(np.int64), gen.epoch)
    assert np.array_equal(fin.commit_index, gen.commit)
    assert np.array_equal(fin.last_index, gen.last)
    roles = np.bincount(fin.role, minlength=3)
    if number in (3, 4):
        assert 0.1 < roles[abi.LEADER] / cfg.groups < 0.4      # the mix stays mixed
    if number == 5:
        assert roles[abi.CANDIDATE] > 0


def test_shards_reproduce_the_whole():
    """1/2/4/8-GPU runs are bit-comparable: any block shard of the group space evolves exactly like the
    same groups inside the full table (counter-based RNG keyed by global group id, no cross-group state)."""
    cfg = workload.config(3, 16384)
    _, whole, _, _ = _replay(cfg, 24)
    for first, count in ((0, 4096), (4096, 4096), (12288, 4096), (2048, 2048)):
        _, part, _, _ = _replay(cfg, 24, first=first, count=count)
        for name in ("current_term", "voted_for", "role", "commit_index", "last_index", "role_epoch"):
            assert np.array_equal(getattr(part, name), getattr(whole, name)[first:first + count]), (name, first)
        F = cfg.cluster - 1
        assert np.array_equal(part.peer_match_index, whole.peer_match_index[first * F:(first + count) * F])


def test_general_handlers_alone_give_the_same_answers(monkeypatch):
    """RG_FAST=0 routes every row through the general handlers: the branch-free fast paths are a strict
    special case of them."""
    cfg = workload.config(3, 8192)
    monkeypatch.setenv("RG_FAST", "0")
    _, slow, _, _ = _replay(cfg, 32)
    monkeypatch.delenv("RG_FAST")
    _, fast, _, _ = _replay(cfg, 32)
    for name in slow.fields():
        assert np.array_equal(getattr(slow, name), getattr(fast, name)), name


def test_rounds_are_order_preserving_across_launch_shapes():
    """Splitting the same stream into launches of different round counts cannot change any result."""
    cfg = workload.config(5, 4096)
    finals = []
    for shape in ((48,), (16, 16, 16), (1,) * 8 + (40,)):
        gen = workload.ReplayGenerator(cfg)
        gpu = engine.Table(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
        gpu.load_state(gen.initial_state())
        for r in shape:
            gpu.submit(gen.next_batch(r))
        finals.append(gpu.read_state())
    for other in finals[1:]:
        for name in other.fields():
            assert np.array_equal(getattr(finals[0], name), getattr(other, name)), name


def test_bench_two_ranks_on_one_gpu():
    """The N>1 code path of bench.py (rank-sharded streams, barrier, MAX/SUM aggregation, one JSON line from rank 0)
    exercised with two gloo ranks sharing the only GPU of the test box; the real run uses RCCL, one GPU per rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--rounds", "8", "--groups-per-gpu", "8192", "--dist-backend", "gloo", "--device", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["groups_total"] == 16384
    assert d["config"]["config_number"] == 4 and d["config"]["seed"] == "0xc0ffee03"      # N > 1 measures BASELINE config 4
    assert d["value"] > 0 and d["cpu_baseline"] is None
    # both ranks' decisions are in the aggregate: about twice one rank's share
    assert 1.8 < d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["decisions_per_step_per_gpu"] < 2.2


def test_bench_n_gt_1_defaults_to_config4_shards():
    """the command the driver runs for N > 1 (no workload flags): BASELINE config 4, 131 072-group shards of the 1 048 576-group
    table, seed 0xC0FFEE03, scalars added up through gloo — here with two ranks sharing the one GPU of the test box"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--rounds", "4", "--device", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert (c["config_number"], c["seed"], c["groups_per_gpu"], c["groups_total"]) == (4, "0xc0ffee03", 131072, 262144)
    assert c["workload"].startswith("config4: 1048576 groups") and "no RCCL" in c["parallelism"]


def test_bench_plain_command_with_two_gpus_needs_no_launcher():
    """`python bench.py --gpus 2` exactly as the driver types it for N = 1 — no torch.distributed.run: bench.py starts its own ranks (here both
    on the one GPU of the test box), rank 0 prints the line with both ranks' figures."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rounds", "8", "--groups-per-gpu", "8192",
                        "--device", "0"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["config_number"] == 4 and d["config"]["groups_total"] == 16384 and "itself" in d["launcher"]
    assert len(d["per_gpu"]) == 2 and all(g["value"] > 0 and g["device"] == 0 for g in d["per_gpu"])
    assert d["value"] <= sum(g["value"] for g in d["per_gpu"]) * 1.001
