"""Cases run by tests/test_devemu_cpu.py in a subprocess whose RG_LIB points at tests/devemu/libraftgpu_emu.so: the
product's device decision code and C-ABI host code, compiled for the host (lane-serial grid, see hip/hip_runtime.h),
against the oracle. TEST INFRASTRUCTURE — a way to exercise rg_device.hpp without a GPU, not a CPU path of the product.
Not collected by a plain `pytest tests` (the file name matches no test pattern); never run against the real library."""
import os

import numpy as np
import pytest

assert os.environ.get("RG_LIB", "").endswith("libraftgpu_emu.so"), "these cases are for the host emulation library only"
assert os.environ.get("RG_SPLIT") == "0", "the two-wavefront kernel needs a real barrier"

from rafting_amd import abi, engine, workload  # noqa: E402
from tests import fuzz, kat_scenarios, oracle_lib  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402
from tests.helpers import compare_outcomes, compare_states  # noqa: E402

NEEDS_WAVEFRONT = {"timers_follow_reset_timer"}          # rg_timers_expired compacts with wavefront ballots


@pytest.mark.parametrize("scenario", [s for s in kat_scenarios.SCENARIOS if s.__name__ not in NEEDS_WAVEFRONT],
                         ids=lambda f: f.__name__)
def test_kat(scenario):
    scenario(T.mk_gpu)


@pytest.mark.parametrize("cluster,self_slot,pre_vote,seed", [(3, 0, True, 11), (5, 2, True, 12), (5, 4, False, 13), (2, 1, True, 14),
                                                             (4, 0, False, 15), (7, 3, True, 16), (6, 5, True, 17)])
def test_fuzz_lockstep_with_hints(cluster, self_slot, pre_vote, seed):
    _, _, _, hist, misses, _ = T._lockstep(192, cluster, self_slot, pre_vote, 100, seed, allow_miss=True)
    assert {abi.OK, abi.DROPPED_STALE_ROLE, abi.NOT_LEADER} <= set(np.flatnonzero(hist).tolist())


@pytest.mark.parametrize("cluster,self_slot,seed", [(9, 4, 61), (15, 0, 63)])
def test_fuzz_lockstep_on_clusters_above_seven_nodes(cluster, self_slot, seed):
    T._lockstep(64, cluster, self_slot, True, 40, seed, allow_miss=True)


def test_fuzz_general_handlers_only(monkeypatch):
    monkeypatch.setenv("RG_FAST", "0")
    _, _, _, hist, _, _ = T._lockstep(192, 5, 0, True, 80, 41, allow_miss=True)
    assert hist[abi.OK] > 0


def test_unfenced_timeouts_are_refused_where_the_table_requires_fences():
    T.fenced_timeouts_case()


def test_abi4_misuse_is_reported():
    T.abi4_misuse_case()


def test_multi_round_launch_on_resident_buffers():
    G, P = 256, 5
    st0, batches, outs, _, misses, _ = T._lockstep(G, P, 1, True, 48, 21, allow_miss=False)
    assert misses == 0
    big, ref = fuzz.concat_batches(batches), fuzz.concat_outcomes(outs)
    gpu = engine.Table(G, P, 1, True)
    gpu.load_state(st0)
    db = engine.DeviceBatch(gpu, big)
    gpu.submit_device(db)
    gpu.sync()
    compare_outcomes(ref, db.outcome(), "multi-round")
    db.free()


def test_workload_replay_configs():
    import dataclasses
    for number in (3, 5, -3):
        cfg = workload.config(abs(number), 1000)         # not a multiple of the wavefront size: shadow lanes
        if number < 0:                                   # config 3 with conflicting AppendEntries: truncate + append in the general handlers
            cfg = dataclasses.replace(cfg, p_conflict=0.008, name=cfg.name + " + conflicts")
        gen = workload.ReplayGenerator(cfg)
        gpu = engine.Table(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
        orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
        st0 = gen.initial_state()
        gpu.load_state(st0)
        orc.load_state(st0)
        for _ in range(3):
            b = gen.next_batch(16)
            compare_outcomes(orc.submit(b), gpu.submit(b), cfg.name)
        compare_states(orc.read_state(), gpu.read_state(), cfg.name)


def test_replicate_kernel_on_fuzzed_state():
    T.test_replicate_matches_oracle_on_fuzzed_state()          # N1: one lane per row, no cross-lane work


def test_health_kernels_in_a_closed_loop():
    T.test_health_replay_matches_oracle()                      # N4b: health_update / health_failure / ready kernels
    T.test_health_multi_round_and_sparse_fold()


def test_sparse_rows_and_need_host_protocol():
    T.test_sparse_rows_only_touch_their_groups()
    T.test_need_host_blocks_later_rounds()


def test_device_memspace_variants_equal_the_host_ones():
    """RG_MEM_DEVICE entry points of N1 / N4 (the caller's buffers already live on the device): on the emulation device
    memory is the heap, so numpy arrays can stand in for resident buffers; results must equal the RG_MEM_HOST path."""
    import ctypes as C
    L = engine.lib()
    G, P = 320, 4
    F = P - 1
    st0 = fuzz.random_initial_state(G, P, 1, 5)
    a, b = engine.Table(G, P, 1, True), engine.Table(G, P, 1, True)
    fz = fuzz.Fuzzer(G, P, 1, 5, allow_miss=False)
    for t in (a, b):
        t.load_state(st0)
        t.timers_configure(900, 300, 3)
        t.timers_arm(1000)
    for r in range(12):
        now = 1000 + 170 * r
        batch = abi.Batch(1, G)
        fz.round(a.read_state(), batch, 0)
        oa, ob = a.submit(batch), b.submit(batch)
        compare_outcomes(oa, ob, "same input")
        nows = np.array([now], dtype=np.int64)
        a.timers_update(1, G, oa.reply, nows)
        a.health_update(batch, oa.reply, nows)
        assert L.rg_timers_update(b._h, 1, G, None, ob.reply.ctypes.data, nows.ctypes.data, abi.MEM_DEVICE) == 0
        assert L.rg_health_update(b._h, 1, G, None, batch.head.ctypes.data, ob.reply.ctypes.data, nows.ctypes.data, abi.MEM_DEVICE) == 0
        assert np.array_equal(a.timers_read(), b.timers_read())
        for x, y in zip(a.health_read(), b.health_read()):
            assert np.array_equal(x, y)
        ready_dev = np.zeros(G, dtype=np.uint8)
        assert L.rg_ready(b._h, now + 5, 1, 100, ready_dev.ctypes.data, abi.MEM_DEVICE) == 0
        b.sync()
        assert np.array_equal(a.ready(now + 5, 1, 100), ready_dev)
        head_h, send_h = a.replicate(heartbeat=r % 2)
        hb = np.full(G, r % 2, dtype=np.uint8)
        head_d, send_d = np.zeros(G, dtype=abi.SEND_HEAD_DT), np.zeros(G * F, dtype=abi.SEND_DT)
        assert L.rg_replicate(b._h, G, None, hb.ctypes.data, None, head_d.ctypes.data, send_d.ctypes.data, abi.MEM_DEVICE) == 0
        b.sync()
        assert np.array_equal(head_h, head_d) and np.array_equal(send_h, np.ascontiguousarray(send_d.reshape(F, G).T))


def test_api_misuse_and_small_entry_points():
    T.test_api_misuse_is_reported()
    assert engine.Table(8, 3).copy_bandwidth(1 << 22, 2) > 0       # the copy kernel runs (its speed means nothing here)


def test_split_kernel_is_refused_not_hung(monkeypatch):
    monkeypatch.setenv("RG_SPLIT", "1")
    t = engine.Table(64, 3, 0, True)
    with pytest.raises(engine.EngineError):
        t.submit(abi.Batch(1, 64))


def test_ingress_repairs_need_host_inside_multi_round_batches():
    """A history whose rows DO leave the device's cached term runs, through the ingress: the multi-round batch comes back with RG_NEED_HOST rows
    and RG_SKIPPED_AFTER_NEED_HOST rows behind them; rw_ingress_repair (hints from the host's log — a lossless shadow oracle here —, one sparse
    submit per step for all broken groups, the group's later rows one by one) must leave every group with the rows, order, replies, response
    frames and final state of the history decided row by row. (Lane-serial emulation: the wide step kernel decides the batches.)"""
    from rafting_amd import wirelib
    from tests import ingress_flow
    G, P = 96, 5
    st0, batches, outs, final = ingress_flow.history(G, P, 2, True, 60, 95, view=engine.Table(G, P, 2, True), allow_miss=True)
    gpu = engine.Table(G, P, 2, True)
    gpu.load_state(st0)
    shadow = oracle_lib.OracleTable(G, P, 2, True)
    shadow.load_state(st0)
    nodes = [("10.1.0.%d" % i, 7000 + i) for i in range(P)]
    sealed, repaired = ingress_flow.drive(lambda b32: gpu.submit(wirelib.unpack32(b32)), lambda sp: ingress_flow.decide_sparse_with_hints(gpu, shadow, sp),
                                          G, P, batches, outs, 64, nodes, shadow=shadow,
                                          raw_submit=lambda inp, outp: engine.lib().rg_submit(gpu._h, inp, outp, abi.MEM_HOST))
    assert repaired > 0, "the history never left the cached term runs: nothing was repaired"
    compare_states(final, gpu.read_state(), "after the ingress and its repairs")
    compare_states(final, shadow.read_state(), "the host's log")
