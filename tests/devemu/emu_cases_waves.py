"""Cases for the WAVEFRONT mode of the host emulation (RG_EMU_WAVES=1: every lane of a workgroup is an OS thread, so
shuffles, ballots and barriers really meet): the two-wavefront step kernel with its LDS hand-over protocol, the decision
counters, the ballot-compacted expiry list. Run by tests/test_devemu_cpu.py in a subprocess; TEST INFRASTRUCTURE.
One thing this mode cannot do: a collective executed by only SOME lanes of a wavefront (the GPU's exec mask) — rows that
are skipped after a NEED_HOST never reach the ballot inside tier 1, so multi-round launches with blocked lanes stay with
the lane-serial mode and the GPU tests."""
import os

import numpy as np
import pytest

assert os.environ.get("RG_LIB", "").endswith("libraftgpu_emu.so"), "these cases are for the host emulation library only"
assert os.environ.get("RG_EMU_WAVES") == "1" and os.environ.get("RG_SPLIT") == "1"

from rafting_amd import abi, engine  # noqa: E402
from tests import fuzz, kat_scenarios, oracle_lib  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402
from tests.helpers import compare_outcomes, compare_states  # noqa: E402


@pytest.mark.parametrize("scenario", kat_scenarios.SCENARIOS, ids=lambda f: f.__name__)
def test_kat_on_the_two_wavefront_kernel(scenario):
    scenario(T.mk_gpu)                                     # includes timers_follow_reset_timer: ballot + popcount compaction


@pytest.mark.parametrize("cluster,self_slot,pre_vote,seed", [(3, 0, True, 11), (5, 2, True, 12), (7, 3, True, 16)])
def test_fuzz_lockstep_with_hints(cluster, self_slot, pre_vote, seed):
    _, _, _, hist, misses, gpu = T._lockstep(128, cluster, self_slot, pre_vote, 24, seed, allow_miss=True)
    c = gpu.counters()
    assert c[0] > 0 and c[1] > 0 and c[2] > 0


def test_fuzz_lockstep_on_an_eleven_node_cluster():
    """clusters above seven nodes: the two-wavefront wide-row kernel (their follower records fill 25 KB of its LDS at 15 nodes)"""
    _, _, _, hist, misses, gpu = T._lockstep(64, 11, 10, False, 30, 62, allow_miss=True)
    assert gpu.step_kernel() == "rg::step_split_kernel" and gpu.counters()[0] > 0
    with pytest.raises(engine.EngineError, match="wide rows"):
        gpu.submit32(abi.Batch(1, 64))


def test_multi_round_launch_with_exact_counters():
    """ONE 32-round launch of the two-wavefront kernel: outcomes, final state and the I/O wavefront's tallies"""
    G, P = 128, 5
    st0, batches, outs, _, misses, _ = T._lockstep(G, P, 1, True, 32, 21, allow_miss=False)
    assert misses == 0
    big, ref = fuzz.concat_batches(batches), fuzz.concat_outcomes(outs)
    gpu = engine.Table(G, P, 1, True)
    assert gpu.step_kernel() == "rg::step_split_kernel"
    gpu.load_state(st0)
    db = engine.DeviceBatch(gpu, big)
    gpu.submit_device(db)
    gpu.sync()
    compare_outcomes(ref, db.outcome(), "multi-round")
    orc = oracle_lib.OracleTable(G, P, 1, True)
    orc.load_state(st0)
    orc.submit(big)
    compare_states(orc.read_state(), gpu.read_state(), "multi-round final")
    c = gpu.counters()
    assert c[0] == int(np.count_nonzero(big.head["hdr"] & 0xF))
    assert c[1] == int(np.count_nonzero(ref.reply["flags"] & abi.F_REPLIED))
    assert c[2] == int(np.count_nonzero(ref.reply["flags"] & abi.F_ROLE_CHANGED))
    assert c[3] == int(np.count_nonzero(ref.reply["flags"] & abi.F_COMMIT))
    db.free()


def test_expired_timers_are_compacted_in_order():
    G = 300                                                # two workgroups of the timer kernels, a ragged last wavefront
    gpu, orc = engine.Table(G, 3, 0, True), oracle_lib.OracleTable(G, 3, 0, True)
    for t in (gpu, orc):
        t.timers_configure(900, 300, 7)
        t.timers_arm(1000)
    for now in (1900, 2300, 2800, 4000):
        eg, ng = gpu.timers_expired(now, capacity=G if now != 2300 else 17)
        eo, no = orc.timers_expired(now, capacity=G if now != 2300 else 17)
        assert ng == no and np.array_equal(eg, eo)
        assert np.array_equal(gpu.timers_read(), orc.timers_read())


def test_packed_pipeline_formats():
    """rg_submit_async_packed end to end (compact uploads decided as they arrive by step32_kernel, row-ordered packed lists written into
    page-locked memory, capacity overflow, refusal of pageable list memory)."""
    T.packed_pipeline_case(320, 3)


# ---- compact rows (rg_batch32_t) through step32_kernel: the 32-bit body, and the 64-bit body it falls back to ------------------------
@pytest.fixture(params=["narrow", "forced-wide", "narrow-out32", "forced-wide-out32"])
def compact_route(request, monkeypatch):
    """Table.submit packs every batch that fits (no hints, values < 2^31) and sends it through rg_submit32 — or, `out32`, through rg_submit32c
    (compact outcome rows, ABI 4) and back through rg_outcome32_unpack"""
    if request.param.startswith("forced-wide"):
        monkeypatch.setenv("RG_FORCE_WIDE", "1")
    T.route_through_compact(monkeypatch, out32=request.param.endswith("out32"))
    return request.param


@pytest.mark.parametrize("scenario", kat_scenarios.SCENARIOS, ids=lambda f: f.__name__)
def test_kat_on_compact_rows(scenario, monkeypatch):
    # (the 32-bit body; the 64-bit body of the same kernel answers the fuzz cases below on this emulation and every scenario on the GPU:
    # tests/test_gpu_parity.py runs all of them on both routes)
    T.route_through_compact(monkeypatch)
    scenario(T.mk_gpu)


@pytest.mark.parametrize("scenario", kat_scenarios.SCENARIOS, ids=lambda f: f.__name__)
def test_kat_on_compact_rows_with_compact_outcomes(scenario, monkeypatch):
    T.route_through_compact(monkeypatch, out32=True)
    scenario(T.mk_gpu)


@pytest.mark.parametrize("cluster,self_slot,pre_vote,seed", [(3, 0, True, 11), (5, 2, True, 12), (5, 4, False, 13), (2, 1, True, 14),
                                                             (7, 3, True, 16)])
def test_fuzz_lockstep_on_compact_rows(cluster, self_slot, pre_vote, seed, compact_route):
    _, _, _, hist, misses, gpu = T._lockstep(128, cluster, self_slot, pre_vote, 24, seed, allow_miss=True)
    assert hist[abi.OK] > 0


def test_fuzz_general_handlers_only_on_compact_rows(monkeypatch, compact_route):
    monkeypatch.setenv("RG_FAST", "0")
    _, _, _, hist, _, _ = T._lockstep(128, 5, 0, True, 24, 41, allow_miss=True)
    assert hist[abi.OK] > 0


def test_compact_multi_round_launch_and_domain_exits():
    T.compact_multi_round_case(192, 5, 24)


def test_compact_workload_replays():
    T.compact_workload_case(groups=200, rounds=12)


@pytest.mark.parametrize("route", ["wide-rows", "narrow", "forced-wide", "narrow-out32"])
def test_unfenced_timeouts_are_refused_where_the_table_requires_fences(route, monkeypatch):
    if route != "wide-rows":
        T.route_through_compact(monkeypatch, out32=route.endswith("out32"))
    if route == "forced-wide":
        monkeypatch.setenv("RG_FORCE_WIDE", "1")
    T.fenced_timeouts_case()


def test_the_once_per_tick_graph_matches_the_oracle():
    T.tick_path_case(G=128, ticks=12)


def test_the_device_resident_tick_matches_the_oracle():
    T.tick2_case(G=128, ticks=14)
    T.tick2_case(G=64, ticks=6, seed=9, device_resident=True)
    T.tick2_case(G=200, ticks=8, seed=10, nodes=2)
    T.tick2_case(G=200, ticks=8, seed=11, nodes=4)
    T.tick2_case(G=200, ticks=8, seed=12, nodes=1)          # (a ragged last workgroup)


def test_groups_at_two_to_the_forty_stay_on_the_32_bit_body():
    T.index_base_case(G=192, rounds=24)
    T.index_base_workload_case(groups=320, rounds=12)


def test_groups_at_two_to_the_forty_on_the_64_bit_body(monkeypatch):
    """the same traffic with the 64-bit body forced: it takes the relative rows off the bases itself (the fallback of a workgroup that left the domain)"""
    monkeypatch.setenv("RG_FORCE_WIDE", "1")
    monkeypatch.setattr(engine.Table, "wide_body_workgroups", lambda self, reset=False: 0)
    T.index_base_workload_case(groups=192, rounds=8)


@pytest.mark.parametrize("forced_wide", [False, True])
def test_adverse_mix_stream(forced_wide, monkeypatch):
    if forced_wide:
        monkeypatch.setenv("RG_FORCE_WIDE", "1")
    T.adverse_mix_case(320, 16)


def test_ingress_batches_are_decided_like_the_history_row_by_row():
    """N2 + a8's host half in front of the kernels: a fuzzed history as wire frames and local rows -> rafting_amd/host/ingress -> the sealed
    multi-round compact batch through rg_submit32 (step32_kernel), its wide leftovers through a sparse rg_submit; tests/ingress_flow.py
    holds every group's rows, replies and response frames to the oracle's row-by-row decisions."""
    from tests import ingress_flow
    G, P = 64, 5
    st0, batches, outs, final = ingress_flow.history(G, P, 2, True, 24, 91, view=engine.Table(G, P, 2, True))
    gpu = engine.Table(G, P, 2, True)
    gpu.load_state(st0)
    nodes = [("10.1.0.%d" % i, 7000 + i) for i in range(P)]
    ingress_flow.drive(lambda b32: gpu.submit32(b32), lambda sp: gpu.submit(sp), G, P, batches, outs, 64, nodes)
    compare_states(final, gpu.read_state(), "after the ingress")


def test_replication_loop_over_frames_equals_the_in_memory_loop():
    """N1 joined to N2 on the kernels: rg_replicate's plans as request frames (Ingress.encode_sends), the followers' ingress batches through
    step32_kernel, their response frames matched to the leader's invocation records, its ack rows — against the same loop with rows built
    directly from plans and replies on the oracle."""
    from tests import ingress_flow
    mem = ingress_flow.replication_loop(lambda g, p, s, pv: oracle_lib.OracleTable(g, p, s, pv), 40, 8, 7, over_the_wire=False)
    net = ingress_flow.replication_loop(lambda g, p, s, pv: engine.Table(g, p, s, pv), 40, 8, 7, over_the_wire=True)
    for node in range(3):
        compare_states(mem[node], net[node], "node %d" % node)
