"""N2 + a8's host half in front of the real kernels: wire frames and the host's own rows -> rafting_amd/host/ingress -> the sealed multi-round
compact batch through rg_submit32 (step32_kernel on the GPU), its wide leftovers through a sparse rg_submit. tests/ingress_flow.py holds every
group's rows, replies and response frames to the oracle's row-by-row decisions (the same flow runs on the host emulation of the kernels in
the CPU suite: tests/devemu/emu_cases_waves.py)."""
import pytest

from rafting_amd import engine
from tests import ingress_flow
from tests.helpers import compare_states

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("groups,cluster,self_slot,pre_vote,rounds,max_rounds,seed", [(192, 5, 2, True, 48, 64, 91), (1000, 3, 0, True, 20, 6, 92),
                                                                                       (130, 7, 6, False, 30, 1, 93)])
def test_ingress_batches_are_decided_like_the_history_row_by_row(groups, cluster, self_slot, pre_vote, rounds, max_rounds, seed):
    st0, batches, outs, final = ingress_flow.history(groups, cluster, self_slot, pre_vote, rounds, seed, view=engine.Table(groups, cluster, self_slot, pre_vote))
    gpu = engine.Table(groups, cluster, self_slot, pre_vote)
    gpu.load_state(st0)
    nodes = [("10.1.0.%d" % i, 7000 + i) for i in range(cluster)]
    sealed = ingress_flow.drive(lambda b32: gpu.submit32(b32), lambda sp: gpu.submit(sp), groups, cluster, batches, outs, max_rounds, nodes)
    assert sealed >= (rounds + max_rounds - 1) // max_rounds
    compare_states(final, gpu.read_state(), "after the ingress")


@pytest.mark.parametrize("compact", [True, False], ids=["compact-rows", "wide-rows"])
def test_ingress_repairs_need_host_inside_multi_round_batches(compact):
    """A history whose rows DO leave the device's cached term runs: the multi-round batch comes back with RG_NEED_HOST rows and
    RG_SKIPPED_AFTER_NEED_HOST rows behind them; rw_ingress_repair (hints from the host's log — a lossless shadow oracle here — one sparse
    rg_submit per step for all broken groups, a group's later rows one by one) leaves every group with the rows, order, replies, response
    frames and final state of the history decided row by row."""
    from rafting_amd import abi, wirelib
    from tests import oracle_lib
    G, P = 320, 5
    st0, batches, outs, final = ingress_flow.history(G, P, 2, True, 60, 95, view=engine.Table(G, P, 2, True), allow_miss=True)
    gpu = engine.Table(G, P, 2, True)
    gpu.load_state(st0)
    shadow = oracle_lib.OracleTable(G, P, 2, True)
    shadow.load_state(st0)
    nodes = [("10.1.0.%d" % i, 7000 + i) for i in range(P)]
    decide = (lambda b32: gpu.submit32(b32)) if compact else (lambda b32: gpu.submit(wirelib.unpack32(b32)))
    sealed, repaired = ingress_flow.drive(decide, lambda sp: ingress_flow.decide_sparse_with_hints(gpu, shadow, sp), G, P, batches, outs, 64, nodes,
                                          shadow=shadow, raw_submit=lambda inp, outp: engine.lib().rg_submit(gpu._h, inp, outp, abi.MEM_HOST))
    assert repaired > 0, "the history never left the cached term runs: nothing was repaired"
    compare_states(final, gpu.read_state(), "after the ingress and its repairs")
    compare_states(final, shadow.read_state(), "the host's log")


def test_replication_loop_over_frames_equals_the_in_memory_loop():
    """N1 joined to N2 on the GPU: rg_replicate's plans leave the leader as request frames with filed invocation records
    (Ingress.encode_sends), the followers decide them from their ingress batches (rg_submit32) and answer with response frames, the leader's
    ingress matches every response to its invocation and decides the ack rows — three tables on one GPU, 12 ticks with client commands —
    against the same loop on the oracle with rows built directly from plans and reply rows."""
    from tests import oracle_lib
    mem = ingress_flow.replication_loop(lambda g, p, s, pv: oracle_lib.OracleTable(g, p, s, pv), 200, 12, 7, over_the_wire=False)
    net = ingress_flow.replication_loop(lambda g, p, s, pv: engine.Table(g, p, s, pv), 200, 12, 7, over_the_wire=True)
    for node in range(3):
        compare_states(mem[node], net[node], "node %d" % node)


def test_one_ingress_in_front_of_several_tables():
    """SURVEY 8(e) at the ingress: one set of connections, three tables (block partition gpu = gid / ceil(G / N); here three tables on the one
    GPU of the test box, a smaller last shard). Rows are routed by group id as they are placed; every shard's sealed batch goes to ITS table
    through rg_submit32 — rows, order, replies, response frames and final state as the history decided row by row on one table of all groups."""
    groups, shards, P, self_slot = 200, 3, 5, 1
    st0, batches, outs, final = ingress_flow.history(groups, P, self_slot, True, 30, 303, view=engine.Table(groups, P, self_slot, True))
    per = -(-groups // shards)
    tables = []
    for k in range(shards):
        first, count = k * per, min(per, groups - k * per)
        t = engine.Table(count, P, self_slot, True)
        t.load_state(ingress_flow.slice_state(st0, first, count))
        tables.append(t)
    nodes = [("10.1.0.%d" % i, 7000 + i) for i in range(P)]
    ingress_flow.drive([(lambda b32, t=t: t.submit32(b32)) for t in tables], [(lambda sp, t=t: t.submit(sp)) for t in tables], groups, P, batches, outs, 8,
                       nodes, shards=shards)
    for k, t in enumerate(tables):
        compare_states(ingress_flow.slice_state(final, k * per, t.groups), t.read_state(), "shard %d" % k)


def test_three_nodes_that_exchange_nothing_but_wire_bytes(tmp_path):
    """BASELINE configs[0] on the wire path only (tests/devemu/ingress_cluster_flow.cpp, built against libraftgpu.so): three nodes — table +
    Ingress + IngressFlusher + MemoryLogs each, three tables on the one GPU — start from nothing, time out, PreVote, RequestVote, elect a leader
    per group, take client commands, replicate (rg_replicate -> encode_sends), commit; a leader is cut off and comes back. Every batch is decided
    by rg_submit32 (step32_kernel), every message is a frame of the reference's protocol; election safety and the stability / agreement of
    committed entries are checked at every tick."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host, lib_dir = os.path.join(root, "rafting_amd", "host"), os.path.join(root, "rafting_amd")
    exe = str(tmp_path / "ingress_cluster_flow")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + host, "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "devemu", "ingress_cluster_flow.cpp")] +
                   [os.path.join(host, f) for f in ("ingress_flusher.cpp", "ingress.cpp", "wire.cpp", "kryo_body.cpp", "raft_host.cpp", "stable_store.cpp")] +
                   ["-L" + lib_dir, "-lraftgpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-pthread", "-o", exe], check=True)
    for args in (["6", "500", "compact"], ["200", "400", "compact"]):
        p = subprocess.run([exe] + args + [str(tmp_path / "cluster")], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "ingress cluster ok=1" in p.stdout, p.stdout + p.stderr[-3000:]
