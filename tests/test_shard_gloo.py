"""The N>1 layout on CPU: two gloo ranks each own a block of the group space, replay their shard of the
stream (through the oracle, there is no GPU here) and agree, bit for bit, with one process replaying the
whole table; the only cross-rank traffic is the scalar aggregation bench.py uses."""
import hashlib
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from rafting_amd import shard, workload
from tests import oracle_lib

GROUPS, ROUNDS, WORLD = 6000, 24, 2       # 6000 does not divide evenly into wavefronts or ranks


def _digest(st, names=("current_term", "voted_for", "role", "commit_index", "last_index", "role_epoch", "peer_match_index")):
    h = hashlib.sha256()
    for n in names:
        h.update(np.ascontiguousarray(getattr(st, n)).tobytes())
    return h.hexdigest()


def _replay(cfg, first, count):
    gen = workload.ReplayGenerator(cfg, first, count)
    orc = oracle_lib.OracleTable(count, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc.load_state(gen.initial_state())
    b = gen.next_batch(ROUNDS)
    orc.submit(b)
    return orc.read_state(), workload.batch_stats(b, cfg.cluster - 1)[0]


def _rank_main(rank, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    cfg = workload.config(3, GROUPS)
    first, count = shard.block_partition(GROUPS, WORLD, rank)
    st, decisions = _replay(cfg, first, count)
    elapsed, (total,) = shard.aggregate(0.5 + rank, [decisions])
    q.put((rank, first, count, _digest(st), decisions, elapsed, total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_table():
    assert shard.block_partition(10, 4, 3) == (9, 1) and shard.block_partition(10, 4, 0) == (0, 3)
    assert shard.owner_of([0, 2999, 3000, 5999], GROUPS, WORLD).tolist() == [0, 0, 1, 1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = workload.config(3, GROUPS)
    whole, whole_dec = _replay(cfg, 0, GROUPS)
    assert sum(g[4] for g in got) == whole_dec
    for rank, first, count, digest, decisions, elapsed, total in got:
        assert (first, count) == shard.block_partition(GROUPS, WORLD, rank)
        part, _ = _replay(cfg, first, count)
        assert digest == _digest(part)
        F = cfg.cluster - 1
        assert np.array_equal(part.current_term, whole.current_term[first:first + count])
        assert np.array_equal(part.commit_index, whole.commit_index[first:first + count])
        assert np.array_equal(part.peer_match_index, whole.peer_match_index[first * F:(first + count) * F])
        assert elapsed == 1.5 and total == whole_dec          # MAX over ranks, SUM over ranks
