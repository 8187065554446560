"""Multi-GPU layout of the decision path: raft groups are independent (a RaftContext shares nothing with
other contexts — context/ContextManager.java:41,112-120), so the group space is BLOCK-partitioned over the
ranks, one table + one stream per GPU, and no collective ever touches the data path.  The only cross-rank
traffic is the handful of scalars a benchmark or a monitor wants to add up; that goes through whatever
torch.distributed backend is initialised (RCCL on the GPU box, gloo in CPU tests)."""
import numpy as np


def block_partition(groups_total, world, rank):
    """gpu = gid // ceil(G / world). Returns (first_gid, count) of `rank` (count may be 0)."""
    per = -(-groups_total // world)
    first = min(rank * per, groups_total)
    return first, min(per, groups_total - first)


def owner_of(gid, groups_total, world):
    return np.asarray(gid) // (-(-groups_total // world))


def aggregate(elapsed_s, sums, device=None):
    """Whole-job view of per-rank measurements: MAX of the elapsed time, SUM of every entry of `sums`.
    Single-process when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), [float(x) for x in sums]
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor(list(sums), dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in s.tolist()]


def gather_rows(row, device=None):
    """Every rank's `row` (a few floats) on every rank, in rank order: the per-GPU figures next to the whole-job sum."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [[float(x) for x in row]]
    mine = torch.tensor(list(row), dtype=torch.float64, device=device)
    rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, mine)
    return [[float(x) for x in r.tolist()] for r in rows]
