#!/usr/bin/env python
"""Regenerate tests/golden/replay_digests.json: SHA-256 digests of the canonical outcomes and final state of BASELINE
replays, computed BY THE REFERENCE'S OWN CODE — oracle/_ref/libref.so, the Java decision classes translated mechanically
by tools/make_ref.py (needs /root/reference).  The reference ships no vectors for this path and cannot travel to the GPU
box; these digests are how its answers travel: the oracle must reproduce them (CPU suite) and the HIP path must reproduce
them with neither oracle nor reference in the loop (GPU suite), including the metric's own configuration at full size
(config 3, 65 536 groups).  usage: python tools/make_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rafting_amd import abi, workload  # noqa: E402

CASES = [("config2", 2, 1024, 24), ("config3", 3, 2048, 32), ("config5", 5, 2048, 32),
         ("config3_full_size", 3, 65536, 16), ("config5_full_shard", 5, 131072, 8),
         # round 4: what is benchmarked — the first launch of bench.py's default stream (config 3, 65 536 groups x 64 rounds; bench.py
         # compares its own first launch with this digest: "golden": "ok"), a shard of config 4 as `bench.py --gpus 8` gives every GPU, config 2's
         # mirrored follower view
         ("config3_bench_launch", 3, 65536, 64), ("config4_shard", 4, 131072, 8), ("config2f", "2f", 4096, 24)]
# round 5: what `bench.py --gpus N` runs on EVERY rank — the first 64-round launch of shards 0..7 (131 072 groups each, streams keyed by
# global group id) of the 1 048 576-group tables of configs 4 and 5; rank r of a multi-GPU run compares its own first launch with case r and the
# line carries the verdicts. (name, config number, groups, rounds, shard index)
SHARD_CASES = [("config%d_shard%d_bench_launch" % (number, k), number, 131072, 64, k) for number in (4, 5) for k in range(8)]
# ... and the same thing small, for the CPU suite's two-rank run of bench.py on the host emulation of the kernels (tests/test_devemu_cpu.py): blocks 0 and 1 of
# 256 groups x 4 rounds — there the per-rank verdicts, and the non-zero exit on a mismatch, are exercised without a GPU
SHARD_CASES += [("config4_shard%d_emulation_launch" % k, 4, 256, 4, k) for k in range(2)]


def canonical_outcome_digest(out):
    """digest of an Outcome with every field that is not flagged valid forced to zero"""
    f = out.reply["flags"]
    rep = out.reply.copy()
    rep["resp_term"][(f & abi.F_REPLIED) == 0] = 0
    lfx = out.logfx.copy()
    lfx["commit_index"][(f & (abi.F_COMMIT | abi.F_LOG_APPEND | abi.F_LOG_TRUNC)) == 0] = 0
    lfx["log_from"][(f & (abi.F_LOG_APPEND | abi.F_LOG_TRUNC)) == 0] = 0
    per = out.persist.copy()
    per[(f & abi.F_PERSIST) == 0] = (0, 0, 0)
    h = hashlib.sha256()
    for a in (rep, lfx, per):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def state_digest(st):
    from tests.helpers import canonical_state
    st = canonical_state(st)           # fields naming objects the reference does not have at that moment are zeroed
    h = hashlib.sha256()
    for n in ("current_term", "voted_for", "role", "current_leader", "timeout_detected", "repl_prepared", "role_epoch", "votes",
              "elected_epoch", "elected_term", "commit_index", "epoch_index", "epoch_term", "first_index", "last_index",
              "peer_last_epoch", "peer_next_index", "peer_match_index", "peer_rejection", "peer_pending"):
        h.update(np.ascontiguousarray(getattr(st, n)).tobytes())
    return h.hexdigest()


def replay(make_table, number, groups, rounds, shard=None):
    if shard is None:
        cfg = workload.config(number, groups)
        gen = workload.ReplayGenerator(cfg)
    else:                                  # block `shard` of THE config's table (what bench.py --gpus N gives rank `shard`)
        cfg = workload.CONFIGS[number]
        gen = workload.ReplayGenerator(cfg, first_gid=shard * groups, count=groups)
    t = make_table(gen.n, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    t.load_state(gen.initial_state())
    b = gen.next_batch(rounds)
    inputs = hashlib.sha256(b.head.tobytes() + b.ab.tobytes() + b.cd.tobytes() + b.entry_terms[: b.entry_count].tobytes()).hexdigest()
    out = t.submit(b)
    return {"inputs": inputs, "outcomes": canonical_outcome_digest(out), "state": state_digest(t.read_state())}


def _one(case):
    from tests import ref_lib
    mk = lambda g, p, s, v: ref_lib.RefTable(g, p, s, v)   # noqa: E731
    name, number, groups, rounds = case[:4]
    shard = case[4] if len(case) > 4 else None
    d = dict(number=number, groups=groups, rounds=rounds, **replay(mk, number, groups, rounds, shard))
    if shard is not None:
        d["shard"] = shard
    return name, d


def main():
    """python tools/make_golden.py [--shards-only] [--jobs N]: every case is one single-threaded run of the translated reference (the 16 shard
    launches are 8.4 M rows each, about a minute apiece), run in N processes"""
    import multiprocessing
    path = os.path.join(ROOT, "tests", "golden", "replay_digests.json")
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else max(1, (os.cpu_count() or 2) - 1)
    doc = {"generator": "tools/make_golden.py over oracle/_ref/libref.so (the reference's Java decision classes, "
                        "mechanically translated by tools/make_ref.py)", "cases": {}}
    cases = list(SHARD_CASES)
    if "--shards-only" in sys.argv:
        doc = json.load(open(path))        # keep the other cases as committed
    else:
        cases = list(CASES) + cases
    if "--missing-only" in sys.argv:       # (a run that lost a worker: keep what is there)
        doc = json.load(open(path))
        cases = [c for c in cases if c[0] not in doc["cases"]]
    with multiprocessing.get_context("spawn").Pool(jobs, maxtasksperchild=1) as pool:
        for name, d in pool.imap_unordered(_one, cases):
            doc["cases"][name] = d
            json.dump(doc, open(path, "w"), indent=1, sort_keys=True)      # after every case: a killed worker costs one case, not the run
            print("done", name, flush=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
