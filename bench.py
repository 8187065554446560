#!/usr/bin/env python
"""bench.py — raft-group decisions/sec of the HIP decision path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                 (N > 1 without a launcher: bench.py starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One STEP = one launch of the step kernel over one batch of the synthetic RPC replay: `--rounds`
consecutive rounds x (one event per raft group), inputs and outputs resident in HBM; every step consumes
FRESH rounds of the stream, so no step sees cached or replayed state.
  N = 1   BASELINE config 3, the configuration the metric is quoted on: 65 536 groups x 5 peers on one GPU,
          ~20 % leader view / ~80 % follower view, 1 % higher-term events, elections keep the mix stationary
          (seed 0xC0FFEE02).
  N > 1   BASELINE config 4: the 1 048 576-group x 5-peer table (seed 0xC0FFEE03) block-partitioned into 131 072-group
          shards, rank r runs shard r (so 8 ranks = the whole of config 4, fewer ranks = its first N shards; a group's
          stream does not depend on the rank count).  `--config 5` selects the leader-churn stream of config 5 the same
          way.  Embarrassingly parallel: no collective on the data path; three scalars (MAX elapsed,
          SUM decisions, SUM bytes) are added up on the host through gloo.  Per-GPU work is fixed: scaling is "weak".

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      HBM bytes the launch really moved (rocprofv3 PMC passes of this very command and library build,
                profiles/traffic.json — a quotation, flagged as such) / average step-kernel launch duration, measured with
                a HIP event pair on the table's stream around the K timed launches (inter-launch gaps included), against
                the 8 TB/s HBM3E peak: `frac` <= 1. SURVEY.md §8(d)'s algorithmic bytes stay beside it as a WORK RATE
                (`work_rate_algorithmic_gbps`): the kernel keeps group state in registers across the rounds of a launch and
                reads 24-byte rows, so that model charges more bytes than any memory system moves.
  cpu_baseline  the C restatement of the reference EventLoop path (oracle/, "port") timed on this box's
                host cores over the same stream (rank 0, N=1 only) — a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=64, help="replay rounds fused into one launch")
    ap.add_argument("--groups-per-gpu", type=int, default=None, help="default: 65536 (config 3) at N=1, 131072 (a config 4/5 shard) at N>1")
    ap.add_argument("--config", default=None, choices=("2", "2f", "3", "4", "5"), help="default: 3 at N=1, 4 at N>1; 2f = config 2's mirrored follower view")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batches", type=int, default=6, help="batches of the stream the CPU baseline replays")
    ap.add_argument("--copy-bw", action="store_true", help="(default now) measure a plain HBM copy kernel in the same run")
    ap.add_argument("--no-copy-bw", action="store_true", help="skip the plain-copy bandwidth measurement")
    ap.add_argument("--copy-bytes", type=int, default=1 << 30, help="size of the plain-copy measurement (tests on the host emulation shrink it)")
    ap.add_argument("--pcie-batches", type=int, default=12, help="host-memory legs: batches per leg (the first two size the staging buffers)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) legs")
    ap.add_argument("--no-int64-pass", action="store_true", help="skip the extra timed pass with the compact kernel's 64-bit body forced (RG_FORCE_WIDE=1)")
    ap.add_argument("--dist-backend", default="gloo", help="how the three result scalars are added up: gloo (default, host sum — the "
                    "data path has no collective) or nccl (= RCCL)")
    ap.add_argument("--device", type=int, default=None, help="test only: HIP device for every rank (default LOCAL_RANK)")
    ap.add_argument("--override", default="", help="experiment only: workload overrides, e.g. leader_frac=0,p_timeout=0")
    ap.add_argument("--wide-outcomes", action="store_true", help="compact rows in, but rg_outcome_t columns out (rg_submit32: 16-byte reply + conditional 16-byte effect "
                    "and persist rows) instead of the default compact outcome rows (rg_submit32c: one 16-byte row per event + persist rows)")
    ap.add_argument("--no-adverse", action="store_true", help="skip the adverse-mix leg (value_adverse_mix: conflicts + cache misses + election churn on the same configuration)")
    ap.add_argument("--adverse-batches", type=int, default=12, help="launches of the adverse-mix leg (the first two are warm-up)")
    ap.add_argument("--index-base-batches", type=int, default=12, help="launches of the long-lived-groups leg (every log compacted at 2^40, index bases set; the first two are "
                    "warm-up; 0 skips the leg)")
    ap.add_argument("--long-launch-rounds", type=int, default=256, help="rounds per launch of the long-launch leg (the same configuration handed over in larger batches: the fixed "
                    "cost of a launch as a share of it; 0 skips the leg)")
    ap.add_argument("--long-launch-batches", type=int, default=5, help="launches of the long-launch leg (the first is warm-up)")
    ap.add_argument("--tick-batches", type=int, default=310, help="single-round ticks per way of the once-per-tick latency leg (the first ten are warm-up; 0 skips the leg; "
                    "1010 gives the >= 1000-tick distribution of profiles/r06*_tick_latency_1000.json)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU over a short re-run of the "
                    "timed leg in child processes: roofline.traffic_measured_in_this_run)")
    ap.add_argument("--pmc-timeout", type=int, default=150, help="seconds one counter pass may take before it is abandoned (the quoted values stand in)")
    ap.add_argument("--wide-rows", action="store_true", help="stage the batches as rg_batch_t (40 B + 8n per row, 64-bit fields) and decide them with the "
                    "wide-row kernels instead of the default compact rows (rg_batch32_t, 24 B per row) / rg::step32_kernel")
    return ap.parse_args()


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: one process per GPU, as torch.distributed.run would start them (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, rendezvous on 127.0.0.1) — groups are independent (context/ContextManager.java:41-47,
    112-120; one event loop per context, support/EventLoopGroup.java:77-80), so a rank is a whole engine of its own and the only thing the
    ranks share is the barrier and the scalars of the result line. Rank 0's line is the job's line; any rank failing fails the run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    entry = os.environ.get("RG_BENCH_ENTRY") or os.path.abspath(__file__)      # (RG_BENCH_ENTRY: the emulation test's wrapper script)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RG_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, entry] + sys.argv[1:], env=env))
    rc = 0
    try:
        deadline = None
        while any(p.poll() is None for p in procs):
            time.sleep(0.05)
            failed = [p for p in procs if p.poll() not in (None, 0)]
            if failed and deadline is None:
                deadline = time.time() + 20.0           # a rank died: its peers hang in the rendezvous / the barrier — give them a moment, then end them
            if deadline is not None and time.time() > deadline:
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
            rc = rc or (p.returncode or 0)
    sys.exit(0 if rc == 0 else 1)


def pmc_passes(args, argv):
    """-> {"FETCH_SIZE": mean per dispatch of the step kernel, ...} measured by rocprofv3 --pmc over child runs of this script, or {"error": ...}"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {"error": "rocprofv3 is not on PATH"}
    keep = []
    skip_next = False
    for a in argv:                                   # the workload-defining flags travel; steps / warm-up and the legs are the child's own
        if skip_next:
            skip_next = False
            continue
        if a in ("--steps", "--warmup", "--cpu-batches", "--pcie-batches", "--adverse-batches", "--index-base-batches", "--tick-batches", "--pmc-timeout", "--copy-bytes", "--long-launch-rounds",
                 "--long-launch-batches"):
            skip_next = True
            continue
        if a.split("=")[0] in ("--steps", "--warmup", "--tick-batches", "--index-base-batches", "--long-launch-rounds", "--long-launch-batches") or a in ("--no-cpu-baseline", "--no-pcie", "--no-int64-pass", "--no-adverse",
                                                                                                       "--no-copy-bw", "--copy-bw", "--no-pmc"):
            continue
        keep.append(a)
    child = [sys.executable, os.path.abspath(__file__)] + keep + ["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-pcie", "--no-int64-pass", "--no-adverse",
                                                                  "--index-base-batches", "0", "--tick-batches", "0", "--long-launch-rounds", "0", "--no-copy-bw", "--no-pmc"]
    out, t_all = {}, time.time()
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"]):
        d = tempfile.mkdtemp(prefix="rg_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, RG_BENCH_UNDER_PMC="1", TMPDIR="/tmp")
            p = subprocess.run([exe, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + child, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=args.pmc_timeout)
            if p.returncode != 0:
                return dict(out, error="rocprofv3 --pmc %s: exit %d: %s" % (" ".join(counters), p.returncode, (p.stderr or p.stdout)[-300:]))
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    if "step32_kernel" in name and "wide" not in name:
                        acc.setdefault(row.get("Counter_Name"), []).append(float(row.get("Counter_Value", 0)))
            for c in counters:
                if c not in acc:
                    return dict(out, error="rocprofv3 --pmc %s: no row for the step kernel" % c)
                out[c] = sum(acc[c]) / len(acc[c])
                out[c + "_dispatches"] = len(acc[c])
        except subprocess.TimeoutExpired:
            return dict(out, error="rocprofv3 --pmc %s: no result within %d s" % (" ".join(counters), args.pmc_timeout))
        except Exception as e:
            return dict(out, error="%s: %s" % (type(e).__name__, str(e)[:300]))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out["seconds"] = time.time() - t_all
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)                          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    import torch                                   # before libraftgpu: both then share one HIP runtime
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
    dev = local_rank if args.device is None else args.device
    if dev >= torch.cuda.device_count() > 0:
        sys.exit("bench.py --gpus %d: rank %d wants HIP device %d but this node shows %d (one rank per GPU; --device D puts every rank on D: tests only)"
                 % (args.gpus, rank, dev, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (gloo announces its connections on STDOUT — "[Gloo] Rank 0 is connected to ..." — and the contract is ONE JSON line there: while the
        # process group comes up, and through its first collective, fd 1 points at stderr)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
            sync_t = torch.zeros(1, device=red_dev)
            dist.all_reduce(sync_t)
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from rafting_amd import abi, engine, shard, workload

    def barrier():
        if world > 1:
            dist.all_reduce(sync_t)
        torch.cuda.synchronize()

    # what is measured: N = 1 -> config 3 (the metric's configuration); N > 1 -> shards of config 4 (or 5)
    number = (args.config if args.config == "2f" else int(args.config)) if args.config is not None else (3 if world == 1 else 4)
    shard_default = 131072 if number in (4, 5) else workload.CONFIGS[number].groups
    gpg = args.groups_per_gpu if args.groups_per_gpu is not None else (shard_default if world > 1 or number != 3 else 65536)
    args.config = number
    full = workload.CONFIGS[number]
    if number in (4, 5) and gpg * world <= full.groups and args.groups_per_gpu in (None, shard_default):
        cfg = full                                      # rank r = shard r of THE config-4/5 table, seeds by global group id
    else:
        cfg = workload.config(number, gpg * world)
    if args.override:
        import ast
        import dataclasses
        kv = dict(item.split("=", 1) for item in args.override.split(";"))
        cfg = dataclasses.replace(cfg, name=cfg.name + " [override %s]" % args.override,
                                  **{k: ast.literal_eval(v) for k, v in kv.items()})
    first_gid, count = rank * gpg, gpg                  # block partition (shard.block_partition when the table is gpg * world)
    assert first_gid + count <= cfg.groups
    gen = workload.ReplayGenerator(cfg, first_gid=first_gid, count=count)
    F = cfg.cluster - 1
    table = engine.Table(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote, device=dev)
    st0 = gen.initial_state()
    table.load_state(st0)

    # ---- stage (warmup + steps) fresh batches of the stream in HBM -------------------------------
    t_gen = time.time()
    nb = args.warmup + args.steps
    dbatches, stats, keep_host = [], [], []
    compact_out = not args.wide_rows and not args.wide_outcomes
    for i in range(nb):
        b = gen.next_batch(args.rounds)
        stats.append(workload.batch_stats(b, F)[:2])
        dbatches.append(engine.DeviceBatch(table, b) if args.wide_rows else engine.DeviceBatch32(table, b, compact=compact_out, wide=False))
        if rank == 0 and world == 1 and not args.no_cpu_baseline and i < args.cpu_batches:
            keep_host.append(b)
    t_gen = time.time() - t_gen

    def outcomes_of(k):
        """the wide image of the first k launches' outcome rows; compact outcome rows carry the role epoch only where it changes (rg_persist32_t),
        so the epochs are chained from the initial state through the launches (rg_outcome32_unpack)"""
        if not compact_out:
            return [db.outcome() for db in dbatches[:k]]
        outs, ep = [], st0.role_epoch
        for db in dbatches[:k]:
            o, ep = engine.unpack32(db.outcome32(), db.rounds, db.count, ep)
            outs.append(o)
        return outs

    # ---- warmup, then EXACTLY `steps` timed steps -------------------------------------------------
    # The plain-copy yardstick (what a copy kernel reaches on this device in this run) runs FIRST: staging the batches is seconds of host work during
    # which the device sits idle, and the launches right after an idle spell run slow — measured on one box, alternating runs, the yardstick after the timed
    # region against before the warm-up: kernel 0.0623 -> 0.0615 ms, ms_per_step 0.0686 - 0.0707 -> 0.0655 - 0.0658 at 10 steps, 0.0649 - 0.0654 -> 0.0630 - 0.0636 at 20
    # (profiles/r07a_copy_first_ab.jsonl; RG_BENCH_COPY_LAST=1 restores the old order). The timed region itself is what it was: K fresh steps, all of them timed.
    copy_gbps_early = None
    if os.environ.get("RG_BENCH_COPY_LAST") != "1" and not args.no_copy_bw:
        copy_gbps_early = table.copy_bandwidth(args.copy_bytes, int(os.environ.get("RG_BENCH_COPY_ITERS", "10")))
    for i in range(args.warmup):
        table.submit_device(dbatches[i])
    table.sync()
    table.counters(reset=True)
    if not args.wide_rows:
        try:
            table.wide_body_workgroups(reset=True)
        except engine.EngineError:
            pass
    barrier()
    t0 = time.perf_counter()
    table.timing_begin()                 # one HIP event pair on the table's stream around the K launches
    for i in range(args.warmup, nb):
        table.submit_device(dbatches[i])
    kernel_ms = table.timing_end()       # records the stop event and waits for it
    table.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    launches = args.steps
    counters = table.counters()
    try:
        wide_wgs = None if args.wide_rows else table.wide_body_workgroups()   # workgroups of the timed launches that left the 32-bit domain (0 on this stream)
    except engine.EngineError:                                                # (an experiment build of an older tree: RG_LIB)
        wide_wgs = None

    decisions = sum(s[0] for s in stats[args.warmup:])
    alg_bytes = sum(s[1] for s in stats[args.warmup:])
    elapsed_rank = elapsed
    elapsed, (decisions_all, alg_all) = shard.aggregate(elapsed, [decisions, alg_bytes], device=red_dev if world > 1 else None)
    per_gpu = shard.gather_rows([decisions / elapsed_rank, elapsed_rank / args.steps * 1e3, kernel_ms / max(launches, 1), float(dev)],
                                device=red_dev if world > 1 else None)

    # the measured-copy yardstick, same run, same device (taken in front of the warm-up unless RG_BENCH_COPY_LAST=1)
    if copy_gbps_early is not None:
        copy_gbps = copy_gbps_early
    elif args.no_copy_bw:
        copy_gbps = None
    else:
        copy_gbps = table.copy_bandwidth(args.copy_bytes, 10)

    # ---- per-launch spread: the SAME launches once more — initial state reloaded, the same resident batches in the same order, so every
    # launch decides what it decided in the timed region and leaves the same outcome rows — each bracketed by its own HIP event pair on the table's stream
    # (the device is idle between them: a launch's own duration, without the inter-launch gaps the region figure includes). Reported, never `value`.
    spread = None
    if rank == 0:
        try:
            table.load_state(st0)
            each = []
            for i in range(nb):
                table.timing_begin()
                table.submit_device(dbatches[i])
                each.append(table.timing_end())
            timed_each = np.sort(np.asarray(each[args.warmup:]))
            spread = {"launches": int(len(timed_each)), "min_ms": float(timed_each[0]), "median_ms": float(timed_each[len(timed_each) // 2]),
                      "max_ms": float(timed_each[-1]), "mean_ms": float(timed_each.mean()),
                      "note": "one HIP event pair per launch, device idle in between; the same launches as the timed region (state reloaded, same batches)"}
            table.sync()
        except Exception as e:      # a reporting leg must not take the bench line down with it
            spread = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            print("bench: per-launch spread leg failed: %r" % (e,), file=sys.stderr)

    # ---- counters measured IN THIS RUN: rocprofv3 --pmc over a short re-run of the timed leg in a child process, one pass per counter
    # set (FETCH_SIZE; WRITE_SIZE; the SQ instruction counters) — PMC passes only, no trace domain beside them. The child is this script with every other
    # leg switched off; its step kernel has the same name, shape and library. Skipped (the quoted values of profiles/traffic.json stand in) when rocprofv3
    # is not on PATH, under --no-pmc, inside such a child, on more than one rank, or when a pass fails or runs out of time.
    pmc = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")      # (no profiler inside a profiler)
    if rank == 0 and world == 1 and not args.no_pmc and not args.wide_rows and not os.environ.get("RG_BENCH_UNDER_PMC") and not os.environ.get("RG_ALLOW_HOST_EMULATION") \
            and not under_profiler:
        pmc = pmc_passes(args, sys.argv[1:])

    # ---- the stream's FIRST launch against the committed digest of the reference's own run of it (tests/golden/replay_digests.json, made by
    # tools/make_golden.py from oracle/_ref: no oracle in this loop), and what this row / table layout must move per launch -----------------
    # EVERY rank checks its own first launch (a shard of config 4 / 5 has a case of its own per shard index: tools/make_golden.py SHARD_CASES);
    # the verdicts travel to rank 0 with the other per-rank rows, and a mismatch anywhere fails the run.
    golden = None
    floor_bytes = None
    first_out = None
    verdict, case_name = -1.0, None                       # -1 no committed case for this launch shape, 1 ok, 0 MISMATCH
    golden_path = os.environ.get("RG_GOLDEN_FILE") or os.path.join(ROOT, "tests", "golden", "replay_digests.json")     # (RG_GOLDEN_FILE: tests hand in a tampered copy)

    def case_of(shard_index):
        """the committed case for this launch shape and that block of the table, if there is one: (name, case)"""
        if args.override:
            return None, None
        for name, c in json.load(open(golden_path))["cases"].items():
            if (str(c["number"]), c["groups"], c["rounds"]) == (str(args.config), gpg, args.rounds) and c.get("shard", 0) == shard_index \
                    and (("shard" in c) == (cfg.groups != gpg)):
                return name, c
        return None, None
    try:
        from tools import make_golden
        first_out = outcomes_of(1)[0]
        case_name, c = case_of(first_gid // gpg)
        if c is not None:
            verdict = 1.0 if make_golden.canonical_outcome_digest(first_out) == c["outcomes"] else 0.0
    except Exception as e:      # a reporting leg must not take the bench line down with it
        print("bench: golden leg failed on rank %d: %r" % (rank, e), file=sys.stderr)
    verdicts = shard.gather_rows([verdict, float(first_gid // gpg)], device=red_dev if world > 1 else None)
    if rank == 0:
        try:
            names = {0.0: "MISMATCH", 1.0: "ok", -1.0: "no committed case"}
            if world == 1:
                golden = None if case_name is None else {"case": case_name, "outcomes": names[verdict]}
            else:
                golden = [{"rank": r, "case": case_of(int(row[1]))[0] if row[0] >= 0 else None, "outcomes": names[float(row[0])]} for r, row in enumerate(verdicts)]
            # layout floor: event rows in (8 + 16 B, or 40 B + entry terms on wide rows), reply rows out (16 B), the log / commit and persist rows that
            # exist (16 B each, counted on the stream's first launch), the table once per launch (5 + 4 columns of 16 B in, 5 out, run columns
            # and the follower records of leading groups when dirty: bounded here by "all of them")
            f0 = first_out.reply["flags"]
            rows0 = len(f0)
            lfx_rows = int(np.count_nonzero((f0 & (abi.F_COMMIT | abi.F_LOG_APPEND | abi.F_LOG_TRUNC)) != 0))
            per_rows = int(np.count_nonzero((f0 & abi.F_PERSIST) != 0))
            ev_bytes = dbatches[0].bytes_in if hasattr(dbatches[0], "bytes_in") else rows0 * 40
            state_bytes = gpg * (9 * 16 + 5 * 16 + 4 * 16) + int(gpg * cfg.leader_frac) * F * 32 * 2
            # (an ESTIMATE of what this row / table layout moves, from the first launch's row counts and every table column once — not a lower bound:
            # ADVICE r4. With compact outcome rows every event has one 16-byte row and only the persist rows are conditional.)
            floor_bytes = ev_bytes + rows0 * 16 + ((per_rows if compact_out else lfx_rows + per_rows) * 16) + state_bytes
        except Exception as e:      # a reporting leg must not take the bench line down with it
            print("bench: layout-estimate leg failed: %r" % (e,), file=sys.stderr)
    if world > 1 or rank == 0:
        if any(float(row[0]) == 0.0 for row in verdicts):
            if rank == 0:
                print("bench: a rank's first launch does not reproduce the reference's digest: %r" % ([list(map(float, r)) for r in verdicts],), file=sys.stderr)
            sys.exit(3)

    # ---- the same step with caller-owned HOST buffers (PCIe both ways): reported, never `value` -----------------------------
    # Three ways over the link, each from the same table state with the same FRESH batches of the stream (B0, B1 size the staging
    # buffers, the rest are timed):  serial = rg_submit(RG_MEM_HOST), H2D -> kernel -> D2H back to back;  pipelined = rg_submit_async,
    # RG_PIPELINE_DEPTH batches in flight (upload of k+1 over kernel + download of k);  packed = rg_submit_async_packed, the same
    # pipeline with int32 event fields up and packed logfx / persist lists down. The packed replies must equal the wide ones.
    pcie = pcie_pipe = pcie_packed = pcie_gbps = packed_gbps = None
    pcie_bytes = {}
    if rank == 0 and world == 1 and not args.no_pcie:
        nb_h, n_size = args.pcie_batches, 2
        gen_p = workload.ReplayGenerator(cfg, first_gid=first_gid, count=count)      # the stream again, from its first round
        st_p = st0
        hbs = [gen_p.next_batch(args.rounds) for _ in range(nb_h)]
        timed = hbs[n_size:]
        dec_timed = sum(workload.batch_stats(hb, F)[0] for hb in timed)
        houts, owners = [], []
        for hb in hbs:
            hout = abi.Outcome(hb.rounds * hb.count)
            hb.entry_terms = np.ascontiguousarray(hb.entry_terms[:max(hb.entry_count, 1)])
            houts.append(hout)
        plain = [(np.array(hb.head), np.array(hb.ab), np.array(hb.cd), np.array(hb.entry_terms)) for hb in hbs]   # pageable originals
        for hb, hout in zip(hbs, houts):
            for obj, names in ((hb, ("head", "ab", "cd", "entry_terms")), (hout, ("reply", "logfx", "persist"))):
                for nm in names:                      # page-locked caller buffers (rg_host_alloc), as a JNI host would use
                    view, own = engine.pinned_like(table, getattr(obj, nm))
                    setattr(obj, nm, view)
                    owners.append(own)
        table.load_state(st_p)
        for k in range(n_size):
            table.submit(hbs[k], houts[k])            # first calls size the staging buffers
        t1 = time.perf_counter()
        table.submit(hbs[n_size], houts[n_size])
        dt = time.perf_counter() - t1
        pcie = workload.batch_stats(hbs[n_size], F)[0] / dt
        table.load_state(st_p)
        for k in range(n_size):
            table.submit_async(hbs[k], houts[k])      # sizes the pipeline's staging sets
        for k in range(n_size):
            table.submit_wait()
        moved = sum(sum(getattr(o, nm).nbytes for nm in names) for hb, ho in zip(timed, houts[n_size:])
                    for o, names in ((hb, ("head", "ab", "cd", "entry_terms")), (ho, ("reply", "logfx", "persist"))))
        t1 = time.perf_counter()
        for k in range(n_size, nb_h):
            table.submit_async(hbs[k], houts[k])
        for k in range(n_size, nb_h):
            table.submit_wait()
        dt = time.perf_counter() - t1
        pcie_pipe = dec_timed / dt
        pcie_gbps = moved / dt / 1e9
        pcie_bytes["wide_bytes_per_decision"] = moved / dec_timed
        wide_last = np.array(houts[-1].reply)
        for own in owners:
            own.free()
        for hb, (h_, ab_, cd_, et_) in zip(hbs, plain):
            hb.head, hb.ab, hb.cd, hb.entry_terms = h_, ab_, cd_, et_
        pbs = [engine.PackedBatch(table, hb) for hb in hbs]
        table.load_state(st_p)
        for k in range(n_size):
            table.submit_async_packed(pbs[k])
        for k in range(n_size):
            table.submit_wait()
        t1 = time.perf_counter()
        for k in range(n_size, nb_h):
            table.submit_async_packed(pbs[k])
        for k in range(n_size, nb_h):
            table.submit_wait()
        dt = time.perf_counter() - t1
        pcie_packed = dec_timed / dt
        moved_p = sum(pb.bytes_up + pb.bytes_down for pb in pbs[n_size:])
        packed_gbps = moved_p / dt / 1e9
        pcie_bytes["packed_bytes_per_decision"] = moved_p / dec_timed
        if not np.array_equal(pbs[-1].reply, wide_last):
            raise SystemExit("bench: the packed pipeline's replies differ from the wide pipeline's on the same batches")
        for pb in pbs:
            pb.free()

    # ---- CPU baseline + result check on the same stream (rank 0, N=1) ------------------------------
    cpu = None
    if keep_host:
        from tests import oracle_lib            # the checker: only this leg touches oracle/
        from tests.helpers import compare_outcomes
        cpu_dec = sum(workload.batch_stats(b, F)[0] for b in keep_host)
        best = {}
        many = max(4, min(os.cpu_count() or 4, 64))
        for threads in (1, 3, many):
            orc = oracle_lib.OracleTable(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote)
            orc.load_state(st0)
            secs, outs = 0.0, []
            for b in keep_host:
                out = abi.Outcome(b.rounds * b.count)
                out.reply[:] = 0; out.logfx[:] = 0; out.persist[:] = 0      # touch pages outside the timed call
                s, out = orc.submit_threads(b, threads, out)
                secs += s
                outs.append(out)
            best[threads] = cpu_dec / secs
            orc.close()
        for b, got, ref in zip(keep_host, outcomes_of(len(keep_host)), outs):
            compare_outcomes(ref, got, "bench stream vs oracle")
        use = max(best, key=best.get)
        cpu = {"value": best[use], "unit": "decisions/s", "cores": use, "kind": "port",
               "value_1_thread": best[1], "value_3_threads": best[3], "value_%d_threads" % many: best[many],
               "host_cores": os.cpu_count(),
               "sample": "first %d batches (%d rounds x %d groups = %d decisions) of the same stream; C restatement of "
                         "the reference EventLoop path (oracle/raft_oracle.c), in-memory log, no fsync/Netty/Kryo; "
                         "3 threads mirror EventLoopGroup(3), the widest run uses up to 64 host cores; GPU results on this sample verified bit-identical"
                         % (len(keep_host), args.rounds * len(keep_host), gpg, cpu_dec)}
        # The reference's own decision classes, translated from their Java sources (oracle/_ref/libref.so, tools/make_ref.py), on the first
        # two rounds of the stream: checked against the GPU's rows, and timed for the record only — that library is built for fidelity
        # (refcounted object graph, closures, an ordered-map RocksDB stand-in), not speed, and says nothing about a JIT-compiled JVM.
        try:
            from tests import ref_lib
            if ref_lib.available():
                import copy
                two = copy.copy(keep_host[0])
                n2 = 2 * two.count
                two.rounds = 2
                two.head, two.ab, two.cd = two.head[:n2], two.ab[:n2], two.cd[:n2]
                rt = ref_lib.RefTable(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote)
                rt.load_state(st0)
                t1 = time.perf_counter()
                r_out = rt.submit(two)
                dt = time.perf_counter() - t1
                g_out = outcomes_of(1)[0]
                for col in ("reply", "logfx", "persist"):
                    setattr(g_out, col, getattr(g_out, col)[:n2])
                compare_outcomes(r_out, g_out, "bench stream vs the translated reference")
                cpu["reference_translated"] = {"value_1_thread": workload.batch_stats(two, F)[0] / dt, "sample": "first 2 rounds of the stream (%d rows), "
                                               "GPU rows verified bit-identical to it; fidelity build, not a performance baseline" % n2}
                rt.close()
        except Exception as e:      # the checker leg must not take the bench line down with it: report, do not raise
            cpu["reference_translated"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            print("bench: translated-reference leg failed: %r" % (e,), file=sys.stderr)

    # ---- the same timed steps with the compact kernel's 64-bit body forced (RG_FORCE_WIDE=1): what the step costs when a value of the
    # workgroup has left the 32-bit tier's domain (>= 2^30). Same batches, a second table from the same initial state; its decision counters must
    # equal the first pass's (the outcome rows are bit-identical: tests/test_gpu_parity.py::test_compact_multi_round_launch_and_domain_exits).
    int64_pass = None
    if rank == 0 and world == 1 and not args.wide_rows and not args.no_int64_pass:
        os.environ["RG_FORCE_WIDE"] = "1"
        try:
            t2 = engine.Table(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote, device=dev)
        finally:
            os.environ.pop("RG_FORCE_WIDE", None)
        t2.load_state(st0)
        if not args.no_copy_bw:
            t2.copy_bandwidth(args.copy_bytes, 10)          # (the CPU baseline sits in front of this pass: the device out of its idle state first, as in front of the main leg)
        for i in range(args.warmup):
            t2.submit_device(dbatches[i])
        t2.sync()
        t2.counters(reset=True)
        t2.timing_begin()
        for i in range(args.warmup, nb):
            t2.submit_device(dbatches[i])
        ms2 = t2.timing_end()
        t2.sync()
        c2 = t2.counters()
        int64_pass = {"ms": ms2 / max(args.steps, 1), "value": decisions / (ms2 * 1e-3) if ms2 > 0 else None,
                      "counters_equal_first_pass": [int(x) for x in c2] == [int(x) for x in counters]}
        t2.close()

    # ---- the same configuration under an ADVERSE mix, HBM-resident, same launch shape, a table of its own: 0.5 % of the rows are
    # conflicting AppendEntries of a new leader (conflict -> truncate -> append: general handlers), ~5 % election traffic (timeouts, vote requests and the
    # reply fan-in they start), and 1 % of the follower rows reach below the cached term runs (RG_NEED_HOST; the host parks the group for the rest of
    # the launch, which is what RG_SKIPPED_AFTER_NEED_HOST would do to its rows). `value_adverse_mix` counts APPLIED decisions only.
    adverse = None
    if rank == 0 and world == 1 and not args.no_adverse and not args.wide_rows and not args.override and args.adverse_batches > 2:
        try:
            import dataclasses
            acfg = dataclasses.replace(cfg, name=cfg.name + " [adverse mix]", p_conflict=0.005, p_higher_term=max(cfg.p_higher_term, 0.01), p_miss=0.01,
                                       p_timeout=max(cfg.p_timeout, 0.03), p_vote_req=max(cfg.p_vote_req, 0.008))
            agen = workload.ReplayGenerator(acfg, first_gid=first_gid, count=count)
            t3 = engine.Table(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote, device=dev)
            t3.load_state(agen.initial_state())
            abatches, adec, aelect, arows, amiss = [], 0, 0, 0, 0
            for i in range(args.adverse_batches):
                miss0 = agen.miss_rows
                b = agen.next_batch(args.rounds)
                if i >= 2:
                    kinds = b.head["hdr"] & 0xF
                    adec += workload.batch_stats(b, F)[0] - (agen.miss_rows - miss0)
                    amiss += agen.miss_rows - miss0
                    aelect += int(np.count_nonzero((kinds >= abi.EV_RV_REQ) & (kinds <= abi.EV_TIMEOUT)))
                    arows += int(np.count_nonzero(kinds))
                abatches.append(engine.DeviceBatch32(t3, b, compact=compact_out, wide=False))
            if not args.no_copy_bw:
                t3.copy_bandwidth(args.copy_bytes, 10)      # (the device out of the idle state the staging left it in, as in front of the main leg)
            for i in range(2):
                t3.submit_device(abatches[i])
            t3.sync()
            t3.counters(reset=True)
            t3.timing_begin()
            for i in range(2, args.adverse_batches):
                t3.submit_device(abatches[i])
            ams = t3.timing_end()
            t3.sync()
            ac = t3.counters()
            n_adv = args.adverse_batches - 2
            adverse = {"value": adec / (ams * 1e-3), "avg_kernel_ms": ams / n_adv, "launches": n_adv,
                       "mix": {"p_conflict": acfg.p_conflict, "p_miss": acfg.p_miss, "p_timeout": acfg.p_timeout, "p_vote_req": acfg.p_vote_req,
                               "election_rows_share": aelect / max(arows, 1), "rows_parked_share": 1.0 - arows / float(n_adv * args.rounds * gpg)},
                       "counters": dict(zip(["rows", "replied", "role_conversions", "commit_advances", "asserts", "need_host", "dropped_stale", "log_appends"], ac)),
                       "need_host_rows_expected": amiss, "need_host_as_expected": int(ac[5]) == amiss,
                       "note": "applied decisions only (RG_NEED_HOST rows and the rounds a group is parked for after one are not counted); same table size, rounds per "
                               "launch and outcome format as `value`"}
            for db in abatches:
                db.free()
            t3.close()
        except Exception as e:      # a reporting leg must not take the bench line down with it
            adverse = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            print("bench: adverse-mix leg failed: %r" % (e,), file=sys.stderr)

    # ---- the same configuration for LONG-LIVED groups: every log compacted at 2^40 (epoch.index = 2^40, all live indices above it), the
    # table's index bases at 2^40 - 1, rows packed relative to them (rg_batch32_pack_rel). The 32-bit body must carry it at the headline's speed: no
    # workgroup on the 64-bit body. The first launch is checked against the oracle, which works on the absolute values.
    long_lived = None
    if rank == 0 and world == 1 and args.index_base_batches > 2 and not args.wide_rows and not args.override:
        try:
            import dataclasses
            OFF = 1 << 40
            lcfg = dataclasses.replace(cfg, name=cfg.name + " [index base 2^40]", index_base=OFF)
            lgen = workload.ReplayGenerator(lcfg, first_gid=first_gid, count=count)
            lbase = np.full(gpg, OFF - 1, dtype=np.int64)
            t4 = engine.Table(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote, device=dev)
            t4.set_index_base(lbase)
            lst0 = lgen.initial_state()
            t4.load_state(lst0)
            lbatches, ldec, first_host = [], 0, None
            for i in range(args.index_base_batches):
                b = lgen.next_batch(args.rounds)
                if i == 0:
                    first_host = b
                if i >= 2:
                    ldec += workload.batch_stats(b, F)[0]
                lbatches.append(engine.DeviceBatch32(t4, engine.pack32(b, index_base=lbase), compact=compact_out, wide=False))
            if not args.no_copy_bw:
                t4.copy_bandwidth(args.copy_bytes, 10)      # (as in front of the main leg)
            for i in range(2):
                t4.submit_device(lbatches[i])
            t4.sync()
            t4.wide_body_workgroups(reset=True)
            t4.timing_begin()
            for i in range(2, args.index_base_batches):
                t4.submit_device(lbatches[i])
            lms = t4.timing_end()
            t4.sync()
            # (the oracle's replay of the first launch comes AFTER the timed launches: seconds of host work between the warm-up and the timed region leave the
            #  device idle, and the first launches after an idle spell run slow — profiles/r06w_long_lived_20_launches.jsonl)
            checked = None
            if not args.no_cpu_baseline:
                from tests import oracle_lib
                from tests.helpers import compare_outcomes
                orc = oracle_lib.OracleTable(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote)
                orc.load_state(lst0)
                _, ref0 = orc.submit_threads(first_host, max(4, min(os.cpu_count() or 4, 64)), abi.Outcome(first_host.rounds * first_host.count))
                got0 = engine.unpack32(lbatches[0].outcome32(), lbatches[0].rounds, lbatches[0].count, lst0.role_epoch, index_base=lbase)[0] if compact_out \
                    else lbatches[0].outcome()
                compare_outcomes(ref0, got0, "long-lived groups, first launch vs oracle")
                orc.close()
                checked = "first launch (%d rows) bit-identical to the oracle on the absolute stream" % (first_host.rounds * first_host.count)
            n_l = args.index_base_batches - 2
            long_lived = {"value": ldec / (lms * 1e-3), "avg_kernel_ms": lms / n_l, "launches": n_l, "index_base": OFF - 1, "epoch_index": OFF,
                          "int64_body_workgroups": t4.wide_body_workgroups(), "checked": checked}
            for db in lbatches:
                db.free()
            t4.close()
        except Exception as e:      # a reporting leg must not take the bench line down with it
            long_lived = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            print("bench: long-lived-groups leg failed: %r" % (e,), file=sys.stderr)

    # ---- the same configuration handed over in LARGER batches: `--long-launch-rounds` rounds per launch instead of `--rounds`. A launch costs 7.5 us whatever its length and ends
    # with its slowest workgroup (DESIGN.md sections 6, 9); a host that can hand over more rounds at once pays both less often. Reported beside `value`, never as it.
    long_launch = None
    if rank == 0 and world == 1 and args.long_launch_rounds > 0 and args.long_launch_batches > 1 and not args.wide_rows and not args.override:
        try:
            ggen = workload.ReplayGenerator(cfg, first_gid=first_gid, count=count)
            t5 = engine.Table(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote, device=dev)
            t5.load_state(ggen.initial_state())
            gbatches, gdec = [], 0
            for i in range(args.long_launch_batches):
                b = ggen.next_batch(args.long_launch_rounds)
                if i >= 1:
                    gdec += workload.batch_stats(b, F)[0]
                gbatches.append(engine.DeviceBatch32(t5, b, compact=compact_out, wide=False))
                del b
            if not args.no_copy_bw:
                t5.copy_bandwidth(args.copy_bytes, 10)      # (as in front of the main leg)
            t5.submit_device(gbatches[0])
            t5.sync()
            t5.wide_body_workgroups(reset=True)
            t5.timing_begin()
            for i in range(1, args.long_launch_batches):
                t5.submit_device(gbatches[i])
            gms = t5.timing_end()
            t5.sync()
            n_g = args.long_launch_batches - 1
            long_launch = {"value": gdec / (gms * 1e-3), "rounds_per_launch": args.long_launch_rounds, "avg_kernel_ms": gms / n_g, "launches": n_g,
                           "ms_per_%d_rounds" % args.rounds: gms / n_g * args.rounds / args.long_launch_rounds, "int64_body_workgroups": t5.wide_body_workgroups(),
                           "note": "the same stream and table size in launches of %d rounds instead of %d, by HIP events; not `value`" % (args.long_launch_rounds, args.rounds)}
            for db in gbatches:
                db.free()
            t5.close()
        except Exception as e:      # a reporting leg must not take the bench line down with it
            long_launch = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            print("bench: long-launch leg failed: %r" % (e,), file=sys.stderr)

    # ---- the once-per-tick path: ONE round per submission, from page-locked host buffers to page-locked host buffers, submit -> wait.
    # Two ways: rg_submit_async_packed (nine runtime calls per tick) and rg_tick_launch (the same chain recorded once as a HIP graph, one call per tick);
    # plus what one single-round launch costs on the device when the rows already lie in HBM. A latency figure, never `value`.
    tick = None
    if rank == 0 and world == 1 and args.tick_batches > 10 and not args.wide_rows and not args.override:
        try:
            import gc

            def lat(us):
                us = np.sort(np.asarray(us))
                q = lambda f: float(us[min(len(us) - 1, int(len(us) * f))])   # noqa: E731
                return {"p50_us": q(0.5), "p99_us": q(0.99), "p999_us": q(0.999), "max_us": float(us[-1]), "mean_us": float(us.mean()), "ticks": len(us)}
            res = {}
            tg = workload.ReplayGenerator(cfg, first_gid=first_gid, count=count)
            tick_st0 = tg.initial_state()
            ticks = [tg.next_batch(1) for _ in range(args.tick_batches)]          # ONE list of fresh single-round ticks; every way replays it from the initial state
            packed = [engine.pack32(b) for b in ticks]
            cap = max(b.entry_count for b in ticks) + 64
            for way in ("rg_submit_async_packed", "rg_tick_launch"):
                tt = engine.Table(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote, device=dev)
                tt.load_state(tick_st0)
                pb = engine.PackedBatch(tt, ticks[0], entry_cap=cap)
                tk = engine.Tick(tt, pb) if way == "rg_tick_launch" else None
                # (round 5's driver run showed ONE 20 ms tick among 90 on the nine-call path, none on the graph: the only thing that path does and the graph
                #  does not is grow a staging buffer when a tick carries more entry terms than any before it — hipFree + hipMalloc on the tick's path. The
                #  array's CAPACITY travels now, as the graph has always had it; and the collector is kept out of the timed loop.)
                pb.c_in.entry_count = cap
                us = []
                gc.collect()
                gc.disable()
                try:
                    for i, b32 in enumerate(packed):
                        pb.head[:], pb.abcd[:] = b32.head, b32.abcd
                        pb.entry_terms[:b32.entry_count] = b32.entry_terms[:b32.entry_count]
                        t1 = time.perf_counter()
                        if tk is None:
                            tt.submit_async_packed(pb)
                            tt.submit_wait()
                        else:
                            tk.launch()
                            tk.wait()
                        dt = time.perf_counter() - t1
                        if i >= 10:
                            us.append(dt * 1e6)
                finally:
                    gc.enable()
                res[way] = lat(us)
                if tk is not None:
                    tk.close()
                    # the same single-round launches with the rows resident in HBM: what the device itself spends per tick
                    dbs = [engine.DeviceBatch32(tt, b, compact=compact_out, wide=False) for b in ticks[:20]]
                    tt.submit_device(dbs[0]); tt.submit_device(dbs[1]); tt.sync()
                    tt.timing_begin()
                    for db in dbs[2:]:
                        tt.submit_device(db)
                    res["device_us_per_single_round_launch"] = tt.timing_end() * 1e3 / (len(dbs) - 2)
                    for db in dbs:
                        db.free()
                pb.free()
                tt.close()
            # the DEVICE-RESIDENT tick (ABI 5, rg_tick2_*): decisions -> timers -> health -> fired tickets -> send table -> readiness as ONE graph on compact
            # outcome rows, every large column in HBM; what the device spends per tick, by a HIP event pair around the replays
            # The default recording is ONE kernel node (tick_kernel); RG_TICK_NODES=2 / 4 record step + fused tail / the step-by-step form: all three are timed.
            by_nodes = {}
            saved_nodes = os.environ.get("RG_TICK_NODES")
            for nodes in (4, 2, 1):
                tt = engine.Table(gpg, cfg.cluster, cfg.self_slot, cfg.pre_vote, device=dev)
                tt.load_state(tick_st0)
                tt.timers_configure(900, 300, 1)
                tt.timers_arm(0)
                os.environ["RG_TICK_NODES"] = str(nodes)
                try:
                    t2k = engine.Tick2(tt, 1, entry_cap=cap, expired_cap=gpg, critical_point=1, cool_down_ms=60, device_resident=True)
                finally:
                    if saved_nodes is None:
                        del os.environ["RG_TICK_NODES"]
                    else:
                        os.environ["RG_TICK_NODES"] = saved_nodes
                # Two readings per recording. `queued`: ticks launched back to back without a host wait in between (they are ordered on the table's stream), one
                # event pair around the run — what the DEVICE spends per tick, the way device_us_per_single_round_launch is taken. `idle`: one event pair around one
                # hipGraphLaunch on an idle stream — what a host that waits for every tick sees: it contains the host's own submission path (the start event
                # completes at once, the graph's packets arrive tens of microseconds later).
                n2 = min(len(packed), 60)
                t_idle = 0.0
                for i in range(n2):
                    t2k.refill(packed[i], [300 * (i + 1)])
                    if i == 10:
                        tt.sync()
                    if i < 10:
                        t2k.launch(); t2k.wait()
                    else:
                        tt.timing_begin()
                        t2k.launch()
                        t_idle += tt.timing_end()
                n_q = 4 if os.environ.get("RG_ALLOW_HOST_EMULATION") else 40      # (the CPU suite runs this leg on the host emulation of the kernels)
                t2k.refill(packed[0], [300 * (n2 + 1)])
                tt.sync()
                tt.timing_begin()
                for _ in range(n_q):
                    t2k.launch()
                t_q = tt.timing_end()
                t2k.wait()
                by_nodes[str(nodes)] = {"device_us": t_q * 1e3 / n_q, "idle_stream_us": t_idle * 1e3 / max(n2 - 10, 1)}
                t2k.close()
                tt.close()
            res["device_us_per_resident_tick"] = by_nodes["1"]["device_us"]
            res["resident_tick_us_on_an_idle_stream"] = by_nodes["1"]["idle_stream_us"]
            res["device_us_per_resident_tick_by_graph_nodes"] = by_nodes
            res["resident_tick_steps"] = ("decisions (step32c), timers_update32 + health_update32 + fired tickets, replicate, ready — recorded as ONE kernel node (tick_kernel: the workgroup "
                                          "that decided 64 groups does the rest for them); by_graph_nodes: 2 = step + fused tail, 4 = step, fold, replicate, ready")
            tick = dict(res, groups=gpg, rounds_per_tick=1, bytes_up_per_tick=24 * gpg, note="submit -> wait of ONE round over PCIe, page-locked buffers both ways; "
                        "python call overhead (ctypes, ~2 us per call) included in both ways; gc disabled inside the timed loops")
        except Exception as e:      # a reporting leg must not take the bench line down with it
            tick = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            print("bench: once-per-tick leg failed: %r" % (e,), file=sys.stderr)

    if rank == 0:
        # HBM bytes per launch from the PMC passes of the SAME command (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate
        # passes, gfx950 x2 fetch correction applied; tools/prof.sh writes profiles/traffic.json) — quoted only when the entry
        # describes this very workload, kernel and library build
        traffic = traffic_src = None
        kernel_name = "%s<%d,false>" % (table.step_kernel() if args.wide_rows else "rg::step32_kernel", F)
        out_fmt = "rg_outcome32_t" if compact_out else "rg_outcome_t"
        try:
            for tr in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["entries"]:
                if (tr["config"], tr["groups_per_gpu"], tr["rounds"], tr["kernel"]) == (args.config, gpg, args.rounds, kernel_name) \
                        and not args.override and tr.get("lib_sha16") == engine.library_sha16() and tr.get("outcome_format", "rg_outcome_t") == out_fmt:
                    traffic, traffic_src = tr["traffic_bytes_per_launch"] / 1e9, tr["source"]
        except (OSError, KeyError, ValueError):
            pass
        traffic_here = False
        if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # measured in THIS run (KB per dispatch; gfx950: FETCH_SIZE counts half of what was fetched — /opt/skills/guides/MI355X_MICROARCH.md, HBM section —
            # and profiles/r05j_counter_calibration.txt holds the check on kernels of known byte counts)
            traffic, traffic_src, traffic_here = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 / 1e9, "this run: FETCH_SIZE x 2 + WRITE_SIZE, separate passes", True
        avg_kernel_s = kernel_ms * 1e-3 / max(launches, 1)
        # the vector-ALU side: instructions the launch issues against what the SIMDs can issue (one wave64 instruction per SIMD per four cycles)
        valu = None
        try:
            props = torch.cuda.get_device_properties(dev)
            simds = int(props.multi_processor_count) * 4
            clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
            quoted = None
            try:                                     # (the evidence pass's figure for this very library and workload, where there is one: profiles/valu.json)
                for tr in json.load(open(os.path.join(ROOT, "profiles", "valu.json")))["entries"]:
                    if (tr["config"], tr["groups_per_gpu"], tr["rounds"], tr["kernel"]) == (args.config, gpg, args.rounds, "%s<%d,false>" % ("rg::step32_kernel", F)) \
                            and not args.override and tr.get("lib_sha16") == engine.library_sha16() and tr.get("outcome_format", "rg_outcome_t") == ("rg_outcome32_t" if compact_out else "rg_outcome_t"):
                        quoted = tr
            except (OSError, KeyError, ValueError):
                pass
            insts = pmc["SQ_INSTS_VALU"] if pmc and "SQ_INSTS_VALU" in pmc else (quoted["sq_insts_valu_per_launch"] if quoted else None)
            if insts is not None and avg_kernel_s > 0:
                peak = simds * clock_hz / 4.0
                valu = {"insts_per_launch": insts, "per_64_groups_per_round": insts / max(1, (gpg + 63) // 64) / args.rounds,
                        "achieved_ginst_per_s": insts / avg_kernel_s / 1e9, "peak_ginst_per_s": peak / 1e9, "busy_frac": insts / avg_kernel_s / peak,
                        "simds": simds, "clock_mhz": clock_hz / 1e6, "peak_basis": "one wave64 vector instruction per SIMD per 4 cycles at the device's maximum clock",
                        "salu_per_launch": pmc.get("SQ_INSTS_SALU") if pmc else (quoted or {}).get("sq_insts_salu_per_launch"),
                        "lds_per_launch": pmc.get("SQ_INSTS_LDS") if pmc else (quoted or {}).get("sq_insts_lds_per_launch"),
                        "measured_in_this_run": bool(pmc and "SQ_INSTS_VALU" in pmc), "lib_sha16": engine.library_sha16(),
                        "source": "this run: rocprofv3 --pmc SQ_INSTS_VALU" if pmc and "SQ_INSTS_VALU" in pmc else quoted.get("source")}
        except Exception as e:      # a reporting leg must not take the bench line down with it
            print("bench: valu block failed: %r" % (e,), file=sys.stderr)
        alg_per_launch = alg_bytes / max(args.steps, 1)
        achieved = alg_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        moved_bytes = traffic * 1e9 if traffic is not None else floor_bytes
        moved_gbps = None if (moved_bytes is None or avg_kernel_s <= 0) else moved_bytes / avg_kernel_s / 1e9
        out = {
            "metric": "raft-group decisions/sec (AppendEntries+vote)",
            "value": decisions_all / elapsed,
            "unit": "decisions/s",
            "n_gpus": world,
            "n_gpus_seen": torch.cuda.device_count(),
            "launcher": "bench.py itself (one process per GPU)" if os.environ.get("RG_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "none"),
            "per_gpu": [{"rank": r, "device": int(row[3]), "value": row[0], "ms_per_step": row[1], "avg_kernel_ms": row[2]} for r, row in enumerate(per_gpu)],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {
                "workload": cfg.name if world == 1 else "%s: ranks 0..%d run its %d-group shards 0..%d (seed %#x, streams keyed by global group id)" % (
                    cfg.name, world - 1, gpg, world - 1, cfg.seed),
                "config_number": args.config, "seed": "%#x" % cfg.seed,
                "groups_per_gpu": gpg, "groups_total": gpg * world, "peers": cfg.cluster,
                "rounds_per_step": args.rounds, "decisions_per_step_per_gpu": decisions // max(args.steps, 1),
                "parallelism": "groups block-partitioned over %d GPU(s), no RCCL on the data path" % world,
                "inputs": "HBM-resident event/outcome buffers (RG_MEM_DEVICE); every step consumes fresh rounds",
                "rows": "rg_batch_t (wide: 8 + 16 + 16 B per row + 8 B per entry term)" if args.wide_rows else
                        "rg_batch32_t (compact: 8 + 16 B per row, the term shared by a row's entries in the row)",
                "outcomes": "rg_outcome32_t (ABI 4, rg_submit32c: one 16 B row per event + a 16 B persist row per conversion)" if compact_out else
                            "rg_outcome_t (16 B reply per event + 16 B effect / persist rows where flagged)",
                "arithmetic": "Java long (int64) semantics throughout; the compact-row kernel decides a workgroup's 64 groups with 32-bit instruction forms "
                              "while every value of those groups and of their rows is below 2^30 and redoes the workgroup in 64-bit forms otherwise "
                              "(tests/test_gpu_parity.py::test_compact_multi_round_launch_and_domain_exits) - results are bit-identical either way",
            },
            "roofline": {
                # `achieved` / `frac`: what the memory system moved per launch (rocprofv3 PMC, FETCH_SIZE x 2 + WRITE_SIZE in separate passes of this command
                # and this library build: profiles/traffic.json) over the launch duration measured HERE with HIP events. The byte count is a quotation from
                # the evidence pass (`traffic_measured_in_this_run`: false), the time is this run's. Without a matching entry the layout estimate stands in
                # and `frac_basis` says so. Never above 1.
                # `bound`: what limits the launch at this size is vector-instruction issue, not the memory system: the `valu` block beside
                # the HBM fraction says how busy the vector ALUs are; achieved / peak / unit / frac stay the HBM figures the contract defines
                "bound": "valu", "valu": valu, "achieved": moved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": None if moved_gbps is None else moved_gbps / HBM_PEAK_GBPS,
                "frac_basis": ("rocprofv3 PMC bytes measured in this run" if traffic_here else "rocprofv3 PMC bytes of this library build (profiles/traffic.json)") if traffic is not None else
                              ("layout estimate (no PMC entry for this library build / workload)" if floor_bytes is not None else None),
                "frac_note": "moved bytes over time: compact outcome rows (round 5) move 24 % fewer bytes per launch than round 4's columns (266 -> 203 MB at config 3), "
                             "so this fraction fell while decisions/s rose; traffic is within 6 % of the layout estimate and the launch is bound by vector-instruction "
                             "issue (`valu`), not by the memory system (DESIGN.md section 6). --wide-outcomes reproduces round 4's format." if compact_out else None,
                "traffic": traffic, "traffic_unit": "GB per launch (rocprofv3 PMC: %s)" % traffic_src,
                "traffic_lib_sha16": engine.library_sha16() if traffic is not None else None, "traffic_measured_in_this_run": traffic_here,
                "pmc_passes": pmc,
                "library": os.path.basename(engine.LIB_PATH), "lib_sha16": engine.library_sha16(),      # (same-box A/B records: which build a line is of)
                "per_launch": spread,
                "hbm_bytes_measured": None if traffic is None else traffic * 1e9,
                "hbm_gbps_measured": None if traffic is None else traffic / avg_kernel_s,
                "hbm_frac_measured": None if traffic is None else traffic / avg_kernel_s / HBM_PEAK_GBPS,
                "kernel": kernel_name, "outcome_format": out_fmt,
                "avg_kernel_ms": avg_kernel_s * 1e3, "launches": launches,
                # SURVEY.md 8(d)'s byte model as a WORK RATE: it charges every decision its group's state (which stays in registers across the rounds of a
                # launch) and 40 + 8n bytes per event (the compact row has 24): more than any memory system moves, so not a roofline fraction
                "work_rate_algorithmic_gbps": achieved, "work_rate_algorithmic_over_peak": achieved / HBM_PEAK_GBPS,
                "algorithmic_bytes_per_launch": alg_per_launch,
                "algorithmic_bytes_per_decision": alg_bytes / max(decisions, 1),
                "model_overcharges": bool(achieved > HBM_PEAK_GBPS),
                "measured_copy_gbps": copy_gbps,
                # an estimate of what this row / table layout moves per launch (events in, outcome rows out, every table column once; row counts of the
                # stream's first launch): close to the counters, not a bound
                "layout_estimate_bytes": floor_bytes,
                "layout_estimate_gbps": None if floor_bytes is None else floor_bytes / avg_kernel_s / 1e9,
                "frac_of_layout_estimate": None if floor_bytes is None else floor_bytes / avg_kernel_s / 1e9 / HBM_PEAK_GBPS,
                "ms_int64_body": None if int64_pass is None else int64_pass["ms"],
                "value_int64_body": None if int64_pass is None else int64_pass["value"],
                "int64_body_counters_equal": None if int64_pass is None else int64_pass["counters_equal_first_pass"],
            },
            "tick_latency": tick,
            "int64_body_workgroups": wide_wgs,
            "value_long_lived_groups": None if not long_lived or "value" not in long_lived else long_lived["value"],
            "long_lived_groups": long_lived,
            "value_long_launches": None if not long_launch or "value" not in long_launch else long_launch["value"],
            "long_launches": long_launch,
            "value_adverse_mix": None if not adverse or "value" not in adverse else adverse["value"],
            "adverse_mix": adverse,
            "golden": golden,
            "cpu_baseline": cpu,
            "pcie_inclusive_value": pcie_packed if pcie_packed is not None else (pcie_pipe if pcie_pipe is not None else pcie),
            "pcie_inclusive": {"serial_rg_submit": pcie, "pipelined_rg_submit_async": pcie_pipe, "pipelined_rg_submit_async_packed": pcie_packed,
                               "link_gbytes_per_s_both_ways": {"wide": pcie_gbps, "packed": packed_gbps}, **pcie_bytes,
                               "timed_batches": None if pcie is None else args.pcie_batches - 2,
                               "note": "decisions/s with caller-owned page-locked host buffers, H2D + D2H included, fresh batches of the "
                                       "stream from one table state for all three legs; pcie_inclusive_value = the packed pipeline "
                                       "(int32 event fields up, packed logfx / persist lists down, replies checked equal to the wide "
                                       "pipeline's): a transport figure, never `value`"},
            "counters": dict(zip(["rows", "replied", "role_conversions", "commit_advances", "asserts", "need_host",
                                  "dropped_stale", "log_appends"], counters)),
            "stage_seconds": t_gen,
        }
        print(json.dumps(out), flush=True)

    for db in dbatches:
        db.free()
    table.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
