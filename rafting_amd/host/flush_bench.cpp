// flush_bench.cpp — what a host that drains once per tick gets out of ContextManager::flush, with and without RG_NEED_HOST round trips
// (VERDICT r2 "measure what is built but unmeasured"; needs a GPU). One drain = every context queues one AppendEntries, flush() submits
// them as ONE sparse single-round rg_submit, looks the missing terms of the RG_NEED_HOST rows up in its own RaftLog (MemoryLog here,
// RocksDB in the reference: storage/RocksLog.java:122-128) and resubmits those rows with hints, then applies log effects / commit.
//   contexts hold a log of SIX term runs (the device caches the newest four): a request whose prevLogIndex lies in the two oldest runs
//   cannot be answered from the cache.
// usage: flush_bench [contexts=65536] [drains=20] [old_prev_share_percent=5]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "raft_host.hpp"

using namespace raftgpu::host;

int main(int argc, char **argv)
{
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536;
    const int drains = argc > 2 ? atoi(argv[2]) : 20;
    const int old_pct = argc > 3 ? atoi(argv[3]) : 5;
    const int P = 5; const ID self = 2, leader = 0;
    ContextManager mgr(0, N, P, self, true);
    std::vector<RaftContext *> ctx;
    for (uint32_t i = 0; i < N; i++) ctx.push_back(&mgr.createContext("ctx-" + std::to_string(i)));
    // six runs of two entries each: terms 1..6, indices 1..12
    for (int t = 1; t <= 6; t++) {
        for (uint32_t i = 0; i < N; i++) {
            const int64_t prev = 2 * (t - 1);
            ctx[i]->appendEntries(t, leader, prev, prev == 0 ? 0 : t - 1, {{prev + 1, t}, {prev + 2, t}}, prev);
        }
        for (const Outcome &o : mgr.flush()) if (o.status != RG_OK) { fprintf(stderr, "setup: status %u\n", o.status); return 1; }
    }
    std::mt19937_64 rng(42);
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int pass = 0; pass < 2; pass++) {               // pass 0: every request at the log tail; pass 1: old_pct % of them reach below the cache
        const uint64_t hints0 = mgr.hintsServed(), rows0 = mgr.rowsDecided();
        uint64_t rows = 0, refused = 0;
        const auto t0 = now();
        for (int d = 0; d < drains; d++) {
            for (uint32_t i = 0; i < N; i++) {
                const bool old = pass == 1 && (int)(rng() % 100) < old_pct;
                // heartbeat whose prevLog is entry 2 (term 1, the oldest run) or the tail (12, term 6)
                if (old) ctx[i]->appendEntries(6, leader, 2, 1, {}, 0); else ctx[i]->appendEntries(6, leader, 12, 6, {}, 12);
            }
            for (const Outcome &o : mgr.flush()) { rows++; refused += !(o.response && o.response->success); }
        }
        const double s = std::chrono::duration<double>(now() - t0).count();
        printf("%s: %u contexts x %d drains = %llu rows in %.3f s: %.3e rows/s through ContextManager::flush; %llu hint round trips (%.2f %% of the rows), "
               "%llu device rows incl. resubmissions, %llu not successful\n", pass == 0 ? "tail only" : "with cache misses", N, drains, (unsigned long long)rows, s,
               rows / s, (unsigned long long)(mgr.hintsServed() - hints0), 100.0 * (mgr.hintsServed() - hints0) / rows,
               (unsigned long long)(mgr.rowsDecided() - rows0), (unsigned long long)refused);
        if (refused) return 1;
        if (pass == 1 && old_pct > 0 && mgr.hintsServed() == hints0) { fprintf(stderr, "no hint was served: the stream did not miss the cache\n"); return 1; }
    }
    return 0;
}
