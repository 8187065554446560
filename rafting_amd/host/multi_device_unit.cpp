// multi_device_unit.cpp — MultiDeviceManager against a single ContextManager (needs a GPU; run by tests/test_gpu_host_cluster.py).
// The same contexts receive the same random protocol traffic (a) in ONE table and (b) block-partitioned over several tables,
// each drained by its own feeder thread (here: several tables on the one GPU of the test box — the data path is identical, only
// the device ordinal differs on a multi-GPU node). Every outcome and every mirror must be identical, context by context.
// usage: multi_device_unit [contexts=96] [shards=3] [rounds=200]      exit code 0 = identical
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>

#include "multi_device.hpp"

using namespace raftgpu::host;

struct Driver {                                   // the traffic one context sees, a function of (context, round) only
    std::mt19937_64 rng;
    int64_t last = 0, lastTerm = 0;
    void play(RaftContext &c, int P, ID self)
    {
        const auto l = c.replicatedLog().last();
        last = l ? l->index : 0; lastTerm = l ? l->term : 0;
        const ID other = (ID)((self + 1 + rng() % (P - 1)) % P);
        const uint64_t x = rng() % 100;
        const int64_t t = c.currentTerm();
        if (c.role() == RG_LEADER) {
            if (x < 40) c.acceptCommand(1 + (uint32_t)(rng() % 2));
            else if (x < 80) c.onAppendEntriesResponse(other, {t, rng() % 8 != 0}, 0, last - (int64_t)(rng() % 2), c.roleEpoch());
            else if (x < 90) c.onTimeout();
            else c.appendEntries(t + 1, other, last, lastTerm, {}, 0);
        } else if (c.role() == RG_CANDIDATE) {
            if (x < 70) c.onVoteResponse(false, other, {t, rng() % 4 != 0}, c.roleEpoch());
            else if (x < 85) c.onTimeout();
            else c.appendEntries(t, other, last, lastTerm, {}, 0);
        } else {
            if (x < 45) {
                std::vector<Entry> e;
                for (uint64_t k = 0, n = rng() % 3; k < n; k++) e.push_back({last + 1 + (int64_t)k, std::max<int64_t>(t, 1)});
                c.appendEntries(std::max<int64_t>(t, 1), (ID)((self + 1) % P), last, lastTerm, e, last);
            } else if (x < 60) c.onTimeout();
            else if (x < 85) c.onVoteResponse(true, other, {t + 1, rng() % 4 != 0}, c.roleEpoch());
            else if (x < 93) c.requestVote(t + 1, other, last + (int64_t)(rng() % 2), lastTerm);
            else c.preVote(t + 1, other, last, lastTerm);
        }
    }
};

static bool same(const Outcome &a, const Outcome &b)
{
    return a.status == b.status && a.flags == b.flags && a.roleEpoch == b.roleEpoch && a.role == b.role &&
           a.response.has_value() == b.response.has_value() &&
           (!a.response || (a.response->term == b.response->term && a.response->success == b.response->success));
}

int main(int argc, char **argv)
{
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 96;
    const size_t S = argc > 2 ? (size_t)atoi(argv[2]) : 3;
    const int rounds = argc > 3 ? atoi(argv[3]) : 200;
    const int P = 5; const ID self = 2;
    ContextManager single(0, N, P, self, true);
    MultiDeviceManager multi(std::vector<int>(S, 0), N, P, self, true);
    std::vector<RaftContext *> a, b;
    std::vector<Driver> da(N), db(N);
    for (uint32_t i = 0; i < N; i++) {
        const std::string id = "ctx-" + std::to_string(i);
        a.push_back(&single.createContext(id));
        b.push_back(&multi.createContext(id));
        da[i].rng.seed(1000 + i); db[i].rng.seed(1000 + i);
        if (multi.globalGid(id) != i || multi.shardOf(id) != i / ((N + S - 1) / S)) { fprintf(stderr, "routing of %s is off\n", id.c_str()); return 1; }
    }
    uint64_t rows = 0, leaders = 0;
    int bad = 0;
    for (int r = 0; r < rounds && !bad; r++) {
        for (uint32_t i = 0; i < N; i++) { da[i].play(*a[i], P, self); db[i].play(*b[i], P, self); }
        std::vector<Outcome> oa = single.flush();                       // ticket i = context i (every context queued one row)
        std::vector<std::vector<Outcome>> ob = multi.flushAll();        // per shard, ticket = position inside the shard
        std::vector<size_t> next(S, 0);
        for (uint32_t i = 0; i < N && !bad; i++) {
            const size_t k = multi.shardOf(a[i]->ctxID());
            const Outcome &x = oa[i], &y = ob[k][next[k]++];
            if (!same(x, y) || a[i]->role() != b[i]->role() || a[i]->currentTerm() != b[i]->currentTerm() || a[i]->votedFor() != b[i]->votedFor() ||
                a[i]->roleEpoch() != b[i]->roleEpoch() || a[i]->replicatedLog().lastCommitted() != b[i]->replicatedLog().lastCommitted()) {
                fprintf(stderr, "round %d context %u: single table and shard %zu disagree (status %u/%u flags %x/%x)\n", r, i, k, x.status, y.status, x.flags, y.flags);
                bad = 1;
            }
            rows++;
            leaders += a[i]->role() == RG_LEADER;
        }
    }
    // createContext / getContext from a second thread WHILE flushAll runs: N2 more contexts appear on a manager whose shards
    // keep draining; every one of them must exist afterwards, be routed by its creation index, and the drains must not have been disturbed
    if (!bad) {
        const uint32_t N2 = N;
        MultiDeviceManager busy(std::vector<int>(S, 0), N + N2, P, self, true);
        std::vector<RaftContext *> c;
        for (uint32_t i = 0; i < N; i++) c.push_back(&busy.createContext("old-" + std::to_string(i)));
        std::atomic<bool> stop{false};
        std::atomic<int> made{0}, wrong{0};
        std::atomic<uint64_t> flushes{0};
        std::thread creator([&] {
            for (uint32_t i = 0; i < N2; i++) {
                while (flushes.load() < i / 4u + 1u) std::this_thread::yield();      // keep the two threads interleaved: a few creations per drain
                RaftContext &x = busy.createContext("new-" + std::to_string(i));
                if (busy.getContext("new-" + std::to_string(i)) != &x || busy.globalGid("new-" + std::to_string(i)) != N + i) wrong++;
                made++;
            }
            stop = true;
        });
        std::mt19937_64 rng(7);
        uint64_t drained = 0;
        while (!stop) {
            for (uint32_t i = 0; i < N; i++) if (rng() % 3 == 0) c[i]->onTimeout();
            for (auto &v : busy.flushAll()) drained += v.size();
            flushes++;
        }
        creator.join();
        for (uint32_t i = 0; i < N2; i++) if (!busy.getContext("new-" + std::to_string(i))) wrong++;
        if (made != (int)N2 || wrong != 0 || drained == 0) { fprintf(stderr, "concurrent createContext: made %d of %u, %d wrong, %llu rows drained\n", made.load(), N2, wrong.load(), (unsigned long long)drained); bad = 1; }
    }
    printf("multi-device ok=%d contexts=%u shards=%zu rounds=%d rows=%llu leader_rows=%llu\n", !bad, N, S, rounds, (unsigned long long)rows,
           (unsigned long long)leaders);
    return bad || leaders == 0;
}
