// context_index_churn.cpp — regression for ADVICE r3 (medium): ContextIndex under context churn. A table of 64 groups sees thousands of
// insert / erase / reclaim cycles with DISTINCT ids while lookups (hits and misses) run: every probe must end, live ids must keep resolving,
// erased ids must stop resolving, tombstones must stay bounded, and lookups on other threads must survive the slot array being rebuilt.
// usage: context_index_churn [capacity=64] [cycles=20000]      prints "context index churn ok=1"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "ingress.hpp"

using namespace rafting::wire;

int main(int argc, char **argv)
{
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 64;
    const uint32_t CYCLES = argc > 2 ? (uint32_t)atoi(argv[2]) : 20000;
    ContextIndex index(G);
    std::vector<std::string> live(G);
    uint64_t serial = 0;
    auto fresh = [&] { return "ctx-" + std::to_string(serial++); };
    for (uint32_t g = 0; g < G; g++) { live[g] = fresh(); if (!index.insert(live[g].data(), live[g].size(), g)) { printf("initial insert %u failed\n", g); return 1; } }
    // groups [0, G/4) are never erased: reader threads look them up all the time, also across rebuilds
    // the owner's contract (ingress.hpp): reclaim() after a seal that began after the erase — the seal excludes every feed(), i.e. every
    // lookup; `feeding` stands in for the ingress's reader / writer lock
    std::shared_mutex feeding;
    std::atomic<bool> stop{false}, bad{false};
    std::vector<std::thread> readers;
    for (int t = 0; t < 2; t++)
        readers.emplace_back([&, t] {
            uint64_t x = 12345 + (uint64_t)t;
            while (!stop.load(std::memory_order_relaxed)) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                const uint32_t g = (uint32_t)(x >> 33) % (G / 4);
                uint32_t got = ~0u;
                std::shared_lock<std::shared_mutex> feed(feeding);
                if (!index.find(live[g].data(), live[g].size(), got) || got != g) bad = true;
                const std::string miss = "nobody-" + std::to_string(x >> 40);
                if (index.find(miss.data(), miss.size(), got)) bad = true;           // (a miss walks its whole run: this is what used to spin)
            }
        });
    uint64_t y = 99;
    uint32_t max_tombs = 0;
    for (uint32_t c = 0; c < CYCLES && !bad; c++) {
        y = y * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t g = G / 4 + (uint32_t)(y >> 33) % (G - G / 4);
        const std::string old = live[g];
        if (index.erase(old.data(), old.size()) != g) { printf("cycle %u: erase of a live id failed\n", c); return 1; }
        uint32_t got;
        if (index.find(old.data(), old.size(), got)) { printf("cycle %u: an erased id still resolves\n", c); return 1; }
        if (index.insert(fresh().data(), 5, g)) { printf("cycle %u: a retired gid was handed out before reclaim\n", c); return 1; }
        const size_t n = index.retired();
        { std::unique_lock<std::shared_mutex> seal(feeding); }   // the seal barrier: lookups that began before the erase are over
        index.reclaim(n);
        live[g] = fresh();
        if (!index.insert(live[g].data(), live[g].size(), g)) { printf("cycle %u: insert after reclaim failed\n", c); return 1; }
        if (!index.find(live[g].data(), live[g].size(), got) || got != g) { printf("cycle %u: a fresh id does not resolve\n", c); return 1; }
        if (index.tombstones() > max_tombs) max_tombs = index.tombstones();
        if (c % 97 == 0)
            for (uint32_t k = 0; k < G; k++)
                if (!index.find(live[k].data(), live[k].size(), got) || got != k) { printf("cycle %u: live id of group %u lost\n", c, k); return 1; }
    }
    stop = true;
    for (auto &t : readers) t.join();
    if (bad) { printf("a reader thread saw a wrong answer\n"); return 1; }
    if (index.size() != G) { printf("size %u != %u\n", index.size(), G); return 1; }
    printf("context index churn ok=1  cycles=%u  max tombstones %u  rebuilds %u\n", CYCLES, max_tombs, index.rebuilds());
    return 0;
}
