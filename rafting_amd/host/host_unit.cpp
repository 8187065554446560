// host_unit.cpp — CPU-only checks of the host-owned RaftLog stand-in (MemoryLog): it must show the observable
// behaviour of command/storage/RocksLog.java that the decision path relies on (quirks Q8-Q10).
// Needs no GPU and no libraftgpu.so. exit code 0 = all checks passed.
#include <unistd.h>

#include <cstdio>
#include <ctime>
#include <string>

#include "raft_host.hpp"

using namespace raftgpu::host;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)
template <class F> static bool throws(F f) { try { f(); } catch (const std::exception &) { return true; } return false; }

int main()
{
    MemoryLog log;
    CHECK(!log.last() && log.epoch().index == 0 && log.epoch().term == 0);
    CHECK(log.newEntry(3).index == 1);                                   // RocksLog.java:83-84: first key is 1
    log.append({{2, 3}, {3, 3}, {4, 4}});
    CHECK(log.last()->index == 4 && log.last()->term == 4 && log.get(2)->term == 3 && !log.get(5) && !log.get(0));
    CHECK(!log.conflict({{3, 3}, {4, 4}, {5, 4}}));                      // stops at the first absent key (:206-209)
    CHECK(log.conflict({{3, 3}, {4, 5}})->index == 4 && log.conflict({{3, 3}, {4, 5}})->term == 4);
    log.append({{3, 3}, {4, 4}});                                        // stale duplicate never shortens the log (Q10)
    CHECK(log.last()->index == 4);
    log.append({{4, 4}, {5, 4}, {6, 5}});
    CHECK(log.last()->index == 6 && log.runs().size() == 3);
    CHECK(throws([&] { log.append({{9, 5}}); }));                        // gap: "log index is not continuous" (:184-188)
    log.truncate(5);
    CHECK(log.last()->index == 4);
    log.truncate(9);                                                     // beyond last: nothing (:221)
    CHECK(log.last()->index == 4);
    CHECK(log.markCommitted(3) && !log.markCommitted(3) && log.lastCommitted() == 3);
    CHECK(throws([&] { log.markCommitted(2); }));                        // "rollback is not allowed" (Q8, :101-103)
    log.flush(3, 3);                                                     // deleteRange [0,3): key 3 == new epoch survives (:235)
    CHECK(log.epoch().index == 3 && log.firstIndex() == 3 && log.get(3) && !log.get(2) && log.last()->index == 4);
    CHECK(throws([&] { log.flush(2, 3); }));                             // IndexOutOfBounds (:230-233)
    log.flush(10, 6);                                                    // beyond last: emptied (snapshot install)
    CHECK(!log.last() && log.epoch().index == 10);
    CHECK(throws([&] { log.append({{12, 6}}); }));                       // "should follow epoch closely" (:175-177)
    log.append({{11, 6}});
    CHECK(log.firstIndex() == 11 && log.last()->index == 11);
    MemoryLog fresh;
    fresh.flush(7, 2);
    CHECK(fresh.newEntry(5).index == 1);                                 // Q14: index 1 whatever the epoch — why the device refuses this row
    // ---- N3: StableStore ---------------------------------------------------------------------------------
    {
        const std::string path = "/tmp/rg_stable_unit_" + std::to_string((long)getpid()) + ".journal";
        ::unlink(path.c_str());
        {
            StableStore st(path);
            CHECK(!st.restore(3, nullptr, nullptr));
            st.persist({{3, 7, 1}, {9, 2, RG_NO_NODE}, {3, 8, 2}});         // one batch = one fdatasync; last record of a group wins
            st.persist({{5, 1, 0}});
            CHECK(st.syncs() == 2 && st.records() == 4 && st.groups() == 3);
        }
        {
            StableStore st(path);                                             // "restart": replay the journal
            int64_t t = 0; int32_t v = 0;
            CHECK(st.restore(3, &t, &v) && t == 8 && v == 2);
            CHECK(st.restore(9, &t, &v) && t == 2 && v == RG_NO_NODE);
            CHECK(st.restore(5, &t, &v) && t == 1 && v == 0 && !st.restore(4, &t, &v));
            st.compact();
            CHECK(st.restore(3, &t, &v) && t == 8);
            st.persist({{4, 11, 1}});
        }
        {
            FILE *f = fopen(path.c_str(), "ab");                              // crash in the middle of a record: torn tail
            fwrite("\x01\x02\x03\x04\x05\x06\x07", 1, 7, f);
            fclose(f);
            StableStore st(path);
            int64_t t = 0; int32_t v = 0;
            CHECK(st.groups() == 4 && st.restore(4, &t, &v) && t == 11 && v == 1);
            st.persist({{4, 12, 2}});                                         // journal is writable again after the truncation
        }
        {
            StableStore st(path);
            int64_t t = 0;
            CHECK(st.restore(4, &t, nullptr) && t == 12);
        }
        // what the barrier buys: N role conversions, one fdatasync per conversion vs one per flush
        const int N = 256;
        auto time_it = [&](bool batched) {
            const std::string p2 = path + (batched ? ".b" : ".s");
            ::unlink(p2.c_str());
            StableStore st(p2);
            std::vector<StableStore::Record> all;
            for (int i = 0; i < N; i++) all.push_back({(uint32_t)i, 5, 1});
            timespec a, b;
            clock_gettime(CLOCK_MONOTONIC, &a);
            if (batched) st.persist(all); else for (auto &r : all) st.persist({r});
            clock_gettime(CLOCK_MONOTONIC, &b);
            ::unlink(p2.c_str());
            return (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
        };
        const double per_group = time_it(false), batched = time_it(true);
        printf("StableStore: %d conversions, one fdatasync each %.3f ms vs one batched fdatasync %.3f ms\n", N, per_group, batched);
        ::unlink(path.c_str());
    }
    printf("host_unit: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
