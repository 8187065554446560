// stable_store_unit.cpp — crash-ordering checks for rafting_amd/host/stable_store.cpp (ContextManager.restore / StableLock.persist,
// RaftContext.java:199-216 in the reference keeps (term, votedFor) durable before any reply leaves). CPU only.
#include <sys/resource.h>
#include <sys/stat.h>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../rafting_amd/host/stable_store.hpp"

using raftgpu::host::StableStore;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static off_t size_of(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0 ? st.st_size : -1; }

int main(int argc, char **argv)
{
    const std::string path = std::string(argc > 1 ? argv[1] : "/tmp") + "/stable_unit.log";
    ::remove(path.c_str());
    int64_t term = 0; int32_t voted = 0;
    {
        StableStore s(path);                        // first create: header + directory entry synced
        s.persist({{7, 11, 3}, {9, 4, -1}});
        CHECK(s.restore(7, &term, &voted) && term == 11 && voted == 3);
    }
    const off_t good = size_of(path);
    {
        StableStore s(path);
        CHECK(s.restore(9, &term, &voted) && term == 4 && voted == -1);
        // make the next append fail half way: the file may not grow past good + 10 bytes
        signal(SIGXFSZ, SIG_IGN);
        struct rlimit old, lim;
        getrlimit(RLIMIT_FSIZE, &old);
        lim = old; lim.rlim_cur = (rlim_t)good + 10;
        setrlimit(RLIMIT_FSIZE, &lim);
        bool threw = false;
        try { s.persist({{7, 12, 5}}); } catch (const std::exception &) { threw = true; }
        setrlimit(RLIMIT_FSIZE, &old);
        CHECK(threw);
        CHECK(s.restore(7, &term, &voted) && term == 11 && voted == 3);      // memory still says what is durable
        CHECK(size_of(path) == good);                                         // the torn record was cut off again
        s.persist({{7, 13, 6}});                                              // and the store goes on working
        CHECK(s.restore(7, &term, &voted) && term == 13 && voted == 6);
    }
    {
        StableStore s(path);                        // replay sees both good batches, nothing of the failed one
        CHECK(s.restore(7, &term, &voted) && term == 13 && voted == 6);
        CHECK(s.restore(9, &term, &voted) && term == 4);
        s.compact();
        CHECK(size_of(path + ".tmp") == -1);
    }
    {
        StableStore s(path);
        CHECK(s.restore(7, &term, &voted) && term == 13 && voted == 6);
        CHECK(s.restore(9, &term, &voted) && term == 4 && voted == -1);
    }
    ::remove(path.c_str());
    printf("stable-store ok=1\n");
    return 0;
}
