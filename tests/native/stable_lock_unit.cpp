// stable_lock_unit.cpp — the reference's StableLock file (support/StableLock.java:47-91) read into (milestone, term, votedFor) and handed to the journal
// of the batched path (StableStore), and written back. The expected bytes below are spelled out by hand from the file layout and Kryo 4.0.2's rules for a
// NodeID (class by name, reference marker, ASCII string with the end mark on its last byte, zig-zag varint port); CPU only.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../rafting_amd/host/stable_lock_file.hpp"
#include "../../rafting_amd/host/stable_store.hpp"

using namespace raftgpu::host;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, err.c_str()); return 1; } } while (0)

static void put_file(const std::string &p, const std::string &bytes) { FILE *f = fopen(p.c_str(), "wb"); fwrite(bytes.data(), 1, bytes.size(), f); fclose(f); }
static std::string get_file(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); std::string s; char b[4096]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) s.append(b, n); fclose(f); return s; }

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    std::string err;
    StableLockImage img;
    // 1. a fresh file: 28 zero bytes (StableLock.<init>) = term 0, no vote, no milestone
    put_file(dir + "/fresh.lock", std::string(28, '\0'));
    CHECK(read_stable_lock(dir + "/fresh.lock", &img, &err) && img.term == 0 && !img.has_vote && img.milestone_index == 0);
    // 2. persist(7, NodeID("127.0.0.1", 6002)) after persist(Snapshot 40:3): header big-endian, then Kryo's image of the id
    std::string id;
    id += (char)1; id += (char)0;                                                // class by name, name id 0 (first name of this stream)
    const std::string cls = "io.lubricant.consensus.raft.transport.event.NodeID";
    id += cls.substr(0, cls.size() - 1); id += (char)(cls.back() | 0x80);       // ASCII string: end mark on the last byte
    id += "\x01";                                                                // reference marker: first occurrence
    id += "\x01";                                                                // hostname: not null
    id += "127.0.0."; id += (char)('1' | 0x80);
    id += "\xE4\x5D";                                                            // port 6002, zig-zag (12004) as a varint: 0xE4 0x5D
    std::string file;
    const unsigned char hdr[28] = {0, 0, 0, 0, 0, 0, 0, 40, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 0, 7, 0, 0, 0, (unsigned char)id.size()};
    file.assign(reinterpret_cast<const char *>(hdr), 28); file += id;
    put_file(dir + "/voted.lock", file + "stale tail of a longer id");            // (persist never truncates)
    CHECK(read_stable_lock(dir + "/voted.lock", &img, &err));
    CHECK(img.milestone_index == 40 && img.milestone_term == 3 && img.term == 7 && img.has_vote && img.vote_host == "127.0.0.1" && img.vote_port == 6002);
    // 3. the writer produces exactly those bytes
    CHECK(write_stable_lock(dir + "/again.lock", img, &err) && get_file(dir + "/again.lock") == file);
    // 4. persist(9, null): Kryo's null is one byte
    StableLockImage none; none.term = 9;
    CHECK(write_stable_lock(dir + "/none.lock", none, &err) && get_file(dir + "/none.lock").size() == 29 && get_file(dir + "/none.lock")[27] == 1 && get_file(dir + "/none.lock")[28] == 0);
    CHECK(read_stable_lock(dir + "/none.lock", &img, &err) && img.term == 9 && !img.has_vote);
    // 5. refused: a short file, a length beyond the file, bytes that are no NodeID
    put_file(dir + "/short.lock", std::string(20, '\0'));
    CHECK(!read_stable_lock(dir + "/short.lock", &img, &err) && err.find("28-byte") != std::string::npos);
    std::string bad = file; bad[27] = (char)200;
    put_file(dir + "/bad1.lock", bad.substr(0, 28 + id.size()));
    CHECK(!read_stable_lock(dir + "/bad1.lock", &img, &err) && err.find("beyond") != std::string::npos);
    bad = file; bad[30] = 'X';
    put_file(dir + "/bad2.lock", bad);
    CHECK(!read_stable_lock(dir + "/bad2.lock", &img, &err) && err.find("NodeID") != std::string::npos);
    // 6. the switch-over: every context's lock file -> one record of the node's journal (peer slot = position in the cluster list), and nothing is lost
    const std::vector<std::pair<std::string, int>> cluster = {{"127.0.0.1", 6001}, {"127.0.0.1", 6002}, {"127.0.0.1", 6003}};
    ::remove((dir + "/journal.log").c_str());
    {
        StableStore store(dir + "/journal.log");
        std::vector<StableStore::Record> batch;
        const char *files[] = {"fresh.lock", "voted.lock", "none.lock"};
        for (uint32_t gid = 0; gid < 3; gid++) {
            CHECK(read_stable_lock(dir + "/" + files[gid], &img, &err));
            int32_t slot = -1;
            for (size_t s = 0; img.has_vote && s < cluster.size(); s++) if (cluster[s].first == img.vote_host && cluster[s].second == img.vote_port) slot = (int32_t)s;
            CHECK(!img.has_vote || slot >= 0);
            batch.push_back({gid, img.term, slot});
        }
        store.persist(batch);                                                     // one write, one fdatasync for the whole node
        CHECK(store.syncs() == 1);
    }
    {
        StableStore store(dir + "/journal.log");
        int64_t term; int32_t voted;
        CHECK(store.restore(0, &term, &voted) && term == 0 && voted == -1);
        CHECK(store.restore(1, &term, &voted) && term == 7 && voted == 1);
        CHECK(store.restore(2, &term, &voted) && term == 9 && voted == -1);
    }
    printf("stable-lock ok=1\n");
    return 0;
}
