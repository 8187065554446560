// ingress_race.cpp — the ingress (rafting_amd/host/ingress.hpp) with its threads as a deployment has them, for ThreadSanitizer / ASan:
// C feeder threads (one connection each, random piece sizes), a sender thread that registers pending requests while their responses are
// already arriving on other connections, a thread that adds contexts to the index while lookups run, and the flush thread sealing,
// emitting and recycling without waiting for anybody. Checks: every row comes out exactly once, rows of one (connection, group) in order.
// usage: ingress_race [groups=256] [conns=6] [rows per conn=20000] [max rounds=4] [shards=1]      prints "ingress race ok=1"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "ingress.hpp"

using namespace rafting::wire;

int main(int argc, char **argv)
{
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 256, C = argc > 2 ? (uint32_t)atoi(argv[2]) : 6;
    const uint32_t N = argc > 3 ? (uint32_t)atoi(argv[3]) : 20000, R = argc > 4 ? (uint32_t)atoi(argv[4]) : 4, SHARDS = argc > 5 ? (uint32_t)atoi(argv[5]) : 1;
    {   // the invocation ring with its two threads: the sender files requests while the reader takes the responses of earlier ones out of the SAME
        // slots (capacity 16: the sender laps the reader all the time). A take that succeeds must return exactly what was filed for that sequence.
        PendingRing ring(16);
        std::atomic<int32_t> filed{-1};
        std::atomic<bool> bad{false};
        const int32_t M = 200000;
        std::thread sender([&] { for (int32_t q = 0; q < M; q++) { ring.put(q, M_APPEND_ENTRIES, (uint32_t)q & 1023u, Pending{(uint32_t)q, (int64_t)q * 3, (int64_t)q * 7}); filed.store(q); } });
        uint64_t taken = 0;
        for (int32_t q = 0; q < M; q++) {
            while (filed.load() < q) std::this_thread::yield();
            Pending p;
            if (ring.take(q, M_APPEND_ENTRIES, (uint32_t)q & 1023u, p)) {
                taken++;
                if (p.role_epoch != (uint32_t)q || p.epoch_at_send != (int64_t)q * 3 || p.last_index_sent != (int64_t)q * 7) bad = true;
            }
            if (ring.take(q, M_APPEND_ENTRIES, (uint32_t)q & 1023u, p)) bad = true;      // removed: a duplicate response finds nothing
        }
        sender.join();
        if (bad || taken == 0) { printf("pending ring: a take returned another request's record (or nothing was ever taken: %llu)\n", (unsigned long long)taken); return 1; }
    }
    const KryoBodyCodec codec({{"a", 1}, {"b", 2}, {"c", 3}});
    ContextIndex index(G);
    auto id_of = [](uint32_t g) { return "group/" + std::to_string(g); };
    for (uint32_t g = 0; g < G / 2; g++) index.insert(id_of(g).data(), id_of(g).size(), g);       // the other half arrives while frames do
    const size_t cells = (size_t)G * R;
    std::vector<rg_ev_head_t> head[2] = {std::vector<rg_ev_head_t>(cells), std::vector<rg_ev_head_t>(cells)};
    std::vector<rg_ev_quad32_t> abcd[2] = {std::vector<rg_ev_quad32_t>(cells), std::vector<rg_ev_quad32_t>(cells)};
    std::vector<int32_t> terms[2] = {std::vector<int32_t>(4096), std::vector<int32_t>(4096)};
    Ingress ing(G, R, C + 1, codec, index, Ingress::Buffers{head[0].data(), abcd[0].data(), terms[0].data(), terms[0].size()},
                Ingress::Buffers{head[1].data(), abcd[1].data(), terms[1].data(), terms[1].size()}, 1u << 16, SHARDS);
    ing.retain_bodies(true);
    for (uint32_t c = 0; c < C; c++) ing.set_peer(c, (int32_t)(c % 3));

    // streams: requests only for groups of the first half (always known); the tag of a row = its sequence number on its connection
    std::vector<std::string> stream(C);
    std::vector<std::map<uint32_t, std::vector<int32_t>>> sent(C + 1);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (uint32_t c = 0; c < C; c++)
        for (uint32_t k = 0; k < N; k++) {
            const uint32_t g = (uint32_t)(rnd() % (rnd() % 3 ? G / 2 : 4));
            Frame f;
            f.type = ENQ; f.sequence = (int32_t)k;
            const Method m = (rnd() & 1) ? M_APPEND_ENTRIES : M_REQUEST_VOTE;                        // (half of them AppendEntries with one entry: their bodies are retained)
            f.head = make_scope(m, id_of(g));
            Request q;
            q.term = (rnd() % 200 == 0) ? (int64_t)1 << 40 : 5; q.node = 1; q.x = k; q.y = 2;       // a few rows outside the compact format
            if (m == M_APPEND_ENTRIES) q.entry_terms.assign(1, 5);
            codec.encode_request(m, q, f.body);
            encode_frame(f, false, stream[c]);
            sent[c][g].push_back((int32_t)k);
        }
    std::atomic<uint32_t> running{C + 2};
    std::vector<std::thread> th;
    for (uint32_t c = 0; c < C; c++)
        th.emplace_back([&, c] {
            uint64_t y = 1000 + c;
            for (size_t at = 0; at < stream[c].size();) {
                y = y * 6364136223846793005ull + 1442695040888963407ull;
                const size_t n = std::min<size_t>(1 + (y >> 33) % 5000, stream[c].size() - at);
                if (ing.feed(c, reinterpret_cast<const uint8_t *>(stream[c].data()) + at, n) < 0) abort();
                at += n;
            }
            running--;
        });
    th.emplace_back([&] {                                         // contexts created while traffic flows (ContextManager.createContext)
        for (uint32_t g = G / 2; g < G; g++) { index.insert(id_of(g).data(), id_of(g).size(), g); uint32_t got; if (!index.find(id_of(g).data(), id_of(g).size(), got) || got != g) abort(); }
        running--;
    });
    th.emplace_back([&] {                                         // the host's own rows on a connection number of their own
        for (uint32_t k = 0; k < N; k++) {
            const uint32_t g = k % (G / 2);
            ing.add_row(C, g, rg_ev_head_t{RG_HDR_MAKE(RG_EV_TIMEOUT, 0, 0, 0), 0}, 5, k, 0, 0, Origin{C, (int32_t)k});
            sent[C][g].push_back((int32_t)k);
        }
        running--;
    });
    std::vector<std::map<uint32_t, std::vector<int32_t>>> got(C + 1);
    std::vector<rg_reply_t> reply(cells, rg_reply_t{5, RG_F_REPLIED, 1});
    uint64_t total = 0, frames = 0, batches = 0;
    auto take = [&](const SealedBatch &b) {
        for (const SealedShard &sh : b.shard)
            for (uint32_t r = 0; r < sh.batch.rounds; r++)
                for (uint32_t g = 0; g < sh.batch.count; g++) {
                    const size_t cell = (size_t)r * sh.batch.count + g;
                    if (RG_HDR_KIND(sh.batch.head[cell].hdr) == RG_EV_NONE) continue;
                    const Origin o = sh.origin[cell];
                    if (sh.batch.abcd[cell].b != o.sequence) abort();
                    size_t blen = 0;
                    const char *kept = ing.body(b, (uint32_t)(&sh - b.shard.data()), cell, blen);
                    if ((RG_HDR_KIND(sh.batch.head[cell].hdr) == RG_EV_AE_REQ) != (kept != nullptr && blen > 40)) abort();     // an AppendEntries row has its body, no other row has one
                    got[o.conn][sh.first_gid + g].push_back(o.sequence);
                    total++;
                }
        for (const HeldRow &h : b.wide) { got[h.from.conn][h.gid].push_back(h.from.sequence); total++; }
        std::vector<std::string> out(C + 1);
        for (uint32_t k = 0; k < b.shard.size(); k++) frames += ing.emit(b, reply.data(), out, 0, (size_t)-1, NO_CONN, k);
        ing.recycle(b);
        batches++;
    };
    while (running.load() != 0) take(ing.seal());
    for (std::thread &t : th) t.join();
    for (;;) {
        const SealedBatch &b = ing.seal();
        if (b.rows == 0 && b.wide.empty()) { ing.recycle(b); break; }
        take(b);
    }
    const bool ok = total == (uint64_t)(C + 1) * N && got == sent && ing.held() == 0 && ing.refused() == 0 && index.size() == G;
    printf("ingress race ok=%d (%llu rows in %llu batches, %llu response frames)\n", (int)ok, (unsigned long long)total, (unsigned long long)batches,
           (unsigned long long)frames);
    return ok ? 0 : 1;
}
