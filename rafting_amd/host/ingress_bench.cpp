// ingress_bench.cpp — how fast do K host threads turn the wire streams of C connections into ONE multi-round compact batch, and the batch's
// replies back into response frames? (N2 + the host half of a8; CPU only, no GPU involved.)
// Config 3's mix on `groups` contexts: 80 % follower view (AppendEntries requests with 0 / 1 / 2 / 4 entries of one term from the leader's
// connection), 20 % leader view (acks from the followers' connections), `rounds` rows per group, Kryo-format bodies, 64 KiB reads.
// usage: build/ingress_bench [groups=65536] [rounds=16] [conns=8] [threads=1,2,4,8]
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ingress.hpp"

using namespace rafting::wire;

// thread t on CPU t: this box's scheduler leaves freshly started threads on one core for hundreds of milliseconds (a plain spin loop on 2 threads takes
// as long as on 1 until the balancer wakes up), which would be measured as "no scaling"
static void pin(int t)
{
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(t % (int)std::thread::hardware_concurrency(), &set);
    pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536, R = argc > 2 ? (uint32_t)atoi(argv[2]) : 16, C = argc > 3 ? (uint32_t)atoi(argv[3]) : 8;
    std::vector<int> ks;
    for (const char *p = argc > 4 ? argv[4] : "1,2,4,8"; *p;) { ks.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p) p++; }
    const KryoBodyCodec codec({{"10.0.0.1", 7001}, {"10.0.0.2", 7002}, {"10.0.0.3", 7003}, {"10.0.0.4", 7004}, {"10.0.0.5", 7005}});
    ContextIndex index(G);
    std::vector<std::string> ids(G);
    for (uint32_t g = 0; g < G; g++) {
        char b[48];
        snprintf(b, sizeof b, "orders/partition-%05u", g);
        ids[g] = b;
        if (!index.insert(ids[g].data(), ids[g].size(), g)) return 2;
    }
    // the streams: a group's rows all travel on one connection (its leader's, or — leader view — spread over its followers')
    std::vector<std::string> stream(C);
    struct Put { uint32_t conn; int32_t seq; uint32_t gid; Pending p; };
    std::vector<Put> puts;
    std::vector<int32_t> seq(C, 0);
    uint64_t x = 0x9E3779B97F4A7C15ull, rows = 0;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (uint32_t r = 0; r < R; r++)
        for (uint32_t g = 0; g < G; g++) {
            const uint64_t h = (uint64_t)g * 0x9E3779B97F4A7C15ull;
            const bool leader_view = (h >> 32) % 5 == 0;
            Frame f;
            f.head = make_scope(M_APPEND_ENTRIES, ids[g]);
            uint32_t conn;
            if (!leader_view) {
                conn = (uint32_t)((h >> 40) % C);
                Request q;
                q.term = 7; q.node = 1 + (int32_t)(conn % 4); q.x = 1000 + r * 4; q.y = 7; q.leader_commit = q.x;
                const uint64_t e = rnd() % 4;
                q.entry_terms.assign(e == 3 ? 4 : e, 7);
                f.type = ENQ;
                codec.encode_request(M_APPEND_ENTRIES, q, f.body);
            } else {
                conn = (uint32_t)(rnd() % C);
                f.type = ACK;
                codec.encode_response(Response{7, true}, f.body);
                puts.push_back(Put{conn, seq[conn], g, Pending{3, 0, (int64_t)(1000 + r)}});
            }
            f.sequence = seq[conn]++;
            encode_frame(f, false, stream[conn]);
            rows++;
        }
    size_t bytes = 0;
    for (const std::string &s : stream) bytes += s.size();
    printf("%u groups x %u rounds = %llu rows on %u connections, %.1f MB of frames (%.0f B per row)\n", G, R, (unsigned long long)rows, C, bytes / 1e6,
           (double)bytes / rows);

    const size_t cells = (size_t)G * R;
    std::vector<rg_ev_head_t> head[2] = {std::vector<rg_ev_head_t>(cells), std::vector<rg_ev_head_t>(cells)};
    std::vector<rg_ev_quad32_t> abcd[2] = {std::vector<rg_ev_quad32_t>(cells), std::vector<rg_ev_quad32_t>(cells)};
    std::vector<int32_t> terms[2] = {std::vector<int32_t>(1 << 20), std::vector<int32_t>(1 << 20)};
    std::vector<rg_reply_t> reply(cells);
    for (size_t i = 0; i < cells; i++) reply[i] = rg_reply_t{7, RG_F_REPLIED | RG_F_SUCCESS | RG_F_RESET_TIMER, 1};
    const size_t CH = 64 * 1024;

    for (int K : ks) {
        Ingress ing(G, R, C, codec, index, Ingress::Buffers{head[0].data(), abcd[0].data(), terms[0].data(), terms[0].size()},
                    Ingress::Buffers{head[1].data(), abcd[1].data(), terms[1].data(), terms[1].size()}, 1u << 22);   // (every request of the run is 'in flight' at once here)
        for (uint32_t c = 0; c < C; c++) ing.set_peer(c, 1 + (int32_t)(c % 4));
        for (const Put &p : puts) ing.pending(p.conn).put(p.seq, M_APPEND_ENTRIES, p.gid, p.p);
        // (1) decode + place: thread t owns connections t, t + K, ...
        const double t0 = now_s();
        {
            std::vector<std::thread> th;
            for (int t = 0; t < K; t++)
                th.emplace_back([&, t] {
                    pin(t);
                    for (uint32_t c = (uint32_t)t; c < C; c += (uint32_t)K)
                        for (size_t off = 0; off < stream[c].size(); off += CH)
                            ing.feed(c, reinterpret_cast<const uint8_t *>(stream[c].data()) + off, std::min(CH, stream[c].size() - off));
                });
            for (std::thread &t : th) t.join();
        }
        const double t1 = now_s();
        const SealedBatch &b = ing.seal();
        const double t2 = now_s();
        if (b.rows != rows || ing.refused() || ing.held() || b.batch.rounds != R) {
            fprintf(stderr, "K=%d: %llu of %llu rows, %llu refused, %llu held, %u rounds\n", K, (unsigned long long)b.rows, (unsigned long long)rows,
                    (unsigned long long)ing.refused(), (unsigned long long)ing.held(), b.batch.rounds);
            return 1;
        }
        // (2) replies -> response frames: thread t takes a slice of the cells
        std::vector<size_t> made(K, 0), out_bytes(K, 0);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < K; t++)
                th.emplace_back([&, t] {
                    pin(t);
                    std::vector<std::string> out(C);
                    made[t] = ing.emit(b, reply.data(), out, cells * t / K, cells * (t + 1) / K);
                    for (const std::string &s : out) out_bytes[t] += s.size();
                });
            for (std::thread &t : th) t.join();
        }
        const double t3 = now_s();
        ing.recycle(b);
        const double t4 = now_s();
        size_t frames = 0, ob = 0;
        for (int t = 0; t < K; t++) { frames += made[t]; ob += out_bytes[t]; }
        printf("threads %d: frames -> batch %.3f s = %.2f M rows/s (%.2f GB/s of frames; seal %.1f ms) | replies -> %zu frames %.3f s = %.2f M/s (%.0f MB) | "
               "recycle %.1f ms\n", K, t1 - t0, rows / (t1 - t0) / 1e6, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, frames, t3 - t2, frames / (t3 - t2) / 1e6,
               ob / 1e6, (t4 - t3) * 1e3);
    }
    // (3) both directions at once, as a deployment runs them: K reader threads keep feeding while ONE flusher seals batches of at most RB
    // rounds, "decides" them (every request answered: the table is not what this measures), emits the responses itself and recycles — the
    // readers fill the other bank meanwhile. End-to-end rows/s of the host side alone.
    const uint32_t RB = R >= 4 ? R / 4 : 1;
    for (int K : ks) {
        if ((unsigned)K + 1 > std::thread::hardware_concurrency()) continue;               // the flusher wants a core of its own
        Ingress ing(G, RB, C, codec, index, Ingress::Buffers{head[0].data(), abcd[0].data(), terms[0].data(), terms[0].size()},
                    Ingress::Buffers{head[1].data(), abcd[1].data(), terms[1].data(), terms[1].size()}, 1u << 22);
        for (uint32_t c = 0; c < C; c++) ing.set_peer(c, 1 + (int32_t)(c % 4));
        for (const Put &p : puts) ing.pending(p.conn).put(p.seq, M_APPEND_ENTRIES, p.gid, p.p);
        std::atomic<int> reading{K};
        const double t0 = now_s();
        std::vector<std::thread> th;
        for (int t = 0; t < K; t++)
            th.emplace_back([&, t] {
                pin(t);
                std::vector<size_t> at(C, 0);
                for (bool more = true; more;) {
                    more = false;
                    for (uint32_t c = (uint32_t)t; c < C; c += (uint32_t)K) {
                        if (at[c] >= stream[c].size()) continue;
                        const size_t n = std::min(CH, stream[c].size() - at[c]);
                        ing.feed(c, reinterpret_cast<const uint8_t *>(stream[c].data()) + at[c], n);
                        at[c] += n;
                        more = true;
                        while (ing.held_on(c) > 100000) std::this_thread::yield();         // the flusher is behind: stop reading this socket for a moment
                    }
                }
                reading--;
            });
        pin(K);
        uint64_t done_rows = 0, frames = 0, batches = 0;
        std::vector<std::string> out(C);
        for (;;) {
            const bool finished = reading.load() == 0;
            const SealedBatch &b = ing.seal();
            if (b.rows == 0) {
                ing.recycle(b);
                if (finished && ing.held() == 0) break;
                std::this_thread::yield();
                continue;
            }
            const size_t n = (size_t)b.batch.rounds * G;
            const unsigned helpers = b.rows > 100000 ? (unsigned)std::max(1, K / 2) : 0;      // a big batch: emitter threads share its cells with the flusher
            std::vector<size_t> made(helpers, 0);
            std::vector<std::thread> eth;
            for (unsigned e = 0; e < helpers; e++)
                eth.emplace_back([&, e] { std::vector<std::string> o(C); made[e] = ing.emit(b, reply.data(), o, n * (e + 1) / (helpers + 1), n * (e + 2) / (helpers + 1)); });
            for (std::string &o : out) o.clear();
            frames += ing.emit(b, reply.data(), out, 0, n / (helpers + 1));
            for (std::thread &t : eth) t.join();
            for (size_t m : made) frames += m;
            done_rows += b.rows;
            batches++;
            ing.recycle(b);
        }
        for (std::thread &t : th) t.join();
        const double s = now_s() - t0;
        if (done_rows != rows || ing.refused()) { fprintf(stderr, "steady K=%d: %llu of %llu rows\n", K, (unsigned long long)done_rows, (unsigned long long)rows); return 1; }
        printf("readers %d + 1 flusher (+ readers / 2 emitter threads on batches beyond 100 000 rows; <= %u rounds per batch): %llu rows in and %llu response frames out in %.3f s = %.2f M rows/s end to end, %llu batches\n", K, RB,
               (unsigned long long)done_rows, (unsigned long long)frames, s, rows / s / 1e6, (unsigned long long)batches);
    }
    return 0;
}
