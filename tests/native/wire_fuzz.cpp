// tests/native/wire_fuzz.cpp — the wire decoders (Kryo-format and fixed-layout bodies, the frame splitter) on random bytes, mutated valid bodies and
// truncations, built with ASan + UBSan by tests/test_wire_cpu.py: nothing a peer sends may crash the host or make it read out of bounds.
// TEST INFRASTRUCTURE. usage: wire_fuzz [iterations=300000]
#include <cstdio>
#include <cstdlib>
#include <random>
#include "ingress.hpp"
#include "wire.hpp"
using namespace rafting::wire;
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300000;
    KryoBodyCodec kryo({{"127.0.0.1", 6001}, {"127.0.0.1", 6002}, {"10.0.0.77", 65535}});
    FixedBodyCodec fixed;
    std::mt19937_64 rng(12345);
    size_t ok = 0, bad = 0;
    Request scratch; Response rsp;
    for (int iter = 0; iter < iters; iter++) {
        const BodyCodec &c = (iter & 1) ? static_cast<const BodyCodec &>(kryo) : fixed;
        std::string body;
        int mode = rng() % 4;
        Method m = (Method)(rng() % 4);
        if (mode == 0) { body.resize(rng() % 200); for (auto &ch : body) ch = (char)rng(); }
        else {
            Request q; q.term = (int64_t)rng() >> (rng() % 64); q.node = rng() % 3; q.x = (int64_t)rng() >> (rng() % 64); q.y = (int64_t)(rng() % 1000);
            q.leader_commit = (int64_t)rng() >> (rng() % 64);
            q.entry_terms.assign(rng() % 6, (int64_t)(rng() % 100));
            if (rng() & 1) c.encode_request(m, q, body); else c.encode_response(Response{(int64_t)rng() >> (rng() % 64), (bool)(rng() & 1)}, body);
            if (mode == 2 && !body.empty()) body.resize(rng() % body.size());
            if (mode == 3 && !body.empty()) for (int k = 0; k < 3; k++) body[rng() % body.size()] = (char)rng();
        }
        // exact-size heap copy so that ASan sees any over-read
        std::vector<char> buf(body.begin(), body.end());
        bool a = c.decode_request(m, buf.data(), buf.size(), scratch);
        if (&c == &kryo) {                                   // the fast path must answer exactly like the general reader
            Request ref;
            const bool g = kryo.decode_request_general(m, buf.data(), buf.size(), ref);
            if (g != a || (g && (ref.term != scratch.term || ref.node != scratch.node || ref.x != scratch.x || ref.y != scratch.y ||
                                 ref.leader_commit != scratch.leader_commit || ref.entry_terms != scratch.entry_terms))) {
                printf("fast path and general reader disagree on input %d\n", iter);
                return 1;
            }
        }
        bool b = c.decode_response(buf.data(), buf.size(), rsp);
        (a || b) ? ok++ : bad++;
    }
    // frame splitter on garbage
    for (int iter = 0; iter < iters / 100; iter++) {
        FrameSplitter sp;
        std::vector<Frame> out;
        for (int k = 0; k < 20 && !sp.failed(); k++) {
            std::vector<uint8_t> chunk(rng() % 300);
            for (auto &ch : chunk) ch = (rng() % 4 == 0) ? (uint8_t)(rng() % 8) : (uint8_t)rng();
            sp.feed(chunk.data(), chunk.size(), out);
        }
    }
    // the ingress on streams of valid frames with damage: flipped bytes, cut tails, frames for unknown contexts, responses nobody waits for, values
    // beyond int32, more entries than a row may carry, more rows than rounds. Whatever arrives: no out-of-bounds access, every cell that holds an event
    // belongs to a known group and a known round, rows + held + refused account for every frame that decoded
    size_t ing_rows = 0, ing_refused = 0;
    for (int iter = 0; iter < iters / 300; iter++) {
        const uint32_t G = 1 + rng() % 40, R = 1 + rng() % 5, C = 1 + rng() % 3;
        ContextIndex index(G);
        for (uint32_t g = 0; g < G; g++) { const std::string id = "c" + std::to_string(g); index.insert(id.data(), id.size(), g); }
        const size_t cells = (size_t)G * R;
        std::vector<rg_ev_head_t> head[2] = {std::vector<rg_ev_head_t>(cells), std::vector<rg_ev_head_t>(cells)};
        std::vector<rg_ev_quad32_t> abcd[2] = {std::vector<rg_ev_quad32_t>(cells), std::vector<rg_ev_quad32_t>(cells)};
        const uint64_t cap = rng() % 64;
        std::vector<int32_t> terms[2] = {std::vector<int32_t>(cap + 1), std::vector<int32_t>(cap + 1)};     // exact sizes: ASan sees an overrun of the term array
        Ingress ing(G, R, C, kryo, index, Ingress::Buffers{head[0].data(), abcd[0].data(), terms[0].data(), cap},
                    Ingress::Buffers{head[1].data(), abcd[1].data(), terms[1].data(), cap}, 64);
        ing.retain_bodies(rng() % 2 == 0);
        for (uint32_t c = 0; c < C; c++) if (rng() % 4) ing.set_peer(c, (int32_t)(rng() % 3));
        for (uint32_t c = 0; c < C; c++) {
            std::string st;
            const int frames = (int)(rng() % 200);
            for (int k = 0; k < frames; k++) {
                Frame f;
                f.type = (rng() % 3) ? ENQ : ACK; f.sequence = (int32_t)(rng() % 100);
                const Method m = (Method)(1 + rng() % 4);
                f.head = make_scope(m, "c" + std::to_string(rng() % (G + 3)));
                if (f.type == ENQ) {
                    Request q; q.term = (rng() % 20) ? (int64_t)(rng() % 1000) : (int64_t)rng(); q.node = rng() % 3; q.x = (int64_t)(rng() % 5000); q.y = (int64_t)(rng() % 1000);
                    q.leader_commit = q.x;
                    q.entry_terms.assign((rng() % 50) ? rng() % 6 : 190 + rng() % 30, (int64_t)(rng() % 100));
                    if (rng() % 3 == 0 && !q.entry_terms.empty()) q.entry_terms.back() += 1;
                    kryo.encode_request(m, q, f.body);
                } else {
                    kryo.encode_response(Response{(int64_t)(rng() % 1000), (bool)(rng() & 1)}, f.body);
                    if (rng() % 2) ing.pending(c).put(f.sequence, m, (uint32_t)(rng() % G), Pending{(uint32_t)rng(), (int64_t)(rng() % 100), (int64_t)(rng() % 100)});
                }
                if (rng() % 40 == 0 && !f.body.empty()) f.body[rng() % f.body.size()] = (char)rng();
                encode_frame(f, false, st);
            }
            if (rng() % 5 == 0 && !st.empty()) st[rng() % st.size()] = (char)rng();          // damage to the framing itself: the connection dies there
            if (rng() % 7 == 0 && !st.empty()) st.resize(rng() % st.size());
            for (size_t at = 0; at < st.size();) {
                const size_t n = std::min<size_t>(1 + rng() % 700, st.size() - at);
                std::vector<uint8_t> piece(st.begin() + (long)at, st.begin() + (long)(at + n));
                ing.feed(c, piece.data(), piece.size());
                at += n;
                if (rng() % 20 == 0) ing.add_row(c, (uint32_t)(rng() % (G + 2)), rg_ev_head_t{RG_HDR_MAKE(rng() % 12, 0, 0, 0), 0}, (int64_t)(rng() % 100), 0, 0, 0, Origin{NO_CONN, 0});
            }
        }
        for (int round = 0; round < 400; round++) {
            const SealedBatch &b = ing.seal();
            if (b.batch.rounds > R || b.batch.count != G || b.batch.entry_count > cap) { printf("ingress: batch out of shape\n"); return 1; }
            size_t events = 0;
            std::vector<rg_reply_t> reply((size_t)b.batch.rounds * G, rg_reply_t{7, RG_F_REPLIED, 1});
            for (size_t cell = 0; cell < (size_t)b.batch.rounds * G; cell++) {
                const rg_ev_head_t h = b.batch.head[cell];
                if (RG_HDR_KIND(h.hdr) == RG_EV_NONE) continue;
                events++;
                if (RG_HDR_KIND(h.hdr) == RG_EV_AE_REQ && RG_HDR_N(h.hdr) > 0 && !(h.hdr & RG_HDR_SAME_TERM) && (uint64_t)h.aux + RG_HDR_N(h.hdr) > b.batch.entry_count) {
                    printf("ingress: a row's entry terms lie outside the term array\n"); return 1;
                }
                if (RG_HDR_N(h.hdr) > RG_MAX_AE_ENTRIES && RG_HDR_KIND(h.hdr) == RG_EV_AE_REQ) { printf("ingress: oversized row\n"); return 1; }
                size_t blen = 0;
                const char *kept = ing.body(b, 0, cell, blen);
                if (kept) { volatile char first = kept[0], last = kept[blen - 1]; (void)first; (void)last; }      // (ASan: the whole span is readable)
            }
            if (events != b.rows) { printf("ingress: %zu events, %llu counted\n", events, (unsigned long long)b.rows); return 1; }
            std::vector<std::string> out(C);
            ing.emit(b, reply.data(), out);
            ing_rows += events + b.wide.size();
            const bool last = b.rows == 0 && b.wide.empty();
            ing.recycle(b);
            if (last) break;
        }
        if (ing.held() != 0) { printf("ingress: rows still held after 400 batches\n"); return 1; }
        ing_refused += ing.refused();
    }
    printf("fuzz ok: %zu decoded, %zu refused; ingress %zu rows, %zu refused\n", ok, bad, ing_rows, ing_refused);
    return 0;
}
