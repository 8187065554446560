// tests/native/wire_fuzz.cpp — the wire decoders (Kryo-format and fixed-layout bodies, the frame splitter) on random bytes, mutated valid bodies and
// truncations, built with ASan + UBSan by tests/test_wire_cpu.py: nothing a peer sends may crash the host or make it read out of bounds.
// TEST INFRASTRUCTURE. usage: wire_fuzz [iterations=300000]
#include <cstdio>
#include <cstdlib>
#include <random>
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
    printf("fuzz ok: %zu decoded, %zu refused\n", ok, bad);
    return 0;
}
