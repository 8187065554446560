// wire_bench.cpp — how fast does ONE host core turn the reference's wire frames into event rows? (N2, CPU only, no GPU involved)
// Encodes `frames` AppendEntries requests / responses of the fixed-layout body stand-in into one byte stream, then times
//   (1) FrameSplitter::feed over 64 KiB reads of that stream, (2) the same + RowWriter::add into rg_batch_t columns.
// usage: build/wire_bench [frames=2000000]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <algorithm>

#include "wire.hpp"

using namespace rafting::wire;

int main(int argc, char **argv)
{
    const size_t frames = argc > 1 ? (size_t)atoll(argv[1]) : 2000000;
    // bodies in the reference's Kryo format by default; `wire_bench <frames> fixed` takes the fixed-layout test codec
    const FixedBodyCodec fixed;
    const KryoBodyCodec kryo({{"127.0.0.1", 6001}, {"127.0.0.1", 6002}, {"127.0.0.1", 6003}});
    const BodyCodec &codec = (argc > 2 && std::string(argv[2]) == "fixed") ? static_cast<const BodyCodec &>(fixed) : kryo;
    std::string stream;
    stream.reserve(frames * 110);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    size_t n_terms = 0;
    for (size_t i = 0; i < frames; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        Frame f;
        f.sequence = (int32_t)i;
        std::string ctx = "ctx" + std::to_string(x % 4096);
        if (i % 5 != 4) {                                   // 80 % requests with 0..4 entries, 20 % responses (config 3's mix)
            f.type = ENQ;
            f.head = make_scope(M_APPEND_ENTRIES, ctx);
            Request q;
            q.term = 7; q.node = 1; q.x = (int64_t)(x >> 40); q.y = 7; q.leader_commit = q.x;
            q.entry_terms.assign((x >> 8) % 5 == 3 ? 4 : (x >> 8) % 3, 7);
            n_terms += q.entry_terms.size();
            codec.encode_request(M_APPEND_ENTRIES, q, f.body);
        } else {
            f.type = ACK;
            f.head = make_scope(M_APPEND_ENTRIES, ctx);
            codec.encode_response(Response{7, true}, f.body);
        }
        encode_frame(f, false, stream);
    }
    const size_t CH = 64 * 1024;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };

    // (1) split only
    {
        FrameSplitter sp;
        size_t got = 0, bytes = 0;
        const std::function<void(const FrameView &)> sink = [&](const FrameView &f) { bytes += f.head_len + f.body_len; };
        const auto t0 = now();
        for (size_t off = 0; off < stream.size(); off += CH)
            got += sp.feed_views(reinterpret_cast<const uint8_t *>(stream.data()) + off, std::min(CH, stream.size() - off), sink);
        const double s = secs(t0, now());
        if (got != frames || sp.failed() || bytes == 0) { fprintf(stderr, "split: %zu of %zu frames, failed=%d\n", got, frames, (int)sp.failed()); return 1; }
        printf("split      %zu frames, %.1f MB in %.3f s: %.2f M frames/s, %.2f GB/s\n", frames, stream.size() / 1e6, s, frames / s / 1e6, stream.size() / s / 1e9);
    }
    // (2) split + rows
    {
        std::vector<rg_ev_head_t> head(frames);
        std::vector<rg_ev_pair_t> ab(frames), cd(frames);
        std::vector<uint32_t> gid(frames);
        std::vector<int64_t> terms(n_terms + 8);
        RowWriter rw(head.data(), ab.data(), cd.data(), gid.data(), terms.data(), frames, terms.size());
        auto gid_of = [](const std::string &c, uint32_t &g) { g = (uint32_t)atoi(c.c_str() + 3); return true; };
        auto pending_of = [](const std::string &, int32_t, Pending &p) { p.role_epoch = 1; p.epoch_at_send = 0; p.last_index_sent = 5; return true; };
        FrameSplitter sp;
        std::vector<Frame> out;
        out.reserve(4096);
        const auto t0 = now();
        const std::function<bool(const std::string &, uint32_t &)> gid_fn = gid_of;
        const std::function<bool(const std::string &, int32_t, Pending &)> pend_fn = pending_of;
        const std::function<void(const FrameView &)> sink = [&](const FrameView &f) { rw.add(f, 1, codec, gid_fn, pend_fn); };
        for (size_t off = 0; off < stream.size(); off += CH)
            sp.feed_views(reinterpret_cast<const uint8_t *>(stream.data()) + off, std::min(CH, stream.size() - off), sink);
        const double s = secs(t0, now());
        if (rw.rows() != frames) { fprintf(stderr, "rows: %zu of %zu\n", rw.rows(), frames); return 1; }
        printf("split+rows %zu rows (%zu entry terms) in %.3f s: %.2f M rows/s on one core\n", rw.rows(), rw.terms(), s, frames / s / 1e6);
    }
    return 0;
}
