// Ingress::drop_rows_of: after a context is erased, the rows of it that still wait outside a batch are dropped, so that its group id
// can go to another context without that context's group deciding the old one's rows. TEST INFRASTRUCTURE (tests/test_ingress_cpu.py).
// prints "drop rows ok"
#include <cstdio>
#include <string>
#include <vector>

#include "ingress.hpp"

using namespace rafting::wire;

int main()
{
    const uint32_t G = 4, R = 2;
    const KryoBodyCodec codec({{"10.3.0.1", 7301}, {"10.3.0.2", 7302}, {"10.3.0.3", 7303}});
    ContextIndex index(G);
    const std::string old_id = "ledger/old", other_id = "ledger/other", new_id = "ledger/new";
    if (!index.insert(old_id.data(), old_id.size(), 0) || !index.insert(other_id.data(), other_id.size(), 1)) return 2;
    const size_t cells = (size_t)G * R;
    std::vector<rg_ev_head_t> head[2] = {std::vector<rg_ev_head_t>(cells), std::vector<rg_ev_head_t>(cells)};
    std::vector<rg_ev_quad32_t> abcd[2] = {std::vector<rg_ev_quad32_t>(cells), std::vector<rg_ev_quad32_t>(cells)};
    std::vector<int32_t> terms[2] = {std::vector<int32_t>(256), std::vector<int32_t>(256)};
    Ingress ing(G, R, 1, codec, index, Ingress::Buffers{head[0].data(), abcd[0].data(), terms[0].data(), terms[0].size()},
                Ingress::Buffers{head[1].data(), abcd[1].data(), terms[1].data(), terms[1].size()});
    ing.set_peer(0, 0);
    int32_t seq = 0;
    auto ae = [&](const std::string &id, int64_t prev) {
        std::string stream;
        Frame f;
        f.type = ENQ; f.sequence = seq++;
        f.head = make_scope(M_APPEND_ENTRIES, id);
        Request q;
        q.term = 3; q.node = 0; q.x = prev; q.y = 3; q.leader_commit = 0;
        codec.encode_request(M_APPEND_ENTRIES, q, f.body);
        encode_frame(f, false, stream);
        return ing.feed(0, reinterpret_cast<const uint8_t *>(stream.data()), stream.size()) >= 0;
    };
    for (int k = 0; k < 6; k++) if (!ae(old_id, 10 + k)) return 3;           // six rows for a two-round batch: two placed, four held back
    if (!ae(other_id, 50)) return 3;
    const SealedBatch &b1 = ing.seal();                                      // the held rows join the backlog; the new bank takes two more of them
    const uint64_t rows1 = b1.rows;
    ing.recycle(b1);
    const uint64_t waiting = ing.held();                                     // 6 - 2 (first batch) - 2 (moved into the bank being filled) = 2
    const uint32_t gid = index.erase(old_id.data(), old_id.size());
    const size_t retired = index.retired();
    const size_t dropped = ing.drop_rows_of(gid);
    const bool dropped_ok = gid == 0 && dropped == waiting && ing.held() == 0 && ing.refused() == dropped && ing.held_on(0) == 0;
    const SealedBatch &b2 = ing.seal();                                      // (the seal after the erase: its lookups are over)
    const uint64_t rows2 = b2.rows;                                          // the two rows the old context still had in the bank
    ing.recycle(b2);
    index.reclaim(retired);
    if (!index.insert(new_id.data(), new_id.size(), 0)) return 4;            // the group id goes to another context
    if (!ae(new_id, 7)) return 3;
    const SealedBatch &b3 = ing.seal();
    const bool fresh_ok = b3.rows == 1 && RG_HDR_KIND(b3.batch.head[0].hdr) == RG_EV_AE_REQ && b3.batch.abcd[0].b == 7 && ing.held() == 0;   // round 0, group 0: its OWN row
    ing.recycle(b3);
    const bool ok = rows1 == 3 && waiting == 2 && dropped_ok && rows2 == 2 && fresh_ok;
    std::printf("%s: first batch %llu rows, %llu waiting, %zu dropped (refused %llu), second batch %llu rows, new context's row first in its group: %d\n",
                ok ? "drop rows ok" : "drop rows WRONG", (unsigned long long)rows1, (unsigned long long)waiting, dropped, (unsigned long long)ing.refused(),
                (unsigned long long)rows2, (int)fresh_ok);
    return ok ? 0 : 1;
}
