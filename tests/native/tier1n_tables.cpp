// Host-compiled unit test of the pieces of the 32-bit body that are pure functions (rafting_amd/csrc/rg_tier1n.hpp, rg_step.hpp), through the
// header shim of tests/devemu: TEST INFRASTRUCTURE (tests/test_kernel_static_cpu.py builds and runs it). Checks
//   * the sign-word primitives on and around the borders of the domain they are used in;
//   * the predicate word: its two 128-entry tables reproduce expand_predicates() for every 14-bit word, and a general handler's word passes through;
//   * the class word: table entry + per-row corrections against the plain definition of every class bit, for every (kind, slot, flag) and a grid
//     of row fields in and out of the domain, for several cluster shapes.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "hip/hip_runtime.h"
#include "../../rafting_amd/csrc/rg_step.hpp"

using namespace rg;
static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { if (failures++ < 20) { std::printf("FAIL %s:%d  %s  ", __FILE__, __LINE__, #c); std::printf(__VA_ARGS__); std::printf("\n"); } } } while (0)

static void sign_words()
{
    const int32_t top = (int32_t)STATE_LIMIT + (1 << 20);        // what tier 1 can compute from in-domain inputs
    const int32_t v[] = {0, 1, 2, 5, 1000, (1 << 30) - 1, 1 << 30, (int32_t)STATE_LIMIT - 1, top};
    for (int32_t x : v) for (int32_t y : v) {
        CHECK((s_lt(x, y) < 0) == (x < y), "%d %d", x, y);
        CHECK((s_ne(x, y) < 0) == (x != y), "%d %d", x, y);
    }
    for (int32_t x : v) CHECK((s_pos(x) < 0) == (x > 0), "%d", x);
    CHECK(s_pos(-1) >= 0, "-1");
    for (int32_t slot = 0; slot < 16; slot++) {                   // NO_NODE reads as "equal" in s_ne: the one exception tier 1 relies on (leader == null)
        CHECK(s_ne(RG_NO_NODE, slot) >= 0, "%d", slot);
        CHECK(((s_ne(RG_NO_NODE, slot) | RG_NO_NODE) < 0), "%d", slot);      // ... and how votedFor == null is told apart
    }
    uint32_t acc = 0;
    const sw bits[] = {-1, 0, 5, INT32_MIN, 7, -7};
    for (sw b : bits) acc = push_bit(acc, b);
    CHECK(acc == 0b100101u, "%u", acc);
}

static void predicate_word()
{
    std::vector<uint32_t> lutm(128);
    std::vector<uint16_t> lute(128);
    for (uint32_t i = 0; i < 128; i++) { lutm[i] = expand_predicates(i); lute[i] = (uint16_t)(expand_predicates((i << 7) | 0x42u) & 0xFFFFu); }
    for (uint32_t w = 0; w < (1u << 14); w++) {
        const uint32_t main_bits = w & 127u;
        const bool fa = !(main_bits & 64u), fc = !(main_bits & 2u);
        if (fa && (w >> 7)) continue;                             // a lane is in one class: an AppendEntries row sets no election bit, nor does a client append
        if (fc && (w >> 7)) continue;
        CHECK(expand_by_table(lutm.data(), lute.data(), w) == expand_predicates(w), "%#x: %#x vs %#x", w, expand_by_table(lutm.data(), lute.data(), w), expand_predicates(w));
    }
    for (uint32_t f : {0u, 0x00210013u, 0x00FFFFFFu}) CHECK(expand_by_table(lutm.data(), lute.data(), PW_SLOW | f) == f, "%#x", f);
    CHECK(expand_predicates(0x42u) == 0u, "no class decides nothing");
    CHECK(expand_predicates(0x02u) == (RG_F_RESET_TIMER | RG_F_REPLIED | RG_F_SUCCESS), "AE that contains, nothing else");
    CHECK(expand_predicates(0x43u) == ((uint32_t)RG_DROPPED_STALE_ROLE << RG_F_STATUS_SHIFT), "drop");
}

static uint32_t plain_class_bits(const StepParams &p, uint32_t hdr, uint32_t aux, const I32x4 &q, bool &out_of_domain)
{
    const uint32_t kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), n = RG_HDR_N(hdr), flag = RG_HDR_FLAG(hdr);
    const uint32_t P = (uint32_t)p.cluster, self = (uint32_t)p.self;
    const bool same = (hdr & RG_HDR_SAME_TERM) != 0, ae = kind == RG_EV_AE_REQ, peer_ok = slot < P && slot != self;
    const bool aux_counts = (ae && same && slot < P) ||       // (the entries' term of a request tier 1 could decide: one from a slot outside the cluster is BAD_EVENT either way)
                             kind == RG_EV_AE_ACK || kind == RG_EV_IS_ACK || kind == RG_EV_RV_REPLY || kind == RG_EV_PV_REPLY || kind == RG_EV_TIMEOUT;
    auto bad = [](int32_t x) { return (uint32_t)x >= EV_LIMIT; };
    out_of_domain = bad(q.x) || bad(q.y) || bad(q.z) || bad(q.w) || (aux_counts && aux >= EV_LIMIT);
    if (out_of_domain) return flag ? 1u << CW_FLAG : 0u;     // (no class: the flag bit alone decides nothing; the row sends the workgroup to the 64-bit body)
    uint32_t c = 0;
    if (kind == RG_EV_AE_ACK && peer_ok) c |= 1u << CW_ACK;
    if (ae && slot < P && q.z != 0 && n <= RG_MAX_AE_ENTRIES && (n == 0 || same)) c |= 1u << CW_AE;
    if (kind == RG_EV_CLIENT_APPEND && n >= 1) c |= 1u << CW_CLIENT;
    if ((kind == RG_EV_AE_ACK || kind == RG_EV_IS_ACK) && peer_ok) c |= 1u << CW_ACKANY;
    if (flag) c |= 1u << CW_FLAG;
    if (kind >= RG_EV_RV_REQ && kind <= RG_EV_TIMEOUT) c |= 1u << CW_ELK;
    if ((kind == RG_EV_RV_REPLY || kind == RG_EV_PV_REPLY) && peer_ok) c |= 1u << CW_VR;
    if (kind == RG_EV_PV_REPLY) c |= 1u << CW_PV;
    if (kind == RG_EV_TIMEOUT) c |= 1u << CW_TO;
    if ((kind == RG_EV_RV_REQ || kind == RG_EV_PV_REQ) && slot < P) c |= 1u << CW_VQ;
    if (kind == RG_EV_PV_REQ) c |= 1u << CW_PVQ;
    if (kind == RG_EV_NONE) c |= 1u << CW_NONE;
    return c;
}

static void class_word_table()
{
    const int32_t fields[] = {0, 1, 7, 200, (1 << 30) - 1, 1 << 30, -1};
    const uint32_t auxs[] = {0u, 3u, (1u << 30) - 1u, 1u << 30, 0xFFFFFFFFu};
    const uint32_t ns[] = {0u, 1u, 2u, 200u, 201u, RG_MAX_ENTRIES};
    long rows = 0;
    for (int cluster = 2; cluster <= RG_MAX_COMPACT_CLUSTER; cluster++) for (int self = 0; self < cluster; self += (cluster > 3 ? 2 : 1)) {      // (the class word belongs to the compact-row kernels: clusters of up to 7 nodes)
        StepParams p{};
        p.cluster = cluster; p.self = self;
        std::vector<uint32_t> lutc(256);
        for (uint32_t i = 0; i < 256; i++) lutc[i] = class_entry(p, i);
        for (uint32_t kind = 0; kind < 16; kind++) for (uint32_t slot = 0; slot < 16; slot++) for (uint32_t flag = 0; flag < 2; flag++)
            for (uint32_t same = 0; same < 2; same++) for (uint32_t n : ns) for (uint32_t aux : auxs) for (int32_t c : fields) for (int32_t other : {5, 1 << 30}) {
                const uint32_t hdr = kind | (slot << 4) | (flag << 8) | (same ? RG_HDR_SAME_TERM : 0u) | (n << 12);
                Row32 x;
                x.h = U32x2{hdr, aux};
                x.q = I32x4{7, other, c, 9};
                const I32x4 got = class_word(lutc.data(), x);
                bool ood;
                const uint32_t want = plain_class_bits(p, hdr, aux, x.q, ood);
                const uint32_t peer_ok = slot < (uint32_t)cluster && slot != (uint32_t)self;
                const uint32_t j = (kind == RG_EV_AE_ACK && peer_ok) ? (slot < (uint32_t)self ? slot : slot - 1u) : 0u;
                rows++;
                CHECK(((uint32_t)got.x & 0xFFF80000u) == want, "cluster %d self %d hdr %#x aux %#x c %d other %d: class bits %#x, want %#x", cluster, self, hdr, aux, c, other, (uint32_t)got.x & 0xFFF80000u, want);
                CHECK(((uint32_t)got.x & (1u << CW_AUXC)) == 0u, "table-only bit leaked");
                CHECK((((uint32_t)got.x >> 10) & 7u) == j && (((uint32_t)got.x >> 5) & 15u) == slot && ((uint32_t)got.x & 31u) == 31u - j, "hdr %#x: j / slot / shift fields", hdr);
                CHECK((uint32_t)got.y == aux && got.z == (int32_t)n, "aux / n");
                CHECK(RG_HDR_KIND((uint32_t)got.w) == (ood ? KIND_OUT_OF_DOMAIN : kind) && (ood || (uint32_t)got.w == hdr), "header handed to the general handlers: hdr %#x aux %#x c %d other %d got %#x ood %d", hdr, aux, c, other, (uint32_t)got.w, (int)ood);
            }
    }
    std::printf("class word: %ld rows checked\n", rows);
}

int main()
{
    sign_words();
    predicate_word();
    class_word_table();
    if (failures) { std::printf("%d failure(s)\n", failures); return 1; }
    std::printf("tier1n tables ok\n");
    return 0;
}
