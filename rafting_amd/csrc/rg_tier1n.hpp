// rg_tier1n.hpp — tier 1 of the 32-bit body (step32_kernel's deciding wavefront), written in SIGN WORDS. Included by rg_step.hpp only.
//
// Why a second statement of tier 1. A round costs (vector instructions of the wavefronts that share a SIMD) x 4 cycles plus whatever latency
// nothing hides, and profiles/r04b_issue_bench.txt prices the idiom the compiler makes of `bool` predicates — v_cmp into an SGPR pair,
// s_and / s_or on the pairs, v_cndmask on the result — at 9.5 cycles per instruction when the three depend on each other (the scalar unit
// waits for the vector pipeline and back), against 4.4 for a chain that stays in the vector unit. rg_device.hpp's tier1<> spends ~100 v_cmp +
// ~100 s_and/s_or + ~110 v_cndmask per round that way. Here a predicate is a 32-bit WORD whose SIGN BIT is the truth value ("sign word",
// `sw`): an order test is one subtraction (`x - y` is negative iff x < y — every value of this body is below 2^30 + 2^29, so the difference
// cannot wrap), an equality test is one v_xad_u32 ((x ^ y) + 0x7fffffff is negative iff x != y), a header bit is one shift, and ANY boolean
// function of three of them is ONE v_bitop3_b32 — the compiler forms it from plain `&`, `|`, `~` on the words; only the sign bits mean
// anything, the low 31 bits are noise. A word becomes a lane mask (one v_cmp against zero) only where a value is selected.
//
// The row classes and their preconditions are those of tier1<> (rg_device.hpp), statement for statement; each is a strict special case of
// Stepper::run, and a row that misses a precondition is left untouched for it. What is new besides the encoding:
//   * the state-only part of every precondition ("a Follower that has a log above its epoch", "a prepared Leader", ...) is kept in three cached
//     words (GroupN::stf_n / stl_n / stc_n), refreshed where a row can change them — the rare blocks, the election block, the general handlers;
//   * the state-independent part comes from the I/O wavefront as ONE class word per row (CW_* below) beside {aux, n, header};
//   * the outcome flags are not assembled here: the truth values they are made of are shifted into a PREDICATE WORD (one v_alignbit each)
//     that the I/O wavefront expands through two LDS tables while it stores the row (expand_predicates, expand_by_table in rg_step.hpp).
// Reference code decided here (paths under /root/reference/src/main/java/io/lubricant/consensus/raft/): member/Follower.java:35-88 (AppendEntries),
// member/Leader.java:218-237 + member/Leadership.java:75-130 + member/Leader.java:247-280 (ack, majorIndices, tryCommit), member/Leader.java:128-140
// (client append), member/Candidate.java:121-134 / member/Follower.java:258-270 (vote replies), member/Follower.java:156-168 / Candidate.java:82-88 /
// Leader.java:120-126 (timeouts), member/Follower.java:91-127,193-207 (vote requests), context/RaftRoutine.java:140-216 (conversions).
#pragma once
#include "rg_device.hpp"

namespace rg {

typedef int32_t sw;                 // sign word: bit 31 is the truth value, bits 30..0 are noise

__device__ __forceinline__ sw s_lt(int32_t x, int32_t y) { return (int32_t)((uint32_t)x - (uint32_t)y); }                  // x < y   (|x - y| < 2^31)
__device__ __forceinline__ sw s_ne(int32_t x, int32_t y) { return (int32_t)(((uint32_t)x ^ (uint32_t)y) + 0x7FFFFFFFu); }   // x != y  (x ^ y < 2^31: operands that differ in bit 31 read as EQUAL)
__device__ __forceinline__ sw s_pos(int32_t x) { return (int32_t)(0u - (uint32_t)x); }                                      // x > 0   (x > -2^31)
__device__ __forceinline__ uint32_t push_bit(uint32_t acc, sw w) { return (acc << 1) | ((uint32_t)w >> 31); }              // v_alignbit_b32

// ---- the class word of a row (made by the I/O wavefront: class_word() in rg_step.hpp) --------------------------------------
// bit 31 .. 19: one bit per state-independent fact; 12..10: follower index j of an ack's responder (0 for every other row); 8..5: slot;
// 4..0: 31 - j (the shift that brings bit j of `pending` to the sign position)
constexpr int CW_ACK = 31;          // AE_ACK from a remote peer
constexpr int CW_AE = 30;           // HDR_AE_OK
constexpr int CW_CLIENT = 29;       // CLIENT_APPEND with n >= 1
constexpr int CW_ACKANY = 28;       // AE_ACK / IS_ACK from a remote peer
constexpr int CW_FLAG = 27;         // the header's flag bit
constexpr int CW_ELK = 26;          // RV_REQ .. TIMEOUT
constexpr int CW_VR = 25;           // RV_REPLY / PV_REPLY from a remote peer
constexpr int CW_PV = 24;           // PV_REPLY
constexpr int CW_TO = 23;           // TIMEOUT
constexpr int CW_VQ = 22;           // RV_REQ / PV_REQ with slot < cluster
constexpr int CW_PVQ = 21;          // PV_REQ
constexpr int CW_NONE = 19;         // row not addressed this round
__device__ __forceinline__ sw cw_bit(int32_t cw, int bit) { return (int32_t)((uint32_t)cw << (31 - bit)); }

// ---- the predicate word of a decided row (expanded by the I/O wavefront: expand_predicates() below) -------------------------
// main block, bits 6..0 (pushed in this order, so the first is the highest): fa_n, x_ct, conv, append, commit, fc_n, drop
// election block, bits 13..7 (same rule): vq, vq_success, reset, conv, to_lead, to_pre, to_candidate
constexpr uint32_t PW_SLOW = 1u << 31;     // not a predicate word: bits 23..0 are the flags | status << 16 of a general handler
__device__ __forceinline__ uint32_t expand_predicates(uint32_t w)
{
    auto m = [&](int bit) { return (uint32_t)((int32_t)(w << (31 - bit)) >> 31); };     // v_bfe_i32: 0 / ~0
    const uint32_t fa = ~m(6), contains = ~m(5), conv = m(4) | m(10), app = m(3), commit = m(2), hb = ~m(1) | m(9), drop = m(0);
    const uint32_t vq = m(13), succ = (fa & contains) | m(12), reset = fa | m(11) | conv, to_pre = m(8), cand = m(7);
    const uint32_t fast = (succ & RG_F_SUCCESS) | ((fa | vq) & RG_F_REPLIED) | (conv & (RG_F_PERSIST | RG_F_ROLE_CHANGED)) | (reset & RG_F_RESET_TIMER) |
                          (commit & RG_F_COMMIT) | (app & RG_F_LOG_APPEND) | (hb & (RG_EMIT_HEARTBEAT << RG_F_EMIT_SHIFT)) |
                          (to_pre & (RG_EMIT_PREVOTE << RG_F_EMIT_SHIFT)) | (cand & (RG_EMIT_REQVOTE << RG_F_EMIT_SHIFT)) |
                          (drop & ((uint32_t)RG_DROPPED_STALE_ROLE << RG_F_STATUS_SHIFT));
    return (w & PW_SLOW) ? (w & 0x00FFFFFFu) : fast;
}

// ---- the group image of the 32-bit body ------------------------------------------------------------------------------------------
// GroupT's fields in 32 bits, the four booleans as sign words (0 / -1 wherever this file writes them) and the cached precondition words.
struct GroupN {
    int32_t term, commit, epoch_index, epoch_term, first, last, elected_term;
    int32_t s0, s1, s2, s3, t0, t1, t2, t3, lt, top;
    int32_t voted_for, leader, votes, role, rc;
    uint32_t role_epoch, elected_epoch, pending;
    sw td, prepared, log_dirty, peers_dirty;
    sw nallow;                      // tier 1 is off for this lane: the launch runs the general handlers only, or the group is blocked after a NEED_HOST
    sw stf_n, stl_n, stc_n;         // NOT (allowed & ...): a Follower with a log above its epoch that is not `prepared` / a prepared Leader / a Leader with a log

    __device__ __forceinline__ void recache()
    {
        const int32_t r = role, n = rc, l = last, e = epoch_index;
        const sw na = nallow, pr = prepared;
        const sw not_l = s_ne(r, RG_LEADER), no_log = s_lt(n, 1);
        stf_n = na | s_pos(r) | pr | no_log | ~s_lt(e, l);
        stl_n = na | not_l | ~pr;
        stc_n = na | not_l | no_log;
    }
    // GroupT::push (db.put(last+1, t)) for a log that has entries, same statements — except that `last` is the caller's (tier 1 has set it to
    // the index of the row's last entry by the time it comes here)
    __device__ __forceinline__ void push_run(int32_t index, int32_t t)
    {
        int n = rc;
        int32_t x0 = s0, x1 = s1, x2 = s2, x3 = s3, y0 = t0, y1 = t1, y2 = t2, y3 = t3;
        const int32_t f = first, cur_lt = lt, cur_top = top;
        const bool empty = n == 0;
        const bool newrun = empty | (cur_lt != t);
        const bool shift = newrun & (n == K);
        x0 = shift ? x1 : x0; y0 = shift ? y1 : y0;
        x1 = shift ? x2 : x1; y1 = shift ? y2 : y1;
        x2 = shift ? x3 : x2; y2 = shift ? y3 : y2;
        n = shift ? K - 1 : n;
        const bool w0 = newrun & (n == 0), w1 = newrun & (n == 1), w2 = newrun & (n == 2), w3 = newrun & (n == 3);
        x0 = w0 ? index : x0; y0 = w0 ? t : y0;
        x1 = w1 ? index : x1; y1 = w1 ? t : y1;
        x2 = w2 ? index : x2; y2 = w2 ? t : y2;
        x3 = w3 ? index : x3; y3 = w3 ? t : y3;
        n += newrun ? 1 : 0;
        s0 = x0; s1 = x1; s2 = x2; s3 = x3; t0 = y0; t1 = y1; t2 = y2; t3 = y3;
        lt = t; top = newrun ? index : cur_top;
        rc = n;
        first = empty ? index : f;
        log_dirty = -1;
    }
};

__device__ __forceinline__ Group widen(const GroupN &n)
{
    Group g;
    g.term = n.term; g.commit = n.commit; g.epoch_index = n.epoch_index; g.epoch_term = n.epoch_term; g.first = n.first; g.last = n.last;
    g.elected_term = n.elected_term;
    g.s0 = n.s0; g.s1 = n.s1; g.s2 = n.s2; g.s3 = n.s3; g.t0 = n.t0; g.t1 = n.t1; g.t2 = n.t2; g.t3 = n.t3; g.lt = n.lt; g.top = n.top;
    g.voted_for = n.voted_for; g.leader = n.leader; g.votes = n.votes; g.role = n.role; g.rc = n.rc;
    g.role_epoch = n.role_epoch; g.elected_epoch = n.elected_epoch; g.pending = n.pending;
    g.td = n.td < 0; g.prepared = n.prepared < 0; g.log_dirty = n.log_dirty < 0; g.peers_dirty = n.peers_dirty < 0;
    return g;
}
// (the caller sets nallow and calls recache())
__device__ __forceinline__ void narrow_into(GroupN &n, const Group &g)
{
    n.term = (int32_t)g.term; n.commit = (int32_t)g.commit; n.epoch_index = (int32_t)g.epoch_index; n.epoch_term = (int32_t)g.epoch_term;
    n.first = (int32_t)g.first; n.last = (int32_t)g.last; n.elected_term = (int32_t)g.elected_term;
    n.s0 = (int32_t)g.s0; n.s1 = (int32_t)g.s1; n.s2 = (int32_t)g.s2; n.s3 = (int32_t)g.s3;
    n.t0 = (int32_t)g.t0; n.t1 = (int32_t)g.t1; n.t2 = (int32_t)g.t2; n.t3 = (int32_t)g.t3; n.lt = (int32_t)g.lt; n.top = (int32_t)g.top;
    n.voted_for = g.voted_for; n.leader = g.leader; n.votes = g.votes; n.role = g.role; n.rc = g.rc;
    n.role_epoch = g.role_epoch; n.elected_epoch = g.elected_epoch; n.pending = g.pending;
    n.td = g.td ? -1 : 0; n.prepared = g.prepared ? -1 : 0; n.log_dirty = g.log_dirty ? -1 : 0; n.peers_dirty = g.peers_dirty ? -1 : 0;
}
// The small fields this file compares with sign words have a domain of their own (beside fits32 for the terms and indices): epochs and the
// vote count in [0, 2^30), node ids in [-1, 2^30) — s_ne() needs operands that agree in bit 31, and NO_NODE is the one exception it is used with.
__device__ __forceinline__ bool small_fields_fit(const Group &g)
{
    const uint32_t w = g.role_epoch | g.elected_epoch | (uint32_t)g.votes | (uint32_t)(g.leader + 1) | (uint32_t)(g.voted_for + 1);
    return (w < EV_LIMIT) & ((uint32_t)g.role <= (uint32_t)RG_LEADER);
}

// what a decided row hands to the I/O wavefront
struct OutN {
    uint32_t pw;                    // predicate word, or PW_SLOW | flags | status << 16
    int32_t resp;                   // RaftResponse.term (read iff REPLIED)
    int32_t log_from;               // (read iff LOG_APPEND / LOG_TRUNC)
};

// MUST be called by every lane of the wavefront (converged code). Returns a sign word: the row was decided here.
template <int F>
__device__ __forceinline__ sw tier1n(const StepParams &p, GroupN &g, PeersNarrow<F> &pe, OutN &out, int32_t cw, uint32_t aux_u, int32_t n,
                                     int32_t a, int32_t b, int32_t c, int32_t d)
{
    const int32_t aux = (int32_t)aux_u;
    const int32_t term = g.term, last = g.last, commit = g.commit, epoch = g.epoch_index, lt = g.lt, top = g.top;
    const int32_t role = g.role, rc = g.rc, leader = g.leader, votes = g.votes, voted = g.voted_for;
    const int32_t repoch = (int32_t)g.role_epoch;
    const sw td = g.td, prep = g.prepared, nallow = g.nallow, stl_n = g.stl_n;
    const int32_t slot = (int32_t)(((uint32_t)cw >> 5) & 15u);
    const sw flg = cw_bit(cw, CW_FLAG);
    const sw x_ta = s_lt(term, a);                                  // a > currentTerm
    const sw x_aux = s_ne(aux, repoch);                             // the row names another participant (AsyncHead aborted)
    const sw x_tl = s_ne(lt, term);                                 // the tail entry is not of currentTerm
    // The ack block's LDS reads are issued first and the AppendEntries block — which needs nothing from LDS — is placed between them and
    // their use: left to itself the scheduler opens the round with the ack block and a wait for the reads.
    const uint32_t j = ((uint32_t)cw >> 10) & 7u;                   // follower index of an ack's responder, 0 for any other row
    const I32x4 st = pe.rec[j * BLOCK];
    int32_t m[F];
    pe.load_matches(m);
    __builtin_amdgcn_sched_barrier(0);

    // ---- AppendEntries request at a follower (member/Follower.java:35-88) ---------------------------------------------------------
    const sw x_ct = s_ne(c, lt);                                    // NOT contains: prevLogTerm != term of the tail
    const sw rf = x_ta | td;                                        // refresh: switchTo(Follower, term, lastCandidate)
    const bool contains = x_ct >= 0;
    const int32_t n_eff = contains ? n : 0;
    const int32_t ae_last = last + n_eff;                           // (b == last wherever this is used)
    const int32_t ae_x = vmin<int32_t>(d, ae_last);
    const sw x_de = s_lt(epoch, d);                                 // leaderCommit > epoch.index
    // NOT: an allowed row at such a Follower | term >= currentTerm | the known leader, or none, or a refresh | prevLogIndex == last |
    //      no commit roll-back
    const sw fa_n = (~cw_bit(cw, CW_AE) | g.stf_n | s_lt(a, term)) | ((s_ne(leader, slot) & ~rf) | s_ne(b, last)) |
                    (~x_ct & x_de & s_lt(ae_x, commit));
    const bool m_fa = fa_n >= 0;
    const sw ae_ref = ~fa_n & rf;
    const bool m_ref = ae_ref < 0;
    const sw ae_wc = ~fa_n & ~x_ct & x_de;                          // commits up to min(leaderCommit, last)
    const sw app_ae = ~fa_n & s_pos(n_eff);

    // ---- AppendEntries ack at a prepared leader (member/Leader.java:218-237, Leadership.java:75-130, Leader.java:247-280) --------
    __builtin_amdgcn_sched_barrier(0);
    const int32_t s_epoch = st.x, s_next = st.y, s_match = st.z, s_rej = st.w;
    const sw pend = (int32_t)(g.pending << ((uint32_t)cw & 31u));   // bit j of `pending`
    const sw adv = flg & s_lt(s_match, c);
    const bool m_adv = adv < 0;
    const int32_t n_match = m_adv ? c : s_match;
    const int32_t n_next = m_adv ? (int32_t)((uint32_t)c + 1u) : s_next;
#pragma unroll
    for (int i = 0; i < F; i++) m[i] = (j == (uint32_t)i) ? n_match : m[i];
    int32_t full, major;
    major_indices<F, int32_t>(m, full, major);
    const sw lookup = flg & s_pos(major);
    const sw bad_major = lookup & (s_lt(rc, 1) | s_lt(major, top) | s_lt(last, major));      // the quorum index is outside the newest run
    const int32_t ct0 = (lt == term) ? major : full;
    const sw rollback = lookup & s_pos(ct0) & s_lt(ct0, commit);
    const sw fk_n = (~cw | stl_n | x_aux) | (x_ta | s_ne(b, s_epoch) | pend) | (s_lt(c, s_match) | (~flg & ~s_pos(s_match))) |
                    (~s_lt(b, n_next) | bad_major | rollback);
    const bool m_fk = fk_n >= 0;
    const bool m_ct = (~fk_n & lookup) < 0;
    const int32_t ack_ct = m_ct ? ct0 : 0;                          // commit_to; 0, or >= commitIndex
    const sw ackany = cw_bit(cw, CW_ACKANY) & ~nallow;
    const sw drop = ackany & x_aux;

    // ---- client append at a leader (member/Leader.java:128-140) ---------------------------------------------------------------------
    const int32_t last_n = last + n;
    const sw fc_n = ~cw_bit(cw, CW_CLIENT) | g.stc_n | ~s_lt(last_n, (int32_t)STATE_LIMIT);
    const bool m_fc = fc_n >= 0;

    // ---- apply -----------------------------------------------------------------------------------------------------------------------
    // (the record is written back by every lane — its own values where the row is no such ack: cheaper than an exec-mask branch around two stores)
    pe.store_ack(j, s_epoch, m_fk ? n_next : s_next, m_fk ? n_match : s_match, m_fk ? ((flg < 0) ? 0 : (int32_t)((uint32_t)s_rej + 1u)) : s_rej);
    g.term = m_fa ? a : term;
    g.role_epoch = (uint32_t)repoch + (m_ref ? 1u : 0u);
    g.td = td & ~ae_ref;
    g.votes = m_ref ? 1 : votes;
    g.leader = m_fa ? slot : leader;
    g.last = m_fa ? ae_last : (m_fc ? last_n : last);
    const sw app = app_ae | ~fc_n;
    g.log_dirty = g.log_dirty | app;
    g.peers_dirty = g.peers_dirty | ~fk_n;
    const int32_t cand = (ae_wc < 0) ? ae_x : ack_ct;
    const int32_t new_commit = vmax<int32_t>(commit, cand);
    g.commit = new_commit;
    uint32_t pw = push_bit(0u, fa_n);
    pw = push_bit(pw, x_ct);
    pw = push_bit(pw, ae_ref);
    pw = push_bit(pw, app);
    pw = push_bit(pw, s_lt(commit, new_commit));
    pw = push_bit(pw, fc_n);
    out.resp = a;
    out.log_from = last + 1;
    // (a row that is not addressed is done whatever the lane's mode: after a NEED_HOST the host parks the group — RG_EV_NONE rows — and a wavefront
    // that visited the general handlers for each of them would pay a slow round per parked round; their answer for such a row is this one)
    sw done = ~fa_n | ~fk_n | ~fc_n | drop | cw_bit(cw, CW_NONE);
    sw any_drop = drop;
    uint32_t pw_el = 0u;

    // ---- what steady replication does not carry, behind ONE wave-uniform branch -------------------------------------------------------
    // rare: a new term run at the log tail (the first entries of a new leader's term), prepareReplication after a leader's first entry;
    // election traffic: decided AND applied here (a lane is in at most one class, and the classes above left the state of every other
    // lane as it was: the snapshot is still valid)
    const sw x_pl = s_ne(aux, lt);                                  // the carried entries are not of the tail's term
    const sw rare = (app_ae & x_pl) | (~fc_n & (x_tl | ~prep));
    const sw ack_down = ackany & ~x_aux & ~stl_n & x_ta;            // Leader -> Follower(result.term, responder)
    const sw election = (cw_bit(cw, CW_ELK) & ~nallow) | ack_down;
    // The ballots of this block are issued BACK TO BACK ahead of the branches that test them: a ballot whose scalar result is branched on at once costs the
    // VALU -> SALU round trip (50 ticks for ballot + taken branch against 20 for a taken branch on a scalar that is already there,
    // profiles/r04b_issue_bench.txt), and an election round has up to seven of them. Same-box: config 3 0.0687 -> 0.0673 ms, config 4's shard 0.1058 -> 0.1047
    // (profiles/r05i_hoisted_ballots_ab.jsonl).
    const uint64_t b_rare = __builtin_amdgcn_ballot_w64(rare < 0), b_el = __builtin_amdgcn_ballot_w64(election < 0);
    if ((b_rare | b_el) != 0) {
        if (b_rare != 0) {
            const bool ae_newrun = (app_ae & x_pl) < 0, fc_newrun = (~fc_n & x_tl) < 0, fc_prepare = (~fc_n & ~prep) < 0;
            if (ae_newrun | fc_newrun) g.push_run(last + 1, ae_newrun ? aux : term);
            if (fc_prepare) {                                           // Leader.prepareReplication after the FIRST new entry: nextIndex = that entry + 1
                pe.store_prepare(epoch, last + 2);
                g.pending = 0;
                g.prepared = -1; g.peers_dirty = -1;
            }
        }
        if (b_el != 0) {
            // One sub-block per row class, each behind its own ballot, and the conversion tail behind one more: three election rows of four are
            // vote replies that are merely counted (config 3: 1.19 % of the rows, 54 % of the wave-rounds; timeouts 12 %, vote requests 9 %), and a
            // launch ends with its slowest workgroup — the one that meets such a row in 61 of its 64 rounds.
            const int32_t term1 = term + 1;
            const sw is_pv = cw_bit(cw, CW_PV);
            const sw not_f = s_pos(role), not_c = s_ne(role, RG_CANDIDATE), not_l = s_ne(role, RG_LEADER);
            sw conv = ack_down, conv_self = 0, to_c = 0, win_rv = 0, late_higher = 0, count = 0, el_fast = ack_down;
            sw to_pre = 0, to_lead = 0, vq = 0, vq_success = 0, reset = 0, rv_new = 0, no_vote = 0;
            // vote replies (member/Candidate.java:121-134, member/Follower.java:258-270)
            const sw vr_shape = cw_bit(cw, CW_VR) & ~nallow;
            const sw to_kind = cw_bit(cw, CW_TO) & ~nallow & (s_pos(aux) | ((p.require_fence != 0) ? 0 : -1));      // (an un-fenced row where fences are required: general handlers, RG_BAD_EVENT)
            const sw vq_shape = cw_bit(cw, CW_VQ) & ~nallow;
            const uint64_t b_vr = __builtin_amdgcn_ballot_w64(vr_shape < 0), b_to = __builtin_amdgcn_ballot_w64(to_kind < 0), b_vq = __builtin_amdgcn_ballot_w64(vq_shape < 0);
            if (b_vr != 0) {
                const int32_t el_term = g.elected_term, el_epoch = (int32_t)g.elected_epoch;
                const sw sender_bad = (is_pv & (not_f | ~td)) | (~is_pv & not_c);
                const int32_t T = (is_pv < 0) ? term1 : term;
                const sw vr_cur = vr_shape & ~x_aux & ~sender_bad;
                const sw x_Ta = s_lt(T, a);
                const sw vr_grant = vr_cur & ~x_Ta & flg;
                const sw vr_win = vr_grant & ~s_lt(votes + 1, p.majority);
                const sw x_lt = s_lt(el_term, a);
                const sw late = vr_shape & ~is_pv & x_aux & (s_pos(el_epoch) & ~s_ne(aux, el_epoch));
                const sw late_noop = late & ~x_lt & (~flg | s_lt(el_term, term) | (~s_ne(el_term, term) & ~not_l));
                const sw vote_drop = vr_shape & x_aux & ~late;
                late_higher = late & x_lt;                              // head.abortRequests(); Follower if that is "better"
                win_rv = vr_win & ~is_pv;
                conv_self = vr_win;
                to_c = vr_win & is_pv;
                count = vr_grant & ~vr_win;
                conv = conv | (vr_cur & x_Ta) | vr_win | (late_higher & ~s_lt(a, term));
                el_fast = el_fast | vr_cur | late_higher | late_noop | vote_drop;
                any_drop = any_drop | vote_drop;
            }
            // timeouts (aux 0 = whoever is current; context/RaftRoutine.java:70)
            if (b_to != 0) {
                const sw to_stale = to_kind & s_pos(aux) & x_aux;
                const sw to_live = to_kind & ~to_stale;
                const sw pre = (p.pre_vote != 0) ? -1 : 0;
                const sw to_cand = to_live & ((~not_f & ~pre) | ~not_c);
                to_pre = to_live & ~not_f & pre;
                to_lead = to_live & ~not_l;
                conv_self = conv_self | to_cand;
                to_c = to_c | to_cand;
                conv = conv | to_cand | to_pre;
                reset = to_lead;
                el_fast = el_fast | to_kind;                            // (stale | pre | cand | lead: a timeout always lands in one of them)
                any_drop = any_drop | to_stale;
            }
            // RequestVote / PreVote at a Follower that has a log (member/Follower.java:91-127, 193-207)
            if (b_vq != 0) {
                const sw is_pvq = cw_bit(cw, CW_PVQ);
                vq = vq_shape & ~not_f & ~s_lt(rc, 1);
                const sw utd = s_lt(lt, c) | (~s_ne(c, lt) & ~s_lt(b, last));      // Follower.logUpToDate with a last entry
                const sw pv_judge = vq & is_pvq & x_ta & td;            // else failure(currentTerm), no timer touched
                const sw rv_same = vq & ~is_pvq & ~x_ta & ~s_lt(a, term);
                const sw voted_other = s_ne(voted, slot) | voted;       // (votedFor == NO_NODE is nobody's slot)
                rv_new = vq & ~is_pvq & x_ta;
                no_vote = rv_new & ~utd;
                vq_success = ((pv_judge | rv_new) & utd) | (rv_same & ~voted_other);
                reset = reset | pv_judge;
                conv = conv | rv_new;
                el_fast = el_fast | vq;
            }
            // RaftRoutine.convertTo + RaftMember.<init> for the lanes in `conv`; what else a row changes beyond the vote count
            const sw lead_prepare = to_lead & ~prep;                    // a new Leader's first tick: Leader.prepareReplication (member/Leader.java:30-50)
            if (__builtin_amdgcn_ballot_w64((conv | lead_prepare | late_higher) < 0) != 0) {
                const bool m_conv = conv < 0, m_winrv = win_rv < 0;
                const int32_t new_role = (win_rv < 0) ? RG_LEADER : ((to_c < 0) ? RG_CANDIDATE : RG_FOLLOWER);
                const int32_t new_term = (to_c < 0) ? term1 : (((win_rv | to_pre) < 0) ? term : a);
                const int32_t new_vote = (conv_self < 0) ? p.self : ((to_pre < 0) ? voted : ((no_vote < 0) ? RG_NO_NODE : slot));
                if (lead_prepare < 0) {
                    pe.store_prepare(epoch, ((rc > 0) ? last : epoch) + 1);
                    g.pending = 0;
                }
                g.elected_epoch = m_winrv ? (uint32_t)repoch : ((late_higher < 0) ? 0u : g.elected_epoch);      // Candidate.java:75-79 / head.abortRequests()
                g.elected_term = m_winrv ? term : g.elected_term;
                g.term = m_conv ? new_term : g.term;
                g.role = m_conv ? new_role : g.role;
                g.voted_for = m_conv ? new_vote : g.voted_for;
                g.role_epoch = g.role_epoch + (m_conv ? 1u : 0u);
                g.td = (g.td & ~conv) | to_pre;
                g.votes = m_conv ? 1 : g.votes;
                g.leader = m_conv ? RG_NO_NODE : g.leader;
                g.prepared = (g.prepared & ~conv) | lead_prepare;
                g.peers_dirty = g.peers_dirty | lead_prepare;
            }
            g.votes = g.votes + (int32_t)((uint32_t)count >> 31);       // (a counted vote never converts)
            out.resp = ((vq & ~rv_new) < 0) ? term : a;                 // (only read for the vote requests)
            pw_el = push_bit(0u, vq);
            pw_el = push_bit(pw_el, vq_success);
            pw_el = push_bit(pw_el, reset);
            pw_el = push_bit(pw_el, conv);
            pw_el = push_bit(pw_el, to_lead);
            pw_el = push_bit(pw_el, to_pre);
            pw_el = push_bit(pw_el, to_c);
            done = done | el_fast;
        }
        g.recache();
    }
    pw = push_bit(pw, any_drop);
    out.pw = pw | (pw_el << 7);
    return done;
}

// ---- tier 1.5 (round 6): one more row class, decided on the 32-bit image but OFF the main path -------------------------------------------------------
// An AppendEntries request whose entries OVERWRITE the uncommitted tail of this Follower's newest term run — what a new leader sends a follower that had
// appended entries of the old one (member/Follower.java:35-88 with RaftLog.conflict / truncate / append, storage/RocksLog.java:131-216): prevLogIndex lies
// strictly below the tail and inside the newest run (so its term is the tail's term and every overlapped entry has that term too), the carried entries are of
// ONE other term (the conflict is at the first of them), nothing committed is cut and the commit index does not roll back. Stepper::on_append_entries on that
// input: [refresh on a higher term or a timed-out Follower], leader = sender, truncate(prev + 1), a new run (prev + 1, entries' term), last = prev + n,
// mark_committed(min(leaderCommit, last)), reply(term, true). tier1n leaves such a row open (prevLogIndex != last); narrow_body calls this for the open lanes
// BEFORE it widens the image for the general handlers, and a wave-round whose open rows are all of this class never visits them: config 3 with 0.5 % of such
// rows spent a general-handler visit on every fourth wave-round (bench.py: value_adverse_mix). A strict special case of the general handlers like every class
// of tier 1 (test_general_handlers_alone_give_the_same_answers); not on tier1n's instruction path: plain predicates, a divergent branch.
// The second class is the cache miss itself: an AppendEntries request whose prevLogIndex lies in the log but BELOW the cached term runs. on_append_entries
// finds that out before it touches anything (its pre-check) and answers RG_NEED_HOST with log_from = the index whose term the host must supply; the group is
// parked for the rest of the launch (`park`). Deciding that needs four compares, not a widened image and the general handlers.
// Returns: the lane's row was decided here (state, `out` written).
// Two more, both of them what an election leaves behind in BASELINE's own streams (config 3: 0.018 % of the rows, but 1.2 % of the wave-rounds — and a launch ends
// with its slowest workgroup): (i) an AppendEntries ack that REJECTS while nothing of that follower has matched yet (member/Leader.java:218-237 ->
// Leadership.State.updateIndex, member/Leadership.java:75-114: nextIndex backs off by rejection_step, down to epoch.index + 1; at or below the epoch the
// follower is marked pending for a snapshot) — tier1n's ack class wants a success or a follower that has matched; (ii) a RequestVote / PreVote of a HIGHER
// term at a Candidate (member/Candidate.java:90-119: switchTo(Follower, term, candidateId), then the new Follower answers the same request: same term, it has
// just voted for that candidate -> granted) — tier1n's vote-request class is the Follower's.
template <int F>
__device__ __forceinline__ bool tier15(const StepParams &p, GroupN &g, PeersNarrow<F> &pe, OutN &out, bool &park, const bool open, const int32_t cw,
                                       const int32_t aux, const int32_t n, const int32_t a, const int32_t b, const int32_t c, const int32_t d)
{
    bool fixed = false;
    {
        const int32_t slot = (int32_t)(((uint32_t)cw >> 5) & 15u);
        // (i)
        const uint32_t j = ((uint32_t)cw >> 10) & 7u;
        const bool nack_shape = open & (cw_bit(cw, CW_ACK) < 0) & (cw_bit(cw, CW_FLAG) >= 0) & (g.stl_n >= 0) & (aux == (int32_t)g.role_epoch) & (a <= g.term);
        if (nack_shape) {
            const I32x4 st = pe.rec[j * BLOCK];
            const bool pend = ((g.pending >> j) & 1u) != 0;
            if ((st.z == 0) & (b == st.x) & !pend) {                 // nothing matched, same epoch as at send, no snapshot pending
                const int32_t rej = (int32_t)((uint32_t)st.w + 1u);
                const int32_t next = vmax<int32_t>(st.y - (int32_t)rejection_step(rej), b + 1);
                const int32_t s_next = vmin<int32_t>(st.y - 1, next);
                pe.store_ack(j, st.x, s_next, 0, rej);
                g.pending = g.pending | ((s_next <= b) ? (1u << j) : 0u);
                g.peers_dirty = -1;
                out.pw = PW_SLOW;                                    // no flag, RG_OK
                out.resp = 0; out.log_from = 0;
                fixed = true;
            }
        }
        // (ii)
        const bool cand_vq = open & (cw_bit(cw, CW_VQ) < 0) & (g.nallow >= 0) & (g.role == RG_CANDIDATE) & (g.term < a) & (slot != p.self);
        if (cand_vq) {
            g.role = RG_FOLLOWER; g.term = a; g.voted_for = slot;
            g.role_epoch = g.role_epoch + 1u;
            g.td = 0; g.leader = RG_NO_NODE; g.votes = 1; g.prepared = 0;
            g.recache();
            out.pw = PW_SLOW | RG_F_PERSIST | RG_F_ROLE_CHANGED | RG_F_RESET_TIMER | RG_F_REPLIED | RG_F_SUCCESS;
            out.resp = a; out.log_from = 0;
            fixed = true;
        }
    }
    {
        const bool miss = open & (cw_bit(cw, CW_AE) < 0) & (g.stf_n >= 0) & (a >= g.term) & (g.epoch_index < b) & (b <= g.last) & (b < g.s0);
        if (miss) {
            out.pw = PW_SLOW | ((uint32_t)RG_NEED_HOST << RG_F_STATUS_SHIFT);
            out.resp = 0;
            out.log_from = b;
        }
        park = miss;
        fixed = fixed | miss;
        if (__builtin_amdgcn_ballot_w64(open & !fixed) == 0) return fixed;
    }
    const int32_t term = g.term, last = g.last, commit = g.commit, epoch = g.epoch_index, lt = g.lt, top = g.top;
    const int32_t slot = (int32_t)(((uint32_t)cw >> 5) & 15u);
    const bool refresh = (term < a) | (g.td < 0);                    // switchTo(Follower, term, lastCandidate)
    const int32_t new_last = b + n;
    const int32_t commit_to = vmin<int32_t>(d, new_last);
    const bool with_commit = epoch < d;                              // leaderCommit > epoch.index
    const bool ok = open & (cw_bit(cw, CW_AE) < 0) & (g.stf_n >= 0) & (a >= term) & ((g.leader == slot) | refresh) &
                    (b < last) & (b >= top) & (b > epoch) & (c == lt) & (n >= 1) & (aux != lt) & (b >= commit) & !(with_commit & (commit_to < commit));
    if (ok) {
        g.term = a;
        g.role_epoch = g.role_epoch + (refresh ? 1u : 0u);
        g.td = 0;
        g.votes = refresh ? 1 : g.votes;
        g.leader = slot;
        g.last = b;                                                  // RaftLog.truncate(prev + 1): the newest run keeps its entries up to prev
        g.push_run(b + 1, aux);                                      // RaftLog.append: a run of the new term starts right above it
        g.last = new_last;
        const int32_t new_commit = with_commit ? vmax<int32_t>(commit, commit_to) : commit;
        g.commit = new_commit;
        out.pw = PW_SLOW | RG_F_RESET_TIMER | RG_F_REPLIED | RG_F_SUCCESS | RG_F_LOG_TRUNC | RG_F_LOG_APPEND |
                 (refresh ? (RG_F_PERSIST | RG_F_ROLE_CHANGED) : 0u) | ((new_commit > commit) ? RG_F_COMMIT : 0u);
        out.resp = a;
        out.log_from = b + 1;
    }
    return ok | fixed;
}

}  // namespace rg
