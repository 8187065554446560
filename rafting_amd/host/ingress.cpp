// ingress.cpp — see ingress.hpp.
#include "ingress.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace rafting {
namespace wire {

// ---- ContextIndex ---------------------------------------------------------------------------------------------------------
static uint32_t pow2_at_least(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

ContextIndex::ContextIndex(uint32_t capacity)
    : mask_(pow2_at_least(capacity < 8 ? 16 : capacity * 2) - 1), owner_(new Slot[mask_ + 1]), key_(new Key[capacity]), capacity_(capacity)
{
    for (uint32_t i = 0; i <= mask_; i++) owner_[i].store(0, std::memory_order_relaxed);
    for (uint32_t g = 0; g < capacity; g++) key_[g].len = 0;
    slot_.store(owner_.get(), std::memory_order_release);
}

uint64_t ContextIndex::hash(const char *s, size_t n)
{
    // 8 bytes at a time, multiply-xorshift (the ids are short ASCII names; what matters is that "ctx123" and "ctx124" part ways)
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    while (n >= 8) { uint64_t w; memcpy(&w, s, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; s += 8; n -= 8; }
    uint64_t w = 0;
    memcpy(&w, s, n);
    h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 29);
}

// Every probe loop below is bounded by the table size: the live entries never fill more than half of the slots, but tombstones of erased
// contexts can take the rest (an unbounded probe then spins for ever on a miss). reclaim() keeps them from getting there.
bool ContextIndex::insert(const char *id, size_t len, uint32_t gid)
{
    if (gid >= capacity_ || len == 0 || len >= KEY_BYTES) return false;
    std::lock_guard<std::mutex> lk(mu_);
    if (key_[gid].len != 0) return false;                        // taken (or erased and not reclaimed yet)
    Slot *slot = owner_.get();
    const uint64_t h = hash(id, len);
    uint32_t at = (uint32_t)h & mask_, free_at = mask_ + 1, probes = 0;
    for (; probes <= mask_; at = (at + 1) & mask_, probes++) {
        const uint64_t v = slot[at].load(std::memory_order_relaxed);
        if (v == 0) break;
        if ((uint32_t)v == TOMB) { if (free_at > mask_) free_at = at; continue; }
        const Key &k = key_[(uint32_t)v - 1];
        if ((v >> 32) == (h >> 32) && k.len == len && memcmp(k.bytes, id, len) == 0) return false;
    }
    if (free_at <= mask_) { at = free_at; tombs_--; }
    else if (probes > mask_) return false;                       // (cannot happen while live <= capacity <= half of the slots: no empty slot and no tombstone)
    memcpy(key_[gid].bytes, id, len);
    key_[gid].len = (uint8_t)len;
    slot[at].store((uint64_t)(gid + 1) | (h >> 32 << 32), std::memory_order_release);      // the key bytes are visible to whoever sees the slot
    n_++;
    return true;
}

uint32_t ContextIndex::erase(const char *id, size_t len)
{
    std::lock_guard<std::mutex> lk(mu_);
    Slot *slot = owner_.get();
    const uint64_t h = hash(id, len);
    uint32_t at = (uint32_t)h & mask_;
    for (uint32_t probes = 0; probes <= mask_; at = (at + 1) & mask_, probes++) {
        const uint64_t v = slot[at].load(std::memory_order_relaxed);
        if (v == 0) return capacity_;
        if ((uint32_t)v == TOMB || (v >> 32) != (h >> 32)) continue;
        const uint32_t gid = (uint32_t)v - 1;
        const Key &k = key_[gid];
        if (k.len != len || memcmp(k.bytes, id, len) != 0) continue;
        // A slot whose successor is empty ends its run of occupied slots: no key can lie beyond it, so it may become EMPTY at once — and so
        // may the tombstones right before it, one after the other (a concurrent lookup that meets the new empty slot stops where nothing it
        // could be looking for lies behind). Only a slot in the MIDDLE of a run has to stay a tombstone.
        if (slot[(at + 1) & mask_].load(std::memory_order_relaxed) == 0) {
            slot[at].store(0, std::memory_order_release);
            for (uint32_t b = (at - 1) & mask_; slot[b].load(std::memory_order_relaxed) == (uint64_t)TOMB; b = (b - 1) & mask_) {
                slot[b].store(0, std::memory_order_release);
                tombs_--;
            }
        } else {
            slot[at].store((uint64_t)TOMB, std::memory_order_release);
            tombs_++;
        }
        retired_.push_back(gid);                                 // the key record stays as it is until reclaim(): a lookup may be comparing it
        n_--;
        return gid;
    }
    return capacity_;
}

size_t ContextIndex::retired()
{
    std::lock_guard<std::mutex> lk(mu_);
    return retired_.size();
}

// Under mu_. A fresh slot array with the live entries only, published with one pointer store. Lookups that loaded the old pointer go on in
// the old array — for them the rebuild did not happen yet, which is what a lookup racing an insert / erase may see anyway — so the old array
// must outlive them: it is kept until the NEXT reclaim(), i.e. (by reclaim()'s contract) until a seal that started after this one returned
// has excluded every feed().
void ContextIndex::rebuild()
{
    std::unique_ptr<Slot[]> fresh(new Slot[mask_ + 1]);
    for (uint32_t i = 0; i <= mask_; i++) fresh[i].store(0, std::memory_order_relaxed);
    for (uint32_t i = 0; i <= mask_; i++) {
        const uint64_t v = owner_[i].load(std::memory_order_relaxed);
        if (v == 0 || (uint32_t)v == TOMB) continue;
        const Key &k = key_[(uint32_t)v - 1];
        uint32_t at = (uint32_t)hash(k.bytes, k.len) & mask_;
        while (fresh[at].load(std::memory_order_relaxed) != 0) at = (at + 1) & mask_;       // (live <= half of the slots: ends)
        fresh[at].store(v, std::memory_order_relaxed);
    }
    previous_ = std::move(owner_);
    owner_ = std::move(fresh);
    slot_.store(owner_.get(), std::memory_order_release);
    tombs_ = 0;
    rebuilds_++;
}

void ContextIndex::reclaim(size_t n)
{
    std::lock_guard<std::mutex> lk(mu_);
    previous_.reset();                                           // the array the rebuild BEFORE the last seal replaced: nobody can be in it any more
    n = std::min(n, retired_.size());
    for (size_t i = 0; i < n; i++) key_[retired_[i]].len = 0;
    retired_.erase(retired_.begin(), retired_.begin() + (long)n);
    if ((uint64_t)n_ + tombs_ > (uint64_t)(mask_ + 1) * 7 / 10) rebuild();     // live + tombstones beyond 70 % of the slots: probes are getting long
}

bool ContextIndex::find(const char *id, size_t len, uint32_t &gid) const { return find(hash(id, len), id, len, gid); }

uint32_t ContextIndex::peek(uint64_t h) const
{
    const Slot *slot = slot_.load(std::memory_order_acquire);
    for (uint32_t at = (uint32_t)h & mask_, probes = 0; probes < 4; at = (at + 1) & mask_, probes++) {
        const uint64_t v = slot[at].load(std::memory_order_acquire);
        if (v == 0) break;
        if ((uint32_t)v != TOMB && (v >> 32) == (h >> 32)) return (uint32_t)v - 1;
    }
    return capacity_;
}

bool ContextIndex::find(uint64_t h, const char *id, size_t len, uint32_t &gid) const
{
    const Slot *slot = slot_.load(std::memory_order_acquire);
    uint32_t at = (uint32_t)h & mask_;
    for (uint32_t probes = 0; probes <= mask_; at = (at + 1) & mask_, probes++) {
        const uint64_t v = slot[at].load(std::memory_order_acquire);
        if (v == 0) return false;
        if ((uint32_t)v == TOMB || (v >> 32) != (h >> 32)) continue;
        const Key &k = key_[(uint32_t)v - 1];
        if (k.len == len && memcmp(k.bytes, id, len) == 0) { gid = (uint32_t)v - 1; return true; }
    }
    return false;
}

std::string ContextIndex::id_of(uint32_t gid) const { return gid < capacity_ ? std::string(key_[gid].bytes, key_[gid].len) : std::string(); }

// ---- PendingRing ------------------------------------------------------------------------------------------------------------
PendingRing::PendingRing(uint32_t capacity_pow2) : mask_(pow2_at_least(capacity_pow2) - 1), s_(new Slot[mask_ + 1]) {}

// bit 0 = occupied, bits 1-32 the sequence, 33-35 the method, 36-63 the group id (28 bits: 268 M groups — a table holds far fewer)
uint64_t PendingRing::key_of(int32_t sequence, Method m, uint32_t gid)
{
    return 1ull | ((uint64_t)(uint32_t)sequence << 1) | ((uint64_t)((uint32_t)m & 7u) << 33) | ((uint64_t)(gid & 0x0FFFFFFFu) << 36);
}

void PendingRing::put(int32_t sequence, Method m, uint32_t gid, const Pending &p)
{
    Slot &s = s_[(uint32_t)sequence & mask_];
    s.key.store(0, std::memory_order_relaxed);                   // whoever is reading the old record loses its CAS ...
    std::atomic_thread_fence(std::memory_order_release);         // ... and the cleared key is visible before any word of the new record is (a release
                                                                 // STORE orders what precedes it, not the stores that follow: ADVICE r3)
    s.w0.store(p.role_epoch, std::memory_order_relaxed);
    s.w1.store((uint64_t)p.epoch_at_send, std::memory_order_relaxed);
    s.w2.store((uint64_t)p.last_index_sent, std::memory_order_relaxed);
    s.key.store(key_of(sequence, m, gid), std::memory_order_release);
}

bool PendingRing::take(int32_t sequence, Method m, uint32_t gid, Pending &p)
{
    Slot &s = s_[(uint32_t)sequence & mask_];
    const uint64_t want = key_of(sequence, m, gid);
    if (s.key.load(std::memory_order_acquire) != want) return false;
    p.role_epoch = (uint32_t)s.w0.load(std::memory_order_relaxed);
    p.epoch_at_send = (int64_t)s.w1.load(std::memory_order_relaxed);
    p.last_index_sent = (int64_t)s.w2.load(std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_acquire);         // the words were read before the CAS looks at the key again
    uint64_t expect = want;                                      // the invocation is REMOVED (AsyncService.remove): a duplicate response finds nothing
    return s.key.compare_exchange_strong(expect, 0, std::memory_order_acq_rel);
}

void PendingRing::clear() { for (uint32_t i = 0; i <= mask_; i++) s_[i].key.store(0, std::memory_order_release); }

// ---- Ingress ------------------------------------------------------------------------------------------------------------------
Ingress::Ingress(uint32_t groups, uint32_t max_rounds, uint32_t conns, const BodyCodec &codec, const ContextIndex &index, Buffers bank0, Buffers bank1,
                 uint32_t pending_capacity, uint32_t shards)
    : groups_(groups), rounds_(max_rounds), codec_(codec), index_(index), c_(conns)
{
    if (shards == 0 || shards > groups) shards = 1;
    per_shard_ = (groups + shards - 1) / shards;                  // shard.block_partition: gpu = gid / ceil(G / N)
    const uint64_t term_cap = bank0.entry_cap / shards;
    for (uint32_t s = 0; s * per_shard_ < groups; s++) {
        const uint32_t first = s * per_shard_;
        shard_.push_back(Shard{first, std::min(per_shard_, groups - first), (size_t)first * max_rounds, (uint64_t)s * term_cap, term_cap});
    }
    const Buffers b[2] = {bank0, bank1};
    for (int i = 0; i < 2; i++) {
        bank_[i].buf = b[i];
        bank_[i].depth.reset(new std::atomic<uint32_t>[groups]);
        for (uint32_t g = 0; g < groups; g++) bank_[i].depth[g].store(0, std::memory_order_relaxed);
        bank_[i].origin.assign((size_t)groups * max_rounds, Origin{NO_CONN, 0});
        bank_[i].body.assign((size_t)groups * max_rounds, Bank::BodyRef{NO_CONN, 0, 0});
        bank_[i].terms_used.reset(new std::atomic<uint64_t>[shard_.size()]);
        bank_[i].dirty_rounds.assign(shard_.size(), max_rounds);  // caller memory: content unknown
        wipe(bank_[i]);
        sealed_[i].shard.resize(shard_.size());
    }
    for (Conn &c : c_) {
        c.ring.reset(new PendingRing(pending_capacity));
        c.rows.assign(shard_.size(), 0);
        c.max_depth.assign(shard_.size(), 0);
    }
}

void Ingress::set_peer(uint32_t conn, int32_t peer_slot) { c_[conn].peer = peer_slot; }

void Ingress::reset_connection(uint32_t conn)
{
    Conn &c = c_[conn];
    c.sp = FrameSplitter();
    c.n_staged = 0;
    c.ring->clear();
}

void Ingress::wipe(Bank &bk)
{
    for (size_t s = 0; s < shard_.size(); s++) {
        const Shard &sh = shard_[s];
        const size_t cells = (size_t)bk.dirty_rounds[s] * sh.count;
        memset(bk.buf.head + sh.cell_off, 0, cells * sizeof(rg_ev_head_t));        // RG_EV_NONE
        memset(bk.buf.abcd + sh.cell_off, 0, cells * sizeof(rg_ev_quad32_t));      // (the kernel's 32-bit domain check reads the fields of every cell)
        bk.terms_used[s].store(0, std::memory_order_relaxed);
        bk.dirty_rounds[s] = 0;
    }
    for (uint32_t g = 0; g < groups_; g++) bk.depth[g].store(0, std::memory_order_relaxed);
    if (retain_) for (Conn &c : c_) c.bodies[&bk - bank_].clear();
    bk.wide.clear();
    bk.clean = true;
}

void Ingress::recycle(const SealedBatch &b) { wipe(bank_[bank_of(b)]); }

void Ingress::hold(Conn &c, uint32_t gid, rg_ev_head_t head, int64_t a, int64_t b, int64_t c4, int64_t d, const int64_t *terms, size_t n_terms, Origin from,
                   const char *body, size_t body_len)
{
    HeldRow h{gid, head, a, b, c4, d, from, {}, ticket_.fetch_add(1, std::memory_order_relaxed), {}};
    h.terms.assign(terms, terms + n_terms);
    if (body_len) h.body.assign(body, body_len);
    c.held.push_back(std::move(h));
    c.held_count.store(c.held.size(), std::memory_order_relaxed);      // (what held_on() reads: the vector itself belongs to its reader and to seal())
}

// true: the row sits in the bank (or in its wide list); false: the caller keeps it for the next batch
bool Ingress::place(Bank &bk, Conn &cn, uint32_t gid, rg_ev_head_t head, int64_t a, int64_t b, int64_t c4, int64_t d, const int64_t *terms, size_t n_terms,
                    Origin from, const char *body, size_t body_len)
{
    std::atomic<uint32_t> &depth = bk.depth[gid];
    const uint32_t si = gid / per_shard_;
    const Shard &sh = shard_[si];
    const bool ae = RG_HDR_KIND(head.hdr) == RG_EV_AE_REQ;
    bool same = true;
    for (size_t k = 1; k < n_terms; k++) same &= terms[k] == terms[0];
    uint64_t all = (uint64_t)a | (uint64_t)b | (uint64_t)c4 | (uint64_t)d;
    for (size_t k = 0; k < n_terms; k++) all |= (uint64_t)terms[k];
    if ((all >> 31) || (!same && n_terms > sh.term_cap)) {       // not a compact row (a value beyond int32, or more terms than the shard's term array
                                                                  // holds at all): close the group, hand the row over on its own
        const uint32_t was = depth.fetch_or(CLOSED, std::memory_order_relaxed);
        if (was >= rounds_) return false;                         // the group was full or closed already: wait for the next batch (it stays closed)
        HeldRow h{gid, head, a, b, c4, d, from, {}, 0, {}};
        h.terms.assign(terms, terms + n_terms);
        if (body_len) h.body.assign(body, body_len);
        std::lock_guard<std::mutex> lk(bk.wide_mu);
        bk.wide.push_back(std::move(h));
        return true;
    }
    uint64_t toff = 0;
    if (ae && n_terms > 0 && !same) {                             // entries of several terms: they need room in the batch's term array
        toff = bk.terms_used[si].fetch_add(n_terms, std::memory_order_relaxed);
        if (toff + n_terms > sh.term_cap || toff + n_terms > 0xFFFFFFFFull) {
            depth.fetch_or(CLOSED, std::memory_order_relaxed);
            return false;
        }
    }
    const uint32_t r = depth.fetch_add(1, std::memory_order_relaxed);
    if (r >= rounds_) return false;                               // the group's rounds are used up, or it is closed: either way it stays so
    const size_t cell = sh.cell_off + (size_t)r * sh.count + (gid - sh.first);
    if (ae && n_terms > 0) {
        if (same) { head.hdr |= RG_HDR_SAME_TERM; head.aux = (uint32_t)terms[0]; }
        else {
            head.aux = (uint32_t)toff;                                // an offset into the SHARD's term array (its batch's entry_terms)
            for (size_t k = 0; k < n_terms; k++) bk.buf.entry_terms[sh.term_off + toff + k] = (int32_t)terms[k];
        }
    }
    bk.buf.abcd[cell] = rg_ev_quad32_t{(int32_t)a, (int32_t)b, (int32_t)c4, (int32_t)d};
    bk.buf.head[cell] = head;
    bk.origin[cell] = from;
    if (retain_) {
        if (body_len) {
            std::string &arena = cn.bodies[&bk - bank_];
            bk.body[cell] = Bank::BodyRef{(uint32_t)(&cn - c_.data()), (uint32_t)arena.size(), (uint32_t)body_len};
            arena.append(body, body_len);
        } else bk.body[cell] = Bank::BodyRef{NO_CONN, 0, 0};
    }
    cn.rows[si]++;
    cn.max_depth[si] = std::max(cn.max_depth[si], r + 1);
    return true;
}

// feed() in three passes over up to eight frames of a read: (1) stage — scope, hash, prefetch of the index slot (and of the invocation slot
// of a response); (2) drain, first half — slot -> candidate group, prefetch of its key record and of its round counter; (3) drain, second
// half — key compare, body decode, cell claim. The cache misses of one frame overlap the work on its neighbours.
void Ingress::stage(uint32_t conn, const FrameView &f)
{
    Conn &c = c_[conn];
    Conn::Staged &s = c.staged[c.n_staged++];
    s.f = f;
    s.ok = (f.type == ENQ || f.type == ACK) && parse_scope(f.head, f.head_len, s.m, s.id, s.id_len);
    if (s.ok) {
        s.hash = ContextIndex::hash_of(s.id, s.id_len);
        index_.prefetch(s.hash);
        if (f.type == ACK) c.ring->prefetch(f.sequence);
    }
    if (c.n_staged == sizeof(c.staged) / sizeof(c.staged[0])) drain(conn);
}

void Ingress::drain(uint32_t conn)
{
    Conn &c = c_[conn];
    Bank &bk = bank_[fill_];
    for (uint32_t i = 0; i < c.n_staged; i++) {
        const Conn::Staged &s = c.staged[i];
        if (!s.ok) continue;
        const uint32_t guess = index_.peek(s.hash);
        index_.prefetch_key(guess);
        if (guess < groups_) __builtin_prefetch(&bk.depth[guess]);
    }
    for (uint32_t i = 0; i < c.n_staged; i++) on_frame(conn, c.staged[i]);
    c.n_staged = 0;
}

// One frame -> one row (RowWriter::add's mapping, wire.cpp, with the lookups this class owns)
void Ingress::on_frame(uint32_t conn, const Conn::Staged &st)
{
    Conn &c = c_[conn];
    const FrameView &f = st.f;
    const Method m = st.m;
    uint32_t gid = 0;
    if (!st.ok || !index_.find(st.hash, st.id, st.id_len, gid)) {
        refused_.fetch_add(1, std::memory_order_relaxed);
        return;
    }
    rg_ev_head_t h{0, 0};
    int64_t a = 0, b = 0, c4 = 0, d = 0;
    const int64_t *terms = nullptr;
    size_t n_terms = 0;
    Origin from{NO_CONN, 0};
    if (f.type == ENQ) {
        Request &q = c.q;
        if (!codec_.decode_request(m, f.body, f.body_len, q) || q.node < 0 || q.node > 15 || q.entry_terms.size() > RG_MAX_AE_ENTRIES) {
            refused_.fetch_add(1, std::memory_order_relaxed);
            return;
        }
        from = Origin{conn, f.sequence};
        a = q.term; b = q.x; c4 = q.y;
        switch (m) {
        case M_APPEND_ENTRIES:
            h.hdr = RG_HDR_MAKE(RG_EV_AE_REQ, q.node, 0, q.entry_terms.size());
            d = q.leader_commit; terms = q.entry_terms.data(); n_terms = q.entry_terms.size();
            break;
        case M_PRE_VOTE:         h.hdr = RG_HDR_MAKE(RG_EV_PV_REQ, q.node, 0, 0); break;
        case M_REQUEST_VOTE:     h.hdr = RG_HDR_MAKE(RG_EV_RV_REQ, q.node, 0, 0); break;
        case M_INSTALL_SNAPSHOT: h.hdr = RG_HDR_MAKE(RG_EV_IS_REQ, q.node, 0, 0); break;    // flag 0: the host holds the verdict (raftwire.h)
        default: refused_.fetch_add(1, std::memory_order_relaxed); return;
        }
    } else {
        Response r;
        Pending p;
        if (c.peer < 0 || c.peer > 15 || !codec_.decode_response(f.body, f.body_len, r) || !c.ring->take(f.sequence, m, gid, p)) {
            refused_.fetch_add(1, std::memory_order_relaxed);
            return;
        }
        a = r.term;
        h.aux = p.role_epoch;
        switch (m) {
        case M_APPEND_ENTRIES:   h.hdr = RG_HDR_MAKE(RG_EV_AE_ACK, c.peer, r.success, 0); b = p.epoch_at_send; c4 = p.last_index_sent; break;
        case M_INSTALL_SNAPSHOT: h.hdr = RG_HDR_MAKE(RG_EV_IS_ACK, c.peer, r.success, 0); b = p.epoch_at_send; break;
        case M_PRE_VOTE:         h.hdr = RG_HDR_MAKE(RG_EV_PV_REPLY, c.peer, r.success, 0); break;
        case M_REQUEST_VOTE:     h.hdr = RG_HDR_MAKE(RG_EV_RV_REPLY, c.peer, r.success, 0); break;
        default: refused_.fetch_add(1, std::memory_order_relaxed); return;
        }
    }
    c.queued++;
    // a connection that already holds rows back keeps its order: a group with a held row is closed in this bank, so place() refuses
    const bool keep = retain_ && n_terms > 0;                    // (an AppendEntries request that carries entries: the host will want their payload)
    if (!place(bank_[fill_], c, gid, h, a, b, c4, d, terms, n_terms, from, keep ? f.body : nullptr, keep ? f.body_len : 0))
        hold(c, gid, h, a, b, c4, d, terms, n_terms, from, keep ? f.body : nullptr, keep ? f.body_len : 0);
}

const char *Ingress::body(const SealedBatch &b, uint32_t shard, size_t cell, size_t &len) const
{
    const int k = bank_of(b);
    len = 0;
    if (!retain_ || shard >= shard_.size() || cell >= (size_t)b.shard[shard].batch.rounds * shard_[shard].count) return nullptr;
    const size_t at = shard_[shard].cell_off + cell;
    if (RG_HDR_KIND(bank_[k].buf.head[at].hdr) != RG_EV_AE_REQ) return nullptr;      // (only those cells' references were written for this batch)
    const Bank::BodyRef r = bank_[k].body[at];
    if (r.conn == NO_CONN || r.len == 0 || (size_t)r.off + r.len > c_[r.conn].bodies[k].size()) return nullptr;
    len = r.len;
    return c_[r.conn].bodies[k].data() + r.off;
}

namespace {
struct StandBack {                                                // feeders do not start a new read while somebody wants the exclusive lock
    std::atomic<uint32_t> &f;                                     // (a COUNT: seal() and held() may both be asking — ADVICE r3: with one flag the first to finish
    explicit StandBack(std::atomic<uint32_t> &x) : f(x) { f.fetch_add(1, std::memory_order_acq_rel); }     // let the feeders loose on the other)
    ~StandBack() { f.fetch_sub(1, std::memory_order_acq_rel); }
};
}  // namespace

int Ingress::feed(uint32_t conn, const uint8_t *data, size_t n)
{
    while (sealing_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    std::shared_lock<std::shared_mutex> lk(mu_);
    Conn &c = c_[conn];
    c.queued = 0;
    c.sp.feed_views(data, n, [this, conn](const FrameView &f) { stage(conn, f); }, [this, conn] { drain(conn); });
    return c.sp.failed() ? -1 : c.queued;
}

size_t Ingress::encode_sends(uint32_t conn, int32_t self_slot, uint32_t count, const uint32_t *gid, const rg_send_head_t *head, const rg_send_t *send_j,
                             TermOf &log, std::string &out, uint32_t *need_host)
{
    Conn &c = c_[conn];
    size_t made = 0;
    uint32_t missing = 0;
    Frame f;
    f.type = ENQ;
    Request q;
    q.node = self_slot;
    for (uint32_t i = 0; i < count; i++) {
        const rg_send_t &s = send_j[i];
        const rg_send_head_t &h = head[i];
        if (s.kind != RG_SEND_APPEND && s.kind != RG_SEND_SNAPSHOT && s.kind != RG_SEND_NEED_HOST) continue;
        const uint32_t g = gid ? gid[i] : i;
        const Method m = s.kind == RG_SEND_SNAPSHOT ? M_INSTALL_SNAPSHOT : M_APPEND_ENTRIES;
        int64_t prev_term = s.prev_term;
        if (s.kind == RG_SEND_NEED_HOST) {                        // prevLogIndex lies below the device's cached term runs: its term comes from
            prev_term = log.term_of(g, s.prev_index);            // the log this side owns (what ContextManager::replicateLog does for the mirror)
            missing++;
        }
        q.term = h.term;
        q.entry_terms.clear();
        if (m == M_APPEND_ENTRIES) {
            q.x = s.prev_index; q.y = prev_term; q.leader_commit = h.leader_commit;
            for (uint32_t k = 0; k < s.count; k++) q.entry_terms.push_back(log.term_of(g, s.prev_index + 1 + (int64_t)k));
        } else {
            q.x = h.epoch_index; q.y = h.epoch_term; q.leader_commit = 0;
        }
        f.sequence = c.next_sequence; c.next_sequence = (int32_t)((uint32_t)c.next_sequence + 1u);      // wraps like the reference's AtomicInteger
        f.head.assign(m == M_APPEND_ENTRIES ? "appendEntries:" : "installSnapshot:");
        index_.append_id(g, f.head);
        f.body.clear();
        codec_.encode_request(m, q, f.body);
        encode_frame(f, false, out);
        Pending p;
        p.role_epoch = h.role_epoch; p.epoch_at_send = h.epoch_index; p.last_index_sent = s.last_index;
        c.ring->put(f.sequence, m, g, p);
        made++;
    }
    if (need_host) *need_host = missing;
    return made;
}

void Ingress::add_row(uint32_t conn, uint32_t gid, rg_ev_head_t head, int64_t a, int64_t b, int64_t c4, int64_t d, Origin reply_to)
{
    while (sealing_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    std::shared_lock<std::shared_mutex> lk(mu_);
    Conn &c = c_[conn];
    const uint32_t kind = RG_HDR_KIND(head.hdr);                 // (AppendEntries: entries travel in frames; NONE / unknown kinds are no rows)
    if (gid >= groups_ || kind == RG_EV_AE_REQ || kind == RG_EV_NONE || kind > RG_EV_IS_REQ) { refused_.fetch_add(1, std::memory_order_relaxed); return; }
    if (!place(bank_[fill_], c, gid, head, a, b, c4, d, nullptr, 0, reply_to)) hold(c, gid, head, a, b, c4, d, nullptr, 0, reply_to);
}

const SealedBatch &Ingress::seal()
{
    StandBack standing_back(sealing_);
    std::unique_lock<std::shared_mutex> lk(mu_);
    const int done = fill_, next = fill_ ^ 1;
    // two banks: one being filled, one sealed and with the flusher. Sealing again before that one was recycle()d would hand its memory to the
    // feeders while the flusher still reads it
    if (!bank_[next].clean) throw std::logic_error("Ingress::seal: the batch sealed before this one has not been recycled");
    Bank &bk = bank_[done];
    SealedBatch &s = sealed_[done];
    s.rows = 0;
    for (size_t k = 0; k < shard_.size(); k++) {
        const Shard &sh = shard_[k];
        SealedShard &ss = s.shard[k];
        uint32_t rounds = 0;
        ss.events = 0;
        for (Conn &c : c_) { rounds = std::max(rounds, c.max_depth[k]); ss.events += c.rows[k]; c.rows[k] = 0; c.max_depth[k] = 0; }
        ss.batch.rounds = rounds;
        ss.batch.count = sh.count;
        ss.batch.gid = nullptr;
        ss.batch.head = bk.buf.head + sh.cell_off;
        ss.batch.abcd = bk.buf.abcd + sh.cell_off;
        ss.batch.entry_terms = bk.buf.entry_terms ? bk.buf.entry_terms + sh.term_off : nullptr;
        ss.batch.entry_count = std::min<uint64_t>(bk.terms_used[k].load(std::memory_order_relaxed), sh.term_cap);
        ss.origin = bk.origin.data() + sh.cell_off;
        ss.first_gid = sh.first;
        s.rows += ss.events;
        bk.dirty_rounds[k] = rounds;
    }
    s.batch = s.shard[0].batch;
    s.origin = s.shard[0].origin;
    s.wide = std::move(bk.wide);
    std::sort(s.wide.begin(), s.wide.end(), [](const HeldRow &x, const HeldRow &y) { return x.gid < y.gid; });
    bk.wide.clear();
    bk.clean = false;
    // open the other bank, oldest held rows first
    Bank &nb = bank_[next];
    nb.clean = false;
    fill_ = next;
    // what the readers held back since the last seal joins its group's queue, oldest ticket first ...
    std::vector<Waiting> fresh;
    for (uint32_t i = 0; i < c_.size(); i++) {
        for (HeldRow &h : c_[i].held) fresh.push_back(Waiting{std::move(h), i});
        c_[i].backlogged.fetch_add(c_[i].held.size(), std::memory_order_relaxed);
        c_[i].held.clear();
        c_[i].held_count.store(0, std::memory_order_relaxed);
    }
    std::sort(fresh.begin(), fresh.end(), [](const Waiting &x, const Waiting &y) { return x.row.ticket < y.row.ticket; });
    for (Waiting &w : fresh) backlog_[w.row.gid].push_back(std::move(w));
    // ... and every queue hands over as many of its oldest rows as the new bank takes; a group that keeps a backlog is closed for this batch,
    // so whatever arrives for it now queues behind
    for (auto it = backlog_.begin(); it != backlog_.end();) {
        std::deque<Waiting> &q = it->second;
        while (!q.empty()) {
            Waiting &w = q.front();
            HeldRow &h = w.row;
            if (!place(nb, c_[w.conn], h.gid, h.head, h.a, h.b, h.c, h.d, h.terms.data(), h.terms.size(), h.from, h.body.data(), h.body.size())) break;
            c_[w.conn].backlogged.fetch_sub(1, std::memory_order_relaxed);
            q.pop_front();
        }
        if (q.empty()) it = backlog_.erase(it);
        else { nb.depth[it->first].fetch_or(CLOSED, std::memory_order_relaxed); ++it; }
    }
    return s;
}

uint64_t Ingress::held() const
{
    StandBack standing_back(sealing_);
    std::unique_lock<std::shared_mutex> lk(mu_);
    uint64_t n = 0;
    for (const Conn &c : c_) n += c.held.size();
    for (const auto &q : backlog_) n += q.second.size();
    return n;
}

size_t Ingress::drop_rows_of(uint32_t gid)
{
    StandBack standing_back(sealing_);
    std::unique_lock<std::shared_mutex> lk(mu_);
    size_t n = 0;
    for (Conn &c : c_) {
        const size_t before = c.held.size();
        c.held.erase(std::remove_if(c.held.begin(), c.held.end(), [gid](const HeldRow &h) { return h.gid == gid; }), c.held.end());
        n += before - c.held.size();
        c.held_count.store(c.held.size(), std::memory_order_relaxed);
    }
    auto it = backlog_.find(gid);
    if (it != backlog_.end()) {
        for (const Waiting &w : it->second) c_[w.conn].backlogged.fetch_sub(1, std::memory_order_relaxed);
        n += it->second.size();
        backlog_.erase(it);
        if (gid < groups_) bank_[fill_].depth[gid].fetch_and(~CLOSED, std::memory_order_relaxed);     // (it was closed for this batch because of that backlog)
    }
    refused_.fetch_add(n, std::memory_order_relaxed);
    return n;
}

size_t Ingress::emit(const SealedBatch &b, const rg_reply_t *reply, std::vector<std::string> &out, size_t cell_begin, size_t cell_end, uint32_t only_conn,
                     uint32_t shard) const
{
    static const Method METHOD_OF_KIND[16] = {M_NONE, M_APPEND_ENTRIES, M_NONE, M_NONE, M_REQUEST_VOTE, M_PRE_VOTE, M_NONE, M_NONE,
                                               M_NONE, M_NONE, M_NONE, M_INSTALL_SNAPSHOT, M_NONE, M_NONE, M_NONE, M_NONE};
    static const char *const SCOPE_OF_METHOD[] = {"", "appendEntries:", "preVote:", "requestVote:", "installSnapshot:"};
    size_t made = 0;
    Frame f;
    f.type = ACK;
    const SealedShard &sb = b.shard[shard];
    const size_t cells = std::min(cell_end, (size_t)sb.batch.rounds * sb.batch.count);
    for (size_t cell = cell_begin; cell < cells; cell++) {
        const Method m = METHOD_OF_KIND[RG_HDR_KIND(sb.batch.head[cell].hdr)];
        if (m == M_NONE || !(reply[cell].flags & RG_F_REPLIED)) continue;            // an empty cell, a response row, or a handler that died
        const Origin o = sb.origin[cell];
        if (o.conn == NO_CONN || o.conn >= out.size() || (only_conn != NO_CONN && o.conn != only_conn)) continue;
        f.sequence = o.sequence;
        f.head.assign(SCOPE_OF_METHOD[m]);                                           // "<method>:<contextId>", as the request carried it
        index_.append_id(sb.first_gid + (uint32_t)(cell % sb.batch.count), f.head);
        f.body.clear();
        codec_.encode_response(Response{reply[cell].resp_term, (reply[cell].flags & RG_F_SUCCESS) != 0}, f.body);
        encode_frame(f, false, out[o.conn]);
        made++;
    }
    return made;
}

uint32_t Ingress::emit_wide(const HeldRow &row, const rg_reply_t &reply, std::string &out) const
{
    const char *scope = nullptr;
    switch (RG_HDR_KIND(row.head.hdr)) {
    case RG_EV_AE_REQ: scope = "appendEntries:"; break;
    case RG_EV_PV_REQ: scope = "preVote:"; break;
    case RG_EV_RV_REQ: scope = "requestVote:"; break;
    case RG_EV_IS_REQ: scope = "installSnapshot:"; break;
    default: return NO_CONN;
    }
    if (!(reply.flags & RG_F_REPLIED) || row.from.conn == NO_CONN) return NO_CONN;
    Frame f;
    f.type = ACK;
    f.sequence = row.from.sequence;
    f.head.assign(scope);
    index_.append_id(row.gid, f.head);
    codec_.encode_response(Response{reply.resp_term, (reply.flags & RG_F_SUCCESS) != 0}, f.body);
    encode_frame(f, false, out);
    return row.from.conn;
}

// ---- repair_need_host ----------------------------------------------------------------------------------------------------------------------
static bool has_logfx_item(uint32_t flags)
{
    return (flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) != 0 || RG_F_STATUS(flags) == RG_NEED_HOST;
}

int64_t repair_need_host(const SealedBatch &whole, rg_reply_t *reply, const rg_logfx_t *logfx, bool packed, RepairHost &host, uint32_t shard)
{
    const SealedShard &b = whole.shard[shard];
    const uint32_t G = b.batch.count, R = b.batch.rounds;
    const size_t cells = (size_t)G * R;
    struct Broken { uint32_t gid, round; bool missed; int64_t need; };            // the group's next undecided row; need = logfx.log_from of its miss
    std::vector<Broken> broken;
    {
        std::vector<uint8_t> seen(G, 0);
        size_t item = 0;                                                          // position in the packed list
        for (size_t cell = 0; cell < cells; cell++) {
            if (RG_HDR_KIND(b.batch.head[cell].hdr) == RG_EV_NONE) continue;
            const uint32_t fl = reply[cell].flags, g = (uint32_t)(cell % G);
            const bool marked = has_logfx_item(fl);
            if (RG_F_STATUS(fl) == RG_NEED_HOST && !seen[g]) {
                seen[g] = 1;
                broken.push_back(Broken{g, (uint32_t)(cell / G), true, packed ? logfx[item].log_from : logfx[cell].log_from});
            }
            item += marked;
        }
    }
    if (broken.empty()) return 0;
    std::sort(broken.begin(), broken.end(), [](const Broken &x, const Broken &y) { return x.gid < y.gid; });     // sparse rows ascend
    int64_t decided = 0;
    std::vector<uint32_t> gid;
    std::vector<rg_ev_head_t> head;
    std::vector<rg_ev_pair_t> ab, cd, hint;
    std::vector<int64_t> terms;
    std::vector<rg_reply_t> rep;
    std::vector<rg_logfx_t> lfx;
    std::vector<rg_persist_t> per;
    while (!broken.empty()) {
        const size_t n = broken.size();
        gid.resize(n); head.resize(n); ab.resize(n); cd.resize(n); hint.assign(n, rg_ev_pair_t{0, 0});
        rep.assign(n, rg_reply_t{0, 0, 0}); lfx.assign(n, rg_logfx_t{0, 0}); per.assign(n, rg_persist_t{0, 0, 0});
        terms.clear();
        for (size_t i = 0; i < n; i++) {
            const Broken &k = broken[i];
            const size_t cell = (size_t)k.round * G + k.gid;
            rg_ev_head_t h = b.batch.head[cell];
            const rg_ev_quad32_t q = b.batch.abcd[cell];
            const uint32_t cnt = RG_HDR_N(h.hdr);
            const bool ae = RG_HDR_KIND(h.hdr) == RG_EV_AE_REQ;
            const size_t t0 = terms.size();
            if (ae && cnt > 0) {                                                  // the wide form of the compact row
                for (uint32_t e = 0; e < cnt; e++) terms.push_back((h.hdr & RG_HDR_SAME_TERM) ? (int64_t)h.aux : (int64_t)b.batch.entry_terms[h.aux + e]);
                h.hdr &= ~RG_HDR_SAME_TERM;
                h.aux = (uint32_t)t0;
            }
            if (k.missed) {
                h.hdr |= RG_HDR_HINT_BIT;
                if (ae) {
                    const int64_t prev = q.b, epoch = host.epoch_index(k.gid);
                    int64_t first = prev + 1;
                    uint32_t skip = 0;
                    if (cnt > 0 && first <= epoch) {                              // entries at or below the epoch are purged first (member/Follower.java:209-221)
                        skip = (uint32_t)std::min<int64_t>(cnt, epoch - first + 1);
                        first += skip;
                    }
                    hint[i] = rg_ev_pair_t{host.term_at(k.gid, prev), cnt > skip ? host.conflict(k.gid, first, terms.data() + t0 + skip, cnt - skip) : 0};
                } else {
                    hint[i] = rg_ev_pair_t{k.need, host.term_at(k.gid, k.need)};
                }
            }
            gid[i] = k.gid; head[i] = h; ab[i] = rg_ev_pair_t{q.a, q.b}; cd[i] = rg_ev_pair_t{q.c, q.d};
        }
        rg_batch_t in{};
        in.rounds = 1; in.count = (uint32_t)n; in.gid = gid.data(); in.head = head.data(); in.ab = ab.data(); in.cd = cd.data();
        in.entry_terms = terms.empty() ? nullptr : terms.data(); in.entry_count = terms.size(); in.hint = hint.data();
        const rg_outcome_t out{rep.data(), lfx.data(), per.data()};
        if (host.submit(in, out) != 0) return -1;
        std::vector<Broken> next;
        for (size_t i = 0; i < n; i++) {
            Broken k = broken[i];
            const size_t cell = (size_t)k.round * G + k.gid;
            if (RG_F_STATUS(rep[i].flags) == RG_NEED_HOST) {
                if (k.missed) return -1;                                          // a hinted row must apply
                k.missed = true; k.need = lfx[i].log_from;
                next.push_back(k);
                continue;
            }
            reply[cell] = rep[i];
            host.applied(k.gid, cell, rep[i], lfx[i], per[i]);
            decided++;
            uint32_t r = k.round + 1;
            while (r < R && RG_HDR_KIND(b.batch.head[(size_t)r * G + k.gid].hdr) == RG_EV_NONE) r++;
            if (r < R) next.push_back(Broken{k.gid, r, false, 0});
        }
        broken.swap(next);
    }
    return decided;
}

}  // namespace wire
}  // namespace rafting
