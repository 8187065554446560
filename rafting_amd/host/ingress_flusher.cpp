// ingress_flusher.cpp — see ingress_flusher.hpp.
#include "ingress_flusher.hpp"

#include <algorithm>
#include <stdexcept>
#include <thread>

using raftgpu::host::Entry;
using raftgpu::host::RaftLog;
using raftgpu::host::StableStore;

namespace rafting {
namespace wire {

IngressFlusher::IngressFlusher(rg_table_t *table, Ingress &ing, const KryoBodyCodec &codec, std::function<RaftLog &(uint32_t)> log_of,
                               std::vector<int64_t> term_of_group, StableStore *store, bool wide_kernel)
    : IngressFlusher(std::vector<rg_table_t *>{table}, ing, codec, std::move(log_of), std::move(term_of_group), store, wide_kernel) {}

IngressFlusher::IngressFlusher(std::vector<rg_table_t *> tables, Ingress &ing, const KryoBodyCodec &codec, std::function<RaftLog &(uint32_t)> log_of,
                               std::vector<int64_t> term_of_group, StableStore *store, bool wide_kernel)
    : tables_(std::move(tables)), ing_(ing), codec_(codec), log_of_(std::move(log_of)), term_(std::move(term_of_group)), store_(store), wide_kernel_(wide_kernel)
{
    if (ing_.shards() != tables_.size()) throw std::invalid_argument("IngressFlusher: one table per shard of the ingress");
    rep_.resize(tables_.size()); lfx_.resize(tables_.size()); per_.resize(tables_.size());
    ing_.retain_bodies(true);
    // the ordering contract of INTEGRATION.md section 1, enforced: a timeout row names the participant whose ticket fired (context/RaftRoutine.java:65-77)
    // (this CHANGES the tables for every other user of theirs — an RG_EV_TIMEOUT row with aux == 0 is RG_BAD_EVENT from here on — and is part of what a
    //  recorded tick bakes in: create rg_tick_* / rg_tick2_* recordings on these tables AFTER the flusher; an older one is refused at its next launch)
    for (rg_table_t *t : tables_)
        if (rg_table_option(t, RG_OPT_REQUIRE_FENCED_TIMEOUTS, 1) != 0)
            throw std::runtime_error(std::string("IngressFlusher: rg_table_option(RG_OPT_REQUIRE_FENCED_TIMEOUTS): ") + rg_last_error(t));
}

// One applied row -> the host-owned plugins, in the handler's order.
void IngressFlusher::apply(uint32_t gid, rg_ev_head_t head, int64_t a, int64_t b, const char *body, size_t body_len, const rg_reply_t &rep,
                           const rg_logfx_t &lfx, const rg_persist_t &per, std::vector<StableStore::Record> &dirty)
{
    const uint32_t f = rep.flags, kind = RG_HDR_KIND(head.hdr);
    RaftLog &log = log_of_(gid);
    if (kind == RG_EV_AE_REQ) {
        if (f & RG_F_LOG_TRUNC) { log.truncate(lfx.log_from); st_.truncated++; }
        if (f & RG_F_LOG_APPEND) {
            std::vector<Entry> sub;                                // the entries of ITS request at or above log_from (storage/RocksLog.java:169-196)
            if (body) codec_.entries(body, body_len, [&](int64_t index, int64_t term, const char *, size_t) { if (index >= lfx.log_from) sub.push_back(Entry{index, term}); });
            log.append(sub);
            st_.appended += sub.size();
        }
    } else if (kind == RG_EV_CLIENT_APPEND && (f & RG_F_LOG_APPEND)) {
        for (uint32_t k = 0; k < RG_HDR_N(head.hdr); k++) log.newEntry(term_[gid]);
        st_.appended += RG_HDR_N(head.hdr);
    } else if (kind == RG_EV_LOG_FLUSH && RG_F_STATUS(f) == RG_OK) {
        log.flush(a, b);
    }
    if (f & RG_F_PERSIST) {                                        // durable BEFORE the response leaves: collected, written once per batch
        term_[gid] = per.term;
        dirty.push_back(StableStore::Record{gid, per.term, per.voted_for});
    }
    if (f & RG_F_COMMIT) { log.markCommitted(lfx.commit_index); st_.committed++; }
    // the host's reaction (timers, emits, role changes) runs AFTER the batch's durable write: a new term or self-vote that on_row broadcasts
    // must be on disk first (member/RaftMember.java:25)
    if (on_row) reactions_.push_back(Reaction{gid, head, rep});
}

struct IngressFlusher::Host : RepairHost {               // (the gids repair_need_host hands over are the TABLE's: + first for the host's)
    IngressFlusher &fl;
    const SealedBatch &b;
    std::vector<StableStore::Record> &dirty;
    uint32_t shard, first;
    Host(IngressFlusher &f, const SealedBatch &batch, std::vector<StableStore::Record> &d, uint32_t s) : fl(f), b(batch), dirty(d), shard(s), first(batch.shard[s].first_gid) {}
    int64_t term_at(uint32_t gid, int64_t index) override { auto e = fl.log_of_(first + gid).get(index); return e ? e->term : -1; }
    int64_t conflict(uint32_t gid, int64_t first_index, const int64_t *terms, uint32_t n) override
    {
        std::vector<Entry> es;
        for (uint32_t k = 0; k < n; k++) es.push_back(Entry{first_index + (int64_t)k, terms[k]});
        auto c = fl.log_of_(first + gid).conflict(es);
        return c ? c->index : 0;
    }
    int64_t epoch_index(uint32_t gid) override { return fl.log_of_(first + gid).epoch().index; }
    int submit(const rg_batch_t &in, const rg_outcome_t &out) override { return rg_submit(fl.tables_[shard], &in, &out, RG_MEM_HOST); }
    void applied(uint32_t gid, size_t cell, const rg_reply_t &rep, const rg_logfx_t &lfx, const rg_persist_t &per) override
    {
        size_t n = 0;
        const char *body = fl.ing_.body(b, shard, cell, n);
        const rg_ev_quad32_t q = b.shard[shard].batch.abcd[cell];
        fl.apply(first + gid, b.shard[shard].batch.head[cell], q.a, q.b, body, n, rep, lfx, per, dirty);
        fl.st_.repaired++;
    }
};

// the launch of one shard on its table (synchronous: the outcome rows are in rep_ / lfx_ / per_[s] when it returns)
int IngressFlusher::submit_shard(const SealedBatch &b, uint32_t s)
{
    const rg_batch32_t &sb = b.shard[s].batch;
    const uint32_t G = sb.count, R = sb.rounds;
    const size_t cells = (size_t)G * R;
    rep_[s].assign(cells, rg_reply_t{0, 0, 0}); lfx_[s].assign(cells, rg_logfx_t{0, 0}); per_[s].assign(cells, rg_persist_t{0, 0, 0});
    if (cells == 0) return 0;
    const rg_outcome_t o{rep_[s].data(), lfx_[s].data(), per_[s].data()};
    if (!wide_kernel_) return rg_submit32(tables_[s], &sb, &o, RG_MEM_HOST);
    // the same rows as an rg_batch_t (the inverse of rg_batch32_pack)
    std::vector<rg_ev_head_t> head(sb.head, sb.head + cells);
    std::vector<rg_ev_pair_t> ab(cells), cd(cells);
    std::vector<int64_t> terms;
    for (size_t i = 0; i < cells; i++) {
        const rg_ev_quad32_t q = sb.abcd[i];
        ab[i] = rg_ev_pair_t{q.a, q.b}; cd[i] = rg_ev_pair_t{q.c, q.d};
        const uint32_t n = RG_HDR_N(head[i].hdr);
        if (RG_HDR_KIND(head[i].hdr) == RG_EV_AE_REQ && n > 0) {
            const uint32_t at = (uint32_t)terms.size();
            for (uint32_t k = 0; k < n; k++) terms.push_back((head[i].hdr & RG_HDR_SAME_TERM) ? (int64_t)head[i].aux : (int64_t)sb.entry_terms[head[i].aux + k]);
            head[i].hdr &= ~RG_HDR_SAME_TERM;
            head[i].aux = at;
        }
    }
    rg_batch_t in{};
    in.rounds = R; in.count = G; in.head = head.data(); in.ab = ab.data(); in.cd = cd.data();
    in.entry_terms = terms.empty() ? nullptr : terms.data(); in.entry_count = terms.size();
    return rg_submit(tables_[s], &in, &o, RG_MEM_HOST);
}

int64_t IngressFlusher::flush(std::vector<std::string> &out)
{
    const SealedBatch &b = ing_.seal();
    if (b.rows == 0 && b.wide.empty()) { ing_.recycle(b); return 0; }
    reactions_.clear();
    // every way out recycles the sealed batch: the next flush() seals again (an error leaves the table and the logs where the failed call left
    // them — the caller stops, as with any table error — but does not turn the NEXT call into a logic_error out of seal())
    struct Recycle { Ingress &ing; const SealedBatch &b; bool armed = true; ~Recycle() { if (armed) ing.recycle(b); } } recycle{ing_, b};
    const uint32_t S = (uint32_t)tables_.size();
    std::vector<StableStore::Record> dirty;
    int64_t decided = 0;
    {
        // the shards' launches side by side: tables are independent (one stream each); the first on this thread
        // (the side threads are joined on EVERY way out of this block — an exception out of shard 0's submit, e.g. bad_alloc, would
        // otherwise destroy joinable threads (std::terminate) while they still read the sealed batch; what a shard throws is reported, not lost)
        std::vector<int> rc(S, 0);
        std::vector<std::string> thrown(S);
        std::vector<std::thread> side;
        struct Join { std::vector<std::thread> &ts; ~Join() { for (auto &t : ts) if (t.joinable()) t.join(); } } join_all{side};
        auto run = [&](uint32_t s) {
            try { rc[s] = submit_shard(b, s); }
            catch (const std::exception &e) { rc[s] = -1; thrown[s] = e.what(); }
        };
        for (uint32_t s = 1; s < S; s++) side.emplace_back(run, s);
        run(0);
        for (auto &t : side) t.join();
        std::string all;
        for (uint32_t s = 0; s < S; s++)
            if (rc[s] != 0) all += (all.empty() ? "" : "; ") + ("shard " + std::to_string(s) + ": ") + (thrown[s].empty() ? std::string(rg_last_error(tables_[s])) : thrown[s]);
        if (!all.empty()) { err_ = all; return -1; }
    }
    for (uint32_t s = 0; s < S; s++) {
        const rg_batch32_t &sb = b.shard[s].batch;
        const uint32_t G = sb.count, first = b.shard[s].first_gid;
        const size_t cells = (size_t)G * sb.rounds;
        if (cells == 0) continue;
        // the rows the launch applied, in (round, group) order = every group's own order
        for (size_t cell = 0; cell < cells; cell++) {
            if (RG_HDR_KIND(sb.head[cell].hdr) == RG_EV_NONE) continue;
            decided++;
            const uint32_t st = RG_F_STATUS(rep_[s][cell].flags);
            if (st == RG_NEED_HOST || st == RG_SKIPPED_AFTER_NEED_HOST) continue;
            size_t n = 0;
            const char *body = ing_.body(b, s, cell, n);
            const rg_ev_quad32_t q = sb.abcd[cell];
            apply(first + (uint32_t)(cell % G), sb.head[cell], q.a, q.b, body, n, rep_[s][cell], lfx_[s][cell], per_[s][cell], dirty);
        }
        Host host(*this, b, dirty, s);
        if (repair_need_host(b, rep_[s].data(), lfx_[s].data(), false, host, s) < 0) { err_ = std::string("repair: ") + rg_last_error(tables_[s]); return -1; }
    }
    // rows beside the batch (a value beyond 2^31 ...): one sparse row each, the same hint protocol
    std::vector<std::pair<size_t, rg_reply_t>> wide_replies;
    for (size_t i = 0; i < b.wide.size(); i++) {
        const HeldRow &w = b.wide[i];
        uint32_t ws = S - 1;                                       // the shard (table) the row's group lives in
        while (ws > 0 && b.shard[ws].first_gid > w.gid) ws--;
        const uint32_t tgid = w.gid - b.shard[ws].first_gid;
        rg_table_t *tbl = tables_[ws];
        rg_ev_head_t h = w.head;
        h.hdr &= ~(RG_HDR_SAME_TERM | RG_HDR_HINT_BIT);
        h.aux = RG_HDR_KIND(h.hdr) == RG_EV_AE_REQ ? 0u : h.aux;
        rg_ev_pair_t ab{w.a, w.b}, cd{w.c, w.d}, hint{0, 0};
        rg_reply_t rep{0, 0, 0}; rg_logfx_t lfx{0, 0}; rg_persist_t per{0, 0, 0};
        for (int attempt = 0; attempt < 2; attempt++) {
            rg_batch_t in{};
            in.rounds = 1; in.count = 1; in.gid = &tgid; in.head = &h; in.ab = &ab; in.cd = &cd;
            in.entry_terms = w.terms.empty() ? nullptr : w.terms.data(); in.entry_count = w.terms.size(); in.hint = &hint;
            const rg_outcome_t o{&rep, &lfx, &per};
            if (rg_submit(tbl, &in, &o, RG_MEM_HOST) != 0) { err_ = rg_last_error(tbl); return -1; }
            if (RG_F_STATUS(rep.flags) != RG_NEED_HOST) break;
            if (attempt == 1) { err_ = "a row beside the batch still answers RG_NEED_HOST after its hint"; return -1; }     // as repair_need_host does
            RaftLog &log = log_of_(w.gid);
            h.hdr |= RG_HDR_HINT_BIT;
            if (RG_HDR_KIND(h.hdr) == RG_EV_AE_REQ) {
                auto pt = log.get(w.b);
                std::vector<Entry> rest;
                for (size_t k = 0; k < w.terms.size(); k++) if (w.b + 1 + (int64_t)k > log.epoch().index) rest.push_back(Entry{w.b + 1 + (int64_t)k, w.terms[k]});
                auto cf = rest.empty() ? std::nullopt : log.conflict(rest);
                hint = rg_ev_pair_t{pt ? pt->term : -1, cf ? cf->index : 0};
            } else {
                auto t = log.get(lfx.log_from);
                hint = rg_ev_pair_t{lfx.log_from, t ? t->term : -1};
            }
        }
        apply(w.gid, w.head, w.a, w.b, w.body.empty() ? nullptr : w.body.data(), w.body.size(), rep, lfx, per, dirty);
        wide_replies.emplace_back(i, rep);
        decided++;
        st_.wide++;
    }
    if (store_ && !dirty.empty()) { store_->persist(dirty); st_.persisted += dirty.size(); }      // N3: before any reply of this batch leaves
    for (const Reaction &r : reactions_) on_row(r.gid, r.head, r.reply);                           // ... and before anything on_row sends
    for (uint32_t s = 0; s < S; s++)
        if (!rep_[s].empty()) st_.frames += ing_.emit(b, rep_[s].data(), out, 0, (size_t)-1, NO_CONN, s);
    for (auto &wr : wide_replies) {
        const HeldRow &w = b.wide[wr.first];
        if (w.from.conn < out.size()) st_.frames += ing_.emit_wide(w, wr.second, out[w.from.conn]) != NO_CONN;
    }
    recycle.armed = false;
    ing_.recycle(b);
    st_.batches++;
    st_.rows += (uint64_t)decided;
    return decided;
}

}  // namespace wire
}  // namespace rafting
