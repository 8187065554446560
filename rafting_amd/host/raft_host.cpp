// raft_host.cpp — see raft_host.hpp.
#include "raft_host.hpp"

#include <algorithm>
#include <numeric>

namespace raftgpu {
namespace host {

// ---- MemoryLog: the observable behaviour of command/storage/RocksLog.java --------------------------

std::optional<Entry> MemoryLog::last() const
{
    if (terms_.empty()) return std::nullopt;
    return Entry{first_ + (int64_t)terms_.size() - 1, terms_.back()};
}

std::optional<Entry> MemoryLog::get(int64_t index) const          // RocksLog.java:122-128
{
    if (terms_.empty() || index < first_ || index >= first_ + (int64_t)terms_.size()) return std::nullopt;
    return Entry{index, terms_[(size_t)(index - first_)]};
}

std::optional<Entry> MemoryLog::conflict(const std::vector<Entry> &entries) const   // RocksLog.java:199-216
{
    for (const Entry &e : entries) {
        auto g = get(e.index);
        if (!g) return std::nullopt;
        if (g->term != e.term) return Entry{e.index, g->term};
    }
    return std::nullopt;
}

void MemoryLog::truncate(int64_t index)                            // RocksLog.java:219-225
{
    auto l = last();
    if (!l || l->index < index) return;
    if (index <= first_) { terms_.clear(); return; }
    terms_.resize((size_t)(index - first_));
}

void MemoryLog::append(const std::vector<Entry> &entries)          // RocksLog.java:169-196
{
    if (entries.empty()) return;
    int64_t prevLogIndex = epoch_.index;
    bool valid = false;
    if (!terms_.empty() && first_ <= entries[0].index) {           // seekForPrev
        prevLogIndex = std::min(last()->index, entries[0].index);
        valid = true;
    }
    if (!valid && entries[0].index != prevLogIndex + 1) throw std::logic_error("log index should follow epoch closely");
    const Entry *prev = nullptr;
    for (const Entry &e : entries) {
        if (prev && prev->index != e.index - 1) throw std::logic_error("log index is not continuous");
        if (e.index > prevLogIndex) {
            if (!prev || prev->index == prevLogIndex)
                if (prevLogIndex != e.index - 1) throw std::logic_error("log index is not continuous");
            if (terms_.empty()) { first_ = e.index; terms_.push_back(e.term); }
            else if (e.index <= last()->index) terms_[(size_t)(e.index - first_)] = e.term;
            else terms_.push_back(e.term);
        }
        prev = &e;
    }
}

Entry MemoryLog::newEntry(int64_t term)                            // RocksLog.java:82-89
{
    auto l = last();
    const int64_t index = l ? l->index + 1 : 1;
    if (terms_.empty()) first_ = index;
    terms_.push_back(term);
    return Entry{index, term};
}

bool MemoryLog::markCommitted(int64_t commitIndex)                 // RocksLog.java:100-109
{
    if (commitIndex < commit_) throw std::logic_error("rollback is not allowed");
    if (commitIndex > commit_) { commit_ = commitIndex; return true; }
    return false;
}

void MemoryLog::flush(int64_t index, int64_t term)                 // RocksLog.java:228-242
{
    if (index < epoch_.index) throw std::out_of_range("flush below the epoch");
    if (!terms_.empty()) {
        if (index > last()->index) terms_.clear();
        else if (index > first_) { terms_.erase(terms_.begin(), terms_.begin() + (index - first_)); first_ = index; }
    }
    epoch_ = Entry{index, term};
}

std::vector<Entry> MemoryLog::runs() const
{
    std::vector<Entry> r;
    for (size_t i = 0; i < terms_.size(); i++)
        if (i == 0 || terms_[i] != terms_[i - 1]) r.push_back(Entry{first_ + (int64_t)i, terms_[i]});
    return r;
}

// ---- one-group state image for rg_load_state / rg_read_state -------------------------------------

namespace {
struct StateBuf {
    int64_t current_term = 0, elected_term = 0, commit_index = 0, epoch_index = 0, epoch_term = 0, first_index = 0, last_index = 0;
    int32_t voted_for = RG_NO_NODE, role = RG_FOLLOWER, current_leader = RG_NO_NODE, votes = 1;
    uint8_t timeout_detected = 0, repl_prepared = 0;
    uint32_t role_epoch = 1, elected_epoch = 0, run_count = 0, run_offset = 0;
    std::vector<int64_t> run_start, run_term, pe, pn, pm;
    std::vector<int32_t> pr;
    std::vector<uint8_t> pp;
    explicit StateBuf(uint32_t followers, size_t runs = RG_TERM_RUNS)
        : run_start(runs), run_term(runs), pe(followers), pn(followers), pm(followers), pr(followers), pp(followers) {}
    rg_group_state_t view()
    {
        rg_group_state_t s{};
        s.current_term = &current_term; s.voted_for = &voted_for; s.role = &role; s.current_leader = &current_leader;
        s.timeout_detected = &timeout_detected; s.repl_prepared = &repl_prepared; s.role_epoch = &role_epoch;
        s.votes = &votes; s.elected_epoch = &elected_epoch; s.elected_term = &elected_term; s.commit_index = &commit_index;
        s.epoch_index = &epoch_index; s.epoch_term = &epoch_term; s.first_index = &first_index; s.last_index = &last_index;
        s.run_count = &run_count; s.run_offset = &run_offset; s.run_start = run_start.data(); s.run_term = run_term.data();
        s.peer_last_epoch = pe.data(); s.peer_next_index = pn.data(); s.peer_match_index = pm.data();
        s.peer_rejection = pr.data(); s.peer_pending = pp.data();
        return s;
    }
};
}  // namespace

// ---- RaftContext: every call queues one row -----------------------------------------------------------

Ticket RaftContext::appendEntries(int64_t term, ID leaderId, int64_t prevLogIndex, int64_t prevLogTerm,
                                  const std::vector<Entry> &entries, int64_t leaderCommit)
{
    // the wire carries entry terms only: indices are prevLogIndex+1+k, as Leader.replicateLog builds them
    // (member/Leader.java:192-212); anything else is what RocksLog would reject as "not continuous"
    for (size_t k = 0; k < entries.size(); k++)
        if (entries[k].index != prevLogIndex + 1 + (int64_t)k) throw std::logic_error("log index is not continuous");
    if (entries.size() > RG_MAX_AE_ENTRIES) throw std::length_error("too many entries in one AppendEntries");
    ContextManager::Row r{this, RG_HDR_MAKE(RG_EV_AE_REQ, leaderId, 0, entries.size()), 0, term, prevLogIndex, prevLogTerm,
                          leaderCommit, entries};
    return mgr_->enqueue(*this, std::move(r));
}
Ticket RaftContext::preVote(int64_t term, ID cand, int64_t lastLogIndex, int64_t lastLogTerm)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(RG_EV_PV_REQ, cand, 0, 0), 0, term, lastLogIndex, lastLogTerm, 0, {}});
}
Ticket RaftContext::requestVote(int64_t term, ID cand, int64_t lastLogIndex, int64_t lastLogTerm)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(RG_EV_RV_REQ, cand, 0, 0), 0, term, lastLogIndex, lastLogTerm, 0, {}});
}
Ticket RaftContext::installSnapshot(int64_t term, ID leaderId, int64_t lastIncludedIndex, int64_t lastIncludedTerm, bool installed)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(RG_EV_IS_REQ, leaderId, installed, 0), 0, term, lastIncludedIndex, lastIncludedTerm, 0, {}});
}
Ticket RaftContext::onTimeout(uint32_t ticketEpoch)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(RG_EV_TIMEOUT, 0, 0, 0), ticketEpoch, 0, 0, 0, 0, {}});
}
Ticket RaftContext::onAppendEntriesResponse(ID peer, RaftResponse res, int64_t epochAtSend, int64_t lastIndexSent, uint32_t sentEpoch)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(RG_EV_AE_ACK, peer, res.success, 0), sentEpoch, res.term, epochAtSend, lastIndexSent, 0, {}});
}
Ticket RaftContext::onInstallSnapshotResponse(ID peer, RaftResponse res, int64_t epochAtSend, uint32_t sentEpoch)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(RG_EV_IS_ACK, peer, res.success, 0), sentEpoch, res.term, epochAtSend, 0, 0, {}});
}
Ticket RaftContext::onVoteResponse(bool pre, ID peer, RaftResponse res, uint32_t sentEpoch)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(pre ? RG_EV_PV_REPLY : RG_EV_RV_REPLY, peer, res.success, 0), sentEpoch, res.term, 0, 0, 0, {}});
}
Ticket RaftContext::acceptCommand(uint32_t commands)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(RG_EV_CLIENT_APPEND, 0, 0, commands), 0, 0, 0, 0, 0, {}});
}
Ticket RaftContext::compactLog(int64_t index, int64_t term)
{
    return mgr_->enqueue(*this, {this, RG_HDR_MAKE(RG_EV_LOG_FLUSH, 0, 0, 0), 0, index, term, 0, 0, {}});
}

// ---- ContextManager ----------------------------------------------------------------------------------

ContextManager::ContextManager(int device, uint32_t maxContexts, uint32_t clusterSize, ID self, bool preVote)
    : cluster_(clusterSize), self_(self), capacity_(maxContexts), queued_(maxContexts, 0)
{
    if (rg_table_create(device, maxContexts, clusterSize, (uint32_t)self, preVote ? 1 : 0, &table_) != 0)
        throw std::runtime_error(std::string("rg_table_create: ") + rg_last_error(nullptr));
}

ContextManager::~ContextManager() { rg_table_destroy(table_); }

RaftContext &ContextManager::createContext(const std::string &id, int64_t restoreTerm, ID restoreBallot)
{
    if (by_id_.count(id)) return *by_id_[id];
    if (contexts_.size() >= capacity_) throw std::length_error("context table is full");
    const uint32_t gid = (uint32_t)contexts_.size();
    contexts_.emplace_back(new RaftContext(this, id, gid, std::make_unique<MemoryLog>()));
    RaftContext &c = *contexts_.back();
    if (store_) store_->restore(gid, &restoreTerm, &restoreBallot);            // StableLock.restore
    c.term_ = restoreTerm; c.voted_for_ = restoreBallot;
    StateBuf sb(cluster_ - 1);                                    // RaftContext.initialize: switchTo(Follower, term, ballot)
    sb.current_term = restoreTerm; sb.voted_for = restoreBallot;
    rg_group_state_t v = sb.view();
    if (rg_load_state(table_, gid, 1, &v) != 0) throw std::runtime_error(rg_last_error(table_));
    by_id_[id] = &c;
    return c;
}

RaftContext *ContextManager::getContext(const std::string &id)
{
    auto it = by_id_.find(id);
    return it == by_id_.end() ? nullptr : it->second;
}

bool ContextManager::pending(const RaftContext &c) const { return queued_[c.gid()] != 0; }

Ticket ContextManager::enqueue(RaftContext &c, Row row)
{
    if (queued_[c.gid()]) throw std::logic_error("one event per context per flush: call flush() first");
    queued_[c.gid()] = 1;
    queue_.push_back(std::move(row));
    return queue_.size() - 1;
}

// one sparse single-round rg_submit over queue_ rows `which` (ascending gid); results land at the rows' queue positions
void ContextManager::configureTimers(int64_t electionMs, int64_t heartbeatMs, uint64_t seed)
{
    if (rg_timers_configure(table_, electionMs, heartbeatMs, seed) != 0) throw std::runtime_error(rg_last_error(table_));
}

void ContextManager::armTimers(int64_t now)
{
    if (rg_timers_arm(table_, now) != 0) throw std::runtime_error(rg_last_error(table_));
}

std::vector<std::pair<RaftContext *, uint32_t>> ContextManager::expiredTickets(int64_t now)
{
    std::vector<uint32_t> gids(contexts_.size() ? contexts_.size() : 1), epochs(gids.size());
    uint32_t n = 0;
    if (rg_timers_expired_epochs(table_, now, gids.data(), epochs.data(), (uint32_t)contexts_.size(), &n, RG_MEM_HOST) != 0)
        throw std::runtime_error(rg_last_error(table_));
    std::vector<std::pair<RaftContext *, uint32_t>> out;
    for (uint32_t i = 0; i < n && i < contexts_.size(); i++)
        if (gids[i] < contexts_.size()) out.emplace_back(contexts_[gids[i]].get(), epochs[i]);
    return out;
}

std::vector<RaftContext *> ContextManager::expiredTimers(int64_t now)
{
    std::vector<RaftContext *> out;
    for (auto &pr : expiredTickets(now)) out.push_back(pr.first);
    return out;
}

void ContextManager::submit(std::vector<size_t> &which, bool hinted, std::vector<rg_reply_t> &rep,
                            std::vector<rg_logfx_t> &lfx, std::vector<rg_persist_t> &per, int64_t now)
{
    std::sort(which.begin(), which.end(), [&](size_t x, size_t y) { return queue_[x].ctx->gid() < queue_[y].ctx->gid(); });
    const size_t n = which.size();
    std::vector<uint32_t> gid(n);
    std::vector<rg_ev_head_t> head(n);
    std::vector<rg_ev_pair_t> ab(n), cd(n), hint(hinted ? n : 0);
    std::vector<int64_t> terms;
    for (size_t k = 0; k < n; k++) {
        const Row &r = queue_[which[k]];
        gid[k] = r.ctx->gid();
        head[k] = {r.hdr, r.aux};
        if (RG_HDR_KIND(r.hdr) == RG_EV_AE_REQ) {
            head[k].aux = (uint32_t)terms.size();
            for (const Entry &e : r.entries) terms.push_back(e.term);
        }
        ab[k] = {r.a, r.b};
        cd[k] = {r.c, r.d};
        if (hinted) {
            // the host half of the NEED_HOST protocol: answer from the RaftLog this side owns
            RaftLog &log = r.ctx->replicatedLog();
            head[k].hdr |= RG_HDR_HINT_BIT;
            if (RG_HDR_KIND(r.hdr) == RG_EV_AE_REQ) {
                auto pt = log.get(r.b);
                std::vector<Entry> rest;                            // Follower.purgeEntries
                for (const Entry &e : r.entries) if (e.index > log.epoch().index) rest.push_back(e);
                auto cf = rest.empty() ? std::nullopt : log.conflict(rest);
                hint[k] = {pt ? pt->term : -1, cf ? cf->index : 0};
            } else {
                const int64_t idx = lfx[which[k]].log_from;
                auto t = log.get(idx);
                hint[k] = {idx, t ? t->term : -1};
            }
            hints_served_++;
        }
    }
    std::vector<rg_reply_t> o_rep(n);
    std::vector<rg_logfx_t> o_lfx(n);
    std::vector<rg_persist_t> o_per(n);
    rg_batch_t in{};
    in.rounds = 1; in.count = (uint32_t)n; in.gid = gid.data(); in.head = head.data(); in.ab = ab.data(); in.cd = cd.data();
    in.entry_terms = terms.empty() ? nullptr : terms.data(); in.entry_count = terms.size();
    in.hint = hinted ? hint.data() : nullptr;
    rg_outcome_t out{o_rep.data(), o_lfx.data(), o_per.data()};
    if (rg_submit(table_, &in, &out, RG_MEM_HOST) != 0) throw std::runtime_error(rg_last_error(table_));
    if (now >= 0 && rg_timers_update(table_, 1, (uint32_t)n, gid.data(), o_rep.data(), &now, RG_MEM_HOST) != 0)
        throw std::runtime_error(rg_last_error(table_));
    if (now >= 0 && rg_health_update(table_, 1, (uint32_t)n, gid.data(), head.data(), o_rep.data(), &now, RG_MEM_HOST) != 0)
        throw std::runtime_error(rg_last_error(table_));
    for (size_t k = 0; k < n; k++) { rep[which[k]] = o_rep[k]; lfx[which[k]] = o_lfx[k]; per[which[k]] = o_per[k]; }
    rows_decided_ += n;
}

void ContextManager::statFailure(const std::vector<RpcFailure> &failures, int64_t now)
{
    if (failures.empty()) return;
    std::vector<uint32_t> gid; std::vector<uint8_t> slot, flags;
    for (const RpcFailure &f : failures) {
        gid.push_back(f.ctx->gid()); slot.push_back((uint8_t)f.peer);
        flags.push_back((uint8_t)((f.unreachable ? 1 : 0) | (f.reject ? 2 : 0)));
    }
    if (rg_health_failure(table_, (uint32_t)gid.size(), gid.data(), slot.data(), flags.data(), now) != 0)
        throw std::runtime_error(rg_last_error(table_));
}

std::vector<uint8_t> ContextManager::isReady(int64_t now, int32_t criticalPoint, int64_t coolDownMs)
{
    std::vector<uint8_t> ready(capacity_);
    if (rg_ready(table_, now, criticalPoint, coolDownMs, ready.data(), RG_MEM_HOST) != 0) throw std::runtime_error(rg_last_error(table_));
    ready.resize(contexts_.size());
    return ready;
}

std::vector<Outcome> ContextManager::flush(int64_t now)
{
    const size_t n = queue_.size();
    std::vector<Outcome> result(n);
    if (n == 0) return result;
    std::vector<rg_reply_t> rep(n);
    std::vector<rg_logfx_t> lfx(n);
    std::vector<rg_persist_t> per(n);
    std::vector<size_t> which(n);
    std::iota(which.begin(), which.end(), 0);
    submit(which, false, rep, lfx, per, now);
    for (int attempt = 0; attempt < 2; attempt++) {                 // term-run cache misses: look up, resubmit with hints
        std::vector<size_t> miss;
        for (size_t i = 0; i < n; i++) if (RG_F_STATUS(rep[i].flags) == RG_NEED_HOST) miss.push_back(i);
        if (miss.empty()) break;
        submit(miss, true, rep, lfx, per, now);
    }
    if (store_) {                                                     // the durability barrier of the whole flush
        std::vector<StableStore::Record> dirty;
        for (size_t i = 0; i < n; i++)
            if (rep[i].flags & RG_F_PERSIST) dirty.push_back({queue_[i].ctx->gid(), per[i].term, per[i].voted_for});
        store_->persist(dirty);
    }
    for (size_t i = 0; i < n; i++) {
        Row &r = queue_[i];
        RaftContext &c = *r.ctx;
        const uint32_t f = rep[i].flags, kind = RG_HDR_KIND(r.hdr);
        RaftLog &log = c.replicatedLog();
        // 1. log mutations, in handler order (truncate -> append / newEntry / flush)
        if (kind == RG_EV_AE_REQ) {
            if (f & RG_F_LOG_TRUNC) log.truncate(lfx[i].log_from);
            if (f & RG_F_LOG_APPEND) {
                std::vector<Entry> sub;
                for (const Entry &e : r.entries) if (e.index >= lfx[i].log_from) sub.push_back(e);
                log.append(sub);
            }
        } else if (kind == RG_EV_CLIENT_APPEND && (f & RG_F_LOG_APPEND)) {
            for (uint32_t k = 0; k < RG_HDR_N(r.hdr); k++) log.newEntry(c.term_);
        } else if (kind == RG_EV_LOG_FLUSH && RG_F_STATUS(f) == RG_OK) {
            log.flush(r.a, r.b);
        }
        // 2. durable (term, votedFor) BEFORE the response may leave (RaftMember.java:25)
        if (f & RG_F_PERSIST) {
            c.term_ = per[i].term; c.voted_for_ = per[i].voted_for;
            if (persist_) persist_(c, per[i].term, per[i].voted_for);
        }
        c.role_ = (int)RG_F_ROLE(f);
        c.role_epoch_ = rep[i].role_epoch;
        // 3. commit (RaftContext.commitLog -> RaftLog.markCommitted -> RaftRoutine.commitState)
        if (f & RG_F_COMMIT) {
            log.markCommitted(lfx[i].commit_index);
            if (commit_) commit_(c, lfx[i].commit_index);
        }
        Outcome &o = result[i];
        o.status = RG_F_STATUS(f); o.flags = f; o.roleEpoch = rep[i].role_epoch; o.role = c.role_;
        if (f & RG_F_REPLIED) o.response = RaftResponse{rep[i].resp_term, (f & RG_F_SUCCESS) != 0};
        queued_[c.gid()] = 0;
    }
    queue_.clear();
    return result;
}

std::vector<SendPlan> ContextManager::replicateLog(const std::vector<RaftContext *> &ctxs, const std::vector<uint8_t> &heartbeat,
                                                   const std::vector<uint16_t> &inFlight)
{
    const size_t n = ctxs.size(), F = cluster_ - 1;
    std::vector<SendPlan> plans(n);
    if (n == 0) return plans;
    if (heartbeat.size() != n || (!inFlight.empty() && inFlight.size() != n * F)) throw std::invalid_argument("replicateLog: array sizes");
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return ctxs[x]->gid() < ctxs[y]->gid(); });
    std::vector<uint32_t> gid(n);
    std::vector<uint8_t> hb(n);
    std::vector<uint16_t> fl(inFlight.empty() ? 0 : n * F);
    for (size_t k = 0; k < n; k++) {
        gid[k] = ctxs[order[k]]->gid();
        hb[k] = heartbeat[order[k]];
        for (size_t j = 0; j < F && !fl.empty(); j++) fl[j * n + k] = inFlight[order[k] * F + j];   // wire is follower-major
    }
    std::vector<rg_send_head_t> head(n);
    std::vector<rg_send_t> send(n * F);
    if (rg_replicate(table_, (uint32_t)n, gid.data(), hb.data(), fl.empty() ? nullptr : fl.data(), head.data(), send.data(), RG_MEM_HOST) != 0)
        throw std::runtime_error(rg_last_error(table_));
    for (size_t k = 0; k < n; k++) {
        SendPlan &pl = plans[order[k]];
        pl.head = head[k];
        pl.to.resize(F);
        for (size_t j = 0; j < F; j++) pl.to[j] = send[j * n + k];
        for (rg_send_t &s : pl.to)
            if (s.kind == RG_SEND_NEED_HOST) {                       // prevLogTerm straight from the log this side owns
                auto e = ctxs[order[k]]->replicatedLog().get(s.prev_index);
                s.prev_term = e ? e->term : 0;
                s.kind = RG_SEND_APPEND;
                hints_served_++;
            }
    }
    return plans;
}

std::vector<PeerProgress> ContextManager::progress(const RaftContext &c)
{
    StateBuf sb(cluster_ - 1);
    rg_group_state_t v = sb.view();
    if (rg_read_state(table_, c.gid(), 1, &v) != 0) throw std::runtime_error(rg_last_error(table_));
    std::vector<PeerProgress> p(cluster_ - 1);
    for (uint32_t j = 0; j + 1 < cluster_; j++) p[j] = {sb.pe[j], sb.pn[j], sb.pm[j], sb.pp[j] != 0};
    return p;
}

}  // namespace host
}  // namespace raftgpu
