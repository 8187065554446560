// ref_driver.cpp — drives the mechanically translated reference classes (oracle/_ref/gen/) through the same wire structs
// as the C-ABI (include/raftgpu.h) and the hand-written oracle (oracle/raft_oracle.h).  TEST INFRASTRUCTURE ONLY.
//
// One RefEnv = one RaftContext of the reference with faked plugins (env.hpp).  An event row is played the way the
// reference would receive it:
//   requests            on the EventLoop thread:  ctx.participant().appendEntries / requestVote / preVote / installSnapshot
//                       (transport/NettyCluster.java:59-90 -> NettyNode.prepareLocalInvocation)
//   vote replies        the REAL closure the participant registered with Async.on (member/Candidate.java:121-134,
//                       member/Follower.java:258-270), called off-loop like a Netty thread would; queued loop tasks drain after
//   AE / IS responses   the same lambda body lifted into a method (captured values come from the row: the host kept them),
//                       member/Leader.java:174-188,218-237
//   TIMEOUT             RaftRoutine.electionTimeout / keepAlive on the participant's live TimerTicket (context/RaftRoutine.java:53-77)
//   CLIENT_APPEND       Leader.acceptCommand (command/RaftStub.java:79-91 minus the isReady gate, which is rg_ready's job)
//   LOG_FLUSH           RaftLog.flush
// What the row reports (flags, persist, log effects) is OBSERVED from the fakes: StableLock.persist calls, puts and
// deleteRange on the fake RocksDB, commitIndex before/after, sends recorded by the RaftService stubs, resetTimer calls.
// An AssertionError escaping the handler is mapped to RG_A_* by the file:line of its `throw` in the reference.
#include "env.hpp"

#include "../../include/raftgpu.h"

#include <cstdlib>

enum { REF_UNMAPPED_THROW = 60, REF_SWALLOWED_EXCEPTION = 61 };

struct ref_table;
struct PartRec { uint32_t epoch; Ref<RaftParticipant> p; Ref<AsyncHead> head; Ref<AtomicInteger> votes; };

struct RaftRoutineObserved : RaftRoutine {
    jboolean resetTimer(Ref<RaftContext> context, Ref<RaftParticipant> participant, jboolean muted) override;
};

struct RefEnv {
    ref_table *t = nullptr; uint32_t gid = 0;
    Ref<RaftContext> ctx; Ref<RaftRoutineObserved> routine; Ref<RaftCluster> cluster; Ref<RaftConfig> cfg;
    Ref<RocksDB> db; Ref<RocksLog> log; Ref<StableLock> lock; Ref<ContextEventLoop> loop;
    uint32_t epoch_counter = 0;
    std::vector<PartRec> parts;
    // observations of one event
    int persists = 0; jlong p_term = 0; int p_vote = RG_NO_NODE; bool reset_timer = false; uint32_t emit = RG_EMIT_NONE;
    Ref<AtomicInteger> last_votes;
    std::vector<Ref<PendingCall>> outbox;
    bool install_result = true;
    uint32_t status = RG_OK;
    bool timer_armed = false;                         // rg_timers_* contract: no ticket until rg_timers_arm / the first reset
    std::deque<std::function<void()>> held;           // loop tasks queued by a fired timer, waiting for their RG_EV_TIMEOUT row
};

struct ref_table {
    uint32_t groups, cluster, self; int pre_vote;
    std::vector<std::unique_ptr<RefEnv>> g;
    const int64_t *clock = nullptr;
    int64_t election_ms = 900, heartbeat_ms = 300; uint64_t timer_seed = 0;
    bool hold = false;                                // ref_hold(): play() leaves what the row queued on the loop undrained (see there)
};

static RefEnv *g_env = nullptr;
static const char *&last_error_fmt() { static const char *f = nullptr; return f; }

// ---- shim pieces that need the environment -------------------------------------------------------------------------
jboolean RaftRoutineObserved::resetTimer(Ref<RaftContext> context, Ref<RaftParticipant> participant, jboolean muted)
{
    env->reset_timer = true;
    env->timer_armed = true;
    return RaftRoutine::resetTimer(context, participant, muted);
}
void RaftRoutine::commitState(Ref<RaftContext>, const std::function<Ref<Promise>(Ref<Entry>)> &, jint) {}
jboolean RaftRoutine::installSnapshot(Ref<RaftContext>, jlong, jlong, const std::function<void()> &) { return env->install_result; }
jboolean RaftContext::installSnapshot(Ref<ID>, jlong lastIncludedIndex, jlong lastIncludedTerm)
{
    if (!inEventLoop()) throw jat(AssertionError("install snapshot should be performed in event loop"), "RaftContext.java:272");
    return stillRunning_ && routine->installSnapshot(this, lastIncludedIndex, lastIncludedTerm, [] {});
}
void StableLock::persist(jlong term, Ref<ID> candidate)
{
    env->epoch_counter++;                 // one call per participant object (member/RaftMember.java:25)
    env->persists++;
    env->p_term = term;
    env->p_vote = candidate == nullptr ? RG_NO_NODE : candidate->slot;
    env->emit = RG_EMIT_NONE;
    env->last_votes = nullptr;
}
RaftCluster::RaftCluster(RefEnv *e, jint cluster, jint self_slot) : env(e), n(cluster)
{
    remotes = jnew<Set<ID>>();
    for (jint s = 0; s < cluster; s++) {
        Ref<ID> id = jnew<ID>(s);
        svc.push_back(jnew<RaftService>(e, (int)s));
        if (s == self_slot) self = id; else remotes->add(id);
    }
}
static Ref<Async<RaftResponse>> sent(RefEnv *env, int kind, int peer, jlong term, jlong a, jlong b, jlong commit, jint count,
                                     jlong first_term, jlong last_index, uint32_t emit)
{
    Ref<Async<RaftResponse>> as = jnew<Async<RaftResponse>>();
    as->env = env;
    as->call = jnew<PendingCall>();
    as->call->kind = kind; as->call->peer = peer; as->call->term = term; as->call->a = a; as->call->b = b;
    as->call->commit = commit; as->call->count = count; as->call->first_term = first_term; as->call->last_index = last_index;
    env->outbox.push_back(as->call);
    env->emit = emit;
    return as;
}
Ref<Async<RaftResponse>> RaftService::appendEntries(jlong term, Ref<ID>, jlong prevLogIndex, jlong prevLogTerm, JArr<Ref<Entry>> entries, jlong leaderCommit)
{
    jint n = entries == nullptr ? 0 : entries->length;
    return sent(env, REF_RPC_AE, peer, term, prevLogIndex, prevLogTerm, leaderCommit, n, n ? entries[0]->term() : 0,
                n ? entries[n - 1]->index() : prevLogIndex, RG_EMIT_HEARTBEAT);
}
Ref<Async<RaftResponse>> RaftService::preVote(jlong term, Ref<ID>, jlong lastLogIndex, jlong lastLogTerm)
{ return sent(env, REF_RPC_PV, peer, term, lastLogIndex, lastLogTerm, 0, 0, 0, 0, RG_EMIT_PREVOTE); }
Ref<Async<RaftResponse>> RaftService::requestVote(jlong term, Ref<ID>, jlong lastLogIndex, jlong lastLogTerm)
{ return sent(env, REF_RPC_RV, peer, term, lastLogIndex, lastLogTerm, 0, 0, 0, 0, RG_EMIT_REQVOTE); }
Ref<Async<RaftResponse>> RaftService::installSnapshot(jlong term, Ref<ID>, jlong lastIncludedIndex, jlong lastIncludedTerm)
{ return sent(env, REF_RPC_IS, peer, term, lastIncludedIndex, lastIncludedTerm, 0, 0, 0, lastIncludedIndex, RG_EMIT_HEARTBEAT); }

template <class T> void Async<T>::on(Ref<AsyncHead> head, jlong, AsyncCallback cb)
{
    call->cb = cb;
    call->head = head.get();
    if (head->aborted) { call->done = true; cb(nullptr, nullptr, true); return; }   // Async.java:200-203: born cancelled
    head->calls.push_back(call);
}

// ---- status of an escaped Throwable: by the reference line that threw it ---------------------------------------------
static uint32_t status_of(const Throwable &e)
{
    static const struct { const char *where; uint32_t st; } map[] = {
        {"Follower.java:49", RG_A_TWO_LEADERS}, {"Follower.java:180", RG_A_PREV_ZERO_MISMATCH},
        {"Follower.java:185", RG_A_EPOCH_TERM_MISMATCH}, {"Follower.java:201", RG_A_IMPOSSIBLE_LOG},
        {"RocksLog.java:102", RG_A_COMMIT_ROLLBACK}, {"RocksLog.java:176", RG_A_LOG_NOT_CONTINUOUS},
        {"RocksLog.java:181", RG_A_LOG_NOT_CONTINUOUS}, {"RocksLog.java:186", RG_A_LOG_NOT_CONTINUOUS},
        {"RocksLog.java:204", RG_A_LOG_NOT_CONTINUOUS}, {"Leader.java:72", RG_A_LEADER_SELF_AE},
        {"Leader.java:80", RG_A_SAME_TERM_LEADER}, {"Leader.java:104", RG_A_LEADER_NOT_SELF_VOTE},
        {"Candidate.java:54", RG_A_CAND_SELF_RV}, {"Candidate.java:65", RG_A_CAND_NOT_SELF_VOTE},
        {"Membership.java:89", RG_A_LEADER_UNCHANGED}, {"Membership.java:104", RG_A_CAND_BALLOT},
        {"Leadership.java:77", RG_A_MATCH_ROLLBACK}, {"Leader.java:252", RG_A_IMPOSSIBLE_REPLICATION},
        {"RocksLog.java:231", RG_FLUSH_OUT_OF_BOUNDS},
        {"RaftMember.java:63", RG_A_INSTALL_BEFORE_AE}, {"Follower.java:139", RG_A_INSTALL_BEFORE_AE},
        {"RaftRoutine.java:171", RG_A_NO_DOWNGRADE},
    };
    for (auto &m : map) if (!strcmp(m.where, e.where)) return m.st;
    if (getenv("REF_TRACE")) fprintf(stderr, "ref: unmapped %s at %s: %s\n", e.kind(), e.where, e.msg.c_str());
    return REF_UNMAPPED_THROW;
}

// ---- event plumbing -----------------------------------------------------------------------------------------------------
static void note(RefEnv &e, uint32_t st) { if (e.status == RG_OK) e.status = st; }

static void drain(RefEnv &e)
{
    e.loop->in_loop = true;
    while (!e.loop->q.empty()) {
        std::function<void()> task = e.loop->q.front();
        e.loop->q.pop_front();
        try { task(); }                                   // EventLoopExecutor.run catches Throwable (support/EventLoopGroup.java:40-44)
        catch (const Throwable &t) { note(e, status_of(t)); }
    }
}

static Ref<AsyncHead> head_of(const Ref<RaftParticipant> &p)
{
    if (Follower *f = dynamic_cast<Follower *>(p.get())) return f->qualifier;
    if (Candidate *c = dynamic_cast<Candidate *>(p.get())) return c->election;
    if (Leader *l = dynamic_cast<Leader *>(p.get())) return l->replication;
    return nullptr;
}
static int role_of(const Ref<RaftParticipant> &p)
{
    if (dynamic_cast<Leader *>(p.get())) return RG_LEADER;
    if (dynamic_cast<Candidate *>(p.get())) return RG_CANDIDATE;
    return RG_FOLLOWER;
}
static PartRec *find_rec(RefEnv &e, uint32_t epoch)
{
    for (auto it = e.parts.rbegin(); it != e.parts.rend(); ++it) if (it->epoch == epoch) return &*it;
    return nullptr;
}
static void remember_participant(RefEnv &e)
{
    Ref<RaftParticipant> cur = e.ctx->participant();
    if (!e.parts.empty() && e.parts.back().p == cur) return;
    e.parts.push_back(PartRec{e.epoch_counter, cur, head_of(cur), e.last_votes});
    if (e.parts.size() > 24) {                            // forget fenced participants (their closures are dead)
        std::vector<PartRec> keep;
        for (size_t i = 0; i < e.parts.size(); i++)
            if (i + 8 >= e.parts.size() || !e.parts[i].head->aborted) keep.push_back(e.parts[i]);
        e.parts.swap(keep);
    }
}
// AsyncFuture arms a timer of broadcastTimeout per request (transport/rpc/Async.java:41-47,225-229) and RaftConfig insists on
// broadcast < heartbeat < election (support/RaftConfig.java:116-118): by the time a LATER election has been won, every
// RequestVote of an earlier winner has long completed with a TimeoutException.  The replay format carries no time, so the
// driver applies that fact when the later win happens (the C-ABI keeps one un-aborted winner head per group for the same reason).
static void expire_older_winners(RefEnv &e)
{
    int newest = -1;
    for (size_t i = 0; i < e.parts.size(); i++) {
        Candidate *c = dynamic_cast<Candidate *>(e.parts[i].p.get());
        if (c && c->elected) newest = (int)i;
    }
    for (int i = 0; i < newest; i++) {
        Candidate *c = dynamic_cast<Candidate *>(e.parts[(size_t)i].p.get());
        if (!c || !c->elected || e.parts[(size_t)i].head->aborted) continue;
        std::vector<Ref<PendingCall>> live;
        live.swap(e.parts[(size_t)i].head->calls);
        for (auto &pc : live) if (!pc->done) { pc->done = true; pc->cb(nullptr, jnew<Exception>("TimeoutException"), false); }
    }
}
static Ref<ID> id_of(RefEnv &e, uint32_t slot)
{
    if ((jint)slot == e.cluster->self->slot) return e.cluster->self;
    for (auto &r : e.cluster->remotes->v) if (r->slot == (jint)slot) return r;
    return jnew<ID>((jint)slot);
}
static uint64_t timer_mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static void new_context(ref_table *t, RefEnv &e)
{
    e.t = t;
    e.cfg = jnew<RaftConfig>(); e.cfg->pre_vote = t->pre_vote != 0;
    e.cfg->election_ms = t->election_ms; e.cfg->heartbeat_ms = t->heartbeat_ms;
    RefEnv *ep = &e;
    // RaftConfig.electionTimeout() is ThreadLocalRandom in [E, 2E] (support/RaftConfig.java:187-190); the draw used here is
    // the counter-based one of the C-ABI (include/raftgpu.h, rg_timers_update) so that deadlines can be compared
    e.cfg->election_draw = [ep]() -> jlong {
        ref_table *tt = ep->t;
        uint64_t h = timer_mix(tt->timer_seed ^ timer_mix((uint64_t)ep->gid * 0xD1342543DE82EF95ull ^ ((uint64_t)ep->epoch_counter << 32) ^ (uint64_t)jrt::now_ms()));
        return tt->election_ms + (jlong)(h % (uint64_t)(tt->election_ms + 1));
    };
    e.timer_armed = false; e.held.clear();
    e.cluster = jnew<RaftCluster>(&e, (jint)t->cluster, (jint)t->self);
    e.db = jnew<RocksDB>();
    e.log = jnew<RocksLog>(e.db, jnew<RocksSerializer>());
    e.lock = jnew<StableLock>(&e);
    e.loop = jnew<ContextEventLoop>();
    e.routine = jnew<RaftRoutineObserved>(); e.routine->env = &e;
    e.ctx = jnew<RaftContext>(); e.ctx->env = &e;
    e.ctx->envConfig_ = e.cfg; e.ctx->replicatedLog_ = e.log; e.ctx->stateMachine_ = jnew<RaftMachine>();
    e.ctx->stableStorage_ = e.lock; e.ctx->snapArchive_ = jnew<SnapshotArchive>(); e.ctx->cluster_ = e.cluster;
    e.ctx->routine = e.routine; e.ctx->eventLoop_ = e.loop;
    e.parts.clear(); e.epoch_counter = 0;
}
// put a participant in place without going through isBetter (state loading only)
static void force_participant(RefEnv &e, Ref<Class> role, jlong term, Ref<ID> ballot, uint32_t epoch)
{
    e.epoch_counter = epoch - 1;
    Ref<Membership> m = jnew<Membership>(role, term, ballot);
    e.ctx->membershipFilter->set(jnew<Membership>(m));    // applied copy, as RaftRoutine.switchTo leaves it (:175)
    e.routine->convertTo(e.ctx, m);
    remember_participant(e);
}

static void begin_event(RefEnv &e)
{
    g_env = &e;
    e.persists = 0; e.reset_timer = false; e.emit = RG_EMIT_NONE; e.status = RG_OK; e.outbox.clear();
    e.db->observe_reset();
    last_error_fmt() = nullptr;
    e.loop->in_loop = true;
    if (Leader *l = dynamic_cast<Leader *>(e.ctx->participant().get()))      // requestInFlight is host-owned at this boundary
        if (l->followerStatus != nullptr) for (auto &s : l->followerStatus->values()) s->requestInFlight = 0;
}

static void play(ref_table *t, RefEnv &e, const rg_batch_t *in, size_t row, rg_reply_t *rep, rg_logfx_t *lfx, rg_persist_t *per)
{
    const uint32_t hdr = in->head[row].hdr, aux = in->head[row].aux;
    const uint32_t kind = RG_HDR_KIND(hdr), slot = RG_HDR_SLOT(hdr), flag = RG_HDR_FLAG(hdr), n = RG_HDR_N(hdr);
    const int64_t a = in->ab[row].x, b = in->ab[row].y, c = in->cd[row].x, d = in->cd[row].y;
    begin_event(e);
    if (t->clock) jrt::now_ms() = t->clock[row / in->count];
    Ref<RaftParticipant> before = e.ctx->participant();
    const jlong commit_before = e.log->commitIndex;
    Ref<RaftResponse> resp;
    jlong client_from = 0; bool client_appended = false;
    try {
        switch (kind) {
        case RG_EV_NONE: break;
        case RG_EV_AE_REQ: {
            if (slot >= t->cluster || n > RG_MAX_AE_ENTRIES || (n > 0 && (in->entry_terms == NULL || (uint64_t)aux + n > in->entry_count))) { e.status = RG_BAD_EVENT; break; }
            JArr<Ref<Entry>> entries = JArr<Ref<Entry>>::make(n);
            for (uint32_t k = 0; k < n; k++) {
                jlong term = in->entry_terms[aux + k];
                entries[k] = jnew<RocksEntry>(term, (jlong)((uint64_t)b + 1 + k), RocksLog::longToBytes(term));
            }
            resp = before->appendEntries(a, id_of(e, slot), b, c, entries, d);
            break;
        }
        case RG_EV_RV_REQ:
        case RG_EV_PV_REQ:
            if (slot >= t->cluster) { e.status = RG_BAD_EVENT; break; }
            resp = kind == RG_EV_PV_REQ ? before->preVote(a, id_of(e, slot), b, c) : before->requestVote(a, id_of(e, slot), b, c);
            break;
#ifdef RG_EV_IS_REQ_DEFINED
        case RG_EV_IS_REQ:
            if (slot >= t->cluster) { e.status = RG_BAD_EVENT; break; }
            e.install_result = flag != 0;
            resp = before->installSnapshot(a, id_of(e, slot), b, c);
            break;
#endif
        case RG_EV_AE_ACK:
        case RG_EV_IS_ACK: {
            if (slot >= t->cluster || slot == t->self) { e.status = RG_BAD_EVENT; break; }
            PartRec *rec = find_rec(e, aux);
            if (!rec || rec->p != before) { e.status = RG_DROPPED_STALE_ROLE; break; }
            Leader *l = dynamic_cast<Leader *>(before.get());
            if (!l || l->followerStatus == nullptr) { e.status = RG_BAD_EVENT; break; }
            Ref<ID> id = id_of(e, slot);
            Ref<State> st = l->followerStatus->get(id);
            Ref<Entry> epoch = jnew<EntryKey>((jlong)b, (jlong)0);
            Ref<RaftResponse> result = RaftResponse::reply(a, flag != 0);
            e.loop->in_loop = false;                          // callbacks run on the thread completing the invocation
            const bool canceled = l->replication->isAborted();
            if (kind == RG_EV_IS_ACK) l->onInstallSnapshotResponse(id, st, epoch, l->replication, result, nullptr, canceled);
            else l->onAppendEntriesResponse(id, st, epoch, c, 0, l->replication, result, nullptr, canceled);
            break;
        }
        case RG_EV_RV_REPLY:
        case RG_EV_PV_REPLY: {
            if (slot >= t->cluster || slot == t->self) { e.status = RG_BAD_EVENT; break; }
            PartRec *rec = find_rec(e, aux);
            if (!rec || rec->head->aborted) { e.status = RG_DROPPED_STALE_ROLE; break; }
            const int want = kind == RG_EV_RV_REPLY ? REF_RPC_RV : REF_RPC_PV;
            Ref<PendingCall> call;
            for (auto &pc : rec->head->calls) if (pc->peer == (int)slot && pc->kind == want) call = pc;
            if (call == nullptr) { e.status = rec->p == before ? RG_BAD_EVENT : RG_DROPPED_STALE_ROLE; break; }
            e.loop->in_loop = false;
            call->cb(RaftResponse::reply(a, flag != 0), nullptr, false);
            break;
        }
        case RG_EV_TIMEOUT: {
            if (aux != 0) {                                   // the ticket that fired belonged to the participant of that epoch
                PartRec *rec = find_rec(e, aux);
                if (!rec || rec->p != before) { e.status = RG_DROPPED_STALE_ROLE; e.held.clear(); break; }   // context/RaftRoutine.java:70
            }
            if (!e.held.empty()) {                            // rg_timers_expired already fired the ticket: run what it queued
                e.loop->q.swap(e.held);
                e.held.clear();
                break;
            }
            Ref<TimerTicket> ticket = e.ctx->ticketHolder->get();
            e.loop->in_loop = false;                          // timer pool thread
            if (dynamic_cast<Leader *>(before.get())) e.routine->keepAlive(e.ctx, ticket);
            else e.routine->electionTimeout(e.ctx, ticket);
            break;
        }
        case RG_EV_CLIENT_APPEND: {
            Leader *l = dynamic_cast<Leader *>(before.get());
            if (!l) { e.status = RG_NOT_LEADER; break; }      // command/RaftStub.java:89
            if (n == 0) break;
            if (e.log->last() == nullptr && e.log->epoch()->index() > 0) { e.status = RG_UNSUPPORTED_LOG_STATE; break; }
            for (uint32_t k = 0; k < n; k++) {
                Ref<Entry> last = e.log->last();
                if (k == 0) client_from = last == nullptr ? 1 : last->index() + 1;
                l->acceptCommand(jnew<Command>(), jnew<Promise>());
                client_appended = true;
            }
            break;
        }
        case RG_EV_LOG_FLUSH:
            e.log->flush(a, b);
            break;
        default:
            e.status = RG_BAD_EVENT;
            break;
        }
    } catch (const Throwable &th) {
        note(e, status_of(th));
    }
    // Delivery modes. Normally the loop is drained right after the row: the one interleaving in which nothing else of this group runs between
    // a callback's off-loop part (the CAS of the membership filter, context/RaftRoutine.java:140-151) and the loop task it queues at the head
    // (context/RaftContext.java:205-215, support/EventLoop.java:87-101). With ref_hold() the queue is LEFT AS IT IS: the next row of the group
    // then runs its handler first — the loop thread was already inside that task when the callback's thread did its CAS — and only then
    // drains, the held urgent tasks first. (tests/test_ref_parity.py: the interleaving differential)
    if (!t->hold) drain(e);
    if (e.status == RG_OK && last_error_fmt()) {
        if (strstr(last_error_fmt(), "try commit failed")) e.status = RG_NPE_MAJOR_NULL;    // member/Leader.java:277-279
        else { if (getenv("REF_TRACE")) fprintf(stderr, "ref: swallowed: %s\n", last_error_fmt()); e.status = REF_SWALLOWED_EXCEPTION; }
    }
    remember_participant(e);
    expire_older_winners(e);

    Ref<RaftParticipant> after = e.ctx->participant();
    uint32_t flags = 0;
    if (resp != nullptr) flags |= RG_F_REPLIED | (resp->success() ? RG_F_SUCCESS : 0);
    if (e.persists) flags |= RG_F_PERSIST;
    if (after != before) flags |= RG_F_ROLE_CHANGED;
    if (e.reset_timer) {
        flags |= RG_F_RESET_TIMER;
        Ref<TimerTicket> tk = e.ctx->ticketHolder->get();            // still muted after the handler: deadline Long.MAX_VALUE (:101-107)
        if (role_of(after) != RG_LEADER && tk != nullptr && tk->get() == Long::MAX_VALUE) flags |= RG_F_TIMER_MUTED;
    }
    if (e.log->commitIndex != commit_before) flags |= RG_F_COMMIT;
    jlong log_from = 0;
    if (kind == RG_EV_AE_REQ) {
        if (e.db->truncated) { flags |= RG_F_LOG_TRUNC; log_from = e.db->trunc_from; }
        if (e.db->puts_new) { flags |= RG_F_LOG_APPEND; if (!e.db->truncated) log_from = e.db->first_new_key; }
    } else if (client_appended) {
        flags |= RG_F_LOG_APPEND; log_from = client_from;
    }
    flags |= e.emit << RG_F_EMIT_SHIFT;
    flags |= (uint32_t)role_of(after) << RG_F_ROLE_SHIFT;
    flags |= e.status << RG_F_STATUS_SHIFT;
    rep->resp_term = resp != nullptr ? resp->term() : 0;
    rep->flags = flags;
    rep->role_epoch = e.epoch_counter;
    if (flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC)) { lfx->commit_index = e.log->commitIndex; lfx->log_from = log_from; }
    if (flags & RG_F_PERSIST) { per->term = e.p_term; per->voted_for = e.p_vote; per->role = role_of(after); }
    g_env = nullptr;
}

// ---- C API -----------------------------------------------------------------------------------------------------------------
extern "C" {

ref_table *ref_table_create(uint32_t groups, uint32_t cluster, uint32_t self_slot, int pre_vote)
{
    if (groups == 0 || cluster < RG_MIN_CLUSTER || cluster > RG_MAX_CLUSTER || self_slot >= cluster) return nullptr;
    static bool once = false;
    if (!once) {
        once = true;
        jrt::on_new_atomic_integer() = [](Object *o) { if (g_env) g_env->last_votes = Ref<AtomicInteger>(dynamic_cast<AtomicInteger *>(o)); };
        jrt::on_logger_error() = [](const char *f) { last_error_fmt() = f; };
        Follower_class->make = [](Ref<RaftContext> c, jlong t, Ref<ID> b, Ref<Membership> m) { return Ref<RaftParticipant>(jnew<Follower>(c, t, b, m)); };
        Candidate_class->make = [](Ref<RaftContext> c, jlong t, Ref<ID> b, Ref<Membership> m) { return Ref<RaftParticipant>(jnew<Candidate>(c, t, b, m)); };
        Leader_class->make = [](Ref<RaftContext> c, jlong t, Ref<ID> b, Ref<Membership> m) { return Ref<RaftParticipant>(jnew<Leader>(c, t, b, m)); };
    }
    ref_table *t = new ref_table();
    t->groups = groups; t->cluster = cluster; t->self = self_slot; t->pre_vote = pre_vote;
    for (uint32_t i = 0; i < groups; i++) {
        t->g.emplace_back(new RefEnv());
        RefEnv &e = *t->g.back();
        e.gid = i;
        new_context(t, e);
        g_env = &e;
        force_participant(e, Follower_class, 0, nullptr, 1);      // RaftContext.initialize: switchTo(Follower, restore.term, restore.ballot)
        e.timer_armed = false;
        g_env = nullptr;
    }
    return t;
}

void ref_table_destroy(ref_table *t)
{
    if (!t) return;
    for (auto &e : t->g) {                                         // break the closure <-> participant cycles
        for (auto &p : e->parts) if (p.head != nullptr) { p.head->aborted = true; p.head->calls.clear(); }
        e->loop->q.clear();
    }
    delete t;
}

int ref_load_state(ref_table *t, uint32_t first, uint32_t count, const rg_group_state_t *s)
{
    if (!t || !s || (uint64_t)first + count > t->groups) return -1;
    const uint32_t F = t->cluster - 1;
    for (uint32_t i = 0; i < count; i++) {
        RefEnv &e = *t->g[first + i];
        for (auto &p : e.parts) if (p.head != nullptr) { p.head->aborted = true; p.head->calls.clear(); }
        new_context(t, e);
        g_env = &e;
        try {
            // the log: every key of the window with its term (the value's first 8 bytes, storage/RocksLog.java:85-86)
            uint32_t rc = s->run_count[i], ro = s->run_offset[i];
            if (rc) {
                if (s->run_start[ro] != s->first_index[i] || s->last_index[i] < s->run_start[ro + rc - 1]) return -2;
                if (s->first_index[i] != s->epoch_index[i] && s->first_index[i] != s->epoch_index[i] + 1) return -3;
                for (uint32_t k = 0; k < rc; k++) {                 // straight into the fake's run representation
                    jlong end = k + 1 < rc ? s->run_start[ro + k + 1] - 1 : s->last_index[i];
                    e.db->kv[(uint64_t)s->run_start[ro + k]] = RocksDB::Run{(uint64_t)end, jkey(RocksLog::longToBytes(s->run_term[ro + k]))};
                }
            }
            e.log->epochEntry = jnew<EntryKey>((jlong)s->epoch_index[i], (jlong)s->epoch_term[i]);
            e.log->lastEntry_ = e.log->lastEntry();
            e.log->commitIndex = s->commit_index[i];
            // Q13: a Candidate that won keeps its election head un-aborted (member/Candidate.java:75-79)
            const int role = s->role[i];
            Ref<ID> ballot = s->voted_for[i] == RG_NO_NODE ? Ref<ID>(nullptr) : id_of(e, (uint32_t)s->voted_for[i]);
            if (s->elected_epoch[i] != 0) {
                if (s->elected_epoch[i] >= s->role_epoch[i] || s->elected_term[i] > s->current_term[i]) return -5;
                force_participant(e, Candidate_class, s->elected_term[i], e.cluster->self, s->elected_epoch[i]);
                Candidate *c = dynamic_cast<Candidate *>(e.ctx->participant().get());
                c->elected = true;
                if (e.parts.back().votes != nullptr) e.parts.back().votes->set((jint)(t->cluster / 2 + 1));
            }
            force_participant(e, role == RG_LEADER ? Leader_class : role == RG_CANDIDATE ? Candidate_class : Follower_class,
                              s->current_term[i], ballot, s->role_epoch[i]);
            Ref<RaftParticipant> p = e.ctx->participant();
            if (Follower *f = dynamic_cast<Follower *>(p.get())) {
                if (s->current_leader[i] != RG_NO_NODE) f->currentLeader_ = id_of(e, (uint32_t)s->current_leader[i]);
                if (s->timeout_detected[i]) {
                    f->prepareElection();                          // what Follower.onTimeout runs on the fresh Follower (:162)
                    e.parts.back().votes = e.last_votes;
                }
            }
            if (e.parts.back().votes != nullptr) e.parts.back().votes->set(s->votes[i]);
            if (Leader *l = dynamic_cast<Leader *>(p.get())) {
                if (s->repl_prepared[i]) {
                    l->prepareReplication();
                    for (uint32_t j = 0; j < F; j++) {
                        uint32_t slot = j < t->self ? j : j + 1;
                        Ref<State> st = l->followerStatus->get(id_of(e, slot));
                        st->lastEpoch = s->peer_last_epoch[(size_t)i * F + j];
                        st->nextIndex = s->peer_next_index[(size_t)i * F + j];
                        st->matchIndex = s->peer_match_index[(size_t)i * F + j];
                        st->recentRejection = s->peer_rejection[(size_t)i * F + j];
                        st->pendingInstallation = s->peer_pending[(size_t)i * F + j] != 0;
                    }
                }
            }
            e.loop->q.clear();
            e.timer_armed = false;
        } catch (const Throwable &th) {
            if (getenv("REF_TRACE")) fprintf(stderr, "ref_load_state: %s at %s: %s\n", th.kind(), th.where, th.msg.c_str());
            g_env = nullptr;
            return -6;
        }
        g_env = nullptr;
    }
    return 0;
}

int ref_read_state(ref_table *t, uint32_t first, uint32_t count, rg_group_state_t *d)
{
    if (!t || !d || (uint64_t)first + count > t->groups) return -1;
    const uint32_t F = t->cluster - 1;
    for (uint32_t i = 0; i < count; i++) {
        RefEnv &e = *t->g[first + i];
        Ref<RaftParticipant> p = e.ctx->participant();
        d->current_term[i] = p->currentTerm();
        Ref<ID> v = p->votedFor();
        d->voted_for[i] = v == nullptr ? RG_NO_NODE : v->slot;
        d->role[i] = role_of(p);
        d->current_leader[i] = RG_NO_NODE; d->timeout_detected[i] = 0; d->repl_prepared[i] = 0;
        if (Follower *f = dynamic_cast<Follower *>(p.get())) {
            if (f->currentLeader_ != nullptr) d->current_leader[i] = f->currentLeader_->slot;
            d->timeout_detected[i] = f->timeoutDetected;
        }
        d->role_epoch[i] = e.epoch_counter;
        PartRec *cur = find_rec(e, e.epoch_counter);
        d->votes[i] = (cur && cur->votes != nullptr) ? cur->votes->get() : 1;
        d->elected_epoch[i] = 0; d->elected_term[i] = 0;
        for (auto &r : e.parts) {                                  // newest winner whose head is still live
            Candidate *c = dynamic_cast<Candidate *>(r.p.get());
            if (c && c->elected && !r.head->aborted && !r.head->calls.empty() && r.p != p) { d->elected_epoch[i] = r.epoch; d->elected_term[i] = c->currentTerm(); }
        }
        d->commit_index[i] = e.log->commitIndex;
        d->epoch_index[i] = e.log->epoch()->index();
        d->epoch_term[i] = e.log->epoch()->term();
        // term runs of the stored key window
        std::vector<std::pair<jlong, jlong>> runs;
        jlong firstk = 0, lastk = 0; bool any = false;
        for (auto &kv : e.db->kv) {
            jlong idx = (jlong)kv.first, term = RocksDB::key_to_long(kv.second.val);
            if (!any) { firstk = idx; any = true; }
            lastk = (jlong)kv.second.end;
            if (runs.empty() || runs.back().second != term) runs.emplace_back(idx, term);
        }
        uint32_t rc = (uint32_t)std::min<size_t>(runs.size(), RG_TERM_RUNS);
        d->run_count[i] = rc; d->run_offset[i] = i * RG_TERM_RUNS;
        d->first_index[i] = any ? firstk : 0; d->last_index[i] = any ? lastk : 0;
        for (uint32_t k = 0; k < RG_TERM_RUNS; k++) {
            bool have = k < rc;
            d->run_start[(size_t)i * RG_TERM_RUNS + k] = have ? runs[runs.size() - rc + k].first : 0;
            d->run_term[(size_t)i * RG_TERM_RUNS + k] = have ? runs[runs.size() - rc + k].second : 0;
        }
        Leader *l = dynamic_cast<Leader *>(p.get());
        if (l && l->followerStatus != nullptr) d->repl_prepared[i] = 1;
        for (uint32_t j = 0; j < F; j++) {
            size_t o = (size_t)i * F + j;
            d->peer_last_epoch[o] = d->peer_next_index[o] = d->peer_match_index[o] = 0; d->peer_rejection[o] = 0; d->peer_pending[o] = 0;
            if (l && l->followerStatus != nullptr) {
                Ref<State> st = l->followerStatus->get(id_of(e, j < t->self ? j : j + 1));
                d->peer_last_epoch[o] = st->lastEpoch; d->peer_next_index[o] = st->nextIndex; d->peer_match_index[o] = st->matchIndex;
                d->peer_rejection[o] = st->recentRejection; d->peer_pending[o] = st->pendingInstallation;
            }
        }
    }
    return 0;
}

int ref_submit(ref_table *t, const rg_batch_t *in, const rg_outcome_t *out)
{
    if (!t || !in || !out || !in->head || !in->ab || !in->cd || !out->reply || !out->logfx || !out->persist || in->rounds == 0) return -1;
    if (in->gid) { if (in->rounds != 1 || in->count > t->groups) return -1; }
    else if (in->count != t->groups) return -1;
    for (uint32_t r = 0; r < in->rounds; r++)
        for (uint32_t i = 0; i < in->count; i++) {
            size_t row = (size_t)r * in->count + i;
            uint32_t gid = in->gid ? in->gid[i] : i;
            if (gid >= t->groups) return -1;
            play(t, *t->g[gid], in, row, &out->reply[row], &out->logfx[row], &out->persist[row]);
        }
    return 0;
}

int ref_clock(ref_table *t, const int64_t *now_per_round) { if (!t) return -1; t->clock = now_per_round; return 0; }
int ref_hold(ref_table *t, int on) { if (!t) return -1; t->hold = on != 0; return 0; }


// ---- N1: Leader.replicateLog (member/Leader.java:142-245) — what the reference itself sends ----------------------------
int ref_replicate(ref_table *t, uint32_t count, const uint32_t *gid, const uint8_t *heartbeat, const uint16_t *in_flight,
                  rg_send_head_t *head, rg_send_t *send)
{
    if (!t || !head || !send) return -1;
    if (gid ? count > t->groups : count != t->groups) return -1;
    const uint32_t F = t->cluster - 1;
    for (uint32_t i = 0; i < count; i++) {
        if (gid && (gid[i] >= t->groups || (i && gid[i] <= gid[i - 1]))) return -1;
        RefEnv &e = *t->g[gid ? gid[i] : i];
        begin_event(e);
        Ref<RaftParticipant> p = e.ctx->participant();
        Leader *l = dynamic_cast<Leader *>(p.get());
        rg_send_head_t *h = &head[i];
        h->term = p->currentTerm(); h->leader_commit = e.log->commitIndex;
        h->epoch_index = e.log->epoch()->index(); h->epoch_term = e.log->epoch()->term();
        h->role_epoch = e.epoch_counter; h->is_leader = l != nullptr; h->reserved = 0;
        for (uint32_t j = 0; j < F; j++) send[(size_t)j * count + i] = rg_send_t{0, 0, 0, 0, RG_SEND_NONE};
        if (!l) { g_env = nullptr; continue; }
        const bool hb = heartbeat && heartbeat[i];
        int rc = 0;
        try {
            l->prepareReplication();                                       // so that the host-owned in-flight counts have a home
            for (uint32_t j = 0; j < F; j++)
                l->followerStatus->get(id_of(e, j < t->self ? j : j + 1))->requestInFlight = in_flight ? in_flight[(size_t)j * count + i] : 0;
            l->replicateLog(hb);
        } catch (const Throwable &th) { rc = (int)status_of(th); }
        if (rc) { g_env = nullptr; return -100 - rc; }
        for (uint32_t j = 0; j < F; j++) {                                 // no send recorded = the in-flight gate (:162-166)
            rg_send_t o = {h->epoch_index, h->epoch_term, h->epoch_index, 0, RG_SEND_GATED};
            const int slot = (int)(j < t->self ? j : j + 1);
            for (auto &pc : e.outbox) {
                if (pc->peer != slot) continue;
                if (pc->kind == REF_RPC_IS) { o.kind = RG_SEND_SNAPSHOT; }
                else { o.prev_index = pc->a; o.prev_term = pc->b; o.last_index = pc->last_index; o.count = (uint32_t)pc->count; o.kind = RG_SEND_APPEND; }
            }
            send[(size_t)j * count + i] = o;
        }
        l->replication->calls.clear();                                     // nobody answers these through the closures
        g_env = nullptr;
    }
    return 0;
}

// ---- N4b: Leadership.State statistics, State.isReady, Leader.isReady (member/Leadership.java:40-73, member/Leader.java:52-64) ---
int ref_health_failure(ref_table *t, uint32_t n, const uint32_t *gid, const uint8_t *slot, const uint8_t *flags, int64_t now)
{
    if (!t || (n && (!gid || !slot || !flags))) return -1;
    for (uint32_t i = 0; i < n; i++) {
        if (gid[i] >= t->groups || slot[i] >= t->cluster || slot[i] == t->self) continue;
        RefEnv &e = *t->g[gid[i]];
        Leader *l = dynamic_cast<Leader *>(e.ctx->participant().get());
        if (!l || l->followerStatus == nullptr) continue;
        l->followerStatus->get(id_of(e, slot[i]))->statFailure(now, (flags[i] & 1u) != 0, (flags[i] & 2u) != 0);
    }
    return 0;
}

int ref_ready(ref_table *t, int64_t now, int32_t critical_point, int64_t cool_down_ms, uint8_t *ready)
{
    if (!t || !ready) return -1;
    jrt::now_ms() = now;
    for (uint32_t i = 0; i < t->groups; i++) {
        RefEnv &e = *t->g[i];
        e.cfg->critical_point = critical_point; e.cfg->cool_down = cool_down_ms;
        Leader *l = dynamic_cast<Leader *>(e.ctx->participant().get());
        ready[i] = l ? (l->isReady() ? 1 : 0) : 0;                         // command/RaftStub.java:80-87: only a Leader is asked
    }
    return 0;
}

int ref_health_read(ref_table *t, uint32_t first, uint32_t count, int64_t *request_success, int64_t *request_failure, int32_t *recent_failure)
{
    if (!t || (uint64_t)first + count > t->groups) return -1;
    const uint32_t F = t->cluster - 1;
    for (uint32_t i = 0; i < count; i++) {
        RefEnv &e = *t->g[first + i];
        Leader *l = dynamic_cast<Leader *>(e.ctx->participant().get());
        for (uint32_t j = 0; j < F; j++) {
            size_t o = (size_t)i * F + j;
            request_success[o] = request_failure[o] = 0; recent_failure[o] = 0;
            if (l && l->followerStatus != nullptr) {
                Ref<State> st = l->followerStatus->get(id_of(e, j < t->self ? j : j + 1));
                request_success[o] = st->requestSuccess; request_failure[o] = st->requestFailure; recent_failure[o] = st->recentFailure;
            }
        }
    }
    return 0;
}

// ---- N4 timers: the reference's own RaftRoutine.resetTimer / electionTimeout / keepAlive on a virtual clock -------------
int ref_timers_configure(ref_table *t, int64_t election_ms, int64_t heartbeat_ms, uint64_t seed)
{
    if (!t || election_ms <= 0 || heartbeat_ms <= 0) return -1;
    t->election_ms = election_ms; t->heartbeat_ms = heartbeat_ms; t->timer_seed = seed;
    for (auto &e : t->g) { e->cfg->election_ms = election_ms; e->cfg->heartbeat_ms = heartbeat_ms; }
    return 0;
}
static Ref<TimerTicket> ticket_of(RefEnv &e) { return e.ctx->ticketHolder->get(); }
int ref_timers_arm(ref_table *t, int64_t now)
{
    if (!t) return -1;
    jrt::now_ms() = now;
    for (auto &ep : t->g) {
        RefEnv &e = *ep;
        if (e.timer_armed) continue;
        g_env = &e;
        Ref<TimerTicket> old = ticket_of(e);
        Ref<RaftParticipant> p = e.ctx->participant();
        if (old != nullptr && old->schedule() != nullptr) old->schedule()->cancel(true);
        e.ctx->ticketHolder->set(nullptr);                        // as convertTo leaves it before the first resetTimer (:198)
        e.loop->in_loop = true;
        e.routine->resetTimer(e.ctx, p, false);
        g_env = nullptr;
    }
    return 0;
}
int ref_timers_read(ref_table *t, uint32_t first, uint32_t count, int64_t *deadline)
{
    if (!t || !deadline || (uint64_t)first + count > t->groups) return -1;
    for (uint32_t i = 0; i < count; i++) {
        RefEnv &e = *t->g[first + i];
        Ref<TimerTicket> tk = ticket_of(e);
        if (!e.timer_armed || tk == nullptr) { deadline[i] = 0; continue; }
        jlong d = tk->get();
        if (d < 0 || tk->schedule()->ran) deadline[i] = -1;       // TimerTicket.TIMEOUT / a heartbeat that fired
        else deadline[i] = (d == Long::MAX_VALUE && role_of(e.ctx->participant()) == RG_LEADER) ? tk->schedule()->due : d;
    }
    return 0;
}
/* the timer pools: every scheduled task that is due runs (electionTimeout / keepAlive, off-loop); the loop task it queues
 * is HELD until the host delivers the group's RG_EV_TIMEOUT row */
int ref_timers_expired_epochs(ref_table *t, int64_t now, uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, uint32_t *out_count);
int ref_timers_expired(ref_table *t, int64_t now, uint32_t *out_gid, uint32_t capacity, uint32_t *out_count)
{
    return ref_timers_expired_epochs(t, now, out_gid, nullptr, capacity, out_count);
}
int ref_timers_expired_epochs(ref_table *t, int64_t now, uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, uint32_t *out_count)
{
    if (!t || !out_count) return -1;
    jrt::now_ms() = now;
    uint32_t n = 0;
    for (auto &ep : t->g) {
        RefEnv &e = *ep;
        Ref<TimerTicket> tk = ticket_of(e);
        if (!e.timer_armed || tk == nullptr || tk->schedule() == nullptr) continue;
        Ref<ScheduledFuture> f = tk->schedule();
        if (f->cancelled || f->ran || f->due > now) continue;
        if (n < capacity) {
            g_env = &e;
            out_gid[n] = e.gid;
            if (out_epoch) out_epoch[n] = e.epoch_counter;
            f->ran = true;
            e.loop->in_loop = false;
            f->task();
            e.loop->in_loop = true;
            e.held.swap(e.loop->q);
            e.loop->q.clear();
            g_env = nullptr;
        }
        n++;
    }
    *out_count = n;
    return 0;
}

// ---- function-level entry points for differential fuzzing --------------------------------------------------------------
/* Membership.isBetter: 1 better, 0 not, <0 = -(RG_A_* status) */
int ref_is_better(int nr, int64_t nt, int32_t nb, int cr, int64_t ct, int32_t cb)
{
    Ref<Class> roles[3] = {Follower_class, Candidate_class, Leader_class};
    Ref<Membership> n = jnew<Membership>(roles[nr], (jlong)nt, nb == RG_NO_NODE ? Ref<ID>(nullptr) : Ref<ID>(jnew<ID>(nb)));
    Ref<Membership> c = jnew<Membership>(roles[cr], (jlong)ct, cb == RG_NO_NODE ? Ref<ID>(nullptr) : Ref<ID>(jnew<ID>(cb)));
    try { return n->isBetter(c) ? 1 : 0; }
    catch (const NullPointerException &) { return -100; }
    catch (const Throwable &th) { return -(int)status_of(th); }
}

void ref_major_indices(const int64_t *match, int n, int64_t out[2])
{
    Ref<Collection<State>> c = jnew<Collection<State>>();
    for (int i = 0; i < n; i++) { Ref<State> s = jnew<State>(); s->matchIndex = match[i]; c->add(s); }
    JArr<jlong> r = State::majorIndices(c);
    out[0] = r[0]; out[1] = r[1];
}

/* State.updateIndex on (lastEpoch, nextIndex, matchIndex, recentRejection, pendingInstallation); returns 0 or the RG_A_* status */
int ref_update_index(int64_t st[3], int32_t *rejection, uint8_t *pending, int64_t epoch, int64_t index, int success, int snapshot)
{
    Ref<State> s = jnew<State>();
    s->lastEpoch = st[0]; s->nextIndex = st[1]; s->matchIndex = st[2]; s->recentRejection = *rejection; s->pendingInstallation = *pending != 0;
    int rc = 0;
    try { s->updateIndex(epoch, index, success != 0, snapshot != 0); }
    catch (const Throwable &th) { rc = (int)status_of(th); }
    st[0] = s->lastEpoch; st[1] = s->nextIndex; st[2] = s->matchIndex; *rejection = s->recentRejection; *pending = s->pendingInstallation;
    return rc;
}

/* Math.round(Math.log(Math.E + r)) exactly as member/Leadership.java:105 evaluates it */
int64_t ref_rejection_step(int32_t r)
{
    Ref<State> s = jnew<State>();
    s->lastEpoch = 0; s->nextIndex = (jlong)1 << 40; s->matchIndex = 0; s->recentRejection = r;
    s->updateIndex(0, (jlong)1 << 41, false, false);
    jlong nx = s->nextIndex;                                        // min(next-1, max(next-step, 1)) = next - step for step >= 1
    jlong step = ((jlong)1 << 40) - nx;
    return step;
}

void ref_update_index_batch(uint32_t n, int64_t *st, int32_t *rejection, uint8_t *pending, const int64_t *epoch, const int64_t *index,
                            const uint8_t *success, const uint8_t *snapshot, int32_t *rc)
{
    for (uint32_t i = 0; i < n; i++)
        rc[i] = ref_update_index(st + 3 * (size_t)i, &rejection[i], &pending[i], epoch[i], index[i], success[i], snapshot[i]);
}
void ref_is_better_batch(uint32_t n, const int32_t *nr, const int64_t *nt, const int32_t *nb, const int32_t *cr, const int64_t *ct,
                         const int32_t *cb, int32_t *out)
{
    for (uint32_t i = 0; i < n; i++) out[i] = ref_is_better(nr[i], nt[i], nb[i], cr[i], ct[i], cb[i]);
}
void ref_major_indices_batch(uint32_t n, int f, const int64_t *match, int64_t *out)
{
    for (uint32_t i = 0; i < n; i++) ref_major_indices(match + (size_t)i * f, f, out + 2 * (size_t)i);
}

// ---- the java.lang / java.util stand-ins of jrt.hpp, bare (tests/test_ref_parity.py holds them to Python's exact integer model) -----------
// op: 0 a + b   1 a - b   2 a * b   3 a >>> (b)   4 Long.compare   5 Long.hashCode(a)   6 Math.max   7 Math.min   8 -a   9 (long)(int)a + (int)b as int
void ref_jrt_long_ops(uint32_t n, int op, const int64_t *a, const int64_t *b, int64_t *out)
{
    for (uint32_t i = 0; i < n; i++) {
        const jlong x = a[i], y = b[i];
        switch (op) {
        case 0: out[i] = x + y; break;
        case 1: out[i] = x - y; break;
        case 2: out[i] = x * y; break;
        case 3: out[i] = jushr(x, (int)y); break;
        case 4: out[i] = Long::compare(x, y); break;
        case 5: out[i] = Long::hashCode(x); break;
        case 6: out[i] = Math::max(x, y); break;
        case 7: out[i] = Math::min(x, y); break;
        case 8: out[i] = -x; break;
        default: out[i] = (jint)((jint)x + (jint)y); break;
        }
    }
}
void ref_jrt_int_ops(uint32_t n, int op, const int32_t *a, const int32_t *b, int32_t *out)      // 0 compareUnsigned  1 a >>> b  2 a + b
{
    for (uint32_t i = 0; i < n; i++)
        out[i] = op == 0 ? Integer::compareUnsigned(a[i], b[i]) : op == 1 ? jushr((jint)a[i], (int)b[i]) : (jint)(a[i] + b[i]);
}
void ref_jrt_math_round(uint32_t n, const double *x, int64_t *out) { for (uint32_t i = 0; i < n; i++) out[i] = Math::round(x[i]); }
void ref_jrt_arrays_sort(int64_t *v, uint32_t n)
{
    JArr<jlong> a = JArr<jlong>::make((jint)n);
    for (uint32_t i = 0; i < n; i++) a[(jint)i] = v[i];
    Arrays::sort(a);
    for (uint32_t i = 0; i < n; i++) v[i] = a[(jint)i];
}

}  // extern "C"
