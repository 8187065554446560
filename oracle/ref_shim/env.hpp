// env.hpp — the reference's I/O plugins and thread pools, faked (test infrastructure; hand-written, NOT reference text).
//
// The translated decision classes (oracle/_ref/gen/, made by tools/make_ref.py) call out to RocksDB, StableLock, the
// Netty RaftService stubs, scheduled executors and an EventLoop.  This header provides in-memory stand-ins with the
// same method names so the translated code compiles unchanged:
//   RocksDB                an ordered byte-key map (std::map over 8-byte big-endian keys, bytewise order as RocksDB's
//                          default comparator), iterators, deleteRange [a,b), column family "epoch"
//   RaftCluster/RaftService  sends are recorded in an outbox; Async.on registers the callback with its AsyncHead
//   AsyncHead              abortRequests() cancels every pending callback with (null, null, canceled=true)
//                          (transport/rpc/Async.java:157-171,231-247)
//   ContextEventLoop       a deque; execute(task, urgent) queues at the head (support/EventLoop.java:87-101); the driver
//                          sets inEventLoop and drains
//   ScheduledExecutorService  a virtual-time timer list; the driver advances the clock and fires what is due
//   StableLock.persist     records (term, ballot) and counts participants — RaftMember.<init> calls it once per object
// RaftContext and RaftRoutine are skeletons here: their DATA members and the I/O-side methods (commitState, compactLog,
// installSnapshot, joinSnapshot) are stubs, their decision methods come from the reference (gen/*.decls.inc).
#pragma once
#include "jrt.hpp"

#include <deque>
#include <memory>

// ---- forward declarations ------------------------------------------------------------------------------------------
struct Class; struct ID; struct Command; struct Promise; struct AsyncHead; struct RaftService; struct RaftCluster;
struct RaftConfig; struct StableLock; struct RaftMachine; struct SnapshotArchive; struct ContextEventLoop; struct RocksDB;
struct RocksIterator; struct ColumnFamilyHandle; struct RocksSerializer; struct RocksStateLoader; struct Path;
struct ScheduledFuture; struct ScheduledExecutorService; struct RaftContext; struct RaftRoutine; struct Boolean;
struct NotLeaderException; struct ObsoleteContextException; struct RefEnv;
template <class T> struct Async;
template <class T> struct Future;
#include "gen/ref_fwd.hpp"

// ---- java.lang.Class of the three roles (context/member/Membership.java:47-48) ------------------------------------
struct Class : virtual Object {
    const char *name;
    std::function<Ref<RaftParticipant>(Ref<RaftContext>, jlong, Ref<ID>, Ref<Membership>)> make;
    Class(const char *n) : name(n) {}
    Ref<RaftParticipant> newInstance(Ref<RaftContext> c, jlong term, Ref<ID> ballot, Ref<Membership> m);
    JString getSimpleName() { return JString(name); }
};
template <class T> jboolean jinstanceof(const Ref<T> &x, const Ref<Class> &c) { return x != nullptr && x->klass_() == c.get(); }

struct ID : virtual Object {             // RaftCluster.ID: NodeID(host, port) -> a peer slot
    Class *klass_() override { return nullptr; }
    jint slot;
    ID(jint s) : slot(s) {}
    jboolean equals(Ref<Object> o) { ID *x = dynamic_cast<ID *>(o.get()); return x != nullptr && x->slot == slot; }
};
struct Command : virtual Object { Class *klass_() override { return nullptr; } };
struct Promise : virtual Object {
    Class *klass_() override { return nullptr; }
    int failed = 0;
    void whenTimeout(const std::function<void()> &) {}
    template <class E> void completeExceptionally(const E &) { failed = 1; }
    void finish() {}
};
struct NotLeaderException : virtual Object { Class *klass_() override { return nullptr; } NotLeaderException(Ref<RaftParticipant>); };
struct ObsoleteContextException : virtual Object { Class *klass_() override { return nullptr; } };
struct Boolean : virtual Object { Class *klass_() override { return nullptr; } static inline Ref<Boolean> TRUE_ = nullptr; };
template <class T> struct Future : virtual Object { Class *klass_() override { return nullptr; } };
struct CompletableFuture_ { template <class T> static Ref<Future<Boolean>> completedFuture(const T &) { return jnew<Future<Boolean>>(); } };
#define TRUE TRUE_                       // `Boolean.TRUE` (storage/RocksLog.java:241)
struct Path : virtual Object { Class *klass_() override { return nullptr; } };
struct RocksStateLoader : virtual Object { Class *klass_() override { return nullptr; } };
struct Snapshot : virtual Object { Class *klass_() override { return nullptr; } };

static Ref<Logger> &logger_ref() { static Ref<Logger> l = jnew<Logger>(); return l; }
#define logger (logger_ref())

// ---- transport/rpc/Async.java -----------------------------------------------------------------------------------------
enum { REF_RPC_AE = 1, REF_RPC_PV = 2, REF_RPC_RV = 3, REF_RPC_IS = 4 };
typedef std::function<void(Ref<RaftResponse>, Ref<Throwable>, jboolean)> AsyncCallback;
struct PendingCall : Object {
    int kind = 0, peer = -1;
    jlong term = 0, a = 0, b = 0, commit = 0;     // request fields (AE: prevIndex, prevTerm; votes: lastIndex, lastTerm; IS: epoch)
    jint count = 0;                               // AE: entries shipped
    jlong first_term = 0, last_index = 0;
    AsyncCallback cb;
    AsyncHead *head = nullptr;
    bool done = false;
};
struct AsyncHead : virtual Object {
    Class *klass_() override { return nullptr; }
    bool aborted = false;
    std::vector<Ref<PendingCall>> calls;
    jboolean isAborted() { return aborted; }
    void abortRequests()
    {
        if (aborted) return;
        aborted = true;
        std::vector<Ref<PendingCall>> live;
        live.swap(calls);
        for (auto &c : live)
            if (!c->done) { c->done = true; c->cb(nullptr, nullptr, true); }     // AsyncFuture.onAbort -> cancel -> done()
    }
};
struct RefEnv;
template <class T> struct Async : virtual Object {
    Class *klass_() override { return nullptr; }
    Ref<PendingCall> call;
    RefEnv *env = nullptr;
    void on(Ref<AsyncHead> head, jlong timeout, AsyncCallback cb);
};

struct Async_ { static Ref<AsyncHead> head() { return jnew<AsyncHead>(); } };     // `Async.head()`

struct RaftService : virtual Object {
    Class *klass_() override { return nullptr; }
    RefEnv *env; int peer;
    RaftService(RefEnv *e, int p) : env(e), peer(p) {}
    Ref<Async<RaftResponse>> appendEntries(jlong term, Ref<ID> leaderId, jlong prevLogIndex, jlong prevLogTerm, JArr<Ref<Entry>> entries, jlong leaderCommit);
    Ref<Async<RaftResponse>> preVote(jlong term, Ref<ID> candidateId, jlong lastLogIndex, jlong lastLogTerm);
    Ref<Async<RaftResponse>> requestVote(jlong term, Ref<ID> candidateId, jlong lastLogIndex, jlong lastLogTerm);
    Ref<Async<RaftResponse>> installSnapshot(jlong term, Ref<ID> leaderId, jlong lastIncludedIndex, jlong lastIncludedTerm);
};
struct RaftCluster : virtual Object {
    Class *klass_() override { return nullptr; }
    RefEnv *env; jint n; Ref<ID> self; Ref<Set<ID>> remotes; std::vector<Ref<RaftService>> svc;
    RaftCluster(RefEnv *e, jint cluster, jint self_slot);
    jint size() { return n; }
    Ref<ID> localID() { return self; }
    Ref<Set<ID>> remoteIDs() { return remotes; }
    Ref<RaftService> remoteService(Ref<ID> id, JString) { return svc[(size_t)id->slot]; }
};

struct RaftConfig : virtual Object {
    Class *klass_() override { return nullptr; }
    jboolean pre_vote = true;
    jlong election_ms = 900, heartbeat_ms = 300, broadcast_ms = 150;
    jint critical_point = 0; jlong cool_down = 0;
    std::function<jlong()> election_draw;            // RaftConfig.electionTimeout(): uniform in [E, 2E] (support/RaftConfig.java:187-190)
    jboolean preVote() { return pre_vote; }
    jlong electionTimeout() { return election_draw ? election_draw() : election_ms; }
    jlong heartbeatInterval() { return heartbeat_ms; }
    jlong broadcastTimeout() { return broadcast_ms; }
    jint availableCriticalPoint() { return critical_point; }
    jlong recoveryCoolDownMills() { return cool_down; }
};
struct StableLock : virtual Object {
    Class *klass_() override { return nullptr; }
    RefEnv *env;
    StableLock(RefEnv *e) : env(e) {}
    void persist(jlong term, Ref<ID> candidate);     // support/StableLock.java:69-80
};
struct RaftMachine : virtual Object { Class *klass_() override { return nullptr; } jlong lastApplied() { return 0; } };
struct SnapshotArchive : virtual Object { Class *klass_() override { return nullptr; } void cleanPending() {} };

// ---- support/EventLoop.java -------------------------------------------------------------------------------------------
struct ContextEventLoop : virtual Object {
    Class *klass_() override { return nullptr; }
    bool in_loop = true;
    std::deque<std::function<void()>> q;
    jboolean inEventLoop() { return in_loop; }
    jboolean isAvailable() { return true; }
    void execute(const std::function<void()> &task, jboolean urgent = false) { if (urgent) q.push_front(task); else q.push_back(task); }
};

// ---- java.util.concurrent.ScheduledExecutorService on a virtual clock --------------------------------------------
struct TimeUnit { static constexpr int MILLISECONDS = 0; };
struct ScheduledFuture : virtual Object {
    Class *klass_() override { return nullptr; }
    jlong due = 0; std::function<void()> task; bool cancelled = false, ran = false;
    jboolean cancel(jboolean) { if (cancelled || ran) return false; cancelled = true; return true; }
    jboolean isCancelled() { return cancelled; }
};
struct ScheduledExecutorService : virtual Object {
    Class *klass_() override { return nullptr; }
    std::vector<Ref<ScheduledFuture>> pending;
    Ref<ScheduledFuture> schedule(const std::function<void()> &task, jlong delay, int)
    {
        Ref<ScheduledFuture> f = jnew<ScheduledFuture>();
        f->due = System::currentTimeMillis() + (delay < 0 ? 0 : delay);
        if (delay > 0 && f->due < 0) f->due = Long::MAX_VALUE;     // muted timers: now + Long.MAX_VALUE - now
        f->task = task;
        pending.push_back(f);
        if (pending.size() > 64) {                                  // forget the cancelled ones
            std::vector<Ref<ScheduledFuture>> keep;
            for (auto &p : pending) if (!p->cancelled && !p->ran) keep.push_back(p);
            pending.swap(keep);
        }
        return f;
    }
};

// ---- org.rocksdb ---------------------------------------------------------------------------------------------------------
inline JArr<jbyte> jbytes(const char *s) { size_t n = strlen(s); JArr<jbyte> a = JArr<jbyte>::make((jlong)n); for (size_t i = 0; i < n; i++) a[(jlong)i] = (jbyte)s[i]; return a; }
inline std::string jkey(const JArr<jbyte> &a) { std::string k; for (jint i = 0; i < a->length; i++) k.push_back((char)a[i]); return k; }
inline JArr<jbyte> jval(const std::string &s) { JArr<jbyte> a = JArr<jbyte>::make((jlong)s.size()); for (size_t i = 0; i < s.size(); i++) a[(jlong)i] = (jbyte)s[i]; return a; }
struct ColumnFamilyHandle : virtual Object { Class *klass_() override { return nullptr; } };
struct RocksDB : virtual Object {
    Class *klass_() override { return nullptr; }
    static constexpr jint NOT_FOUND = -1;
    std::map<std::string, std::string> kv, cf;       // default column family, "epoch" column family
    // observations for the driver (what the host-owned RaftLog plugin was told to do during one event)
    jlong puts_new = 0; jlong first_new_key = 0; bool truncated = false; jlong trunc_from = 0;
    static jlong key_to_long(const std::string &k) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | (uint8_t)k[(size_t)i]; return (jlong)v; }
    void observe_reset() { puts_new = 0; first_new_key = 0; truncated = false; trunc_from = 0; }
    JArr<jbyte> get(const JArr<jbyte> &key) { auto it = kv.find(jkey(key)); return it == kv.end() ? JArr<jbyte>(nullptr) : jval(it->second); }
    jint get(const JArr<jbyte> &key, const JArr<jbyte> &buf)
    {
        auto it = kv.find(jkey(key));
        if (it == kv.end()) return NOT_FOUND;
        for (jint i = 0; i < buf->length && (size_t)i < it->second.size(); i++) buf[i] = (jbyte)it->second[(size_t)i];
        return (jint)it->second.size();
    }
    void put(const JArr<jbyte> &key, const JArr<jbyte> &val)
    {
        std::string k = jkey(key);
        if (!kv.count(k)) { if (puts_new == 0) first_new_key = key_to_long(k); puts_new++; }
        kv[k] = jkey(val);
    }
    void put(Ref<ColumnFamilyHandle>, const JArr<jbyte> &key, const JArr<jbyte> &val) { cf[jkey(key)] = jkey(val); }
    void flushWal(jboolean) {}
    void deleteRange(const JArr<jbyte> &a, const JArr<jbyte> &b)
    {
        std::string ka = jkey(a), kb = jkey(b);
        if (ka >= kb) return;                                     // RocksDB: empty or inverted range deletes nothing
        auto lo = kv.lower_bound(ka), hi = kv.lower_bound(kb);
        if (lo != hi) { truncated = true; trunc_from = key_to_long(ka); }
        kv.erase(lo, hi);
    }
    Ref<List<JArr<jbyte>>> multiGetAsList(Ref<List<JArr<jbyte>>> keys)
    {
        Ref<List<JArr<jbyte>>> out = jnew<ArrayList<JArr<jbyte>>>();
        for (auto &k : keys->v) out->add(get(k));
        return out;
    }
    Ref<RocksIterator> newIterator();
};
struct RocksIterator : virtual Object {
    Class *klass_() override { return nullptr; }
    RocksDB *db; std::map<std::string, std::string>::iterator it; bool valid = false;
    RocksIterator(RocksDB *d) : db(d) {}
    void seekToLast() { valid = !db->kv.empty(); if (valid) it = std::prev(db->kv.end()); }
    void seekForPrev(const JArr<jbyte> &key)                      // last entry whose key <= target
    {
        auto ub = db->kv.upper_bound(jkey(key));
        valid = ub != db->kv.begin();
        if (valid) it = std::prev(ub);
    }
    jboolean isValid() { return valid; }
    JArr<jbyte> key() { return jval(it->first); }
    JArr<jbyte> value() { return jval(it->second); }
};
inline Ref<RocksIterator> RocksDB::newIterator() { return jnew<RocksIterator>(this); }
struct RocksSerializer : virtual Object {                          // value = 8-byte term prefix + payload (none here)
    Class *klass_() override { return nullptr; }
    JArr<jbyte> serialize(const JArr<jbyte> &prefix, Ref<Command>) { return prefix; }
};

#include "gen/ref_decls.hpp"

// ---- skeletons around the extracted methods -------------------------------------------------------------------------
struct RaftRoutine : virtual Object {
    Class *klass_() override { return nullptr; }
    RefEnv *env = nullptr;
    Ref<ScheduledExecutorService> electionTimer = jnew<ScheduledExecutorService>();     // context/RaftRoutine.java:39-40
    Ref<ScheduledExecutorService> heartbeatKeeper = jnew<ScheduledExecutorService>();
#include "gen/RaftRoutine.decls.inc"
    // I/O side, not decisions: stubs
    void commitState(Ref<RaftContext>, const std::function<Ref<Promise>(Ref<Entry>)> &, jint);   // :224-306 apply loop
    void compactLog(Ref<RaftContext>) {}                                                          // :308-400 snapshot policy
    jboolean installSnapshot(Ref<RaftContext>, jlong, jlong, const std::function<void()> &);      // :408-541 download + apply
};
struct RaftContext : virtual Object {
    Class *klass_() override { return nullptr; }
    RefEnv *env = nullptr;
    JString id = "ctx";
    Ref<RaftConfig> envConfig_; Ref<RaftLog> replicatedLog_; Ref<RaftMachine> stateMachine_; Ref<StableLock> stableStorage_;
    Ref<SnapshotArchive> snapArchive_; Ref<RaftCluster> cluster_; Ref<RaftRoutine> routine; Ref<ContextEventLoop> eventLoop_;
    Ref<Map<EntryKey, Promise>> commandPromises = jnew<ConcurrentHashMap<EntryKey, Promise>>();   // context/RaftContext.java:49
    Ref<AtomicReference<TimerTicket>> ticketHolder = jnew<AtomicReference<TimerTicket>>();        // :51
    Ref<AtomicReference<Membership>> membershipFilter = jnew<AtomicReference<Membership>>();      // :52
    jboolean stillRunning_ = true;
#include "gen/RaftContext.decls.inc"
    void joinSnapshot() { snapArchive_->cleanPending(); }                                         // :283-297 (no installation pending)
    jboolean installSnapshot(Ref<ID> leaderId, jlong lastIncludedIndex, jlong lastIncludedTerm);   // :270-278 -> routine (host I/O)
};

inline NotLeaderException::NotLeaderException(Ref<RaftParticipant>) {}
inline Ref<RaftParticipant> Class::newInstance(Ref<RaftContext> c, jlong term, Ref<ID> ballot, Ref<Membership> m) { return make(c, term, ballot, m); }

// Leader.prepareReplication wraps its map so that values() returns a fixed-order list (member/Leader.java:44-49)
template <class K, class V> struct FixedValuesMap : ConcurrentHashMap<K, V> {
    Ref<List<V>> fixed;
    Ref<Collection<V>> values() override { return fixed; }
};
template <class K, class V> Ref<Map<K, V>> jrt_fixed_values_map(Ref<Map<K, V>> map, Ref<List<V>> states)
{
    Ref<FixedValuesMap<K, V>> m = jnew<FixedValuesMap<K, V>>();
    m->kv = map->kv;
    m->fixed = states;
    return m;
}

#include "gen/ref_defs.hpp"

#define REF_CLASS(X) Ref<Class> X##_class = jnew<Class>(#X);
REF_CLASS(RaftResponse) REF_CLASS(RaftParticipant) REF_CLASS(Entry) REF_CLASS(EntryKey) REF_CLASS(RaftLog) REF_CLASS(RocksEntry)
REF_CLASS(Membership) REF_CLASS(State) REF_CLASS(Leadership) REF_CLASS(RaftMember) REF_CLASS(TimerTicket) REF_CLASS(Follower)
REF_CLASS(Candidate) REF_CLASS(Leader) REF_CLASS(RocksLog)
