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
// The default column family holds only 8-byte keys (storage/RocksLog.java:86,190); equal neighbouring values are stored as
// one run [start, end] so that a log of 2^60 entries costs one node.  Order = RocksDB's bytewise comparator = unsigned
// order of the big-endian key.  This is a representation of the fake, invisible through get / put / deleteRange / iterators.
struct RocksDB : virtual Object {
    Class *klass_() override { return nullptr; }
    static constexpr jint NOT_FOUND = -1;
    struct Run { uint64_t end; std::string val; };
    std::map<uint64_t, Run> kv;                      // start -> run
    std::map<std::string, std::string> cf;           // "epoch" column family
    // observations for the driver (what the host-owned RaftLog plugin was told to do during one event)
    jlong puts_new = 0; jlong first_new_key = 0; bool truncated = false; jlong trunc_from = 0;
    static uint64_t ukey(const JArr<jbyte> &a)
    {
        if (a->length != 8) throw IllegalArgumentException("fake RocksDB: default column family keys are 8 bytes");
        uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | (uint8_t)a[i]; return v;
    }
    static JArr<jbyte> kbytes(uint64_t v) { JArr<jbyte> a = JArr<jbyte>::make(8); for (int i = 7; i >= 0; i--) { a[i] = (jbyte)(v & 0xFF); v >>= 8; } return a; }
    static jlong key_to_long(const std::string &k) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | (uint8_t)k[(size_t)i]; return (jlong)v; }
    void observe_reset() { puts_new = 0; first_new_key = 0; truncated = false; trunc_from = 0; }
    std::map<uint64_t, Run>::iterator find_run(uint64_t k)
    {
        auto it = kv.upper_bound(k);
        if (it == kv.begin()) return kv.end();
        --it;
        return it->second.end >= k ? it : kv.end();
    }
    void erase_range(uint64_t a, uint64_t b_incl)                 // remove keys a..b_incl
    {
        auto it = kv.upper_bound(a);
        if (it != kv.begin()) --it;
        while (it != kv.end() && it->first <= b_incl) {
            uint64_t s0 = it->first, e0 = it->second.end; std::string v = it->second.val;
            if (e0 < a) { ++it; continue; }
            it = kv.erase(it);
            if (s0 < a) kv[s0] = Run{a - 1, v};
            if (e0 > b_incl) { kv[b_incl + 1] = Run{e0, v}; break; }
        }
    }
    JArr<jbyte> get(const JArr<jbyte> &key) { auto it = find_run(ukey(key)); return it == kv.end() ? JArr<jbyte>(nullptr) : jval(it->second.val); }
    jint get(const JArr<jbyte> &key, const JArr<jbyte> &buf)
    {
        auto it = find_run(ukey(key));
        if (it == kv.end()) return NOT_FOUND;
        const std::string &v = it->second.val;
        for (jint i = 0; i < buf->length && (size_t)i < v.size(); i++) buf[i] = (jbyte)v[(size_t)i];
        return (jint)v.size();
    }
    void put(const JArr<jbyte> &key, const JArr<jbyte> &val)
    {
        uint64_t k = ukey(key); std::string v = jkey(val);
        auto it = find_run(k);
        if (it == kv.end()) { if (puts_new == 0) first_new_key = (jlong)k; puts_new++; }
        else if (it->second.val == v) return;
        else erase_range(k, k);
        uint64_t s0 = k, e0 = k;
        auto nx = kv.find(k + 1);
        if (k != UINT64_MAX && nx != kv.end() && nx->second.val == v) { e0 = nx->second.end; kv.erase(nx); }
        if (k != 0) {
            auto pv = find_run(k - 1);
            if (pv != kv.end() && pv->second.val == v) { s0 = pv->first; kv.erase(pv); }
        }
        kv[s0] = Run{e0, v};
    }
    void put(Ref<ColumnFamilyHandle>, const JArr<jbyte> &key, const JArr<jbyte> &val) { cf[jkey(key)] = jkey(val); }
    void flushWal(jboolean) {}
    void deleteRange(const JArr<jbyte> &a, const JArr<jbyte> &b)
    {
        uint64_t ka = ukey(a), kb = ukey(b);
        if (ka >= kb) return;                                     // RocksDB: an empty or inverted range deletes nothing
        auto it = kv.upper_bound(kb - 1);                         // is any key inside [ka, kb)?
        bool any = false;
        if (it != kv.begin()) { --it; any = it->second.end >= ka; }
        if (any) { truncated = true; trunc_from = (jlong)ka; }
        erase_range(ka, kb - 1);
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
    RocksDB *db; uint64_t k = 0; std::string v; bool valid = false;
    RocksIterator(RocksDB *d) : db(d) {}
    void seekToLast() { valid = !db->kv.empty(); if (valid) { auto it = std::prev(db->kv.end()); k = it->second.end; v = it->second.val; } }
    void seekForPrev(const JArr<jbyte> &key)                      // last entry whose key <= target
    {
        uint64_t t = RocksDB::ukey(key);
        auto it = db->kv.upper_bound(t);
        valid = it != db->kv.begin();
        if (valid) { --it; k = it->second.end < t ? it->second.end : t; v = it->second.val; }
    }
    jboolean isValid() { return valid; }
    JArr<jbyte> key() { return RocksDB::kbytes(k); }
    JArr<jbyte> value() { return jval(v); }
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
