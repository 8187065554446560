// ingress.hpp — N2 + the host half of a8: bytes from many peer connections -> ONE multi-round compact batch the step kernel takes as it is,
// and the replies of that batch -> response frames on the connections the requests came from. Host C++, any number of decoder threads.
//
// What it replaces in the reference (paths relative to src/main/java/io/lubricant/consensus/raft/):
//   ContextIndex     the contextId -> RaftContext map                                context/ContextManager.java:41
//   PendingRing      AsyncService's invocations by (scope, sequence), one sequence counter per node
//                                                                                    transport/rpc/AsyncService.java:18-24,91-104
//   Ingress::feed    NettyCluster.on(PingEvent / PongEvent) -> context.eventLoop().execute(...): a decoded request / response becomes a task
//                    of ITS context's loop                                           transport/NettyCluster.java:59-105, NettyNode.java:109-158
//   rounds           the per-context FIFO of support/EventLoopGroup.java:32-46,77-80: tasks of one context run in the order they were
//                    queued; tasks of different contexts are unordered. Here: the k-th row a batch holds for a group lies in round k
//                    (cell [k][gid] of a dense [round][group] batch), rows of one connection keep their order, rows of different
//                    connections interleave in arrival order — exactly what execute() from several Netty threads gives.
//   Ingress::emit    reply(...) -> channel.writeAndFlush(new PongEvent(...))           transport/NettyCluster.java:75-90
//
// Threading: feed(conn, ...) may run on one thread per connection at a time, any number of connections concurrently — a row claims its
// cell with ONE relaxed fetch_add on its group's depth counter and writes 24 bytes nobody else touches; there is no lock on that path
// beyond a shared lock that only seal() takes exclusively. seal() / emit() / the pending rings' put side belong to the flush thread.
//
// Rows that cannot sit in the compact format (a value outside [0, 2^31)) or that find their group's rounds used up are kept on their
// connection and go first into the next batch, in the order they were held back (one ticket counter for all connections: a group's rows
// stay in arrival order across batches too); a group is closed for the rest of a batch once one of its rows was held back, so a later
// row never overtakes an earlier one.
#pragma once
#include <atomic>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "wire.hpp"

namespace rafting {
namespace wire {

// contextId -> gid, fixed capacity (the table's group count is fixed at rg_table_create). insert() under a mutex, find() lock-free
// (a slot is published with a release store after its key bytes are in place).
class ContextIndex {
public:
    explicit ContextIndex(uint32_t capacity);
    bool insert(const char *id, size_t len, uint32_t gid);      // false: gid beyond the capacity or taken, id empty / longer than MAX_HEAD_SIZE / present
    bool find(const char *id, size_t len, uint32_t &gid) const;
    // ContextManager.exitContext / destroyContext (context/ContextManager.java:126-171): the id stops resolving at once (its slot becomes a
    // tombstone), the group id can be given to another context after reclaim() — which the owner calls when no lookup that began before the
    // erase can still be running: an Ingress::seal() that STARTED after the erase has returned (it excludes every feed()). Returns the gid the
    // id had, or capacity.
    uint32_t erase(const char *id, size_t len);
    size_t retired();                                            // erased ids not reclaimed yet: take this BEFORE the seal ...
    void reclaim(size_t n = (size_t)-1);                         // ... and free that many (the oldest) after it: later erasures wait for the next seal
    // the same in steps, for a caller that looks several ids up at once and wants the cache misses of one to overlap the work on another:
    // hash_of, then prefetch(hash) some frames ahead, then find(hash, id, len, gid)
    static uint64_t hash_of(const char *id, size_t len) { return hash(id, len); }
    void prefetch(uint64_t h) const { __builtin_prefetch(&slot_.load(std::memory_order_acquire)[(uint32_t)h & mask_]); }
    bool find(uint64_t h, const char *id, size_t len, uint32_t &gid) const;
    void prefetch_key(uint32_t gid) const { if (gid < capacity_) __builtin_prefetch(&key_[gid]); }
    uint32_t peek(uint64_t h) const;                            // the gid of the first slot whose tag matches (capacity = none): a guess for prefetch_key
    std::string id_of(uint32_t gid) const;                      // "" if that gid was never inserted
    void append_id(uint32_t gid, std::string &to) const { if (gid < capacity_) to.append(key_[gid].bytes, key_[gid].len); }
    uint32_t size() const { return n_; }
    uint32_t tombstones() const { return tombs_; }              // (tests) slots of erased contexts that probing still has to step over
    uint32_t rebuilds() const { return rebuilds_; }

private:
    // open addressing; a slot holds gid + 1 (0 = empty) and 32 bits of the key's hash, so a probe touches the key bytes — one fixed-size
    // record per gid in ONE array, no pointer to chase — only for the slot that matches
    static constexpr size_t KEY_BYTES = 128;                    // MAX_HEAD_SIZE: a scope "<method>:<contextId>" is at most that long
    struct Key { uint8_t len; char bytes[KEY_BYTES - 1]; };     // (ids of 128 bytes cannot occur: the method name and the colon take at least 8)
    static uint64_t hash(const char *s, size_t n);
    uint32_t mask_;
    // The slot array is replaced as a whole when tombstones have piled up (rebuild(), from reclaim()): readers load the pointer once per
    // lookup; the array they may still be walking is freed one reclaim() later (see erase() for why that is late enough).
    typedef std::atomic<uint64_t> Slot;
    std::atomic<Slot *> slot_;                                  // (gid + 1) | tag << 32
    std::unique_ptr<Slot[]> owner_, previous_;
    std::unique_ptr<Key[]> key_;                                // by gid
    uint32_t capacity_;
    std::mutex mu_;
    uint32_t n_ = 0, tombs_ = 0, rebuilds_ = 0;
    void rebuild();                                             // under mu_
    std::vector<uint32_t> retired_;                             // erased, not yet reclaimed: their key records still belong to lookups in flight
    static constexpr uint32_t TOMB = 0xFFFFFFFFu;               // low half of a slot whose context was erased: probing goes on
};

// What the host remembered when it sent request `sequence` on a connection. One ring per connection; put() by the sending side,
// take() by the thread decoding that connection's responses. A slot is matched on (sequence, method, gid): a response whose request
// was overwritten (more than `capacity` requests in flight) or never sent is refused, as AsyncService.remove returns null.
class PendingRing {
public:
    explicit PendingRing(uint32_t capacity_pow2 = 1u << 16);
    void put(int32_t sequence, Method m, uint32_t gid, const Pending &p);
    bool take(int32_t sequence, Method m, uint32_t gid, Pending &p);
    void prefetch(int32_t sequence) const { __builtin_prefetch(&s_[(uint32_t)sequence & mask_]); }
    void clear();                                               // drop every filed invocation

private:
    // key = 1 | sequence << 1 | method << 33 | gid << 36 (see .cpp). The record is three relaxed atomic words between two key stores (put) /
    // a key load and a key CAS (take): a take that overlaps the put of a later request to the same slot reads garbage and then loses the CAS
    struct Slot { std::atomic<uint64_t> key{0}, w0{0}, w1{0}, w2{0}; };
    static uint64_t key_of(int32_t sequence, Method m, uint32_t gid);
    uint32_t mask_;
    std::unique_ptr<Slot[]> s_;
};

struct Origin { uint32_t conn; int32_t sequence; };             // who gets the reply of a cell (conn == NO_CONN: nobody — a response row)
constexpr uint32_t NO_CONN = 0xFFFFFFFFu;

struct HeldRow {                                                  // a row kept on its connection, in the wide form
    uint32_t gid; rg_ev_head_t head; int64_t a, b, c, d; Origin from;
    std::vector<int64_t> terms;
    uint64_t ticket = 0;                                          // order in which rows were held back, across connections
    std::string body;                                             // the request's body, when the ingress retains them (retain_bodies)
};

// One sealed batch: what rg_submit32 / rg_submit_async_packed take (dense, `rounds` rounds of `count` cells) plus where its replies go.
// An ingress in front of SEVERAL tables (block partition gpu = gid / ceil(G / N), one table and one feeder per GPU) seals one such
// batch per shard: rows are routed by their group id as they are placed, every shard's cells are a [round][group of the shard] array of their own.
struct SealedShard {
    rg_batch32_t batch;                 // gid == NULL (dense over the shard's groups); pointers into the bank
    const Origin *origin;               // [rounds * count]
    uint32_t first_gid;                 // group id of the shard's cell 0 (a table's group g is first_gid + g)
    uint64_t events;                    // cells that hold an event
};
struct SealedBatch {
    rg_batch32_t batch;                 // shard 0 — the whole batch of an unsharded ingress
    const Origin *origin;
    uint64_t rows;                      // cells that hold an event, all shards together
    std::vector<SealedShard> shard;     // [shards]; shard[0].batch is `batch`
    std::vector<HeldRow> wide;          // rows the compact format cannot express (a value beyond 2^31, entry terms of several terms that outnumber
                                        // the shard's term array), at most one per group, ascending (global) gid: the host decides them
                                        // with ONE sparse rg_submit per table AFTER this batch (their groups took no later row into the batch)
};

class Ingress {
public:
    struct Buffers {                    // caller-owned memory of ONE bank, ideally page-locked (rg_host_alloc): the batch is uploaded from it
        rg_ev_head_t *head; rg_ev_quad32_t *abcd; int32_t *entry_terms; uint64_t entry_cap;
    };
    // groups, max_rounds: the shape of a batch. conns: number of peer connections. banks: two sets of caller buffers, filled alternately.
    // shards: tables behind this ingress (block partition: shard s holds groups [s * per, min(groups, (s + 1) * per)), per = ceil(groups / shards));
    // a bank's head / abcd arrays hold the shards one after the other (shard s at cell s * per * max_rounds), its term array is split evenly.
    Ingress(uint32_t groups, uint32_t max_rounds, uint32_t conns, const BodyCodec &codec, const ContextIndex &index, Buffers bank0, Buffers bank1,
            uint32_t pending_capacity = 1u << 16,      // requests in flight per connection whose responses can still be matched
            uint32_t shards = 1);
    uint32_t shards() const { return (uint32_t)shard_.size(); }
    // The command payload never reaches the table, but the host has to write it to its RaftLog once the row answers RG_F_LOG_APPEND
    // (storage/RocksLog.java:169-225): with retain_bodies(true) — before the first feed() — the body of every AppendEntries request that
    // carries entries is kept next to its cell (one arena per connection and bank, freed by recycle()) and body() hands it back.
    void retain_bodies(bool on) { retain_ = on; }
    const char *body(const SealedBatch &b, uint32_t shard, size_t cell, size_t &len) const;     // nullptr: nothing kept for that cell

    void set_peer(uint32_t conn, int32_t peer_slot);             // the node at the other end (its slot in the cluster list)
    PendingRing &pending(uint32_t conn) { return *c_[conn].ring; }
    int32_t &send_sequence(uint32_t conn) { return c_[conn].next_sequence; }   // the sequence number encode_sends gives its next request on `conn`

    // The TCP connection behind `conn` was closed (by the peer, or by this side after feed() returned -1) and a new one takes its place: the
    // frame state machine starts over, the invocations still filed are dropped — their responses can no longer arrive; the reference fails
    // them with the channel (transport/rpc/AsyncService.java: the node's pending invocations time out) — rows already queued stay queued.
    // Called by the thread that reads that connection (or with no reader on it).
    void reset_connection(uint32_t conn);
    // Bytes as they arrive on `conn`. Returns the number of rows this call queued (placed or held), -1 once the stream broke the grammar.
    int feed(uint32_t conn, const uint8_t *data, size_t n);
    // N1's output on the wire: what rg_replicate decided for follower j of `count` leader rows (send_j = the rg_send_t of that follower, one per
    // row; head = the rows' rg_send_head_t; gid NULL = rows 0..count-1 are groups 0..count-1) as the request frames Leader.replicateLog ships —
    // appendEntries(term, self, prevLogIndex, prevLogTerm, entries, leaderCommit) / installSnapshot(term, self, epoch.index, epoch.term)
    // (member/Leader.java:168-245, transport/NettyNode.java:54-73) — appended to `out` for connection `conn`, each under the connection's next
    // sequence number with its invocation record (role epoch, epoch.index at send, lastIndex) filed for the response
    // (transport/rpc/AsyncService.java:91-104). term_of(gid, index) reads an entry's term from the host's RaftLog (the command payload is the
    // transport's and is not modelled: an entry travels as the 8 bytes of its term, as KryoBodyCodec writes it). Rows with RG_SEND_NONE /
    // RG_SEND_GATED send nothing; an RG_SEND_NEED_HOST row (prevLogIndex below the device's cached term runs) is completed here — its
    // prevLogTerm is term_of(gid, prev_index) — and counted in *need_host. Returns the frames written. One caller per connection at a time.
    struct TermOf { virtual ~TermOf() {} virtual int64_t term_of(uint32_t gid, int64_t index) = 0; };
    size_t encode_sends(uint32_t conn, int32_t self_slot, uint32_t count, const uint32_t *gid, const rg_send_head_t *head, const rg_send_t *send_j, TermOf &log,
                        std::string &out, uint32_t *need_host = nullptr);
    // A row that does not come off the wire — RG_EV_TIMEOUT from rg_timers_expired, RG_EV_CLIENT_APPEND, RG_EV_LOG_FLUSH, an installSnapshot
    // request released with the host's verdict — queued like a row of connection `conn` (give local sources connection numbers of their own:
    // one caller per connection at a time, as for feed()). reply_to: where an RG_F_REPLIED answer goes ({NO_CONN, 0}: nowhere).
    void add_row(uint32_t conn, uint32_t gid, rg_ev_head_t head, int64_t a, int64_t b, int64_t c, int64_t d, Origin reply_to);
    // Close the bank being filled and open the other one (rows held back by the closed bank are placed first). At most ONE sealed batch is
    // with the flusher at a time: sealing again before the previous batch was recycle()d throws std::logic_error. The sealed batch stays
    // valid until its recycle().
    const SealedBatch &seal();
    // After the batch was decided: one PongEvent frame per cell whose reply carries RG_F_REPLIED, appended to out[conn] in cell order.
    // N3's rule — no reply before its (term, votedFor) is durable (member/RaftMember.java:25) — is the caller's: StableStore::persist of the
    // batch's RG_F_PERSIST rows comes before this call. Cells [cell_begin, cell_end) only: several threads may share a batch, each with its
    // own `out` (responses are matched by sequence number, their order on a connection carries no meaning).
    // shard: which of the batch's shards `reply` belongs to (its rows [rounds * count] as that table returned them).
    size_t emit(const SealedBatch &b, const rg_reply_t *reply, std::vector<std::string> &out, size_t cell_begin = 0, size_t cell_end = (size_t)-1,
                uint32_t only_conn = NO_CONN, uint32_t shard = 0) const;       // only_conn: the frames of that connection alone
    // The response frame of ONE row decided outside the batch — a wide row (b.wide[i], decided by the host's sparse submit): appended to `out`
    // when its reply carries RG_F_REPLIED and a requester waits for it; returns the connection it belongs to, NO_CONN when nothing was written.
    uint32_t emit_wide(const HeldRow &row, const rg_reply_t &reply, std::string &out) const;
    // The batch is done with (decided, effects applied, replies emitted): wipe the cells it used so that its bank can be filled again.
    // Touches only that bank: runs beside feed() without a lock.
    void recycle(const SealedBatch &b);
    int bank_of(const SealedBatch &b) const { return &b == &sealed_[0] ? 0 : 1; }

    // rows of `conn` waiting for the next batch (the reader thread of that connection asks: a reader whose rows pile up because the flusher is
    // behind stops reading its socket until the next seal — TCP pushes back on the peer — instead of letting the list grow)
    size_t held_on(uint32_t conn) const { return c_[conn].held_count.load(std::memory_order_relaxed) + c_[conn].backlogged.load(std::memory_order_relaxed); }
    uint64_t refused() const { return refused_.load(std::memory_order_relaxed); }     // frames that were no decision row (unknown context, ...)
    uint64_t held() const;                                                            // rows waiting for the next batch
    // The context behind `gid` was erased (ContextIndex::erase): the rows of it that are still waiting outside a batch — held back by a reader,
    // queued in the backlog — are dropped and counted as refused, so that the group id can be reclaimed and given to ANOTHER context without that
    // context's group deciding the old one's rows. Rows already placed in the batch being filled are decided with it, by the old
    // context's group. Call it after erase() and before reclaim(); excludes the feeders like seal() does. Returns the rows dropped.
    size_t drop_rows_of(uint32_t gid);

private:
    struct Bank {
        Buffers buf;
        std::unique_ptr<std::atomic<uint32_t>[]> depth;          // per group: rows claimed; bit 30 (CLOSED): no more rows in this batch
        std::vector<Origin> origin;                              // meaningful where the cell's head holds an event
        struct BodyRef { uint32_t conn, off, len; };
        std::vector<BodyRef> body;                               // (retain_bodies) where the cell's request body lies: c_[conn].bodies[this bank]
        std::unique_ptr<std::atomic<uint64_t>[]> terms_used;     // per shard
        std::vector<uint32_t> dirty_rounds;                      // per shard: rounds to wipe before the bank is filled again
        bool clean = false;
        std::mutex wide_mu;
        std::vector<HeldRow> wide;
    };
    struct Conn {
        FrameSplitter sp;
        int32_t peer = RG_NO_NODE;
        std::unique_ptr<PendingRing> ring;
        std::vector<HeldRow> held;                               // held back since the last seal, in arrival order
        std::atomic<size_t> held_count{0};                       // held.size(), for held_on() from other threads
        std::atomic<size_t> backlogged{0};                       // its rows in backlog_ (seal() moves them there)
        std::string bodies[2];                                   // (retain_bodies) request bodies of its rows, per bank
        Request q;                                               // decode scratch
        std::string ctx;
        struct Staged { FrameView f; Method m; const char *id; size_t id_len; uint64_t hash; bool ok; };
        Staged staged[8];                                        // frames of the current read whose lookups are in flight (feed() works on
        uint32_t n_staged = 0;                                   // eight at a time so that the cache misses of one overlap the decode of another)
        int queued = 0;
        int32_t next_sequence = 0;                               // of the requests this side sends on the connection (AsyncService.sequence)
        std::vector<uint64_t> rows;                              // per shard: rows this connection placed into the bank being filled, and the
        std::vector<uint32_t> max_depth;                         // deepest round it reached (folded by seal(): no shared counter on the row path)
    };
    struct Shard { uint32_t first, count; size_t cell_off; uint64_t term_off, term_cap; };
    static constexpr uint32_t CLOSED = 1u << 30;
    bool place(Bank &bk, Conn &c, uint32_t gid, rg_ev_head_t head, int64_t a, int64_t b, int64_t c4, int64_t d, const int64_t *terms, size_t n_terms,
               Origin from, const char *body = nullptr, size_t body_len = 0);
    void hold(Conn &c, uint32_t gid, rg_ev_head_t head, int64_t a, int64_t b, int64_t c4, int64_t d, const int64_t *terms, size_t n_terms, Origin from,
              const char *body = nullptr, size_t body_len = 0);
    void stage(uint32_t conn, const FrameView &f);
    void drain(uint32_t conn);
    void on_frame(uint32_t conn, const Conn::Staged &s);
    void wipe(Bank &bk);

    const uint32_t groups_, rounds_;
    bool retain_ = false;
    uint32_t per_shard_ = 0;
    std::vector<Shard> shard_;
    const BodyCodec &codec_;
    const ContextIndex &index_;
    std::vector<Conn> c_;
    // rows held back over more than one batch, one FIFO per group (ticket order): seal() places at most max_rounds of a group per batch, so what a
    // seal costs follows the rows it can place, not the size of the backlog (a peer that floods one group must not make every seal walk its flood)
    struct Waiting { HeldRow row; uint32_t conn; };
    std::unordered_map<uint32_t, std::deque<Waiting>> backlog_;
    Bank bank_[2];
    SealedBatch sealed_[2];
    int fill_ = 0;                                               // bank being filled (changed under the exclusive lock)
    mutable std::shared_mutex mu_;
    mutable std::atomic<uint32_t> sealing_{0};                           // seal() is waiting for / holds the exclusive lock: feeders stand back (the rwlock
                                                                 // alone prefers readers, a stream of overlapping feed() calls would starve the flusher)
    std::atomic<uint64_t> refused_{0}, ticket_{0};
};

// ---- RG_NEED_HOST inside a multi-round batch -------------------------------------------------------------------------------------------
// A row whose term lookup leaves the device's cached runs comes back RG_NEED_HOST, unapplied, and the later rows of its group in the same
// launch come back RG_SKIPPED_AFTER_NEED_HOST (include/raftgpu.h). Before the next batch may be submitted those rows have to be decided, in
// order: the missed row again with a hint read from the host's RaftLog (rg_batch_t.hint), then the group's later rows of the batch one by
// one — each can miss in its turn. repair_need_host does that with one sparse single-round rg_submit per step for ALL broken groups
// together, writes the final replies over reply[cell] (so that emit() answers the requests) and hands every repaired row's outcome to the
// host at once: the hint of a group's next row is read from the log as that row left it.
struct RepairHost {
    virtual ~RepairHost() {}
    virtual int64_t term_at(uint32_t gid, int64_t index) = 0;                   // RaftLog.get(index).term(), -1: no such entry (storage/RocksLog.java:122-128)
    virtual int64_t conflict(uint32_t gid, int64_t first_index, const int64_t *terms, uint32_t n) = 0;   // RaftLog.conflict(entries).index(), 0: none (:199-225)
    virtual int64_t epoch_index(uint32_t gid) = 0;                              // RaftLog.epoch().index()
    virtual int submit(const rg_batch_t &in, const rg_outcome_t &out) = 0;      // rg_submit(table, &in, &out, RG_MEM_HOST); 0 = ok
    // a repaired row was applied: its log effects / (term, votedFor) go to the host-owned plugins NOW, before the group's next row is hinted
    virtual void applied(uint32_t gid, size_t cell, const rg_reply_t &reply, const rg_logfx_t &logfx, const rg_persist_t &persist) = 0;
};
// logfx: the batch's log-effect rows — dense [rounds * groups] (rg_submit32) or, packed = true, the row-ordered list of
// rg_submit_async_packed. Returns the number of rows decided here (0: the batch had no RG_NEED_HOST), -1 when a submit failed or a hinted
// row still missed.
// shard: the shard of `b` that reply / logfx belong to; the gids handed to `host` are the TABLE's (0 .. count-1 of that shard).
int64_t repair_need_host(const SealedBatch &b, rg_reply_t *reply, const rg_logfx_t *logfx, bool packed, RepairHost &host, uint32_t shard = 0);

}  // namespace wire
}  // namespace rafting
