// raft_host.hpp — host side of the GPU decision path, above the C-ABI (include/raftgpu.h).
//
// The reference is Java; there is no JDK in this environment, so the host layer a Java deployment would
// write as a ContextManager subclass + JNI shim is written here in C++, mirroring the
// reference's own interface names, argument meaning and error behaviour for this path:
//
//   RaftResponse        RaftResponse.java:8-24            (term, success)
//   RaftLog / MemoryLog command/RaftLog.java:72-132, with the observable semantics of
//                       command/storage/RocksLog.java:82-253 (host-owned plugin: the device never sees payloads)
//   RaftContext         context/RaftContext.java          one raft group; participant surface =
//                       RaftParticipant.java:9-51 (appendEntries / preVote / requestVote / onTimeout)
//                       plus the response callbacks of Leader.replicateLog (member/Leader.java:218-237),
//                       Candidate.startElection (member/Candidate.java:121-134) and
//                       Follower.prepareElection (member/Follower.java:258-270)
//   ContextManager      context/ContextManager.java:43-177 — here it owns ONE rg_table on one GPU
//                       instead of an EventLoopGroup(3); flush() is the EventLoop drain.
//
// Calls on a RaftContext do not decide anything on the host: each queues one row; ContextManager::flush()
// submits all queued rows in one rg_submit, then — exactly in the order the reference does it inside the
// handler — applies the log effects to the host-owned RaftLog (truncate/append, markCommitted), hands
// (term, votedFor) to the StableLock hook BEFORE the response is released (member/RaftMember.java:25),
// and only then completes the tickets.  Rows answered RG_NEED_HOST are looked up in the RaftLog and
// resubmitted with hints inside the same flush.  There is no CPU decision path: without libraftgpu.so and
// a GPU, ContextManager's constructor throws.
#pragma once

#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/raftgpu.h"
#include "stable_store.hpp"

namespace raftgpu {
namespace host {

using ID = int32_t;                        // RaftCluster.ID as a peer slot; RG_NO_NODE == Java null

struct RaftResponse {                      // RaftResponse.java:8-24
    int64_t term;
    bool success;
    static RaftResponse ok(int64_t t) { return {t, true}; }
    static RaftResponse failure(int64_t t) { return {t, false}; }
    static RaftResponse reply(int64_t t, bool r) { return {t, r}; }
};

struct Entry {                             // RaftLog.Entry: index + term (payload stays with the caller)
    int64_t index;
    int64_t term;
};

class RaftLog {                            // command/RaftLog.java:72-132
  public:
    virtual ~RaftLog() = default;
    virtual Entry epoch() const = 0;
    virtual std::optional<Entry> last() const = 0;
    virtual std::optional<Entry> get(int64_t index) const = 0;
    virtual std::optional<Entry> conflict(const std::vector<Entry> &entries) const = 0;
    virtual void truncate(int64_t index) = 0;
    virtual void append(const std::vector<Entry> &entries) = 0;
    virtual Entry newEntry(int64_t term) = 0;
    virtual int64_t lastCommitted() const = 0;
    virtual bool markCommitted(int64_t commitIndex) = 0;
    virtual void flush(int64_t index, int64_t term) = 0;
};

// In-memory RaftLog with RocksLog's observable behaviour (contiguous key window, key == epoch.index may
// survive a flush, conflict() stops at the first absent key, append() skips what is already stored).
class MemoryLog : public RaftLog {
  public:
    Entry epoch() const override { return epoch_; }
    std::optional<Entry> last() const override;
    std::optional<Entry> get(int64_t index) const override;
    std::optional<Entry> conflict(const std::vector<Entry> &entries) const override;
    void truncate(int64_t index) override;
    void append(const std::vector<Entry> &entries) override;
    Entry newEntry(int64_t term) override;
    int64_t lastCommitted() const override { return commit_; }
    bool markCommitted(int64_t commitIndex) override;
    void flush(int64_t index, int64_t term) override;
    // maximal equal-term runs of the stored window, oldest first (what rg_load_state wants)
    std::vector<Entry> runs() const;
    int64_t firstIndex() const { return first_; }

  private:
    Entry epoch_{0, 0};
    int64_t first_ = 1;                    // index of terms_[0]
    std::deque<int64_t> terms_;
    int64_t commit_ = 0;
};

using Ticket = size_t;

struct Outcome {
    std::optional<RaftResponse> response;  // empty: no RaftResponse (a callback/timer row, or the handler died)
    uint32_t status = RG_OK;               // RG_A_* = the reference would have thrown there
    uint32_t flags = 0;
    uint32_t roleEpoch = 0;                // tag to attach to RPCs emitted from now on
    int role = RG_FOLLOWER;
    unsigned emit() const { return RG_F_EMIT(flags); }
    bool resetTimer() const { return flags & RG_F_RESET_TIMER; }
    bool roleChanged() const { return flags & RG_F_ROLE_CHANGED; }
};

class ContextManager;

class RaftContext {
  public:
    const std::string &ctxID() const { return id_; }
    uint32_t gid() const { return gid_; }
    RaftLog &replicatedLog() { return *log_; }

    // mirrors of the device state, refreshed by every flush that touched this context
    int role() const { return role_; }
    int64_t currentTerm() const { return term_; }
    ID votedFor() const { return voted_for_; }
    uint32_t roleEpoch() const { return role_epoch_; }

    // ---- RaftParticipant (RaftParticipant.java:34-44, :24) ---------------------------------------
    Ticket appendEntries(int64_t term, ID leaderId, int64_t prevLogIndex, int64_t prevLogTerm,
                         const std::vector<Entry> &entries, int64_t leaderCommit);
    Ticket preVote(int64_t term, ID candidateId, int64_t lastLogIndex, int64_t lastLogTerm);
    Ticket requestVote(int64_t term, ID candidateId, int64_t lastLogIndex, int64_t lastLogTerm);
    // lastIncludedIndex/Term as in RaftParticipant.java:49; installed = the verdict of RaftContext.installSnapshot
    // (context/RaftContext.java:270-278: download + apply are the host's job, the decision row carries only the result)
    Ticket installSnapshot(int64_t term, ID leaderId, int64_t lastIncludedIndex, int64_t lastIncludedTerm, bool installed);
    // ticketEpoch: role epoch of the participant whose timer fired (ContextManager::expiredTickets), 0 = whoever is current;
    // a timeout overtaken by a row that replaced the participant is dropped (context/RaftRoutine.java:70)
    Ticket onTimeout(uint32_t ticketEpoch = 0);
    // ---- response callbacks ------------------------------------------------------------------------
    Ticket onAppendEntriesResponse(ID peer, RaftResponse result, int64_t epochIndexAtSend, int64_t lastIndexSent,
                                   uint32_t roleEpochAtSend);
    Ticket onInstallSnapshotResponse(ID peer, RaftResponse result, int64_t epochIndexAtSend, uint32_t roleEpochAtSend);
    Ticket onVoteResponse(bool preVote, ID peer, RaftResponse result, uint32_t roleEpochAtSend);
    // ---- Leader.acceptCommand x n (member/Leader.java:128-140): entries are created by the flush ----
    Ticket acceptCommand(uint32_t commands = 1);
    // ---- RaftLog.flush (log compaction decided by the host) ---------------------------------------
    Ticket compactLog(int64_t index, int64_t term);

  private:
    friend class ContextManager;
    RaftContext(ContextManager *m, std::string id, uint32_t gid, std::unique_ptr<RaftLog> log)
        : mgr_(m), id_(std::move(id)), gid_(gid), log_(std::move(log)) {}
    ContextManager *mgr_;
    std::string id_;
    uint32_t gid_;
    std::unique_ptr<RaftLog> log_;
    int role_ = RG_FOLLOWER;
    int64_t term_ = 0;
    ID voted_for_ = RG_NO_NODE;
    uint32_t role_epoch_ = 1;
};

struct SendPlan {                          // what Leader.replicateLog decided for one leader context
    rg_send_head_t head;
    std::vector<rg_send_t> to;             // one per follower j (j = slot < self ? slot : slot - 1)
};

struct PeerProgress {                      // Leadership.State as the send side needs it
    int64_t lastEpoch, nextIndex, matchIndex;
    bool pendingInstallation;
};

class ContextManager {
  public:
    // persist(ctx, term, votedFor): StableLock.persist — called before the row's response is released
    using PersistHook = std::function<void(RaftContext &, int64_t, ID)>;
    // commit(ctx, commitIndex): RaftRoutine.commitState — entries up to commitIndex may be applied
    using CommitHook = std::function<void(RaftContext &, int64_t)>;

    ContextManager(int device, uint32_t maxContexts, uint32_t clusterSize, ID self, bool preVote);
    ~ContextManager();
    ContextManager(const ContextManager &) = delete;
    ContextManager &operator=(const ContextManager &) = delete;

    // ContextManager.createContext (context/ContextManager.java:112-120): Follower(restoreTerm, restoreBallot)
    RaftContext &createContext(const std::string &id, int64_t restoreTerm = 0, ID restoreBallot = RG_NO_NODE);
    RaftContext *getContext(const std::string &id);
    void onPersist(PersistHook h) { persist_ = std::move(h); }
    // N3: every (term, votedFor) a flush marks RG_F_PERSIST goes to this journal with ONE write + ONE fdatasync,
    // before any response of that flush is released; createContext restores from it.
    void attachStableStore(StableStore *s) { store_ = s; }
    void onCommit(CommitHook h) { commit_ = std::move(h); }

    // N4: timers live on the device. configureTimers = RaftConfig's election / heartbeat intervals; every flush(now) folds
    // the RESET_TIMER / ROLE_CHANGED flags of its rows into the deadlines (RaftRoutine.resetTimer); expiredTimers(now) is
    // what electionTimeout / keepAlive would have fired by `now`: the caller answers each with ctx.onTimeout().
    void configureTimers(int64_t electionMs, int64_t heartbeatMs, uint64_t seed);
    void armTimers(int64_t now);
    std::vector<RaftContext *> expiredTimers(int64_t now);
    // the same with the role epoch of the participant whose ticket fired: pass it to ctx.onTimeout(epoch)
    std::vector<std::pair<RaftContext *, uint32_t>> expiredTickets(int64_t now);

    // N4b: Leadership.State health statistics live next to the timers. flush(now) folds statSuccess for every ack row
    // (member/Leader.java:182,229); statFailure is for RPCs the host saw fail — timeout, transport error, peer service not
    // available (member/Leader.java:187,235,240); isReady is the gate RaftStub.process applies before acceptCommand
    // (command/RaftStub.java:80-87, member/Leader.java:52-64) with RaftConfig's availableCriticalPoint / recoveryCoolDownMills.
    struct RpcFailure { const RaftContext *ctx; ID peer; bool unreachable, reject; };
    void statFailure(const std::vector<RpcFailure> &failures, int64_t now);
    std::vector<uint8_t> isReady(int64_t now, int32_t criticalPoint, int64_t coolDownMs);   // indexed by RaftContext::gid()

    bool pending(const RaftContext &c) const;  // a row for this context is already queued (one per context per flush)
    // The EventLoop drain: decide every queued row on the GPU, apply effects, complete tickets.
    // Outcome i answers ticket i of this flush; tickets restart at 0 afterwards.
    std::vector<Outcome> flush(int64_t now = -1);          // now >= 0: also maintain the device timers
    // Leader.replicateLog (member/Leader.java:142-245) for a set of contexts in one rg_replicate launch: which range to
    // ship to every follower, heartbeat[i] selects the onTimeout limits; inFlight holds State.requestInFlight per
    // (context, follower) or is empty. A send whose prevLogTerm is below the device's cached runs comes back
    // RG_SEND_NEED_HOST and is completed here from the context's RaftLog.
    std::vector<SendPlan> replicateLog(const std::vector<RaftContext *> &ctxs, const std::vector<uint8_t> &heartbeat,
                                       const std::vector<uint16_t> &inFlight = {});
    // replication progress of a leader context (read back from the device)
    std::vector<PeerProgress> progress(const RaftContext &c);

    ID self() const { return self_; }
    uint32_t clusterSize() const { return cluster_; }
    uint64_t rowsDecided() const { return rows_decided_; }
    uint64_t hintsServed() const { return hints_served_; }

  private:
    friend class RaftContext;
    struct Row {
        RaftContext *ctx;
        uint32_t hdr, aux;
        int64_t a, b, c, d;
        std::vector<Entry> entries;        // AE_REQ only
    };
    Ticket enqueue(RaftContext &c, Row row);
    void submit(std::vector<size_t> &which, bool hinted, std::vector<rg_reply_t> &rep, std::vector<rg_logfx_t> &lfx,
                std::vector<rg_persist_t> &per, int64_t now);

    rg_table_t *table_ = nullptr;
    uint32_t cluster_;
    ID self_;
    uint32_t capacity_;
    std::vector<std::unique_ptr<RaftContext>> contexts_;
    std::map<std::string, RaftContext *> by_id_;
    std::vector<Row> queue_;
    std::vector<char> queued_;             // by gid
    PersistHook persist_;
    StableStore *store_ = nullptr;
    CommitHook commit_;
    uint64_t rows_decided_ = 0, hints_served_ = 0;
};

}  // namespace host
}  // namespace raftgpu
