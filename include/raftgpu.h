/*
 * raftgpu.h — C-ABI of the MI355X batched multi-Raft decision engine (libraftgpu.so).
 *
 * This is the drop-in boundary for ONE path of curioloop/rafting: the per-RaftContext
 * EventLoop decision logic (io.lubricant.consensus.raft.context.**).  Everything the Java
 * host would bind through JNI is declared here with plain pointers and sizes; no C++ or
 * torch types appear in any signature.  Paths below are relative to
 *   /root/reference/src/main/java/io/lubricant/consensus/raft/
 *
 * What each entry point replaces
 * ------------------------------
 *   rg_table_create / rg_table_destroy
 *       ContextManager.buildContext + RaftContext.<init>/initialize
 *       (context/ContextManager.java:57-106, context/RaftContext.java:62-113): instead of one
 *       RaftContext object per group, ONE structure-of-arrays table of G groups in HBM.
 *   rg_load_state / rg_read_state
 *       StableLock.restore -> switchTo(Follower, term, ballot) (context/RaftContext.java:96-104),
 *       RaftLog.epoch()/last() (command/storage/RocksLog.java:112-119) and, for leaders,
 *       Leader.prepareReplication (context/member/Leader.java:30-50).
 *   rg_submit
 *       The EventLoop drain itself (support/EventLoopGroup.java:32-46): one row = one task that
 *       the reference would have run on a ContextLoop thread (or, for responses, on a Netty
 *       thread), i.e. one call of
 *         RaftParticipant.appendEntries / preVote / requestVote   (RaftParticipant.java:34-44)
 *         Leader.replicateLog response callbacks                  (member/Leader.java:174-188,218-237)
 *         Candidate.startElection / Follower.prepareElection tally callbacks
 *                                                                (member/Candidate.java:121-134, member/Follower.java:258-270)
 *         RaftParticipant.onTimeout                               (RaftParticipant.java:24)
 *         Leader.acceptCommand -> RaftLog.newEntry                (member/Leader.java:128-140, storage/RocksLog.java:82-89)
 *         RaftLog.flush                                           (storage/RocksLog.java:228-242)
 *       The reply row is the RaftResponse(term, success) (RaftResponse.java:8-24) plus the
 *       instructions the host-owned plugins (RaftLog, StableLock, timers) must carry out.
 *
 * Ordering contract: rows of one group are applied in (round, row) order — the only ordering
 * the reference EventLoop guarantees (one context is pinned to one loop thread,
 * support/EventLoopGroup.java:77-80).  Groups are independent.
 *
 * Error convention: functions return 0 on success, <0 on API misuse / HIP failure
 * (text via rg_last_error).  Reference protocol violations (AssertionError sites) are NOT API
 * errors: they are reported per row in rg_reply_t.flags status bits, bit-exactly as the
 * reference would have hit them (the handler dies, no reply is sent, earlier side effects stay).
 *
 * Environment knobs read by rg_table_create (experiments / tests only): RG_FAST=0 routes every row through the
 * general handlers (no fast-path tier); RG_SPLIT=1|0 forces the two-wavefront (decide + I/O) or the single-wavefront
 * step kernel instead of choosing per launch; RG_LANES=8|16|32|64 sets the raft groups per wavefront of the latter.
 *
 * Threading: a table is not re-entrant; one host thread + one HIP stream per table.  Different
 * tables (different GPUs) are fully independent.  No RCCL, no cross-table traffic.
 */
#ifndef RAFTGPU_H
#define RAFTGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RG_ABI_VERSION      5     /* 2: RG_EV_IS_REQ, RG_EV_TIMEOUT.aux fence, RG_F_TIMER_MUTED, rg_timers_expired_epochs, rg_submit_async(_packed), 48-byte rg_send_head_t
                                     3: rg_submit32 / rg_batch32_pack, RG_HDR_SAME_TERM in rg_batch32_t rows
                                     4: compact OUTCOME rows (rg_out32_t, rg_submit32c, rg_outcome32_unpack, RG_F_WIDE_VALUES); rg_table_option; the index base of the compact formats (rg_index_base_set)
                                     5: the device-resident tick on compact outcome rows: rg_timers_update32, rg_health_update32, rg_tick2_*; clusters of up to 15 nodes */
#define RG_MIN_CLUSTER      2     /* P: cluster size incl. self (RaftCluster.size()) */
#define RG_MAX_CLUSTER      15    /* (ABI 5; the slot field of a row header is 4 bits. Leadership.majorIndices sorts any number of followers, member/Leadership.java:116-130.)
                                     Clusters of up to RG_MAX_COMPACT_CLUSTER nodes have every kernel; larger ones are decided by the wide-row kernels only:
                                     rg_submit / rg_submit_async take them, the compact formats (rg_submit32*, rg_submit_async_packed, rg_tick*) answer -1 */
#define RG_MAX_COMPACT_CLUSTER 7
#define RG_TERM_RUNS        4     /* K: cached term runs of the log tail per group */
#define RG_NO_NODE          (-1)  /* Java null for a RaftCluster.ID */

/* ---- roles: class of RaftContext.participant() (context/RaftContext.java:169) ------------ */
enum { RG_FOLLOWER = 0, RG_CANDIDATE = 1, RG_LEADER = 2 };

/* ---- event kinds (rg_ev_head_t.hdr bits 0-3) ------------------------------------------------ */
enum {
    RG_EV_NONE          = 0,  /* row not addressed this round                                          */
    RG_EV_AE_REQ        = 1,  /* RaftParticipant.appendEntries       a=term b=prevLogIndex c=prevLogTerm d=leaderCommit
                                 slot=leaderId n=#entries aux=offset of entry terms in entry_terms[];
                                 entry k has index prevLogIndex+1+k (what Leader.replicateLog builds, member/Leader.java:192-212) */
    RG_EV_AE_ACK        = 2,  /* Leader AE response callback         a=result.term b=epoch.index at send c=lastIndex sent
                                 slot=responder flag=result.success aux=roleEpoch the request was sent under */
    RG_EV_IS_ACK        = 3,  /* Leader InstallSnapshot callback     a=result.term b=epoch.index at send, slot, flag, aux as AE_ACK */
    RG_EV_RV_REQ        = 4,  /* RaftParticipant.requestVote        a=term b=lastLogIndex c=lastLogTerm slot=candidateId */
    RG_EV_PV_REQ        = 5,  /* RaftParticipant.preVote            same fields */
    RG_EV_RV_REPLY      = 6,  /* Candidate.startElection callback    a=result.term slot=responder flag=result.success aux=roleEpoch */
    RG_EV_PV_REPLY      = 7,  /* Follower.prepareElection callback   same fields */
    RG_EV_TIMEOUT       = 8,  /* RaftParticipant.onTimeout (election timer for F/C, heartbeat tick for L).
                                 aux = role epoch of the participant whose ticket fired (rg_timers_expired_epochs reports it), or 0 =
                                 "whoever is current". The reference runs onTimeout only if ticket.participant() == context.participant()
                                 (context/RaftRoutine.java:70): a row whose aux names a replaced participant is RG_DROPPED_STALE_ROLE */
    RG_EV_CLIENT_APPEND = 9,  /* Leader.acceptCommand x n            n=#commands appended at currentTerm */
    RG_EV_LOG_FLUSH     = 10, /* RaftLog.flush(index, term)          a=index b=term (log compaction / snapshot install moved the epoch) */
    RG_EV_IS_REQ        = 11  /* RaftParticipant.installSnapshot    a=term b=lastIncludedIndex c=lastIncludedTerm slot=leaderId
                                 flag = what RaftContext.installSnapshot returns (context/RaftContext.java:270-278: the host downloads and
                                 applies the snapshot; the decision row only carries its verdict). The term checks, the assertion, the
                                 Follower refresh after a timeout and the timer handling are member/Follower.java:129-152 and
                                 member/RaftMember.java:61-66. After a successful install the host submits the RG_EV_LOG_FLUSH row. */
};
#define RG_EV_IS_REQ_DEFINED 1

#define RG_HDR_KIND(h)        ((uint32_t)(h) & 0xFu)
#define RG_HDR_SLOT(h)        (((uint32_t)(h) >> 4) & 0xFu)
#define RG_HDR_FLAG(h)        (((uint32_t)(h) >> 8) & 0x1u)
#define RG_HDR_HINT(h)        (((uint32_t)(h) >> 9) & 0x1u)
#define RG_HDR_N(h)           ((uint32_t)(h) >> 12)
#define RG_HDR_MAKE(kind, slot, flag, n) \
    (((uint32_t)(kind) & 0xFu) | (((uint32_t)(slot) & 0xFu) << 4) | (((uint32_t)(flag) & 1u) << 8) | ((uint32_t)(n) << 12))
#define RG_HDR_HINT_BIT       (1u << 9)
#define RG_HDR_SAME_TERM      (1u << 10)  /* rg_batch32_t rows only, RG_EV_AE_REQ with n >= 1: ALL n carried entries have the term held in `aux`
                                             (what a leader replicating its own term sends); entry_terms is not consulted for the row.
                                             Must be 0 in rg_batch_t rows. Bit 11 is reserved (0). */
#define RG_MAX_ENTRIES        ((1u << 20) - 1)   /* what the header field can carry */
#define RG_MAX_AE_ENTRIES     200u               /* what a row may carry: 4 x REPLICATE_LIMIT (member/Leadership.java:10; the reference never
                                                    ships more than REPLICATE_LIMIT per request, member/Leader.java:194). Larger -> RG_BAD_EVENT:
                                                    one lane walks the entries of its row, an oversized row would stall its whole wavefront */

/* ---- wire structs: structure-of-small-structs so every lane issues 16-byte accesses -------- */
typedef struct { uint32_t hdr; uint32_t aux; } rg_ev_head_t;       /*  8 B */
typedef struct { int64_t  x;   int64_t  y;   } rg_ev_pair_t;       /* 16 B: (a,b) / (c,d) / hints */

/* reply: always written for every row */
typedef struct {
    int64_t  resp_term;   /* RaftResponse.term (valid iff RG_F_REPLIED) */
    uint32_t flags;       /* RG_F_* | status << 16 */
    uint32_t role_epoch;  /* role epoch AFTER the row: tag the host must attach to RPCs it now emits */
} rg_reply_t;             /* 16 B */

/* log / commit effects: written iff flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC), or status NEED_HOST */
typedef struct {
    int64_t commit_index; /* RaftLog.lastCommitted() after the row (valid iff RG_F_COMMIT)                 */
    int64_t log_from;     /* first log index the host must (re)write from the request's entries, after
                             deleting [log_from, oldLast] when RG_F_LOG_TRUNC (RocksLog.truncate+append);
                             CLIENT_APPEND: index of the first new entry; NEED_HOST: the index whose term is needed */
} rg_logfx_t;             /* 16 B */

/* durable membership: written iff flags & RG_F_PERSIST — what RaftMember.<init> hands to
 * StableLock.persist(currentTerm, lastCandidate) BEFORE the reply may leave (member/RaftMember.java:25) */
typedef struct {
    int64_t term;
    int32_t voted_for;    /* peer slot or RG_NO_NODE */
    int32_t role;
} rg_persist_t;           /* 16 B */

/* reply.flags bits */
#define RG_F_SUCCESS      (1u << 0)   /* RaftResponse.success                                             */
#define RG_F_REPLIED      (1u << 1)   /* a RaftResponse is produced (requests only; absent when an assertion killed the handler) */
#define RG_F_PERSIST      (1u << 2)   /* >=1 role conversion happened: fsync (term, votedFor) before replying */
#define RG_F_ROLE_CHANGED (1u << 3)   /* participant object was replaced (RaftRoutine.convertTo)            */
#define RG_F_RESET_TIMER  (1u << 4)   /* handler re-armed the election timer (ctx.resetTimer / convertTo)   */
#define RG_F_COMMIT       (1u << 5)   /* markCommitted advanced commitIndex                                 */
#define RG_F_LOG_TRUNC    (1u << 6)   /* RaftLog.truncate(log_from) must run before the append              */
#define RG_F_LOG_APPEND   (1u << 7)   /* entries with index >= log_from must be written                     */
#define RG_F_EMIT_SHIFT   8           /* what the new/refreshed participant broadcasts                      */
#define RG_F_EMIT_MASK    (3u << 8)
#define RG_EMIT_NONE      0u
#define RG_EMIT_PREVOTE   1u          /* preVote(currentTerm+1, self, last|epoch)   member/Follower.java:223-279 */
#define RG_EMIT_REQVOTE   2u          /* requestVote(currentTerm, self, last|epoch) member/Candidate.java:90-143 */
#define RG_EMIT_HEARTBEAT 3u          /* Leader.onTimeout -> replicateLog(true)      member/Leader.java:120-126  */
#define RG_F_ROLE_SHIFT   10          /* role after the row                                                 */
#define RG_F_ROLE_MASK    (3u << 10)
#define RG_F_TIMER_MUTED  (1u << 12)  /* the handler left the election timer MUTED: it called resetTimer(this, true) (deadline = Long.MAX_VALUE,
                                         context/RaftRoutine.java:101-107) and returned or threw before the un-muting call — Follower.java:43 then
                                         the throw at :48-50, Follower.java:118 then the throw in logUpToDate, Follower.java:134 then :136-139.
                                         The timer does not fire until a later handler re-arms it. Only set together with RG_F_RESET_TIMER */
#define RG_F_WIDE_VALUES  (1u << 13)  /* rg_out32_t rows only: a value of this row does not fit int32 — the row's numbers are the low 32 bits only,
                                         the full ones are in the wide columns of rg_outcome32_t at the same row index (when the caller gave any) */
#define RG_F_STATUS_SHIFT 16
#define RG_F_STATUS(f)    (((f) >> RG_F_STATUS_SHIFT) & 0xFFu)
#define RG_F_EMIT(f)      (((f) & RG_F_EMIT_MASK) >> RG_F_EMIT_SHIFT)
#define RG_F_ROLE(f)      (((f) & RG_F_ROLE_MASK) >> RG_F_ROLE_SHIFT)

/* per-row status (reply.flags bits 16-23).  A_* = the reference throws at that site: the event
 * task dies (EventLoopGroup.java:40-44), NO reply is sent, side effects made before the throw stay. */
enum {
    RG_OK                       = 0,
    RG_A_TWO_LEADERS            = 1,   /* member/Follower.java:48-50   */
    RG_A_PREV_ZERO_MISMATCH     = 2,   /* member/Follower.java:179-181 */
    RG_A_EPOCH_TERM_MISMATCH    = 3,   /* member/Follower.java:184-186 */
    RG_A_IMPOSSIBLE_LOG         = 4,   /* member/Follower.java:199-204 */
    RG_A_COMMIT_ROLLBACK        = 5,   /* storage/RocksLog.java:101-103 (via Follower.java:80, Leader.java:265/269) */
    RG_A_LOG_NOT_CONTINUOUS     = 6,   /* storage/RocksLog.java:175-177,180-188 */
    RG_A_LEADER_SELF_AE         = 7,   /* member/Leader.java:71-73     */
    RG_A_SAME_TERM_LEADER       = 8,   /* member/Leader.java:79-81     */
    RG_A_LEADER_NOT_SELF_VOTE   = 9,   /* member/Leader.java:101-105   */
    RG_A_CAND_SELF_RV           = 10,  /* member/Candidate.java:53-55  */
    RG_A_CAND_NOT_SELF_VOTE     = 11,  /* member/Candidate.java:64-66  */
    RG_A_LEADER_UNCHANGED       = 12,  /* member/Membership.java:86-90 */
    RG_A_CAND_BALLOT            = 13,  /* member/Membership.java:103-105 */
    RG_A_MATCH_ROLLBACK         = 14,  /* member/Leadership.java:76-81 (AbstractMethodError) */
    RG_A_IMPOSSIBLE_REPLICATION = 15,  /* member/Leader.java:251-253 (unreachable after a sort; kept for completeness) */
    RG_NPE_MAJOR_NULL           = 16,  /* member/Leader.java:256-257,277-279: log.get(majorIndex)==null, caught+logged, no commit */
    RG_DROPPED_STALE_ROLE       = 17,  /* response to a fenced participant (transport/rpc/Async.java:157-171) */
    RG_NOT_LEADER               = 18,  /* command/RaftStub.java:79-91: command submitted to a non-leader */
    RG_FLUSH_OUT_OF_BOUNDS      = 19,  /* storage/RocksLog.java:230-233 (IndexOutOfBoundsException) */
    RG_A_INSTALL_BEFORE_AE      = 20,  /* member/RaftMember.java:61-66, member/Follower.java:138-139: installSnapshot with term >= currentTerm at a
                                          Leader/Candidate, or term > currentTerm at a Follower */
    RG_A_NO_DOWNGRADE           = 21,  /* context/RaftRoutine.java:170-172 (unreachable when every switch goes through trySwitch first, as it does
                                          under one-row-per-group serialisation; kept so every assertion site of the path has a code) */
    RG_NEED_HOST                = 32,  /* term-run cache miss: row NOT applied; host looks up the term of
                                          logfx.log_from in its RaftLog and resubmits the row with a hint */
    RG_SKIPPED_AFTER_NEED_HOST  = 33,  /* later round of a group that hit NEED_HOST in this launch: NOT applied */
    RG_BAD_EVENT                = 34,  /* malformed row (slot out of range / self, kind unknown, response
                                          whose epoch matches a role that cannot have sent it): NOT applied */
    RG_UNSUPPORTED_LOG_STATE    = 35   /* CLIENT_APPEND on an empty log whose epoch.index>0 (RocksLog.newEntry
                                          would write key 1 below the epoch, storage/RocksLog.java:83-84): NOT applied */
};

/* ---- batches -------------------------------------------------------------------------------- */
enum { RG_MEM_HOST = 0, RG_MEM_DEVICE = 1 };

typedef struct {
    uint32_t            rounds;       /* R >= 1; round r occupies rows [r*count, (r+1)*count)             */
    uint32_t            count;        /* rows per round. dense: == number of groups                       */
    const uint32_t     *gid;          /* NULL: dense, row i of every round addresses group i.
                                         else sparse: row -> group id, strictly ascending, rounds must be 1 */
    const rg_ev_head_t *head;         /* [rounds*count] */
    const rg_ev_pair_t *ab;           /* [rounds*count] fields a,b */
    const rg_ev_pair_t *cd;           /* [rounds*count] fields c,d */
    const int64_t      *entry_terms;  /* AE_REQ entry terms, addressed by head.aux; may be NULL if no row has n>0 */
    uint64_t            entry_count;  /* length of entry_terms; at most 2^29 per batch */
    const rg_ev_pair_t *hint;         /* optional [rounds*count]; consulted only for rows with RG_HDR_HINT_BIT:
                                         AE_REQ: x = term of the host log at prevLogIndex (-1 = no such entry),
                                                 y = RaftLog.conflict(entries).index() (0 = none)
                                         AE_ACK: x = index, y = term of the host log at that index (-1 = none) */
} rg_batch_t;

typedef struct {
    rg_reply_t   *reply;    /* [rounds*count], required */
    rg_logfx_t   *logfx;    /* [rounds*count], required */
    rg_persist_t *persist;  /* [rounds*count], required */
} rg_outcome_t;

/* ---- group state exchange (host SoA, one element per group unless noted) -------------------- */
typedef struct {
    int64_t  *current_term;    /* RaftMember.currentTerm                                     */
    int32_t  *voted_for;       /* RaftMember.lastCandidate as peer slot, RG_NO_NODE = null    */
    int32_t  *role;            /* RG_FOLLOWER / RG_CANDIDATE / RG_LEADER                      */
    int32_t  *current_leader;  /* Follower.currentLeader                                      */
    uint8_t  *timeout_detected;/* Follower.timeoutDetected                                    */
    uint8_t  *repl_prepared;   /* Leader.followerStatus != null                               */
    uint32_t *role_epoch;      /* identity of the participant object / its AsyncHead          */
    int32_t  *votes;           /* AtomicInteger votes of the running (pre-)election           */
    uint32_t *elected_epoch;   /* role epoch of the Candidate whose election head stayed un-aborted after it won (0 = none) */
    int64_t  *elected_term;
    int64_t  *commit_index;    /* RocksLog.commitIndex                                        */
    int64_t  *epoch_index;     /* RaftLog.epoch()                                             */
    int64_t  *epoch_term;
    int64_t  *first_index;     /* smallest key stored (epoch.index or epoch.index+1); ignored when log empty */
    int64_t  *last_index;      /* RaftLog.last().index(); log empty <=> run_count == 0        */
    uint32_t *run_count;       /* number of term runs supplied (device keeps the newest RG_TERM_RUNS) */
    uint32_t *run_offset;      /* start of this group's runs in run_start/run_term            */
    int64_t  *run_start;       /* [sum run_count] ascending start index of each maximal equal-term run */
    int64_t  *run_term;
    /* per (group, follower j) with j = slot<self ? slot : slot-1, array index g*(P-1)+j (Leadership.State) */
    int64_t  *peer_last_epoch;
    int64_t  *peer_next_index;
    int64_t  *peer_match_index;
    int32_t  *peer_rejection;  /* recentRejection */
    uint8_t  *peer_pending;    /* pendingInstallation */
} rg_group_state_t;

typedef struct rg_table rg_table_t;

/* ---- life cycle ----------------------------------------------------------------------------- */
int         rg_abi_version(void);
/* device: HIP device ordinal. groups: G. cluster: P (self included). self_slot in [0,P). pre_vote: RaftConfig.preVote(). */
int         rg_table_create(int device, uint32_t groups, uint32_t cluster, uint32_t self_slot,
                            int pre_vote, rg_table_t **out);
int         rg_table_destroy(rg_table_t *t);
const char *rg_last_error(const rg_table_t *t);      /* t may be NULL for create failures */
/* Table options. RG_OPT_REQUIRE_FENCED_TIMEOUTS (default 0): with 1, an RG_EV_TIMEOUT row whose aux is 0 ("whoever is current") is refused as
 * RG_BAD_EVENT instead of being resolved against the participant of the moment. The reference's timer thread checks ticket.participant() ==
 * context.participant() BEFORE the event loop sees the task (context/RaftRoutine.java:65-77); a serial engine can only reproduce that when the
 * row names the participant whose ticket fired (rg_timers_expired_epochs supplies it) — the ordering contract of INTEGRATION.md section 1,
 * enforced instead of merely stated. The C++ host's IngressFlusher turns it on. */
enum { RG_OPT_REQUIRE_FENCED_TIMEOUTS = 1 };
int         rg_table_option(rg_table_t *t, int option, int value);
uint32_t    rg_table_groups(const rg_table_t *t);
uint32_t    rg_table_cluster(const rg_table_t *t);

/* ---- state ---------------------------------------------------------------------------------- */
/* Load groups [first, first+count). Arrays are indexed from 0 for group `first`. Runs beyond the
 * newest RG_TERM_RUNS are not cached (lookups into them answer RG_NEED_HOST). */
int rg_load_state(rg_table_t *t, uint32_t first, uint32_t count, const rg_group_state_t *src);
/* Read groups back. run arrays must hold count*RG_TERM_RUNS elements; run_offset[i] is set to i*RG_TERM_RUNS. */
int rg_read_state(rg_table_t *t, uint32_t first, uint32_t count, rg_group_state_t *dst);

/* ---- the hot path --------------------------------------------------------------------------- */
/* memspace RG_MEM_HOST: caller-owned host buffers (ideally from rg_host_alloc), staged over PCIe and copied back,
 * synchronous; logfx/persist rows that are not flagged valid come back zeroed. Sparse gid lists are validated.
 * memspace RG_MEM_DEVICE: all pointers are device pointers (rg_dev_alloc or any hipMalloc memory on
 * the table's device); the launch is asynchronous on the table's stream — call rg_sync. Unflagged logfx/persist
 * rows keep their previous content; a sparse gid list is trusted (it cannot be inspected from the host). */
int rg_submit(rg_table_t *t, const rg_batch_t *in, const rg_outcome_t *out, int memspace);
int rg_sync(rg_table_t *t);

/* The host-memory path, PIPELINED. rg_submit(RG_MEM_HOST) is one synchronous H2D -> kernel -> D2H chain: the link idles while the
 * kernel runs and each direction idles while the other moves. rg_submit_async stages a batch on a copy-in stream, runs the step kernel
 * on the table's stream and returns the outcome on a copy-out stream, chained by events, and RETURNS AT ONCE; up to
 * RG_PIPELINE_DEPTH batches are in flight (one more call first waits for the oldest), so the upload of batch k+1 overlaps the kernel
 * and the download of batch k — PCIe is full duplex. Batches apply in submission order (their kernels share one stream).
 *   - caller buffers (in->*, out->*) must stay valid and untouched until the batch has been waited for, and should be page-locked
 *     (rg_host_alloc): pageable memory makes every copy synchronous;
 *   - out->reply is written for every row; out->logfx / out->persist rows are meaningful only where the reply's flags say so. The
 *     CONTENT OF THE OTHER ROWS IS UNDEFINED (the columns are copied back densely from staging memory that is not cleared between
 *     batches — unlike rg_submit, which zeroes them): read a logfx / persist row only after checking its reply;
 *   - rg_submit_wait blocks until the OLDEST batch in flight has landed in its buffers; returns 1 when nothing is in flight;
 *   - every other entry point that touches the table first drains the pipeline. */
#define RG_PIPELINE_DEPTH 2
int rg_submit_async(rg_table_t *t, const rg_batch_t *in, const rg_outcome_t *out);
int rg_submit_wait(rg_table_t *t);

/* The same pipeline with COMPACT transfer formats: the link, not the kernel, bounds the host-memory path, so what
 * crosses it is cut to what carries information.
 *   upload    rg_batch32_t: a, b, c, d and the entry terms as int32 (16 + 4n bytes per row instead of 32 + 8n; 16 bytes flat for the
 *             AppendEntries rows whose entries share one term, RG_HDR_SAME_TERM). Legal only while every one of those values is in
 *             [0, 2^31) — raft terms and log indices of any deployment younger than 2^31 entries; a host that meets a larger value
 *             submits that batch through rg_submit_async instead (the two may be mixed freely). The rows are decided as they arrive by
 *             the compact-format step kernel: the decisions are the 64-bit ones, bit for bit (rg_submit32 below). No hint column (hints
 *             answer RG_NEED_HOST rows, which are rare: use the wide format for those batches).
 *   download  rg_outcome_packed_t: reply stays one row per event; logfx and persist come back PACKED, in row order, one item per row whose
 *             reply says it has one —  logfx:   flags & (RG_F_COMMIT | RG_F_LOG_APPEND | RG_F_LOG_TRUNC), or status RG_NEED_HOST
 *                                      persist: flags & RG_F_PERSIST
 *             so a host that walks reply[0 .. rows) takes the next item of a list whenever a row carries the mark. counts[0] / counts[1] =
 *             items produced (logfx / persist); items beyond the capacity given are dropped, counts still say how many there were.
 *             reply, logfx, persist and counts MUST be page-locked (rg_host_alloc): the lists are written by the device straight into
 *             them (their length is not known when the copy would have to be queued). -1 with a message otherwise.
 * Everything else — depth, ordering, rg_submit_wait — is as for rg_submit_async. */
typedef struct { int32_t a, b, c, d; } rg_ev_quad32_t;            /* 16 B */
typedef struct {
    uint32_t              rounds, count;  /* as rg_batch_t */
    const uint32_t       *gid;            /* as rg_batch_t */
    const rg_ev_head_t   *head;           /* [rounds*count] */
    const rg_ev_quad32_t *abcd;           /* [rounds*count] */
    const int32_t        *entry_terms;    /* [entry_count], addressed by head.aux */
    uint64_t              entry_count;
} rg_batch32_t;
typedef struct {
    rg_reply_t   *reply;                  /* [rounds*count] */
    rg_logfx_t   *logfx;                  /* [logfx_cap] packed */
    rg_persist_t *persist;                /* [persist_cap] packed */
    uint32_t     *counts;                 /* [2] */
    uint32_t      logfx_cap, persist_cap;
} rg_outcome_packed_t;
int rg_submit_async_packed(rg_table_t *t, const rg_batch32_t *in, const rg_outcome_packed_t *out);

/* THE ONCE-PER-TICK PATH (ABI 4). A host that flushes once per tick (support/EventLoopGroup.java:32-46 drains whatever is queued; a 300 ms-tick cluster queues
 * one or two events per group) submits small batches of a FIXED shape from the same page-locked buffers over and over, and what it feels is not the kernel
 * but the driver: rg_submit_async_packed is nine runtime calls per batch (copies up, three kernels, copies down, events). rg_tick_create records that
 * whole chain ONCE — upload of the compact rows, the step kernel, the packing of the effect lists, the download of the replies — as a HIP graph bound to
 * the caller's buffers; rg_tick_launch replays it with one call. The host refills `in`'s buffers between a wait and the next launch.
 *   in      shape and buffers are fixed at creation: rounds, count, gid (NULL = dense), head, abcd; entry_terms with entry_count = its CAPACITY (the whole
 *           array is uploaded every tick; 0 when every AppendEntries row carries RG_HDR_SAME_TERM). All page-locked (rg_host_alloc).
 *   out     as for rg_submit_async_packed (reply dense; logfx / persist lists with counts), page-locked.
 * Same decisions as rg_submit_async_packed on the same rows; launches of one table run in the order they were made, with every other submission
 * (they share the table's stream). Table options and whether the table has index bases are read at creation (set both first). At most one launch of a tick is in flight: rg_tick_launch waits for the previous one. Destroy a tick before its table. */
typedef struct rg_tick rg_tick_t;
int rg_tick_create(rg_table_t *t, const rg_batch32_t *in, const rg_outcome_packed_t *out, rg_tick_t **tick);
int rg_tick_launch(rg_tick_t *tick);      /* returns at once */
int rg_tick_wait(rg_tick_t *tick);        /* the launch has landed in `out` */
int rg_tick_destroy(rg_tick_t *tick);

/* The compact rows as a batch format of their own — what a host that keeps its batches in HBM (or builds them there) hands over:
 * 24 bytes per event instead of 40 + 8n, no gathered entry-term loads. Same contract as rg_submit (memspace, dense / sparse, rounds,
 * outcome columns, per-row status), same results: the kernel decides a workgroup's 64 groups on 32-bit values while every value of
 * those groups and of their rows is below 2^30 and restarts that workgroup's rounds in 64-bit arithmetic at the first value that is
 * not — nothing was written to the table by then, outcome rows are written again — so a table may hold any int64 state.
 * Rows whose values do not fit int32 cannot be expressed in this format at all: submit those batches through rg_submit.
 * The 32-bit tier's domain also covers the small fields it compares: role epochs, elected epochs and vote counts below 2^30, node ids >= -1,
 * `aux` below 2^30 where it is a role epoch or the entries' term, fewer than 2^24 rounds per launch — anything else is decided by the 64-bit
 * body as well, with the same results. Limit of the format: at most 2^28 - 1 rows per round (a row is addressed as scalar base + 32-bit lane
 * offset); larger batches are refused (-2, rg_last_error). RG_FORCE_WIDE=1 in the environment of rg_table_create makes every compact batch of
 * that table take the 64-bit body ("rg::step32_wide_kernel" in a profile): differential tests, and bench.py's int64-body pass. */
int rg_submit32(rg_table_t *t, const rg_batch32_t *in, const rg_outcome_t *out, int memspace);
/* Host-side packer (no device involved): rewrites a wide batch into caller buffers head[rows], abcd[rows], entry_terms[<= in->entry_count].
 * AppendEntries rows whose entries all have one term get RG_HDR_SAME_TERM and carry it in aux; the others keep their terms, renumbered.
 * Returns the entry_count of the compact batch, or < 0: -1 missing column, -2 the batch has hints, -3 a value outside [0, 2^31),
 * -4 entry_terms needed but NULL. rounds / count / gid are the wide batch's. */
int64_t rg_batch32_pack(const rg_batch_t *in, rg_ev_head_t *head, rg_ev_quad32_t *abcd, int32_t *entry_terms);

/* THE INDEX BASE OF THE COMPACT FORMATS (ABI 4). Every quantity of the path is a Java long (command/RaftLog.java:72-132); terms stay small (one per
 * election) but a group that lives long enough pushes its log indices past 2^30, which — with absolute 32-bit rows — would put it beyond the compact
 * formats and its whole workgroup on the 64-bit body for good. So every group has an index base (0 after rg_table_create), and in rg_batch32_t rows
 * and rg_out32_t rows a log index x of group g travels RELATIVE to it:
 *       x == 0 ("none": prevLogIndex of an empty log, leaderCommit 0, matchIndex of a follower that has not answered)  ->  0
 *       any other x                                                                                                      ->  x - base[g], in [1, 2^31)
 * The index fields of a row are: AE_REQ b, d; AE_ACK b, c; IS_ACK b; RV_REQ / PV_REQ b; LOG_FLUSH a; IS_REQ b (terms, n and aux travel as they are);
 * of an rg_out32_t row: commit_index and log_from. rg_outcome_t columns, rg_group_state_t and rg_send_t always speak absolute values.
 * The host keeps the base BELOW every non-zero index the group can meet — in practice a little below epoch.index, moved up after a compaction
 * (RaftLog.flush, command/storage/RocksLog.java:228-242) between two launches; a group whose epoch.index is 0 keeps base 0. Decisions never depend
 * on the base: the 32-bit body works on the relative image while every non-zero index of the group lies in (base, base + 2^30) and epoch.index > base,
 * the 64-bit body on absolute values otherwise (a value without an image in an rg_out32_t row: RG_F_WIDE_VALUES). base = 0 is ABI 3's format. */
int rg_index_base_set(rg_table_t *t, uint32_t first, uint32_t count, const int64_t *base);
int rg_index_base_get(rg_table_t *t, uint32_t first, uint32_t count, int64_t *base);
/* rg_batch32_pack with bases: index_base[g] for every group of the table (NULL: all 0); -3 also when an index lies at or below its group's base */
int64_t rg_batch32_pack_rel(const rg_batch_t *in, const int64_t *index_base, rg_ev_head_t *head, rg_ev_quad32_t *abcd, int32_t *entry_terms);
/* COMPACT OUTCOME ROWS (ABI 4). What a reply carries is RaftResponse(term, success) (RaftResponse.java:8-24) plus instructions; while a
 * group's values are below 2^31 all of it fits ONE 16-byte row per event that every lane stores unconditionally — instead of a 16-byte reply
 * plus a conditional 16-byte effect row that half of the lanes store (32-byte write granules: measured 1.17x inflation) — and the role epoch,
 * which only changes with a conversion, moves into the persist row that a conversion writes anyway. Per 64-round launch of 65 536 groups the
 * outcome stream shrinks from ~125 MB to ~68 MB.
 *   rg_out32_t      always written.  resp_term: RaftResponse.term, 0 unless RG_F_REPLIED.  flags: as rg_reply_t.flags.  commit_index:
 *                   RaftLog.lastCommitted() AFTER the row, whatever the flags say (rg_logfx_t only has it for rows that carry an effect).
 *                   log_from: as rg_logfx_t.log_from WHERE the row carries RG_F_LOG_APPEND / RG_F_LOG_TRUNC or the status RG_NEED_HOST; UNDEFINED on every
 *                   other row (the kernel stores it without a select: a raw consumer must test the flags first — rg_outcome32_unpack does).
 *   rg_persist32_t  written iff RG_F_PERSIST (every conversion sets it: RaftMember.<init> persists, member/RaftMember.java:25): the durable
 *                   pair, the role, and the role epoch AFTER the row — the tag for the RPCs the host emits from now on. A row without
 *                   RG_F_PERSIST leaves the group's role epoch where it was.
 *   wide            optional (all three columns or none, [rounds*count] each): where a workgroup that left the 32-bit domain (the 64-bit body,
 *                   see rg_submit32) puts the full rows of an event whose values do not fit — such a row carries RG_F_WIDE_VALUES and the
 *                   low 32 bits. The columns are not touched otherwise. Without them the full values of such a row are lost to the caller
 *                   (rg_read_state still has the group's state).
 * Same batches, same decisions, same table state as rg_submit32; dense batches only (gid == NULL). The device-side timers and health statistics
 * take these rows as they are: rg_timers_update32 / rg_health_update32 (ABI 5), or the whole tick as one graph: rg_tick2_*. */
typedef struct { int32_t resp_term; uint32_t flags; int32_t commit_index; int32_t log_from; } rg_out32_t;          /* 16 B */
typedef struct { int32_t term; int32_t voted_for; uint32_t role_epoch; int32_t role; } rg_persist32_t;            /* 16 B */
typedef struct {
    rg_out32_t     *row;       /* [rounds*count], required */
    rg_persist32_t *persist;   /* [rounds*count], required */
    rg_outcome_t    wide;      /* optional overflow columns */
} rg_outcome32_t;
int rg_submit32c(rg_table_t *t, const rg_batch32_t *in, const rg_outcome32_t *out, int memspace);
/* Host-side (no device involved): compact outcome rows -> the wide columns, for callers written against rg_outcome_t. `in` and `out` are HOST
 * arrays of rounds*count rows (out->logfx / out->persist rows that carry nothing are zeroed, like rg_submit's). role_epoch: [count], the role
 * epoch of every group BEFORE the batch (rg_group_state_t.role_epoch); updated in place to the epochs after it. Rows flagged
 * RG_F_WIDE_VALUES are copied from in->wide (-3 when that is missing). Returns 0, or < 0. */
int rg_outcome32_unpack(const rg_outcome32_t *in, uint32_t rounds, uint32_t count, uint32_t *role_epoch, const rg_outcome_t *out);
/* the same for a table with index bases (below): index_base[count] puts commit_index / log_from back on the groups' bases (NULL: all 0) */
int rg_outcome32_unpack_rel(const rg_outcome32_t *in, uint32_t rounds, uint32_t count, uint32_t *role_epoch, const int64_t *index_base, const rg_outcome_t *out);
/* which step kernel a batch of `count` rows per round is decided by: "rg::step_split_kernel" (a deciding and an I/O
 * wavefront per 64 groups; chosen while the batch has at most one wavefront of groups per SIMD) or "rg::step_kernel";
 * compact batches (rg_submit32, rg_submit_async_packed) are always decided by "rg::step32_kernel" */
const char *rg_step_kernel(rg_table_t *t, uint32_t count);

/* ---- N1: the leader's send side ---------------------------------------------------------------- */
/* Leader.replicateLog (member/Leader.java:142-245) for many leader groups at once: WHAT to send to each follower —
 * heartbeat / entries range / InstallSnapshot — decided from nextIndex, pendingInstallation, the log window and the
 * in-flight gate. The host then reads the payload range [prev_index+1, prev_index+count] from its RaftLog, ships the
 * RPC tagged with head.role_epoch, and keeps (head.epoch_index, send.last_index) for the response row (AE_ACK b, c).
 * Runs Leader.prepareReplication first for a leader that has not sent anything yet (the only state change). */
enum {
    RG_SEND_NONE      = 0,   /* the group is not a Leader: nothing to send                                   */
    RG_SEND_APPEND    = 1,   /* appendEntries(term, self, prev_index, prev_term, entries[count], leaderCommit) */
    RG_SEND_SNAPSHOT  = 2,   /* installSnapshot(term, self, epoch.index, epoch.term)  (Leader.java:168-190)   */
    RG_SEND_GATED     = 3,   /* requestInFlight > IN_FLIGHT_LIMIT / (heartbeat ? 10 : 1)  (Leader.java:162-166) */
    RG_SEND_NEED_HOST = 4    /* term of prev_index is below the cached runs: host reads it from its RaftLog    */
};
#define RG_REPLICATE_LIMIT 50    /* member/Leadership.java:10 */
#define RG_IN_FLIGHT_LIMIT 20    /* member/Leadership.java:11 */

typedef struct {                 /* one per row */
    int64_t  term;               /* currentTerm                                    */
    int64_t  leader_commit;      /* RaftLog.lastCommitted()                        */
    int64_t  epoch_index;        /* RaftLog.epoch() at send time (closure state of the callback) */
    int64_t  epoch_term;
    uint32_t role_epoch;         /* tag for the response rows                      */
    uint32_t is_leader;          /* 0: every send of this row is RG_SEND_NONE      */
    uint64_t reserved;           /* 0; pads the row to three 16-byte words so a wavefront stores whole 1 KiB lines */
} rg_send_head_t;                /* 48 B */

typedef struct {                 /* one per (row, follower j); j = slot<self ? slot : slot-1 */
    int64_t  prev_index;
    int64_t  prev_term;
    int64_t  last_index;         /* lastIndex of Leader.replicateLog: what the success ack will claim */
    uint32_t count;              /* entries to ship: indices prev_index+1 .. prev_index+count         */
    uint32_t kind;               /* RG_SEND_* */
} rg_send_t;                     /* 32 B; stored follower-major: send[j * count + row], so a wavefront writes 2 KiB runs */

/* rows: `count` groups; gid NULL = groups 0..count-1 (count == table groups), else strictly ascending group ids.
 * heartbeat[i] != 0: Leader.onTimeout path (fetch limit 25, in-flight limit 2); else the acceptCommand path (50, 20).
 * in_flight: [(cluster-1) * count] State.requestInFlight, follower-major like `send`; NULL = all zero.
 * head: [count]; send: [(cluster-1) * count], element (j, row) at j * count + row.
 * memspace as rg_submit (RG_MEM_DEVICE is asynchronous). */
int rg_replicate(rg_table_t *t, uint32_t count, const uint32_t *gid, const uint8_t *heartbeat, const uint16_t *in_flight,
                 rg_send_head_t *head, rg_send_t *send, int memspace);

/* ---- N4: election / heartbeat timers on the device ------------------------------------------------ */
/* RaftRoutine.resetTimer / electionTimeout / keepAlive (context/RaftRoutine.java:53-130) for all groups: instead of two
 * ScheduledFuture operations per AppendEntries on the host, one deadline per group lives in HBM.
 *   rg_timers_update  folds the RG_F_RESET_TIMER / RG_F_ROLE_CHANGED flags of a finished batch into the deadlines:
 *                     Follower/Candidate -> now + election timeout drawn uniformly from [E, 2E] (RaftConfig.java:187-190;
 *                     the draw is splitmix64(seed, gid, role_epoch, now), not Java's ThreadLocalRandom), a new Leader ->
 *                     now (first heartbeat at once), a Leader's tick -> now + H. A ticket that already fired is only
 *                     replaced by a new participant (TimerTicket.TIMEOUT makes resetTimer return false).
 *   rg_timers_expired lists, in ascending group order, every group whose deadline has passed (wavefront ballot +
 *                     popcount compaction) and marks its ticket fired: the host turns each into one RG_EV_TIMEOUT row.
 * now / deadlines are milliseconds on any monotonic clock of the host's choosing; deadline 0 = no ticket yet. */
int rg_timers_configure(rg_table_t *t, int64_t election_ms, int64_t heartbeat_ms, uint64_t seed);
/* reply: the rg_reply_t rows of the batch just submitted ([rounds*count], same gid convention as rg_submit);
 * now: [rounds] timestamp of every round. memspace applies to reply and gid (now is always a host array). */
int rg_timers_update(rg_table_t *t, uint32_t rounds, uint32_t count, const uint32_t *gid, const rg_reply_t *reply,
                     const int64_t *now, int memspace);
/* out_gid: caller buffer for up to `capacity` group ids (host or device per memspace); *out_count receives the number of
 * expired groups (may exceed capacity: then only the first `capacity` were written AND marked fired). Synchronous. */
int rg_timers_expired(rg_table_t *t, int64_t now, uint32_t *out_gid, uint32_t capacity, uint32_t *out_count, int memspace);
/* the same, also reporting for every expired group the role epoch of the participant whose ticket fired: the host puts it into
 * the aux field of the RG_EV_TIMEOUT row, so a timeout that is overtaken by a row which replaces the participant is dropped like
 * the reference drops it (context/RaftRoutine.java:70). out_epoch: [capacity], same memspace as out_gid. */
int rg_timers_expired_epochs(rg_table_t *t, int64_t now, uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, uint32_t *out_count,
                             int memspace);
/* The same from COMPACT outcome rows (ABI 5; dense batches, as rg_submit32c): row = the rg_out32_t column, persist32 = the rg_persist32_t column of the
 * batch just decided, [rounds * groups] each. A compact row names its role epoch only where a conversion happened (rg_persist32_t.role_epoch); the
 * rows before a batch's first conversion are chained from the epoch the group had after the PREVIOUS batch, which the table remembers per group —
 * refreshed by rg_load_state, rg_timers_arm and by every rg_timers_update / rg_timers_update32: the timers must see every batch of the table, in
 * order (they must anyway: a skipped batch loses its RG_F_RESET_TIMER flags). Deadlines are those rg_timers_update makes of the unpacked rows. */
int rg_timers_update32(rg_table_t *t, uint32_t rounds, const rg_out32_t *row, const rg_persist32_t *persist32, const int64_t *now, int memspace);
/* arm every group that has no ticket yet (after rg_load_state): role from the table, as rg_timers_update would */
int rg_timers_arm(rg_table_t *t, int64_t now);
int rg_timers_read(rg_table_t *t, uint32_t first, uint32_t count, int64_t *deadline);

/* ---- N4 (second half): follower health and the leader's readiness gate ------------------------------ */
/* Leadership.State.{requestSuccess, requestFailure, recentFailure} with statSuccess / statFailure / isUnhealthy / isReady
 * (member/Leadership.java:28-73) and Leader.isReady (member/Leader.java:52-64), the gate RaftStub.process puts in front of
 * client commands (command/RaftStub.java:83-87). Wall-clock inputs arrive as `now`; nothing here feeds back into decisions.
 *   rg_health_update   fold a finished batch: every AE_ACK / IS_ACK row that reached statSuccess (not fenced, no higher
 *                      term, row applied) sets requestSuccess = max(.., now) and clears recentFailure; a group that
 *                      became Leader gets fresh State objects (all zero).
 *   rg_health_failure  statFailure(now, unreachable, reject) for RPCs that ended in an error / timeout on the host
 *                      (flags bit0 = unreachable, bit1 = reject; reject also bumps recentRejection of the table).
 *   rg_ready           Leader.isReady per group: 1 when self + healthy followers exceed half the followers, else 0
 *                      (0 for non-leaders and for leaders that have not prepared replication). */
int rg_health_update(rg_table_t *t, uint32_t rounds, uint32_t count, const uint32_t *gid, const rg_ev_head_t *head,
                     const rg_reply_t *reply, const int64_t *now, int memspace);
/* rg_health_update from compact outcome rows (ABI 5; dense): head = the batch's event heads, row = its rg_out32_t column */
int rg_health_update32(rg_table_t *t, uint32_t rounds, const rg_ev_head_t *head, const rg_out32_t *row, const int64_t *now, int memspace);
int rg_health_failure(rg_table_t *t, uint32_t n, const uint32_t *gid, const uint8_t *slot, const uint8_t *flags, int64_t now);
int rg_ready(rg_table_t *t, int64_t now, int32_t critical_point, int64_t cool_down_ms, uint8_t *ready, int memspace);
/* request_success / request_failure / recent_failure: [count * (cluster-1)], index g*(P-1)+j like rg_group_state_t peers */
int rg_health_read(rg_table_t *t, uint32_t first, uint32_t count, int64_t *request_success, int64_t *request_failure,
                   int32_t *recent_failure);

/* ---- THE DEVICE-RESIDENT TICK (ABI 5) --------------------------------------------------------------------------------------------------
 * "closes the loop leader-step -> follower-step on device". One tick of a node, recorded ONCE as a HIP graph on the table's stream
 * and replayed with one call:
 *     rg::step32_kernel (compact rows in, compact outcome rows out: rg_submit32c)                        support/EventLoopGroup.java:32-46
 *  -> timers_update32   the batch's RESET_TIMER / ROLE_CHANGED / TIMER_MUTED flags into the deadlines     context/RaftRoutine.java:86-130
 *  -> health_update32   acks that reached statSuccess into requestSuccess / recentFailure                 member/Leadership.java:53-63
 *  -> timers_expired    the tickets that fired by now[rounds - 1], ascending, with their role epochs      context/RaftRoutine.java:53-77
 *  -> replicate         what every leader sends to every follower                                          member/Leader.java:142-245
 *  -> ready             Leader.isReady per group                                                           member/Leader.java:52-64
 * (the last three are optional: leave their outputs NULL). Nothing is copied: every pointer below must be DEVICE-VISIBLE for the life of the tick —
 * device memory (rg_dev_alloc) or page-locked host memory (rg_host_alloc, which the device reads and writes over the link) — and is read / written where
 * it lies, so the caller decides per column what stays in HBM (the rows of a resident replay, the send table a gateway kernel consumes) and what crosses
 * the link (the expired list, a few counters). The per-round clocks are READ from `now` when the graph runs: refill rows and clocks, launch, wait.
 * Decisions, deadlines, statistics and send rows are those of the separate calls (tests/test_gpu_parity.py::test_the_device_resident_tick_matches_the_oracle).
 * The table's options and index bases at creation are part of the recording: rg_tick2_launch refuses (-1) when they have changed since.            */
typedef struct {
    /* in */
    uint32_t              rounds;           /* 1 .. 64; count is the table's group count (dense) */
    const rg_ev_head_t   *head;             /* [rounds * G] */
    const rg_ev_quad32_t *abcd;             /* [rounds * G] */
    const int32_t        *entry_terms;      /* [entry_capacity] or NULL */
    uint64_t              entry_capacity;
    const int64_t        *now;              /* [rounds] clock of every round; now[rounds - 1] is "now" for expiry and readiness */
    const uint8_t        *heartbeat;        /* [G] rg_replicate's per-row flag, or NULL (all 0) */
    const uint16_t       *in_flight;        /* [(P - 1) * G] or NULL */
    int32_t               critical_point;   /* rg_ready's availableCriticalPoint / recoveryCoolDownMills */
    int64_t               cool_down_ms;
    /* out */
    rg_out32_t           *row;              /* [rounds * G] */
    rg_persist32_t       *persist32;        /* [rounds * G] (a row is written iff its flags carry RG_F_PERSIST) */
    uint32_t             *expired_gid;      /* [expired_capacity] or NULL: no expiry step */
    uint32_t             *expired_epoch;    /* [expired_capacity] or NULL */
    uint32_t             *expired_count;    /* [1]: groups that expired (may exceed the capacity: then only the first were listed AND marked fired) */
    uint32_t              expired_capacity;
    rg_send_head_t       *send_head;        /* [G] or NULL: no send step */
    rg_send_t            *send;             /* [(P - 1) * G] */
    uint8_t              *ready;            /* [G] or NULL */
} rg_tick2_io_t;
typedef struct rg_tick2 rg_tick2_t;
int rg_tick2_create(rg_table_t *t, const rg_tick2_io_t *io, rg_tick2_t **tick);
int rg_tick2_launch(rg_tick2_t *tick);      /* asynchronous on the table's stream. A launch issued before rg_tick2_wait of the previous one is ORDERED AFTER it
                                               on that stream and does not wait on the host: with every column in device memory a host can queue ticks whose rows
                                               another kernel produces; it reads `row`, the lists and `ready` of the LAST tick only after rg_tick2_wait */
int rg_tick2_wait(rg_tick2_t *tick);
int rg_tick2_destroy(rg_tick2_t *tick);

/* ---- device memory helpers (so a host without its own HIP binding can keep batches in HBM) --- */
/* Page-locked host memory for RG_MEM_HOST batches (JNI: wrap it with NewDirectByteBuffer): staging then runs at PCIe
 * speed instead of through the driver's pageable bounce buffers. */
int rg_host_alloc(rg_table_t *t, size_t bytes, void **hptr);
int rg_host_free(rg_table_t *t, void *hptr);
int rg_dev_alloc(rg_table_t *t, size_t bytes, void **dptr);
int rg_dev_free(rg_table_t *t, void *dptr);
int rg_copy_to_device(rg_table_t *t, void *dst, const void *src, size_t bytes);
int rg_copy_to_host(rg_table_t *t, void *dst, const void *src, size_t bytes);
void *rg_stream(rg_table_t *t);   /* the table's hipStream_t */

/* ---- measurement ---------------------------------------------------------------------------- */
/* When enabled every rg_submit brackets its step kernel with HIP events on the table's stream. */
int rg_timing_enable(rg_table_t *t, int on);
/* Sum and count of step-kernel durations since the last reset (synchronises the stream). */
int rg_timing_read(rg_table_t *t, uint64_t *launches, double *total_ms, int reset);
/* One event pair around a whole REGION of submissions on the table's stream (cheaper than per-launch pairs, which
 * put two timestamp packets between consecutive kernels): rg_timing_begin records the start event,
 * rg_timing_end records the stop event, synchronises, and returns the elapsed milliseconds. */
int rg_timing_begin(rg_table_t *t);
int rg_timing_end(rg_table_t *t, double *elapsed_ms);
/* Device-side decision counters accumulated by the step kernel (per-lane tallies, summed over the wavefront with a
 * shuffle butterfly at the end of the launch and added to the wavefront's own slot of a table — no atomics): [0]=rows with kind!=NONE, [1]=replied, [2]=role conversions, [3]=commit advances,
 * [4]=assert statuses, [5]=NEED_HOST, [6]=dropped stale, [7]=log appends. */
#define RG_NUM_COUNTERS 8
int rg_counters_read(rg_table_t *t, uint64_t counters[RG_NUM_COUNTERS], int reset);
/* Workgroups (64 groups each) of compact-row launches (rg_submit32 / rg_submit32c / rg_submit_async_packed) that were decided by the 64-bit body since the
 * last reset: a value of theirs left the 32-bit image (include/raftgpu.h, rg_submit32; with index bases: "the index base of the compact formats"). Results
 * are the same either way; the count tells a host that groups have outgrown their bases. Waits for the stream. */
int rg_wide_body_workgroups(rg_table_t *t, uint64_t *count, int reset);
/* Plain streaming-copy kernel over `bytes` of scratch on this device: returns achieved GB/s
 * (read+write) — the measured-copy roofline reported next to the 8 TB/s spec. */
int rg_copy_bandwidth(rg_table_t *t, size_t bytes, int iters, double *gbps);

#ifdef __cplusplus
}
#endif
#endif /* RAFTGPU_H */
