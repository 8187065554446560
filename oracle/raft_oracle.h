/*
 * raft_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the per-RaftContext decision logic of curioloop/rafting
 * (io.lubricant.consensus.raft.context.**), one event at a time, exactly in the order the
 * reference EventLoop would run it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call this; the product path (rafting_amd/, libraftgpu.so) never does.
 *
 * HOW IT IS PINNED: the reference (Java 8 + Maven deps) cannot run here (no JDK) and its own tests hold no
 * golden vector for this path.  Its decision classes are however compiled FROM THEIR OWN
 * SOURCES: tools/make_ref.py translates Follower / Candidate / Leader / Leadership.State / Membership / RaftMember /
 * TimerTicket / RocksLog / RaftRoutine / RaftContext token by token into C++ (oracle/_ref/libref.so, built by
 * `make -C oracle ref` from /root/reference, sha-pinned source ranges, no hand-edited output), and
 * tests/test_ref_parity.py requires this oracle to answer exactly like that library: all 47 known-answer
 * scenarios, >= 10^6 random inputs per pure function, the lockstep fuzzer over seven cluster shapes (outcome rows
 * and full state after every round), the BASELINE replay streams; tests/golden/replay_digests.json is generated
 * by that library, not by this oracle.  Hand-derived KATs (tests/test_oracle_kat.py) and the constants extracted
 * from the reference text (tests/test_reference_pins.py) remain as the second, independent pin.
 *
 * The log model is LOSSLESS: an unbounded run-length encoding of (index -> term) over the contiguous
 * key window RocksLog keeps (storage/RocksLog.java), so it answers RaftLog.get(i) for every index —
 * unlike the device, which caches only the newest RG_TERM_RUNS runs.
 *
 * It consumes the same wire structs as the C-ABI (include/raftgpu.h) so tests feed both sides the
 * identical buffers.
 */
#ifndef RAFT_ORACLE_H
#define RAFT_ORACLE_H

#include "../include/raftgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_table orc_table_t;

orc_table_t *orc_table_create(uint32_t groups, uint32_t cluster, uint32_t self_slot, int pre_vote);
void         orc_table_destroy(orc_table_t *t);

/* same semantics as rg_load_state, but ALL supplied runs are kept (lossless log) */
int orc_load_state(orc_table_t *t, uint32_t first, uint32_t count, const rg_group_state_t *src);
/* same semantics as rg_read_state: reports the newest RG_TERM_RUNS runs of each log */
int orc_read_state(orc_table_t *t, uint32_t first, uint32_t count, rg_group_state_t *dst);

/* same semantics as rg_submit(..., RG_MEM_HOST); never answers RG_NEED_HOST. */
int orc_submit(orc_table_t *t, const rg_batch_t *in, const rg_outcome_t *out);

/* same semantics as rg_replicate(..., RG_MEM_HOST); never answers RG_SEND_NEED_HOST */
int orc_replicate(orc_table_t *t, uint32_t count, const uint32_t *gid, const uint8_t *heartbeat, const uint16_t *in_flight,
                  rg_send_head_t *head, rg_send_t *send);

/* N4 timers: same semantics as rg_timers_configure / _update / _arm / _expired / _read (host memory) */
int orc_timers_configure(orc_table_t *t, int64_t election_ms, int64_t heartbeat_ms, uint64_t seed);
int orc_timers_update(orc_table_t *t, uint32_t rounds, uint32_t count, const uint32_t *gid, const rg_reply_t *reply, const int64_t *now);
int orc_timers_arm(orc_table_t *t, int64_t now);
int orc_timers_expired(orc_table_t *t, int64_t now, uint32_t *out_gid, uint32_t capacity, uint32_t *out_count);
int orc_timers_expired_epochs(orc_table_t *t, int64_t now, uint32_t *out_gid, uint32_t *out_epoch, uint32_t capacity, uint32_t *out_count);
int orc_timers_read(orc_table_t *t, uint32_t first, uint32_t count, int64_t *deadline);

/* N4b health: statSuccess happens inside orc_submit at the place the reference calls it (Leader.java:229), with the clock
 * given by orc_health_clock; the rest mirrors rg_health_failure / rg_ready / rg_health_read */
int orc_health_clock(orc_table_t *t, const int64_t *now_per_round);
int orc_health_failure(orc_table_t *t, uint32_t n, const uint32_t *gid, const uint8_t *slot, const uint8_t *flags, int64_t now);
int orc_ready(orc_table_t *t, int64_t now, int32_t critical_point, int64_t cool_down_ms, uint8_t *ready);
int orc_health_read(orc_table_t *t, uint32_t first, uint32_t count, int64_t *request_success, int64_t *request_failure,
                    int32_t *recent_failure);

/* CPU baseline: apply a dense batch with `threads` worker threads, groups dealt round-robin to the threads
 * like EventLoopGroup.next (support/EventLoopGroup.java:77-80; the reference uses 3), in chunks of 64 groups.
 * Returns wall seconds of the apply phase (thread start/join excluded via a start barrier). */
double orc_submit_threads(orc_table_t *t, const rg_batch_t *in, const rg_outcome_t *out, int threads);

/* The lossless log as the host's RaftLog would answer (tests use it to build NEED_HOST hints):
 * RaftLog.get(index).term() -> returns 1 and *term when the key exists, 0 otherwise. */
int     orc_log_term(const orc_table_t *t, uint32_t gid, int64_t index, int64_t *term);
/* RaftLog.conflict over entries with indices e0..e0+n-1 -> conflicting index or 0 */
int64_t orc_log_conflict(const orc_table_t *t, uint32_t gid, int64_t e0, uint32_t n, const int64_t *terms);

/* exposed pieces for known-answer tests */
int64_t orc_rejection_step(int32_t recent_rejection);                 /* Leadership.java:105 */
void    orc_major_indices(const int64_t *match, int n, int64_t out[2]); /* Leadership.java:116-130 */
/* State.updateIndex (member/Leadership.java:75-114) on st = {lastEpoch, nextIndex, matchIndex}; returns 0 or the RG_A_* status */
int     orc_update_index(int64_t st[3], int32_t *rejection, uint8_t *pending, int64_t epoch, int64_t index, int success, int snapshot);
/* Membership.isBetter (member/Membership.java:74-108): 1 better, 0 not, <0 = -(RG_A_* status) */
int     orc_is_better(int new_role, int64_t new_term, int32_t new_ballot,
                      int cur_role, int64_t cur_term, int32_t cur_ballot);

/* batch forms of the three above (n independent inputs) for differential fuzzing against oracle/_ref */
void    orc_update_index_batch(uint32_t n, int64_t *st, int32_t *rejection, uint8_t *pending, const int64_t *epoch, const int64_t *index,
                               const uint8_t *success, const uint8_t *snapshot, int32_t *rc);
void    orc_is_better_batch(uint32_t n, const int32_t *nr, const int64_t *nt, const int32_t *nb, const int32_t *cr, const int64_t *ct,
                            const int32_t *cb, int32_t *out);
void    orc_major_indices_batch(uint32_t n, int f, const int64_t *match, int64_t *out);

#ifdef __cplusplus
}
#endif
#endif
