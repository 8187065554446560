/*
 * raftwire.h — C-ABI of the host-side wire decoder (N2): the reference's RPC frames -> rows of an rg_batch_t.
 * Host C++ only (rafting_amd/host/wire.cpp -> build/libraftwire.so); no GPU, no torch.  What it replaces in the reference:
 *   rw_splitter_*      EventCodec.FrameDecoder.decode      transport/EventCodec.java:219-335  (streaming frame state machine)
 *   rw_encode_frame    EventCodec.FrameEncoder.encode      transport/EventCodec.java:169-196
 *   rw_rows_add_frame  NettyCluster.on(PingEvent/PongEvent) + NettyNode.parseContextId / prepareLocalInvocation
 *                                                          transport/NettyCluster.java:59-105, transport/NettyNode.java:93-158
 *   rw_ingress_*       the same for MANY connections and contexts at once, laid out as the multi-round batch of the step kernel; with the
 *                      contextId map (context/ContextManager.java:41), the invocation table (transport/rpc/AsyncService.java:18-24,91-104)
 *                      and the reply frames (transport/NettyCluster.java:75-90)
 * The BODY of a frame is Kryo 4.0.2 (third party, a JVM library that is absent here). rafting_amd/host/kryo_body.cpp restates the part of its
 * byte format these bodies use (KryoBodyCodec; rw_kryo_* below) — UNVERIFIED AGAINST A JVM, see tests/golden/kryo_bodies.json and
 * INTEGRATION.md. FixedBodyCodec (rw_fixed_*, rw_rows_add_frame) is a fixed-layout codec for tests only.
 */
#ifndef RAFTWIRE_H
#define RAFTWIRE_H
#include <stddef.h>
#include <stdint.h>

#include "raftgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rw_splitter rw_splitter_t;
rw_splitter_t *rw_splitter_new(void);
void           rw_splitter_free(rw_splitter_t *s);
/* feed bytes as they arrive; returns the number of complete frames now queued, or -1 once the stream violated the grammar */
int            rw_splitter_feed(rw_splitter_t *s, const uint8_t *data, size_t n);
/* pop the oldest queued frame: returns 0 when none. head/body point into memory owned by the splitter, valid until the next call */
int            rw_splitter_pop(rw_splitter_t *s, uint8_t *type, int32_t *sequence, const char **head, size_t *head_len,
                               const uint8_t **body, size_t *body_len);
int            rw_splitter_failed(const rw_splitter_t *s);
size_t         rw_splitter_held(const rw_splitter_t *s);             /* bytes the splitter keeps in memory (bounded: one frame + 64 KiB) */
int            rw_splitter_transparent(const rw_splitter_t *s);      /* an EOT ended the framed protocol on this connection */
size_t         rw_splitter_passthrough(rw_splitter_t *s, const uint8_t **data);   /* bytes after the EOT */

/* writes the frame into out (capacity cap); returns its length, or 0 if it does not fit */
size_t rw_encode_frame(uint8_t type, int32_t sequence, const char *head, size_t head_len, const uint8_t *body, size_t body_len,
                       int ending, uint8_t *out, size_t cap);

/* fixed-layout bodies (stand-in for Kryo): method 1 appendEntries, 2 preVote, 3 requestVote, 4 installSnapshot */
size_t rw_fixed_request(int method, int64_t term, int32_t node, int64_t x, int64_t y, int64_t leader_commit, const int64_t *entry_terms,
                        uint32_t n, uint8_t *out, size_t cap);
size_t rw_fixed_response(int64_t term, int success, uint8_t *out, size_t cap);

/* Kryo-format bodies. nodes = "host:port,host:port,..." in peer-slot order (the <cluster> of the XML config). decode: 1 = ok, 0 = not one of
 * the reference's RPC bodies / unknown node / too many entries for entry_terms[max_terms] */
size_t rw_kryo_request(const char *nodes, int method, int64_t term, int32_t node, int64_t x, int64_t y, int64_t leader_commit,
                       const int64_t *entry_terms, uint32_t n, uint8_t *out, size_t cap);
size_t rw_kryo_response(int64_t term, int success, uint8_t *out, size_t cap);
int    rw_kryo_decode_request(const char *nodes, int method, const uint8_t *body, size_t len, int64_t *term, int32_t *node, int64_t *x, int64_t *y,
                              int64_t *leader_commit, int64_t *entry_terms, uint32_t max_terms, uint32_t *n_terms);
int    rw_kryo_decode_response(const uint8_t *body, size_t len, int64_t *term, int *success);
/* the follower's write path: entry k of an appendEntries body as (index, term, stored value = RocksEntry.data, what RaftLog.append puts under the
 * index): 1 and *data / *n pointing into `body`, 0 when the body is no appendEntries body of this cluster or has no entry k */
int    rw_kryo_entry(const char *nodes, const uint8_t *body, size_t len, uint32_t k, int64_t *index, int64_t *term, const uint8_t **data, size_t *n);

/* one frame -> one row appended at index *rows of the caller's structure of arrays (head/ab/cd/gid/entry_terms as rg_batch_t wants them).
 * context ids are resolved through ctx_ids[n_ctx] (gid = position); a response needs what the host kept about its request.
 * An installSnapshot request becomes an RG_EV_IS_REQ row with flag = 0: the flag is RaftContext.installSnapshot()'s verdict, which only the
 * host knows — it holds the row back until its download finished and sets bit 8 of head.hdr before submitting it.
 * returns 1 row written, 0 frame is not a decision row (unknown context / method, string event, undecodable body, buffers full) */
int rw_rows_add_frame(uint8_t type, int32_t sequence, const char *head, size_t head_len, const uint8_t *body, size_t body_len, int32_t peer,
                      const char *const *ctx_ids, uint32_t n_ctx, uint32_t pending_role_epoch, int64_t pending_epoch_at_send,
                      int64_t pending_last_index_sent, rg_ev_head_t *head_out, rg_ev_pair_t *ab, rg_ev_pair_t *cd, uint32_t *gid,
                      int64_t *entry_terms, size_t max_rows, size_t max_terms, size_t *rows, size_t *terms);

/* ---- many connections -> ONE multi-round compact batch (rafting_amd/host/ingress.hpp) -----------------------------------------------
 * What NettyCluster.on(PingEvent / PongEvent) -> context.eventLoop().execute(...) does for one context at a time
 * (transport/NettyCluster.java:59-105, support/EventLoopGroup.java:32-46,77-80), for all contexts of a table: every decoded RPC becomes the
 * next row of ITS group — cell [k][gid] of a dense [round][group] rg_batch32_t for the k-th row a batch holds for that group — so the batch
 * goes to rg_submit32 / rg_submit_async_packed as it stands. Rows of one connection keep their order per group; rows of different
 * connections interleave in arrival order. Rows beyond `max_rounds` for a group, or behind a row the compact format cannot express, wait on
 * their connection for the next batch.
 *   threads   rw_ingress_feed: one thread per connection at a time, any number of connections at once (a row costs one relaxed fetch_add on
 *             its group's counter). Everything else: the flush thread.
 *   memory    head / abcd / entry_terms of the two banks are the CALLER's (page-locked memory from rg_host_alloc in a deployment),
 *             head and abcd [max_rounds * groups], entry_terms [entry_cap] each.
 *   shards    tables behind the ingress (block partition gpu = gid / ceil(groups / shards), one table, stream and feeder per GPU):
 *             a row is routed by its group id as it is placed; a bank's head / abcd arrays hold the shards one after the other (shard s at cell
 *             s * ceil(groups / shards) * max_rounds), its term array is split evenly; rw_ingress_seal closes all shards at once and
 *             rw_ingress_shard describes each one's dense batch — what THAT table's rg_submit32 / rg_submit_async_packed takes. 1 = one table.
 * nodes: "host:port,..." in peer-slot order for Kryo-format bodies (rw_kryo_*), NULL for the fixed-layout test codec. */
typedef struct rw_ingress rw_ingress_t;
rw_ingress_t *rw_ingress_new(uint32_t groups, uint32_t max_rounds, uint32_t conns, const char *nodes,
                             rg_ev_head_t *head0, rg_ev_quad32_t *abcd0, int32_t *entry_terms0,
                             rg_ev_head_t *head1, rg_ev_quad32_t *abcd1, int32_t *entry_terms1, uint64_t entry_cap, uint32_t shards);
void     rw_ingress_free(rw_ingress_t *g);
/* ContextManager.createContext: contextId -> group id (1 ok, 0 refused: duplicate id / gid, gid out of range, id longer than 128 bytes) */
int      rw_ingress_add_context(rw_ingress_t *g, const char *id, size_t len, uint32_t gid);
/* ContextManager.exitContext / destroyContext (context/ContextManager.java:126-171): the id stops resolving at once; rows already queued for the
 * group stay queued. The group id may be given to another context after the next rw_ingress_seal (which reclaims it). 1, or 0: no such context */
int      rw_ingress_remove_context(rw_ingress_t *g, const char *id, size_t len);
int      rw_ingress_set_peer(rw_ingress_t *g, uint32_t conn, int32_t peer_slot);
/* what the host remembers when it SENDS request `sequence` of `method` for group `gid` on `conn` (AsyncService.invoke,
 * transport/rpc/AsyncService.java:91-104): the response row needs it (RG_EV_AE_ACK b, c and aux) */
int      rw_ingress_sent(rw_ingress_t *g, uint32_t conn, int32_t sequence, int method, uint32_t gid, uint32_t role_epoch, int64_t epoch_at_send,
                         int64_t last_index_sent);
/* The command payload never reaches the table, but the host has to write it to its RaftLog once the row answers RG_F_LOG_APPEND
 * (command/storage/RocksLog.java:169-225). rw_ingress_retain_bodies(g, 1) — before the first byte is fed — keeps the body of every
 * AppendEntries request that carries entries next to its cell (freed by rw_ingress_recycle); rw_ingress_body hands it back: 1 and the
 * Kryo-format Object[] body as it came off the wire (the entries' payload is in it), or 0 when nothing is kept for that cell. */
int      rw_ingress_retain_bodies(rw_ingress_t *g, int on);
int      rw_ingress_body(const rw_ingress_t *g, int bank, uint32_t shard, uint64_t cell, const uint8_t **body, size_t *len);
/* the TCP connection behind `conn` was replaced (closed by the peer, or by this side after rw_ingress_feed returned -1): the frame state machine
 * starts over, invocations still filed for it are dropped (their responses cannot arrive any more), rows already queued stay queued */
int      rw_ingress_reset_conn(rw_ingress_t *g, uint32_t conn);
/* bytes as they arrive: rows queued by this call, or -1 once the connection broke the frame grammar */
int      rw_ingress_feed(rw_ingress_t *g, uint32_t conn, const uint8_t *data, size_t n);
/* N1's output on the wire: what rg_replicate decided for follower j of `count` leader rows (head[count]; send_j = send + j * count, the rg_send_t
 * of that follower; gid NULL = rows are groups 0..count-1) as the request frames Leader.replicateLog ships (member/Leader.java:168-245,
 * transport/NettyNode.java:54-73), written to out[cap] for connection `conn`, each under the connection's next sequence number with its
 * invocation record filed for the response (what rw_ingress_sent does by hand). term_of(user, gid, index) reads an entry's term from the host's
 * RaftLog; the command payload is the transport's and is not modelled. An RG_SEND_NEED_HOST row (prevLogIndex below the device's cached term
 * runs) is completed here — prevLogTerm = term_of(gid, prev_index) — and counted in *need_host.
 * Returns the bytes written, or the bytes needed when that exceeds cap — in which case no sequence number was used (the invocation records that
 * were filed are filed again, identically, by the retry with a larger buffer). */
size_t   rw_ingress_encode_sends(rw_ingress_t *g, uint32_t conn, int32_t self_slot, uint32_t count, const uint32_t *gid, const rg_send_head_t *head,
                                 const rg_send_t *send_j, int64_t (*term_of)(void *user, uint32_t gid, int64_t index), void *user,
                                 uint8_t *out, size_t cap, uint32_t *frames, uint32_t *need_host);
/* a row that does not come off the wire (RG_EV_TIMEOUT, RG_EV_CLIENT_APPEND, RG_EV_LOG_FLUSH, an RG_EV_IS_REQ released with the host's verdict),
 * queued like a row of connection `conn` — give local sources connection numbers of their own. reply_conn = UINT32_MAX: nobody waits for a
 * reply, else the RG_F_REPLIED answer is emitted as the response to (reply_conn, reply_sequence). Not for RG_EV_AE_REQ (entries travel in frames). */
int      rw_ingress_add_row(rw_ingress_t *g, uint32_t conn, uint32_t gid, uint32_t hdr, uint32_t aux, int64_t a, int64_t b, int64_t c, int64_t d,
                            uint32_t reply_conn, int32_t reply_sequence);
/* close the batch being filled: *batch describes shard 0 (dense, gid NULL; rounds may be 0), *rows = cells that hold an event (all shards), *wide = rows that
 * had to stay out of the compact format (read them with rw_ingress_wide_row, decide them with one sparse rg_submit AFTER this batch).
 * Returns the bank (0 / 1) the batch lies in, or -1 when the batch sealed before this one has not been recycled yet (one sealed batch is with
 * the flusher at a time; the feeders fill the other bank meanwhile). The batch stays valid until rw_ingress_recycle(bank). */
int      rw_ingress_seal(rw_ingress_t *g, rg_batch32_t *batch, uint64_t *rows, uint32_t *wide);
/* shard `shard` of that bank's sealed batch: its dense rg_batch32_t (group g of the table = global group *first_gid + g), the cells that hold an
 * event. 1, or 0 for a bad bank / shard */
int      rw_ingress_shard(const rw_ingress_t *g, int bank, uint32_t shard, rg_batch32_t *batch, uint64_t *events, uint32_t *first_gid);
int      rw_ingress_wide_row(const rw_ingress_t *g, int bank, uint32_t i, uint32_t *gid, rg_ev_head_t *head, int64_t abcd[4], int64_t *entry_terms,
                             uint32_t max_terms, uint32_t *reply_conn, int32_t *reply_sequence);
                             /* ascending gid; returns the row's entry count, -1 if it exceeds max_terms; reply_conn UINT32_MAX = a response row */
/* the response frame of wide row i (decided by the host's sparse submit: `reply` = its reply row): the bytes written to out[cap] — 0 when the row
 * was a response row, its handler died (no RG_F_REPLIED) or the frame does not fit — and the connection they belong to in *conn */
size_t   rw_ingress_emit_wide(const rw_ingress_t *g, int bank, uint32_t i, const rg_reply_t *reply, uint32_t *conn, uint8_t *out, size_t cap);
/* who sent the request in cell `cell` (= round * count + group of the shard) of that shard's batch: 1 and (*conn, *sequence), or 0 for a response row */
int      rw_ingress_origin(const rw_ingress_t *g, int bank, uint32_t shard, uint64_t cell, uint32_t *conn, int32_t *sequence);
/* the PongEvent frames of cells [cell_begin, cell_end) of shard `shard` (reply = that table's reply rows) whose reply carries RG_F_REPLIED, for connection `conn`, into out[cap]: returns the
 * bytes written, or the size needed (nothing written) when that exceeds cap. Persist the batch's RG_F_PERSIST rows BEFORE releasing these
 * bytes (member/RaftMember.java:25). */
size_t   rw_ingress_emit(const rw_ingress_t *g, int bank, uint32_t shard, const rg_reply_t *reply, uint64_t cell_begin, uint64_t cell_end, uint32_t conn,
                         uint8_t *out, size_t cap);
/* RG_NEED_HOST inside the batch of that bank (a term lookup left the device's cached runs: the row came back unapplied, the later rows of its
 * group RG_SKIPPED_AFTER_NEED_HOST): decide those rows, in order, BEFORE the next batch is submitted — the missed row again with a hint
 * read from the host's RaftLog, then the group's later rows one by one — with one sparse single-round submit per step for all broken groups
 * together. The final replies are written over reply[cell] (rw_ingress_emit then answers the requests); every repaired row's outcome is
 * handed to `applied` at once, because the hint of the group's next row is read from the log as that row left it.
 *   logfx   the batch's log-effect rows: dense [rounds * groups] (rg_submit32), or packed != 0: the row-ordered list of rg_submit_async_packed
 * returns the rows decided here (0: nothing to repair), -1 when a submit failed or a hinted row still missed. */
typedef struct {
    void    *user;
    int64_t (*term_at)(void *user, uint32_t gid, int64_t index);                 /* RaftLog.get(index).term(), -1 = no such entry */
    int64_t (*conflict)(void *user, uint32_t gid, int64_t first_index, const int64_t *terms, uint32_t n);  /* RaftLog.conflict(entries).index(), 0 = none */
    int64_t (*epoch_index)(void *user, uint32_t gid);                            /* RaftLog.epoch().index() */
    int     (*submit)(void *user, const rg_batch_t *in, const rg_outcome_t *out); /* rg_submit(table, in, out, RG_MEM_HOST) */
    void    (*applied)(void *user, uint32_t gid, uint64_t cell, const rg_reply_t *reply, const rg_logfx_t *logfx, const rg_persist_t *persist);
} rw_repair_host_t;
int64_t  rw_ingress_repair(const rw_ingress_t *g, int bank, uint32_t shard, rg_reply_t *reply, const rg_logfx_t *logfx, int packed, const rw_repair_host_t *host);
/* (the gids the callbacks see are that TABLE's: 0 .. count-1 of the shard) */
/* the batch of that bank is done with: wipe the cells it used (may run beside rw_ingress_feed) */
int      rw_ingress_recycle(rw_ingress_t *g, int bank);
uint64_t rw_ingress_refused(const rw_ingress_t *g);      /* frames that were no decision row: unknown context / method, undecodable body, unmatched response */
uint64_t rw_ingress_held(const rw_ingress_t *g);         /* rows waiting for the next batch (flush thread) */
/* rows of `conn` waiting for the next batch — asked by the thread that reads that connection: while the flusher is behind it stops reading the
 * socket (TCP pushes back on the peer) instead of letting the backlog grow */
uint64_t rw_ingress_held_on(const rw_ingress_t *g, uint32_t conn);

#ifdef __cplusplus
}
#endif
#endif
