/*
 * raftwire.h — C-ABI of the host-side wire decoder (N2): the reference's RPC frames -> rows of an rg_batch_t.
 * Host C++ only (rafting_amd/host/wire.cpp -> build/libraftwire.so); no GPU, no torch.  What it replaces in the reference:
 *   rw_splitter_*      EventCodec.FrameDecoder.decode      transport/EventCodec.java:219-335  (streaming frame state machine)
 *   rw_encode_frame    EventCodec.FrameEncoder.encode      transport/EventCodec.java:169-196
 *   rw_rows_add_frame  NettyCluster.on(PingEvent/PongEvent) + NettyNode.parseContextId / prepareLocalInvocation
 *                                                          transport/NettyCluster.java:59-105, transport/NettyNode.java:93-158
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

/* one frame -> one row appended at index *rows of the caller's structure of arrays (head/ab/cd/gid/entry_terms as rg_batch_t wants them).
 * context ids are resolved through ctx_ids[n_ctx] (gid = position); a response needs what the host kept about its request.
 * An installSnapshot request becomes an RG_EV_IS_REQ row with flag = 0: the flag is RaftContext.installSnapshot()'s verdict, which only the
 * host knows — it holds the row back until its download finished and sets bit 8 of head.hdr before submitting it.
 * returns 1 row written, 0 frame is not a decision row (unknown context / method, string event, undecodable body, buffers full) */
int rw_rows_add_frame(uint8_t type, int32_t sequence, const char *head, size_t head_len, const uint8_t *body, size_t body_len, int32_t peer,
                      const char *const *ctx_ids, uint32_t n_ctx, uint32_t pending_role_epoch, int64_t pending_epoch_at_send,
                      int64_t pending_last_index_sent, rg_ev_head_t *head_out, rg_ev_pair_t *ab, rg_ev_pair_t *cd, uint32_t *gid,
                      int64_t *entry_terms, size_t max_rows, size_t max_terms, size_t *rows, size_t *terms);

#ifdef __cplusplus
}
#endif
#endif
