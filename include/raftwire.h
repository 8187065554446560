/*
 * raftwire.h — C-ABI of the host-side wire decoder (N2): the reference's RPC frames -> rows of an rg_batch_t.
 * Host C++ only (rafting_amd/host/wire.cpp -> build/libraftwire.so); no GPU, no torch.  What it replaces in the reference:
 *   rw_splitter_*      EventCodec.FrameDecoder.decode      transport/EventCodec.java:219-335  (streaming frame state machine)
 *   rw_encode_frame    EventCodec.FrameEncoder.encode      transport/EventCodec.java:169-196
 *   rw_rows_add_frame  NettyCluster.on(PingEvent/PongEvent) + NettyNode.parseContextId / prepareLocalInvocation
 *                                                          transport/NettyCluster.java:59-105, transport/NettyNode.java:93-158
 * The BODY of a frame is Kryo (third party, absent here): these entry points use the fixed-layout stand-in documented in
 * rafting_amd/host/wire.hpp (FixedBodyCodec); a JVM deployment plugs its own BodyCodec in C++.
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
int            rw_splitter_transparent(const rw_splitter_t *s);      /* an EOT ended the framed protocol on this connection */
size_t         rw_splitter_passthrough(rw_splitter_t *s, const uint8_t **data);   /* bytes after the EOT */

/* writes the frame into out (capacity cap); returns its length, or 0 if it does not fit */
size_t rw_encode_frame(uint8_t type, int32_t sequence, const char *head, size_t head_len, const uint8_t *body, size_t body_len,
                       int ending, uint8_t *out, size_t cap);

/* fixed-layout bodies (stand-in for Kryo): method 1 appendEntries, 2 preVote, 3 requestVote, 4 installSnapshot */
size_t rw_fixed_request(int method, int64_t term, int32_t node, int64_t x, int64_t y, int64_t leader_commit, const int64_t *entry_terms,
                        uint32_t n, uint8_t *out, size_t cap);
size_t rw_fixed_response(int64_t term, int success, uint8_t *out, size_t cap);

/* one frame -> one row appended at index *rows of the caller's structure of arrays (head/ab/cd/gid/entry_terms as rg_batch_t wants them).
 * context ids are resolved through ctx_ids[n_ctx] (gid = position); a response needs what the host kept about its request.
 * returns 1 row written, 0 frame is not a decision row (unknown context / method, string event, undecodable body, buffers full) */
int rw_rows_add_frame(uint8_t type, int32_t sequence, const char *head, size_t head_len, const uint8_t *body, size_t body_len, int32_t peer,
                      const char *const *ctx_ids, uint32_t n_ctx, uint32_t pending_role_epoch, int64_t pending_epoch_at_send,
                      int64_t pending_last_index_sent, rg_ev_head_t *head_out, rg_ev_pair_t *ab, rg_ev_pair_t *cd, uint32_t *gid,
                      int64_t *entry_terms, size_t max_rows, size_t max_terms, size_t *rows, size_t *terms);

#ifdef __cplusplus
}
#endif
#endif
