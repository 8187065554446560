/*
 * raftgpu_jni.c — the JNI side of the drop-in boundary: what io.lubricant.consensus.raft.gpu.GpuTable / GpuIngress (integration/java) bind to.
 *
 * No logic lives here: every function unwraps direct ByteBuffers (memory laid out exactly as include/raftgpu.h / include/raftwire.h say) and
 * calls ONE entry point of libraftgpu.so / libraftwire.so.  The Java classes it serves replace, in the reference tree
 * (src/main/java/io/lubricant/consensus/raft/):
 *     RaftParticipant.java:9-51            the per-context handler interface  -> rows of a batch            (GpuTable.submit*)
 *     support/RaftFactory.java:18-36       where the ContextManager is made   -> GpuRaftFactory
 *     context/ContextManager.java:57-106   buildContext / createContext       -> GpuContextManager (loadState, one table per GPU)
 *     support/EventLoopGroup.java:32-46    the drain                          -> one flusher thread per table
 *
 * Build (a machine with a JDK; this image has none — here the file is type-checked against a stand-in jni.h, tests/test_jni_shim_cpu.py,
 * and driven through a fake JNIEnv on the host emulation of the kernels, tests/native/jni_harness.c):
 *     cc -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude integration/jni/raftgpu_jni.c \
 *        -Lrafting_amd -lraftgpu -Lbuild -lraftwire -o libraftgpu_jni.so
 *
 * Conventions: a table / an ingress travels as a jlong holding the pointer; buffers are DIRECT ByteBuffers (GetDirectBufferAddress) — a null
 * reference is a NULL pointer (optional columns); return values are the C-ABI's (0 ok, < 0: GpuTable.lastError(handle)).
 */
#include <jni.h>
#include <stddef.h>
#include <stdint.h>

#include "raftgpu.h"
#include "raftwire.h"

#define J(cls, name) Java_io_lubricant_consensus_raft_gpu_##cls##_##name
#define TABLE(h) ((rg_table_t *)(intptr_t)(h))
#define INGRESS(h) ((rw_ingress_t *)(intptr_t)(h))
#define ADDR(b) ((b) ? (*env)->GetDirectBufferAddress(env, (b)) : NULL)

static void throw_state(JNIEnv *env, const char *msg)
{
    jclass cls = (*env)->FindClass(env, "java/lang/IllegalStateException");
    if (cls) (*env)->ThrowNew(env, cls, msg ? msg : "libraftgpu");
}

/* ---- GpuTable: life cycle ------------------------------------------------------------------------------------------------------------ */

JNIEXPORT jint JNICALL J(GpuTable, abiVersion)(JNIEnv *env, jclass cls)
{
    (void)env; (void)cls;
    return rg_abi_version();
}

JNIEXPORT jlong JNICALL J(GpuTable, create)(JNIEnv *env, jclass cls, jint device, jint groups, jint cluster, jint self_slot, jboolean pre_vote)
{
    (void)cls;
    rg_table_t *t = NULL;
    if (rg_table_create(device, (uint32_t)groups, (uint32_t)cluster, (uint32_t)self_slot, pre_vote ? 1 : 0, &t) != 0) {
        throw_state(env, rg_last_error(NULL));
        return 0;
    }
    return (jlong)(intptr_t)t;
}

JNIEXPORT void JNICALL J(GpuTable, destroy)(JNIEnv *env, jclass cls, jlong h)
{
    (void)env; (void)cls;
    rg_table_destroy(TABLE(h));
}

JNIEXPORT jstring JNICALL J(GpuTable, lastError)(JNIEnv *env, jclass cls, jlong h)
{
    (void)cls;
    return (*env)->NewStringUTF(env, rg_last_error(TABLE(h)));
}

JNIEXPORT jint JNICALL J(GpuTable, option)(JNIEnv *env, jclass cls, jlong h, jint option, jint value)
{
    (void)env; (void)cls;
    return rg_table_option(TABLE(h), option, value);
}

/* Page-locked host memory as a direct ByteBuffer (a ByteBuffer.allocateDirect block cannot be pinned after the fact): the batch columns a
 * flusher fills and the outcome columns it reads live in these, so staging runs at link speed (rg_host_alloc). */
JNIEXPORT jobject JNICALL J(GpuTable, hostAlloc)(JNIEnv *env, jclass cls, jlong h, jlong bytes)
{
    (void)cls;
    void *p = NULL;
    if (bytes <= 0 || rg_host_alloc(TABLE(h), (size_t)bytes, &p) != 0) { throw_state(env, rg_last_error(TABLE(h))); return NULL; }
    return (*env)->NewDirectByteBuffer(env, p, bytes);
}

JNIEXPORT jint JNICALL J(GpuTable, hostFree)(JNIEnv *env, jclass cls, jlong h, jobject buffer)
{
    (void)cls;
    return rg_host_free(TABLE(h), ADDR(buffer));
}

/* ---- GpuTable: state (ContextManager.buildContext -> RaftContext.initialize: StableLock.restore, RaftLog.epoch / last) --------------------- */

/* columns: the 24 arrays of rg_group_state_t in declaration order, each a direct ByteBuffer (or null where the struct allows NULL) */
static int state_columns(JNIEnv *env, jobjectArray columns, rg_group_state_t *s)
{
    void **field = (void **)s;
    const jsize n = (jsize)(sizeof(rg_group_state_t) / sizeof(void *));
    if ((*env)->GetArrayLength(env, columns) != n) return -1;
    for (jsize i = 0; i < n; i++) field[i] = ADDR((*env)->GetObjectArrayElement(env, columns, i));
    return 0;
}

JNIEXPORT jint JNICALL J(GpuTable, loadState)(JNIEnv *env, jclass cls, jlong h, jint first, jint count, jobjectArray columns)
{
    (void)cls;
    rg_group_state_t s;
    if (state_columns(env, columns, &s) != 0) { throw_state(env, "loadState: 24 columns expected (rg_group_state_t)"); return -1; }
    return rg_load_state(TABLE(h), (uint32_t)first, (uint32_t)count, &s);
}

JNIEXPORT jint JNICALL J(GpuTable, readState)(JNIEnv *env, jclass cls, jlong h, jint first, jint count, jobjectArray columns)
{
    (void)cls;
    rg_group_state_t s;
    if (state_columns(env, columns, &s) != 0) { throw_state(env, "readState: 24 columns expected (rg_group_state_t)"); return -1; }
    return rg_read_state(TABLE(h), (uint32_t)first, (uint32_t)count, &s);
}

/* ---- GpuTable: the hot path (support/EventLoopGroup.java:32-46) -------------------------------------------------------------------------------- */

/* wide rows, host buffers, synchronous: a once-per-tick flush of `count` rows (sparse: gid != null, rounds must be 1) */
JNIEXPORT jint JNICALL J(GpuTable, submit)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject head, jobject ab, jobject cd,
                                           jobject entry_terms, jlong entry_count, jobject hint, jobject reply, jobject logfx, jobject persist)
{
    (void)cls;
    rg_batch_t in;
    in.rounds = (uint32_t)rounds; in.count = (uint32_t)count;
    in.gid = (const uint32_t *)ADDR(gid); in.head = (const rg_ev_head_t *)ADDR(head);
    in.ab = (const rg_ev_pair_t *)ADDR(ab); in.cd = (const rg_ev_pair_t *)ADDR(cd);
    in.entry_terms = (const int64_t *)ADDR(entry_terms); in.entry_count = (uint64_t)entry_count;
    in.hint = (const rg_ev_pair_t *)ADDR(hint);
    rg_outcome_t out;
    out.reply = (rg_reply_t *)ADDR(reply); out.logfx = (rg_logfx_t *)ADDR(logfx); out.persist = (rg_persist_t *)ADDR(persist);
    return rg_submit(TABLE(h), &in, &out, RG_MEM_HOST);
}

static void batch32(JNIEnv *env, rg_batch32_t *in, jint rounds, jint count, jobject gid, jobject head, jobject abcd, jobject entry_terms, jlong entry_count)
{
    in->rounds = (uint32_t)rounds; in->count = (uint32_t)count;
    in->gid = (const uint32_t *)ADDR(gid); in->head = (const rg_ev_head_t *)ADDR(head); in->abcd = (const rg_ev_quad32_t *)ADDR(abcd);
    in->entry_terms = (const int32_t *)ADDR(entry_terms); in->entry_count = (uint64_t)entry_count;
}

/* compact rows (24 bytes per event), wide outcome columns */
JNIEXPORT jint JNICALL J(GpuTable, submit32)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject head, jobject abcd,
                                             jobject entry_terms, jlong entry_count, jobject reply, jobject logfx, jobject persist)
{
    (void)cls;
    rg_batch32_t in;
    batch32(env, &in, rounds, count, gid, head, abcd, entry_terms, entry_count);
    rg_outcome_t out;
    out.reply = (rg_reply_t *)ADDR(reply); out.logfx = (rg_logfx_t *)ADDR(logfx); out.persist = (rg_persist_t *)ADDR(persist);
    return rg_submit32(TABLE(h), &in, &out, RG_MEM_HOST);
}

/* compact rows in, compact outcome rows out (ABI 4): row = rg_out32_t[rounds*count], persist32 = rg_persist32_t[rounds*count]; the wide columns
 * are the optional overflow area (all three or none) */
JNIEXPORT jint JNICALL J(GpuTable, submit32c)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject head, jobject abcd, jobject entry_terms,
                                              jlong entry_count, jobject row, jobject persist32, jobject reply, jobject logfx, jobject persist)
{
    (void)cls;
    rg_batch32_t in;
    batch32(env, &in, rounds, count, NULL, head, abcd, entry_terms, entry_count);
    rg_outcome32_t out;
    out.row = (rg_out32_t *)ADDR(row); out.persist = (rg_persist32_t *)ADDR(persist32);
    out.wide.reply = (rg_reply_t *)ADDR(reply); out.wide.logfx = (rg_logfx_t *)ADDR(logfx); out.wide.persist = (rg_persist_t *)ADDR(persist);
    return rg_submit32c(TABLE(h), &in, &out, RG_MEM_HOST);
}

/* rg_outcome32_unpack for callers written against the wide columns; role_epoch: int[count] as a direct buffer, updated in place */
JNIEXPORT jint JNICALL J(GpuTable, unpack32)(JNIEnv *env, jclass cls, jint rounds, jint count, jobject row, jobject persist32, jobject wide_reply,
                                             jobject wide_logfx, jobject wide_persist, jobject role_epoch, jobject reply, jobject logfx, jobject persist)
{
    (void)cls;
    rg_outcome32_t in;
    in.row = (rg_out32_t *)ADDR(row); in.persist = (rg_persist32_t *)ADDR(persist32);
    in.wide.reply = (rg_reply_t *)ADDR(wide_reply); in.wide.logfx = (rg_logfx_t *)ADDR(wide_logfx); in.wide.persist = (rg_persist_t *)ADDR(wide_persist);
    rg_outcome_t out;
    out.reply = (rg_reply_t *)ADDR(reply); out.logfx = (rg_logfx_t *)ADDR(logfx); out.persist = (rg_persist_t *)ADDR(persist);
    return rg_outcome32_unpack(&in, (uint32_t)rounds, (uint32_t)count, (uint32_t *)ADDR(role_epoch), &out);
}

/* the pipelined host-memory path with compact transfer formats: every buffer from hostAlloc (the device writes the lists into them) */
JNIEXPORT jint JNICALL J(GpuTable, submitAsyncPacked)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject head, jobject abcd,
                                                      jobject entry_terms, jlong entry_count, jobject reply, jobject logfx, jint logfx_cap,
                                                      jobject persist, jint persist_cap, jobject counts)
{
    (void)cls;
    rg_batch32_t in;
    batch32(env, &in, rounds, count, gid, head, abcd, entry_terms, entry_count);
    rg_outcome_packed_t out;
    out.reply = (rg_reply_t *)ADDR(reply); out.logfx = (rg_logfx_t *)ADDR(logfx); out.persist = (rg_persist_t *)ADDR(persist);
    out.counts = (uint32_t *)ADDR(counts); out.logfx_cap = (uint32_t)logfx_cap; out.persist_cap = (uint32_t)persist_cap;
    return rg_submit_async_packed(TABLE(h), &in, &out);
}

JNIEXPORT jint JNICALL J(GpuTable, submitWait)(JNIEnv *env, jclass cls, jlong h)
{
    (void)env; (void)cls;
    return rg_submit_wait(TABLE(h));
}

JNIEXPORT jint JNICALL J(GpuTable, sync)(JNIEnv *env, jclass cls, jlong h)
{
    (void)env; (void)cls;
    return rg_sync(TABLE(h));
}

/* ---- GpuTable: N1 the send side (member/Leader.java:142-245), N4 timers and health (context/RaftRoutine.java:53-130, member/Leadership.java:28-73) */

JNIEXPORT jint JNICALL J(GpuTable, replicate)(JNIEnv *env, jclass cls, jlong h, jint count, jobject gid, jobject heartbeat, jobject in_flight, jobject head,
                                              jobject send)
{
    (void)cls;
    return rg_replicate(TABLE(h), (uint32_t)count, (const uint32_t *)ADDR(gid), (const uint8_t *)ADDR(heartbeat), (const uint16_t *)ADDR(in_flight),
                        (rg_send_head_t *)ADDR(head), (rg_send_t *)ADDR(send), RG_MEM_HOST);
}

JNIEXPORT jint JNICALL J(GpuTable, timersConfigure)(JNIEnv *env, jclass cls, jlong h, jlong election_ms, jlong heartbeat_ms, jlong seed)
{
    (void)env; (void)cls;
    return rg_timers_configure(TABLE(h), election_ms, heartbeat_ms, (uint64_t)seed);
}

JNIEXPORT jint JNICALL J(GpuTable, timersArm)(JNIEnv *env, jclass cls, jlong h, jlong now)
{
    (void)env; (void)cls;
    return rg_timers_arm(TABLE(h), now);
}

JNIEXPORT jint JNICALL J(GpuTable, timersUpdate)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject reply, jobject now)
{
    (void)cls;
    return rg_timers_update(TABLE(h), (uint32_t)rounds, (uint32_t)count, (const uint32_t *)ADDR(gid), (const rg_reply_t *)ADDR(reply), (const int64_t *)ADDR(now),
                            RG_MEM_HOST);
}

/* -> number of expired groups; out_gid / out_epoch: int[capacity] direct buffers (the epochs go into the aux of the RG_EV_TIMEOUT rows: the fence) */
JNIEXPORT jint JNICALL J(GpuTable, timersExpired)(JNIEnv *env, jclass cls, jlong h, jlong now, jobject out_gid, jobject out_epoch, jint capacity)
{
    (void)cls;
    uint32_t n = 0;
    const int rc = rg_timers_expired_epochs(TABLE(h), now, (uint32_t *)ADDR(out_gid), (uint32_t *)ADDR(out_epoch), (uint32_t)capacity, &n, RG_MEM_HOST);
    return rc != 0 ? rc : (jint)n;
}

JNIEXPORT jint JNICALL J(GpuTable, healthUpdate)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject head, jobject reply, jobject now)
{
    (void)cls;
    return rg_health_update(TABLE(h), (uint32_t)rounds, (uint32_t)count, (const uint32_t *)ADDR(gid), (const rg_ev_head_t *)ADDR(head),
                            (const rg_reply_t *)ADDR(reply), (const int64_t *)ADDR(now), RG_MEM_HOST);
}

JNIEXPORT jint JNICALL J(GpuTable, ready)(JNIEnv *env, jclass cls, jlong h, jlong now, jint critical_point, jlong cool_down_ms, jobject ready)
{
    (void)cls;
    return rg_ready(TABLE(h), now, critical_point, cool_down_ms, (uint8_t *)ADDR(ready), RG_MEM_HOST);
}

/* ---- GpuIngress: socket bytes -> the [round][group] batch and back (transport/EventCodec.java:169-335, transport/NettyCluster.java:59-105) ------ */

JNIEXPORT jint JNICALL J(GpuIngress, feed)(JNIEnv *env, jclass cls, jlong g, jint conn, jlong address, jint length)
{
    (void)env; (void)cls;      /* address: ByteBuf.memoryAddress() + readerIndex() of a pooled direct Netty buffer — no copy, no Java object per RPC */
    return rw_ingress_feed(INGRESS(g), (uint32_t)conn, (const uint8_t *)(intptr_t)address, (size_t)length);
}

JNIEXPORT jint JNICALL J(GpuIngress, addRow)(JNIEnv *env, jclass cls, jlong g, jint conn, jint gid, jint hdr, jint aux, jlong a, jlong b, jlong c, jlong d,
                                             jint reply_conn, jint reply_sequence)
{
    (void)env; (void)cls;      /* the host's own rows: RG_EV_TIMEOUT (aux = the fired ticket's role epoch), RG_EV_CLIENT_APPEND, RG_EV_LOG_FLUSH, a released
                                  RG_EV_IS_REQ; reply_conn = -1 (UINT32_MAX): nobody waits for a reply */
    return rw_ingress_add_row(INGRESS(g), (uint32_t)conn, (uint32_t)gid, (uint32_t)hdr, (uint32_t)aux, a, b, c, d, (uint32_t)reply_conn, reply_sequence);
}

JNIEXPORT jint JNICALL J(GpuIngress, recycle)(JNIEnv *env, jclass cls, jlong g, jint bank)
{
    (void)env; (void)cls;
    return rw_ingress_recycle(INGRESS(g), bank);
}

JNIEXPORT jlong JNICALL J(GpuIngress, held)(JNIEnv *env, jclass cls, jlong g)
{
    (void)env; (void)cls;
    return (jlong)rw_ingress_held(INGRESS(g));
}
