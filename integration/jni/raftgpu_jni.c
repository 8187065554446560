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
#include <stdio.h>
#include <string.h>

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

/* Every buffer that crosses here is checked BEFORE the library sees its address: a heap ByteBuffer has no address — silently reading
 * that NULL as "column absent" would, e.g., turn a sparse batch into a dense one — and a short one is a native out-of-bounds access inside libraftgpu.
 * BUF(b, bytes, optional, "name"): the address of direct buffer b, which must hold at least `bytes`; a null b is "absent" only where the C-ABI allows it.
 * Anything else leaves an IllegalArgumentException pending and sets `bad`: the native method then returns -1 without calling the library. */
static void *checked(JNIEnv *env, jobject b, jlong need, int optional, const char *what, int *bad)
{
    char msg[160];
    jlong cap = 0;
    void *p = NULL;
    if (!b) {
        if (optional) return NULL;
        snprintf(msg, sizeof msg, "%s: a direct ByteBuffer of at least %lld bytes is required (got null)", what, (long long)need);
    } else if (!(p = (*env)->GetDirectBufferAddress(env, b))) {
        snprintf(msg, sizeof msg, "%s: not a DIRECT ByteBuffer (ByteBuffer.allocateDirect or GpuTable.hostAlloc)", what);
    } else if ((cap = (*env)->GetDirectBufferCapacity(env, b)) < need) {
        snprintf(msg, sizeof msg, "%s: capacity %lld, the call needs %lld bytes", what, (long long)cap, (long long)need);
    } else {
        return p;
    }
    jclass cls = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (cls) (*env)->ThrowNew(env, cls, msg);
    *bad = 1;
    return NULL;
}
#define BUF(b, bytes, optional, what) checked(env, (b), (jlong)(bytes), (optional), (what), &bad)

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
/* bytes per group of every column (runs: per run; peers: per follower): the declaration order of rg_group_state_t */
static const int state_width[] = {8, 4, 4, 4, 1, 1, 4, 4, 4, 8, 8, 8, 8, 8, 8, 4, 4, /* runs */ 8, 8, /* peers */ 8, 8, 8, 4, 1};
enum { FIRST_RUN_COLUMN = 17, FIRST_PEER_COLUMN = 19 };

/* loading: the run columns hold sum(run_count) elements (read from the run_count column once ITS size is known); reading: count * RG_TERM_RUNS */
static int state_columns(JNIEnv *env, jobjectArray columns, rg_group_state_t *s, jlong h, jint count, int loading)
{
    void **field = (void **)s;
    const jsize n = (jsize)(sizeof(rg_group_state_t) / sizeof(void *));
    int bad = 0;
    if ((*env)->GetArrayLength(env, columns) != n || count < 0) return -1;
    const jlong followers = (jlong)rg_table_cluster(TABLE(h)) - 1;
    jlong runs = (jlong)count * RG_TERM_RUNS;
    for (jsize i = 0; i < n && !bad; i++) {
        jlong elems = i < FIRST_RUN_COLUMN ? count : (i < FIRST_PEER_COLUMN ? runs : (jlong)count * followers);
        field[i] = BUF((*env)->GetObjectArrayElement(env, columns, i), elems * state_width[i], 0, "rg_group_state_t column");
        if (loading && i == 15 && !bad) {                /* run_count: from here on the run columns' size is known */
            runs = 0;
            for (jint g = 0; g < count; g++) runs += ((const uint32_t *)field[15])[g];
        }
    }
    return bad ? -2 : 0;
}

JNIEXPORT jint JNICALL J(GpuTable, loadState)(JNIEnv *env, jclass cls, jlong h, jint first, jint count, jobjectArray columns)
{
    (void)cls;
    rg_group_state_t s;
    const int rc = state_columns(env, columns, &s, h, count, 1);
    if (rc == -1) throw_state(env, "loadState: 24 columns expected (rg_group_state_t)");
    if (rc != 0) return -1;
    return rg_load_state(TABLE(h), (uint32_t)first, (uint32_t)count, &s);
}

JNIEXPORT jint JNICALL J(GpuTable, readState)(JNIEnv *env, jclass cls, jlong h, jint first, jint count, jobjectArray columns)
{
    (void)cls;
    rg_group_state_t s;
    const int rc = state_columns(env, columns, &s, h, count, 0);
    if (rc == -1) throw_state(env, "readState: 24 columns expected (rg_group_state_t)");
    if (rc != 0) return -1;
    return rg_read_state(TABLE(h), (uint32_t)first, (uint32_t)count, &s);
}

/* ---- GpuTable: the hot path (support/EventLoopGroup.java:32-46) -------------------------------------------------------------------------------- */

/* wide rows, host buffers, synchronous: a once-per-tick flush of `count` rows (sparse: gid != null, rounds must be 1) */
JNIEXPORT jint JNICALL J(GpuTable, submit)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject head, jobject ab, jobject cd,
                                           jobject entry_terms, jlong entry_count, jobject hint, jobject reply, jobject logfx, jobject persist)
{
    (void)cls;
    int bad = 0;
    const jlong rows = (jlong)rounds * count;
    if (rounds < 0 || count < 0 || entry_count < 0) { throw_state(env, "submit: negative rounds / count / entry_count"); return -1; }
    rg_batch_t in;
    in.rounds = (uint32_t)rounds; in.count = (uint32_t)count;
    in.gid = (const uint32_t *)BUF(gid, 4 * (jlong)count, 1, "gid"); in.head = (const rg_ev_head_t *)BUF(head, 8 * rows, 0, "head");
    in.ab = (const rg_ev_pair_t *)BUF(ab, 16 * rows, 0, "ab"); in.cd = (const rg_ev_pair_t *)BUF(cd, 16 * rows, 0, "cd");
    in.entry_terms = (const int64_t *)BUF(entry_terms, 8 * entry_count, entry_count == 0, "entry_terms"); in.entry_count = (uint64_t)entry_count;
    in.hint = (const rg_ev_pair_t *)BUF(hint, 16 * rows, 1, "hint");
    rg_outcome_t out;
    out.reply = (rg_reply_t *)BUF(reply, 16 * rows, 0, "reply"); out.logfx = (rg_logfx_t *)BUF(logfx, 16 * rows, 0, "logfx");
    out.persist = (rg_persist_t *)BUF(persist, 16 * rows, 0, "persist");
    if (bad) return -1;
    return rg_submit(TABLE(h), &in, &out, RG_MEM_HOST);
}

static int batch32(JNIEnv *env, rg_batch32_t *in, jint rounds, jint count, jobject gid, jobject head, jobject abcd, jobject entry_terms, jlong entry_count)
{
    int bad = 0;
    const jlong rows = (jlong)rounds * count;
    if (rounds < 0 || count < 0 || entry_count < 0) { throw_state(env, "negative rounds / count / entry_count"); return 1; }
    in->rounds = (uint32_t)rounds; in->count = (uint32_t)count;
    in->gid = (const uint32_t *)BUF(gid, 4 * (jlong)count, 1, "gid"); in->head = (const rg_ev_head_t *)BUF(head, 8 * rows, 0, "head");
    in->abcd = (const rg_ev_quad32_t *)BUF(abcd, 16 * rows, 0, "abcd");
    in->entry_terms = (const int32_t *)BUF(entry_terms, 4 * entry_count, entry_count == 0, "entry_terms"); in->entry_count = (uint64_t)entry_count;
    return bad;
}

/* compact rows (24 bytes per event), wide outcome columns */
JNIEXPORT jint JNICALL J(GpuTable, submit32)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject head, jobject abcd,
                                             jobject entry_terms, jlong entry_count, jobject reply, jobject logfx, jobject persist)
{
    (void)cls;
    rg_batch32_t in;
    int bad = batch32(env, &in, rounds, count, gid, head, abcd, entry_terms, entry_count);
    const jlong rows = (jlong)rounds * count;
    rg_outcome_t out;
    out.reply = (rg_reply_t *)BUF(reply, 16 * rows, 0, "reply"); out.logfx = (rg_logfx_t *)BUF(logfx, 16 * rows, 0, "logfx");
    out.persist = (rg_persist_t *)BUF(persist, 16 * rows, 0, "persist");
    if (bad) return -1;
    return rg_submit32(TABLE(h), &in, &out, RG_MEM_HOST);
}

/* compact rows in, compact outcome rows out (ABI 4): row = rg_out32_t[rounds*count], persist32 = rg_persist32_t[rounds*count]; the wide columns
 * are the optional overflow area (all three or none) */
JNIEXPORT jint JNICALL J(GpuTable, submit32c)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject head, jobject abcd, jobject entry_terms,
                                              jlong entry_count, jobject row, jobject persist32, jobject reply, jobject logfx, jobject persist)
{
    (void)cls;
    rg_batch32_t in;
    int bad = batch32(env, &in, rounds, count, NULL, head, abcd, entry_terms, entry_count);
    const jlong rows = (jlong)rounds * count;
    rg_outcome32_t out;
    out.row = (rg_out32_t *)BUF(row, 16 * rows, 0, "row"); out.persist = (rg_persist32_t *)BUF(persist32, 16 * rows, 0, "persist32");
    out.wide.reply = (rg_reply_t *)BUF(reply, 16 * rows, 1, "wide reply"); out.wide.logfx = (rg_logfx_t *)BUF(logfx, 16 * rows, 1, "wide logfx");
    out.wide.persist = (rg_persist_t *)BUF(persist, 16 * rows, 1, "wide persist");
    if (bad) return -1;
    return rg_submit32c(TABLE(h), &in, &out, RG_MEM_HOST);
}

/* rg_outcome32_unpack for callers written against the wide columns; role_epoch: int[count] as a direct buffer, updated in place */
JNIEXPORT jint JNICALL J(GpuTable, unpack32)(JNIEnv *env, jclass cls, jint rounds, jint count, jobject row, jobject persist32, jobject wide_reply,
                                             jobject wide_logfx, jobject wide_persist, jobject role_epoch, jobject reply, jobject logfx, jobject persist)
{
    (void)cls;
    int bad = 0;
    const jlong rows = (jlong)rounds * count;
    if (rounds < 0 || count < 0) { throw_state(env, "unpack32: negative rounds / count"); return -1; }
    rg_outcome32_t in;
    in.row = (rg_out32_t *)BUF(row, 16 * rows, 0, "row"); in.persist = (rg_persist32_t *)BUF(persist32, 16 * rows, 0, "persist32");
    in.wide.reply = (rg_reply_t *)BUF(wide_reply, 16 * rows, 1, "wide reply"); in.wide.logfx = (rg_logfx_t *)BUF(wide_logfx, 16 * rows, 1, "wide logfx");
    in.wide.persist = (rg_persist_t *)BUF(wide_persist, 16 * rows, 1, "wide persist");
    rg_outcome_t out;
    out.reply = (rg_reply_t *)BUF(reply, 16 * rows, 0, "reply"); out.logfx = (rg_logfx_t *)BUF(logfx, 16 * rows, 0, "logfx");
    out.persist = (rg_persist_t *)BUF(persist, 16 * rows, 0, "persist");
    uint32_t *epochs = (uint32_t *)BUF(role_epoch, 4 * (jlong)count, 0, "role_epoch");
    if (bad) return -1;
    return rg_outcome32_unpack(&in, (uint32_t)rounds, (uint32_t)count, epochs, &out);
}

/* the pipelined host-memory path with compact transfer formats: every buffer from hostAlloc (the device writes the lists into them) */
JNIEXPORT jint JNICALL J(GpuTable, submitAsyncPacked)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject head, jobject abcd,
                                                      jobject entry_terms, jlong entry_count, jobject reply, jobject logfx, jint logfx_cap,
                                                      jobject persist, jint persist_cap, jobject counts)
{
    (void)cls;
    rg_batch32_t in;
    int bad = batch32(env, &in, rounds, count, gid, head, abcd, entry_terms, entry_count);
    const jlong rows = (jlong)rounds * count;
    if (logfx_cap < 0 || persist_cap < 0) { throw_state(env, "submitAsyncPacked: negative list capacity"); return -1; }
    rg_outcome_packed_t out;
    out.reply = (rg_reply_t *)BUF(reply, 16 * rows, 0, "reply"); out.logfx = (rg_logfx_t *)BUF(logfx, 16 * (jlong)logfx_cap, logfx_cap == 0, "logfx");
    out.persist = (rg_persist_t *)BUF(persist, 16 * (jlong)persist_cap, persist_cap == 0, "persist");
    out.counts = (uint32_t *)BUF(counts, 8, 0, "counts"); out.logfx_cap = (uint32_t)logfx_cap; out.persist_cap = (uint32_t)persist_cap;
    if (bad) return -1;
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
    int bad = 0;
    const jlong followers = (jlong)rg_table_cluster(TABLE(h)) - 1;
    if (count < 0) { throw_state(env, "replicate: negative count"); return -1; }
    const uint32_t *g = (const uint32_t *)BUF(gid, 4 * (jlong)count, 1, "gid");
    const uint8_t *hb = (const uint8_t *)BUF(heartbeat, count, 1, "heartbeat");
    const uint16_t *fl = (const uint16_t *)BUF(in_flight, 2 * followers * count, 1, "in_flight");
    rg_send_head_t *sh = (rg_send_head_t *)BUF(head, 48 * (jlong)count, 0, "head");
    rg_send_t *ss = (rg_send_t *)BUF(send, 32 * followers * count, 0, "send");
    if (bad) return -1;
    return rg_replicate(TABLE(h), (uint32_t)count, g, hb, fl, sh, ss, RG_MEM_HOST);
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
    int bad = 0;
    if (rounds < 0 || count < 0) { throw_state(env, "timersUpdate: negative rounds / count"); return -1; }
    const uint32_t *g = (const uint32_t *)BUF(gid, 4 * (jlong)count, 1, "gid");
    const rg_reply_t *rep = (const rg_reply_t *)BUF(reply, 16 * (jlong)rounds * count, 0, "reply");
    const int64_t *clk = (const int64_t *)BUF(now, 8 * (jlong)rounds, 0, "now");
    if (bad) return -1;
    return rg_timers_update(TABLE(h), (uint32_t)rounds, (uint32_t)count, g, rep, clk, RG_MEM_HOST);
}

/* the same from compact outcome rows (ABI 5; dense: rounds * groups rows) */
JNIEXPORT jint JNICALL J(GpuTable, timersUpdate32)(JNIEnv *env, jclass cls, jlong h, jint rounds, jobject row, jobject persist32, jobject now)
{
    (void)cls;
    int bad = 0;
    if (rounds < 0) { throw_state(env, "timersUpdate32: negative rounds"); return -1; }
    const jlong rows = (jlong)rounds * rg_table_groups(TABLE(h));
    const rg_out32_t *r = (const rg_out32_t *)BUF(row, 16 * rows, 0, "row");
    const rg_persist32_t *p = (const rg_persist32_t *)BUF(persist32, 16 * rows, 0, "persist32");
    const int64_t *clk = (const int64_t *)BUF(now, 8 * (jlong)rounds, 0, "now");
    if (bad) return -1;
    return rg_timers_update32(TABLE(h), (uint32_t)rounds, r, p, clk, RG_MEM_HOST);
}

JNIEXPORT jint JNICALL J(GpuTable, healthUpdate32)(JNIEnv *env, jclass cls, jlong h, jint rounds, jobject head, jobject row, jobject now)
{
    (void)cls;
    int bad = 0;
    if (rounds < 0) { throw_state(env, "healthUpdate32: negative rounds"); return -1; }
    const jlong rows = (jlong)rounds * rg_table_groups(TABLE(h));
    const rg_ev_head_t *hd = (const rg_ev_head_t *)BUF(head, 8 * rows, 0, "head");
    const rg_out32_t *r = (const rg_out32_t *)BUF(row, 16 * rows, 0, "row");
    const int64_t *clk = (const int64_t *)BUF(now, 8 * (jlong)rounds, 0, "now");
    if (bad) return -1;
    return rg_health_update32(TABLE(h), (uint32_t)rounds, hd, r, clk, RG_MEM_HOST);
}

/* -> number of expired groups; out_gid / out_epoch: int[capacity] direct buffers (the epochs go into the aux of the RG_EV_TIMEOUT rows: the fence) */
JNIEXPORT jint JNICALL J(GpuTable, timersExpired)(JNIEnv *env, jclass cls, jlong h, jlong now, jobject out_gid, jobject out_epoch, jint capacity)
{
    (void)cls;
    int bad = 0;
    uint32_t n = 0;
    if (capacity < 0) { throw_state(env, "timersExpired: negative capacity"); return -1; }
    uint32_t *g = (uint32_t *)BUF(out_gid, 4 * (jlong)capacity, capacity == 0, "outGid");
    uint32_t *e = (uint32_t *)BUF(out_epoch, 4 * (jlong)capacity, 1, "outEpoch");
    if (bad) return -1;
    const int rc = rg_timers_expired_epochs(TABLE(h), now, g, e, (uint32_t)capacity, &n, RG_MEM_HOST);
    return rc != 0 ? rc : (jint)n;
}

JNIEXPORT jint JNICALL J(GpuTable, healthUpdate)(JNIEnv *env, jclass cls, jlong h, jint rounds, jint count, jobject gid, jobject head, jobject reply, jobject now)
{
    (void)cls;
    int bad = 0;
    if (rounds < 0 || count < 0) { throw_state(env, "healthUpdate: negative rounds / count"); return -1; }
    const jlong rows = (jlong)rounds * count;
    const uint32_t *g = (const uint32_t *)BUF(gid, 4 * (jlong)count, 1, "gid");
    const rg_ev_head_t *hd = (const rg_ev_head_t *)BUF(head, 8 * rows, 0, "head");
    const rg_reply_t *rep = (const rg_reply_t *)BUF(reply, 16 * rows, 0, "reply");
    const int64_t *clk = (const int64_t *)BUF(now, 8 * (jlong)rounds, 0, "now");
    if (bad) return -1;
    return rg_health_update(TABLE(h), (uint32_t)rounds, (uint32_t)count, g, hd, rep, clk, RG_MEM_HOST);
}

JNIEXPORT jint JNICALL J(GpuTable, ready)(JNIEnv *env, jclass cls, jlong h, jlong now, jint critical_point, jlong cool_down_ms, jobject ready)
{
    (void)cls;
    int bad = 0;
    uint8_t *out = (uint8_t *)BUF(ready, (jlong)rg_table_groups(TABLE(h)), 0, "ready");
    if (bad) return -1;
    return rg_ready(TABLE(h), now, critical_point, cool_down_ms, out, RG_MEM_HOST);
}

/* ---- the device-resident tick (ABI 5, rg_tick2_*): ONE recorded launch per tick — decisions, RaftRoutine.resetTimer, Leadership.State.statSuccess, the fired
 * tickets, Leader.replicateLog, Leader.isReady (context/RaftRoutine.java:53-130, member/Leadership.java:28-73, member/Leader.java:52-64,142-245). Every column is
 * a direct buffer of GpuTable.hostAlloc (page-locked, addressable by the device) and stays bound to the tick until tick2Destroy. -> tick handle, 0 on failure. */
JNIEXPORT jlong JNICALL J(GpuTable, tick2Create)(JNIEnv *env, jclass cls, jlong h, jint rounds, jobject head, jobject abcd, jobject entry_terms, jlong entry_capacity,
                                                 jobject now, jobject heartbeat, jobject in_flight, jint critical_point, jlong cool_down_ms, jobject row,
                                                 jobject persist32, jobject expired_gid, jobject expired_epoch, jobject expired_count, jint expired_capacity,
                                                 jobject send_head, jobject send, jobject ready)
{
    (void)cls;
    int bad = 0;
    if (rounds <= 0 || entry_capacity < 0 || expired_capacity < 0) { throw_state(env, "tick2Create: rounds must be positive, capacities not negative"); return 0; }
    const jlong G = rg_table_groups(TABLE(h)), followers = (jlong)rg_table_cluster(TABLE(h)) - 1, rows = (jlong)rounds * G;
    rg_tick2_io_t io;
    memset(&io, 0, sizeof io);
    io.rounds = (uint32_t)rounds;
    io.head = (const rg_ev_head_t *)BUF(head, 8 * rows, 0, "head");
    io.abcd = (const rg_ev_quad32_t *)BUF(abcd, 16 * rows, 0, "abcd");
    io.entry_terms = (const int32_t *)BUF(entry_terms, 4 * entry_capacity, entry_capacity == 0, "entryTerms");
    io.entry_capacity = (uint64_t)entry_capacity;
    io.now = (const int64_t *)BUF(now, 8 * (jlong)rounds, 0, "now");
    io.heartbeat = (const uint8_t *)BUF(heartbeat, G, 1, "heartbeat");
    io.in_flight = (const uint16_t *)BUF(in_flight, 2 * followers * G, 1, "inFlight");
    io.critical_point = critical_point;
    io.cool_down_ms = cool_down_ms;
    io.row = (rg_out32_t *)BUF(row, 16 * rows, 0, "row");
    io.persist32 = (rg_persist32_t *)BUF(persist32, 16 * rows, 0, "persist32");
    io.expired_gid = (uint32_t *)BUF(expired_gid, 4 * (jlong)expired_capacity, 1, "expiredGid");
    io.expired_epoch = (uint32_t *)BUF(expired_epoch, 4 * (jlong)expired_capacity, 1, "expiredEpoch");
    io.expired_count = (uint32_t *)BUF(expired_count, 4, expired_gid == NULL, "expiredCount");
    io.expired_capacity = (uint32_t)expired_capacity;
    io.send_head = (rg_send_head_t *)BUF(send_head, 48 * G, 1, "sendHead");
    io.send = (rg_send_t *)BUF(send, 32 * followers * G, send_head == NULL, "send");
    io.ready = (uint8_t *)BUF(ready, G, 1, "ready");
    if (bad) return 0;
    rg_tick2_t *tick = NULL;
    if (rg_tick2_create(TABLE(h), &io, &tick) != 0) { throw_state(env, rg_last_error(TABLE(h))); return 0; }
    return (jlong)(intptr_t)tick;
}

JNIEXPORT jint JNICALL J(GpuTable, tick2Launch)(JNIEnv *env, jclass cls, jlong tick)
{
    (void)env; (void)cls;
    return rg_tick2_launch((rg_tick2_t *)(intptr_t)tick);
}

JNIEXPORT jint JNICALL J(GpuTable, tick2Wait)(JNIEnv *env, jclass cls, jlong tick)
{
    (void)env; (void)cls;
    return rg_tick2_wait((rg_tick2_t *)(intptr_t)tick);
}

JNIEXPORT jint JNICALL J(GpuTable, tick2Destroy)(JNIEnv *env, jclass cls, jlong tick)
{
    (void)env; (void)cls;
    return rg_tick2_destroy((rg_tick2_t *)(intptr_t)tick);
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
