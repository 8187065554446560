/*
 * jni_harness.c — TEST INFRASTRUCTURE: drives integration/jni/raftgpu_jni.c through a FAKE JNIEnv (tests/jni_stub/jni.h: a stand-in, no JVM
 * here) on whatever libraftgpu.so it is linked against — the host emulation of the kernels in the CPU suite — and holds the shim to the
 * C-ABI called directly: the same batch through Java_..._GpuTable_submit and through rg_submit must give byte-identical outcome columns and
 * table state.  A "direct ByteBuffer" here is a struct {address, capacity}; an "Object[]" a struct {n, elements}.
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raftgpu.h"

struct _jobject { int kind; void *addr; jlong cap; jsize n; jobject *elems; const char *text; };
enum { K_BUFFER = 1, K_ARRAY, K_CLASS, K_STRING };
static char thrown[256];

static jclass fake_FindClass(JNIEnv *env, const char *name) { (void)env; static struct _jobject c = {K_CLASS, 0, 0, 0, 0, 0}; c.text = name; return &c; }
static jint fake_ThrowNew(JNIEnv *env, jclass c, const char *msg) { (void)env; (void)c; snprintf(thrown, sizeof thrown, "%s", msg ? msg : ""); return 0; }
static jstring fake_NewStringUTF(JNIEnv *env, const char *utf) { (void)env; jobject o = calloc(1, sizeof *o); o->kind = K_STRING; o->text = utf; return o; }
static jsize fake_GetArrayLength(JNIEnv *env, jarray a) { (void)env; return a->n; }
static jobject fake_GetObjectArrayElement(JNIEnv *env, jobjectArray a, jsize i) { (void)env; return a->elems[i]; }
static jobject fake_NewDirectByteBuffer(JNIEnv *env, void *p, jlong cap) { (void)env; jobject o = calloc(1, sizeof *o); o->kind = K_BUFFER; o->addr = p; o->cap = cap; return o; }
static void *fake_GetDirectBufferAddress(JNIEnv *env, jobject b) { (void)env; return b->kind == K_BUFFER ? b->addr : NULL; }
static jlong fake_GetDirectBufferCapacity(JNIEnv *env, jobject b) { (void)env; return b->cap; }

static const struct JNINativeInterface_ table = {fake_FindClass, fake_ThrowNew, fake_NewStringUTF, fake_GetArrayLength, fake_GetObjectArrayElement,
                                                 fake_NewDirectByteBuffer, fake_GetDirectBufferAddress, fake_GetDirectBufferCapacity};
static JNIEnv env_ = &table;
static JNIEnv *env = &env_;

#define J(cls, name) Java_io_lubricant_consensus_raft_gpu_##cls##_##name
jint J(GpuTable, abiVersion)(JNIEnv *, jclass);
jlong J(GpuTable, create)(JNIEnv *, jclass, jint, jint, jint, jint, jboolean);
void J(GpuTable, destroy)(JNIEnv *, jclass, jlong);
jstring J(GpuTable, lastError)(JNIEnv *, jclass, jlong);
jint J(GpuTable, option)(JNIEnv *, jclass, jlong, jint, jint);
jobject J(GpuTable, hostAlloc)(JNIEnv *, jclass, jlong, jlong);
jint J(GpuTable, hostFree)(JNIEnv *, jclass, jlong, jobject);
jint J(GpuTable, loadState)(JNIEnv *, jclass, jlong, jint, jint, jobjectArray);
jint J(GpuTable, readState)(JNIEnv *, jclass, jlong, jint, jint, jobjectArray);
jint J(GpuTable, submit)(JNIEnv *, jclass, jlong, jint, jint, jobject, jobject, jobject, jobject, jobject, jlong, jobject, jobject, jobject, jobject);
jint J(GpuTable, submit32c)(JNIEnv *, jclass, jlong, jint, jint, jobject, jobject, jobject, jlong, jobject, jobject, jobject, jobject, jobject);
jint J(GpuTable, unpack32)(JNIEnv *, jclass, jint, jint, jobject, jobject, jobject, jobject, jobject, jobject, jobject, jobject, jobject);
jlong J(GpuTable, tick2Create)(JNIEnv *, jclass, jlong, jint, jobject, jobject, jobject, jlong, jobject, jobject, jobject, jint, jlong, jobject, jobject, jobject, jobject,
                               jobject, jint, jobject, jobject, jobject);
jint J(GpuTable, tick2Launch)(JNIEnv *, jclass, jlong);
jint J(GpuTable, tick2Wait)(JNIEnv *, jclass, jlong);
jint J(GpuTable, tick2Destroy)(JNIEnv *, jclass, jlong);

enum { G = 96, P = 3, NCOL = sizeof(rg_group_state_t) / sizeof(void *) };

/* the 24 columns of rg_group_state_t for G fresh followers with a small log (what RaftContext.initialize would restore) */
static void make_state(rg_group_state_t *s)
{
    const size_t sizes[NCOL] = {8, 4, 4, 4, 1, 1, 4, 4, 4, 8, 8, 8, 8, 8, 8, 4, 4, 8 * RG_TERM_RUNS, 8 * RG_TERM_RUNS, 8 * (P - 1), 8 * (P - 1), 8 * (P - 1), 4 * (P - 1), 1 * (P - 1)};
    void **f = (void **)s;
    for (int i = 0; i < NCOL; i++) f[i] = calloc(G, sizes[i]);
    for (uint32_t g = 0; g < G; g++) {
        s->current_term[g] = 3; s->voted_for[g] = 1; s->role[g] = RG_FOLLOWER; s->current_leader[g] = 1; s->role_epoch[g] = 1; s->votes[g] = 1;
        s->commit_index[g] = 8; s->first_index[g] = 1; s->last_index[g] = 10 + g; s->run_count[g] = 1; s->run_offset[g] = g * RG_TERM_RUNS;
        s->run_start[g * RG_TERM_RUNS] = 1; s->run_term[g * RG_TERM_RUNS] = 3;
    }
}

static jobject buf(void *p, size_t n) { return fake_NewDirectByteBuffer(env, p, (jlong)n); }

int main(void)
{
    if (J(GpuTable, abiVersion)(env, NULL) != RG_ABI_VERSION) { fprintf(stderr, "abi\n"); return 1; }
    /* a refused create throws IllegalStateException with the library's message and returns 0 */
    if (J(GpuTable, create)(env, NULL, 0, G, 99, 0, 1) != 0 || !strstr(thrown, "cluster")) { fprintf(stderr, "create(bad cluster) did not throw: '%s'\n", thrown); return 1; }
    const jlong h = J(GpuTable, create)(env, NULL, 0, G, P, 0, 1);
    rg_table_t *direct = NULL;
    if (!h || rg_table_create(0, G, P, 0, 1, &direct) != 0) { fprintf(stderr, "create: %s\n", thrown); return 1; }
    if (J(GpuTable, option)(env, NULL, h, RG_OPT_REQUIRE_FENCED_TIMEOUTS, 1) != 0 || rg_table_option(direct, RG_OPT_REQUIRE_FENCED_TIMEOUTS, 1) != 0) return 1;
    if (J(GpuTable, option)(env, NULL, h, 12345, 1) == 0 || !strstr(J(GpuTable, lastError)(env, NULL, h)->text, "option")) { fprintf(stderr, "unknown option accepted\n"); return 1; }

    rg_group_state_t st;
    make_state(&st);
    jobject cols[NCOL];
    const size_t sizes[NCOL] = {8, 4, 4, 4, 1, 1, 4, 4, 4, 8, 8, 8, 8, 8, 8, 4, 4, 8 * RG_TERM_RUNS, 8 * RG_TERM_RUNS, 8 * (P - 1), 8 * (P - 1), 8 * (P - 1), 4 * (P - 1), 1 * (P - 1)};
    for (int i = 0; i < NCOL; i++) cols[i] = buf(((void **)&st)[i], G * sizes[i]);
    struct _jobject arr = {K_ARRAY, 0, 0, NCOL, cols, 0}, short_arr = {K_ARRAY, 0, 0, NCOL - 1, cols, 0};
    if (J(GpuTable, loadState)(env, NULL, h, 0, G, &short_arr) == 0 || !strstr(thrown, "24 columns")) { fprintf(stderr, "short column list accepted\n"); return 1; }
    if (J(GpuTable, loadState)(env, NULL, h, 0, G, &arr) != 0 || rg_load_state(direct, 0, G, &st) != 0) { fprintf(stderr, "loadState: %s\n", rg_last_error(direct)); return 1; }

    /* one round: AppendEntries at the log tail with two entries (even groups), an un-fenced and a fenced timeout (odd groups) */
    rg_ev_head_t head[G]; rg_ev_pair_t ab[G], cd[G]; int64_t terms[2 * G]; uint64_t nterms = 0;
    for (uint32_t g = 0; g < G; g++) {
        if (g % 2 == 0) {
            head[g].hdr = RG_HDR_MAKE(RG_EV_AE_REQ, 1, 0, 2); head[g].aux = (uint32_t)nterms;
            terms[nterms++] = 3; terms[nterms++] = 3;
            ab[g].x = 3; ab[g].y = 10 + g; cd[g].x = 3; cd[g].y = 11 + g;
        } else {
            head[g].hdr = RG_HDR_MAKE(RG_EV_TIMEOUT, 0, 0, 0); head[g].aux = (g % 4 == 1) ? 0u : 1u;
            ab[g].x = ab[g].y = cd[g].x = cd[g].y = 0;
        }
    }
    rg_reply_t rep_j[G], rep_d[G]; rg_logfx_t lfx_j[G], lfx_d[G]; rg_persist_t per_j[G], per_d[G];
    memset(rep_j, 0, sizeof rep_j); memset(rep_d, 0, sizeof rep_d);
    const jint rc = J(GpuTable, submit)(env, NULL, h, 1, G, NULL, buf(head, sizeof head), buf(ab, sizeof ab), buf(cd, sizeof cd), buf(terms, sizeof terms), (jlong)nterms, NULL,
                                        buf(rep_j, sizeof rep_j), buf(lfx_j, sizeof lfx_j), buf(per_j, sizeof per_j));
    rg_batch_t in = {1, G, NULL, head, ab, cd, terms, nterms, NULL};
    rg_outcome_t out = {rep_d, lfx_d, per_d};
    if (rc != 0 || rg_submit(direct, &in, &out, RG_MEM_HOST) != 0) { fprintf(stderr, "submit: %d %s\n", rc, rg_last_error(direct)); return 1; }
    if (memcmp(rep_j, rep_d, sizeof rep_j) || memcmp(lfx_j, lfx_d, sizeof lfx_j) || memcmp(per_j, per_d, sizeof per_j)) { fprintf(stderr, "outcomes differ\n"); return 1; }
    unsigned ok = 0, bad = 0, app = 0;
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t status = RG_F_STATUS(rep_j[g].flags);
        ok += status == RG_OK; bad += status == RG_BAD_EVENT; app += (rep_j[g].flags & RG_F_LOG_APPEND) != 0;
    }
    if (bad != G / 4 || ok != G - G / 4 || app != G / 2) { fprintf(stderr, "unexpected decisions: ok %u bad %u append %u\n", ok, bad, app); return 1; }

    /* the same state through readState: both tables hold the same image */
    rg_group_state_t a, b;
    make_state(&a); make_state(&b);
    jobject cols_a[NCOL];
    for (int i = 0; i < NCOL; i++) cols_a[i] = buf(((void **)&a)[i], G * sizes[i]);
    struct _jobject arr_a = {K_ARRAY, 0, 0, NCOL, cols_a, 0};
    if (J(GpuTable, readState)(env, NULL, h, 0, G, &arr_a) != 0 || rg_read_state(direct, 0, G, &b) != 0) return 1;
    for (int i = 0; i < NCOL; i++) if (memcmp(((void **)&a)[i], ((void **)&b)[i], G * sizes[i])) { fprintf(stderr, "state column %d differs\n", i); return 1; }

    /* compact rows in, compact outcome rows out (ABI 4) through the shim and directly — on tables that hold the state both just reached — and the unpacking.
     * (step32_kernel is a two-wavefront kernel: the emulation runs it with one OS thread per lane, RG_EMU_WAVES=1; skipped on a lane-serial emulation) */
    if (getenv("RG_EMU_WAVES") || !getenv("RG_ALLOW_HOST_EMULATION")) {
        rg_ev_head_t head32[G]; rg_ev_quad32_t abcd[G]; int32_t terms32[2 * G];
        for (uint32_t g = 0; g < G; g++) {       /* the next AppendEntries at every even group's new tail; a fenced timeout elsewhere */
            if (g % 2 == 0) { ab[g].y = 12 + g; cd[g].y = 13 + g; }
            else head[g].aux = rep_j[g].role_epoch;
        }
        const int64_t nt = rg_batch32_pack(&in, head32, abcd, terms32);
        if (nt < 0) { fprintf(stderr, "pack %lld\n", (long long)nt); return 1; }
        rg_out32_t row_j[G], row_d[G]; rg_persist32_t p32_j[G], p32_d[G];
        memset(p32_j, 0, sizeof p32_j); memset(p32_d, 0, sizeof p32_d);
        const jint rc32 = J(GpuTable, submit32c)(env, NULL, h, 1, G, buf(head32, sizeof head32), buf(abcd, sizeof abcd), nt ? buf(terms32, sizeof terms32) : NULL, (jlong)nt,
                                                 buf(row_j, sizeof row_j), buf(p32_j, sizeof p32_j), NULL, NULL, NULL);
        rg_batch32_t in32 = {1, G, NULL, head32, abcd, nt ? terms32 : NULL, (uint64_t)nt};
        rg_outcome32_t out32 = {row_d, p32_d, {NULL, NULL, NULL}};
        if (rc32 != 0 || rg_submit32c(direct, &in32, &out32, RG_MEM_HOST) != 0) { fprintf(stderr, "submit32c: %d %s\n", rc32, rg_last_error(direct)); return 1; }
        if (memcmp(row_j, row_d, sizeof row_j) || memcmp(p32_j, p32_d, sizeof p32_j)) { fprintf(stderr, "compact outcome rows differ\n"); return 1; }
        uint32_t ep_j[G], ep_d[G];
        for (uint32_t g = 0; g < G; g++) ep_j[g] = ep_d[g] = rep_j[g].role_epoch;
        rg_outcome32_t src = {row_j, p32_j, {NULL, NULL, NULL}};
        if (J(GpuTable, unpack32)(env, NULL, 1, G, buf(row_j, sizeof row_j), buf(p32_j, sizeof p32_j), NULL, NULL, NULL, buf(ep_j, sizeof ep_j), buf(rep_j, sizeof rep_j),
                                  buf(lfx_j, sizeof lfx_j), buf(per_j, sizeof per_j)) != 0 ||
            rg_outcome32_unpack(&src, 1, G, ep_d, &out) != 0) { fprintf(stderr, "unpack32\n"); return 1; }
        if (memcmp(rep_j, rep_d, sizeof rep_j) || memcmp(lfx_j, lfx_d, sizeof lfx_j) || memcmp(per_j, per_d, sizeof per_j) || memcmp(ep_j, ep_d, sizeof ep_j)) { fprintf(stderr, "unpacked outcomes differ\n"); return 1; }
        unsigned appended = 0, converted = 0;
        for (uint32_t g = 0; g < G; g++) { appended += (rep_j[g].flags & RG_F_LOG_APPEND) != 0; converted += (rep_j[g].flags & RG_F_ROLE_CHANGED) != 0; }
        if (appended != G / 2 || converted != G / 2) { fprintf(stderr, "compact round: %u appends, %u conversions\n", appended, converted); return 1; }
        printf("compact outcome rows through the shim: %u appends, %u conversions, identical to the C-ABI\n", appended, converted);

        /* the device-resident tick (ABI 5) through the shim and directly: the same rows once more on both tables, timers armed alike; every output column equal */
        {
            rg_table_t *tj = (rg_table_t *)(intptr_t)h;
            if (rg_timers_configure(tj, 900, 300, 7) || rg_timers_configure(direct, 900, 300, 7) || rg_timers_arm(tj, 1000) || rg_timers_arm(direct, 1000)) return 1;
            enum { NB = 12 };
            const size_t bytes[NB] = {8 * G, 16 * G, 4 * 2 * G, 8, 16 * G, 16 * G, 4 * G, 4 * G, 4, 48 * G, 32 * (P - 1) * G, G};
            jobject jb[NB]; void *db[NB];
            for (int i = 0; i < NB; i++) {
                jb[i] = J(GpuTable, hostAlloc)(env, NULL, h, (jlong)bytes[i]);
                if (!jb[i] || rg_host_alloc(direct, bytes[i], &db[i]) != 0) { fprintf(stderr, "hostAlloc %d\n", i); return 1; }
                memset(jb[i]->addr, 0, bytes[i]); memset(db[i], 0, bytes[i]);
            }
            const int64_t clock = 5000;      /* past every armed deadline: tickets fire */
            for (int side = 0; side < 2; side++) {
                void **b = side ? db : NULL;
                memcpy(side ? b[0] : jb[0]->addr, head32, sizeof head32);
                memcpy(side ? b[1] : jb[1]->addr, abcd, sizeof abcd);
                memcpy(side ? b[2] : jb[2]->addr, terms32, sizeof terms32);
                memcpy(side ? b[3] : jb[3]->addr, &clock, 8);
            }
            const jlong tk = J(GpuTable, tick2Create)(env, NULL, h, 1, jb[0], jb[1], jb[2], (jlong)(2 * G), jb[3], NULL, NULL, 1, 60, jb[4], jb[5], jb[6], jb[7], jb[8], (jint)G,
                                                      jb[9], jb[10], jb[11]);
            rg_tick2_io_t io;
            memset(&io, 0, sizeof io);
            io.rounds = 1; io.head = db[0]; io.abcd = db[1]; io.entry_terms = db[2]; io.entry_capacity = 2 * G; io.now = db[3]; io.critical_point = 1; io.cool_down_ms = 60;
            io.row = db[4]; io.persist32 = db[5]; io.expired_gid = db[6]; io.expired_epoch = db[7]; io.expired_count = db[8]; io.expired_capacity = G;
            io.send_head = db[9]; io.send = db[10]; io.ready = db[11];
            rg_tick2_t *td = NULL;
            if (!tk || rg_tick2_create(direct, &io, &td) != 0) { fprintf(stderr, "tick2Create: %s | %s\n", thrown, rg_last_error(direct)); return 1; }
            if (J(GpuTable, tick2Launch)(env, NULL, tk) || J(GpuTable, tick2Wait)(env, NULL, tk) || rg_tick2_launch(td) || rg_tick2_wait(td)) { fprintf(stderr, "tick2 launch\n"); return 1; }
            for (int i = 4; i < NB; i++)
                if (memcmp(jb[i]->addr, db[i], bytes[i])) { fprintf(stderr, "tick2: output column %d differs\n", i); return 1; }
            const uint32_t fired = *(uint32_t *)db[8];
            if (fired == 0 || fired > G) { fprintf(stderr, "tick2: %u fired tickets\n", fired); return 1; }
            /* a short column is refused before the library sees it */
            thrown[0] = 0;
            if (J(GpuTable, tick2Create)(env, NULL, h, 1, jb[0], jb[1], NULL, 0, jb[3], NULL, NULL, 1, 60, jb[8], jb[5], NULL, NULL, NULL, 0, NULL, NULL, NULL) != 0 || !thrown[0]) {
                fprintf(stderr, "tick2Create took a 4-byte row column\n"); return 1;
            }
            if (J(GpuTable, tick2Destroy)(env, NULL, tk) || rg_tick2_destroy(td)) return 1;
            for (int i = 0; i < NB; i++) { J(GpuTable, hostFree)(env, NULL, h, jb[i]); rg_host_free(direct, db[i]); }
            printf("the device-resident tick through the shim: %u fired tickets, every output column identical to the C-ABI\n", fired);
        }
    }

    /* misuse: a heap ByteBuffer (no direct address), a short buffer, a missing required column — IllegalArgumentException, the library never
     * sees the address; an optional column may be null */
    {
        rg_ev_head_t hd[G]; rg_ev_pair_t xy[G]; rg_reply_t rp[G]; rg_logfx_t lf[G]; rg_persist_t pr[G];
        memset(hd, 0, sizeof hd); memset(xy, 0, sizeof xy);
        struct _jobject heap = {K_BUFFER, NULL, (jlong)sizeof hd, 0, 0, 0};            /* GetDirectBufferAddress answers NULL for it */
        uint32_t gids[G];
        for (uint32_t g = 0; g < G; g++) gids[g] = g;
        struct _jobject heap_gid = {K_BUFFER, NULL, (jlong)sizeof gids, 0, 0, 0};
        thrown[0] = 0;
        if (J(GpuTable, submit)(env, NULL, h, 1, G, &heap_gid, buf(hd, sizeof hd), buf(xy, sizeof xy), buf(xy, sizeof xy), NULL, 0, NULL, buf(rp, sizeof rp), buf(lf, sizeof lf),
                                buf(pr, sizeof pr)) != -1 || !strstr(thrown, "gid") || !strstr(thrown, "DIRECT")) { fprintf(stderr, "heap gid accepted: '%s'\n", thrown); return 1; }
        thrown[0] = 0;
        if (J(GpuTable, submit)(env, NULL, h, 1, G, NULL, &heap, buf(xy, sizeof xy), buf(xy, sizeof xy), NULL, 0, NULL, buf(rp, sizeof rp), buf(lf, sizeof lf),
                                buf(pr, sizeof pr)) != -1 || !strstr(thrown, "head")) { fprintf(stderr, "heap head accepted: '%s'\n", thrown); return 1; }
        thrown[0] = 0;
        if (J(GpuTable, submit)(env, NULL, h, 1, G, NULL, buf(hd, sizeof hd), buf(xy, sizeof xy), buf(xy, sizeof xy), NULL, 0, NULL, buf(rp, sizeof rp - 16), buf(lf, sizeof lf),
                                buf(pr, sizeof pr)) != -1 || !strstr(thrown, "reply") || !strstr(thrown, "capacity")) { fprintf(stderr, "short reply accepted: '%s'\n", thrown); return 1; }
        thrown[0] = 0;
        if (J(GpuTable, submit)(env, NULL, h, 1, G, NULL, buf(hd, sizeof hd), buf(xy, sizeof xy), NULL, NULL, 0, NULL, buf(rp, sizeof rp), buf(lf, sizeof lf),
                                buf(pr, sizeof pr)) != -1 || !strstr(thrown, "cd") || !strstr(thrown, "null")) { fprintf(stderr, "missing cd accepted: '%s'\n", thrown); return 1; }
        jobject short_cols[NCOL];
        for (int i = 0; i < NCOL; i++) short_cols[i] = cols[i];
        short_cols[9] = buf(((void **)&st)[9], G * sizes[9] - 8);                       /* elected_term one element short */
        struct _jobject short_state = {K_ARRAY, 0, 0, NCOL, short_cols, 0};
        thrown[0] = 0;
        if (J(GpuTable, readState)(env, NULL, h, 0, G, &short_state) != -1 || !strstr(thrown, "capacity")) { fprintf(stderr, "short state column accepted: '%s'\n", thrown); return 1; }
        thrown[0] = 0;
    }

    /* page-locked memory as a direct buffer, and back */
    jobject pinned = J(GpuTable, hostAlloc)(env, NULL, h, 4096);
    if (!pinned || pinned->cap != 4096 || !pinned->addr) return 1;
    memset(pinned->addr, 0x5A, 4096);
    if (J(GpuTable, hostFree)(env, NULL, h, pinned) != 0) return 1;

    J(GpuTable, destroy)(env, NULL, h);
    rg_table_destroy(direct);
    printf("jni shim ok: %u rows decided identically through the shim and through the C-ABI (%u refused as un-fenced timeouts)\n", G, bad);
    return 0;
}
