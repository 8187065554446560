// ingress_pipeline.cpp — the flusher's loop of INTEGRATION.md §1 as a program: socket bytes -> decisions on the GPU -> response bytes.
//
//   reader threads   feed the byte streams of C peer connections to the ingress in 64 KiB reads (rafting_amd/host/ingress.hpp)
//   flush thread     seal() -> rg_submit_async_packed (the sealed bank IS the page-locked upload buffer) -> rg_submit_wait ->
//                    [StableStore::persist of the RG_F_PERSIST rows] -> emit() the PongEvent frames on E threads -> recycle()
//                    while the readers fill the other bank
// Workload: config 3's shape with a verifiable end state. 80 % of the groups are Followers of term 7 receiving AppendEntries at their log
// tail (0 / 1 / 2 / 4 entries of term 7, leaderCommit = prevLogIndex) from their leader's connection; 20 % are prepared Leaders receiving
// heartbeat acks from their followers' connections. At the end every Follower's log must end exactly where its requests said, every request
// must have got one successful response frame, no row may have needed the host.
// Needs libraftgpu.so and a GPU (tests/test_devemu_cpu.py runs it against the host emulation of the kernels for its logic only).
// With old_prev_percent > 0 the Followers' logs hold SIX term runs (the device caches the newest four) and that share of the requests are
// heartbeats whose prevLogIndex lies in the oldest run: those rows come back RG_NEED_HOST, the rows behind them RG_SKIPPED_AFTER_NEED_HOST, and
// the flush thread repairs them (repair_need_host: hints from the host's log, sparse resubmission) before it emits the batch's responses.
// usage: ingress_pipeline [groups=65536] [rounds per group=64] [conns=8] [readers=4] [emitters=4] [max rounds per batch=16] [journal path | -]
//                         [old_prev_percent=0]
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ingress.hpp"
#include "stable_store.hpp"

using namespace rafting::wire;

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pin(unsigned t)
{
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(t % std::thread::hardware_concurrency(), &set);
    pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}
// the host's RaftLog for the repair: the Followers' logs are the initial runs plus entries of term `term` behind them
struct PipelineLog : RepairHost {
    rg_table_t *table;
    int64_t term, last0;
    int runs;
    uint64_t repaired = 0, submits = 0;
    int64_t term_at(uint32_t, int64_t index) override
    {
        if (index < 1) return -1;
        if (index > last0) return term;
        return term - runs + 1 + std::min<int64_t>(runs - 1, (index - 1) / (last0 / runs));
    }
    int64_t conflict(uint32_t, int64_t, const int64_t *, uint32_t) override { return 0; }      // every shipped entry extends the log in its own term
    int64_t epoch_index(uint32_t) override { return 0; }
    int submit(const rg_batch_t &in, const rg_outcome_t &out) override { submits++; return rg_submit(table, &in, &out, RG_MEM_HOST); }
    void applied(uint32_t, size_t, const rg_reply_t &, const rg_logfx_t &, const rg_persist_t &) override { repaired++; }
};

#define RG(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, rg_last_error(table)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536, ROUNDS = argc > 2 ? (uint32_t)atoi(argv[2]) : 64, C = argc > 3 ? (uint32_t)atoi(argv[3]) : 8;
    const unsigned READERS = argc > 4 ? (unsigned)atoi(argv[4]) : 4, EMITTERS = argc > 5 ? (unsigned)atoi(argv[5]) : 4;
    const uint32_t R = argc > 6 ? (uint32_t)atoi(argv[6]) : 16;
    const char *journal = argc > 7 && strcmp(argv[7], "-") != 0 ? argv[7] : nullptr;
    const int OLD_PCT = argc > 8 ? atoi(argv[8]) : 0;
    const int RUNS = OLD_PCT > 0 ? 6 : 1;                       // term runs of a Follower's log: terms TERM - RUNS + 1 .. TERM, LAST0 / RUNS entries each
    const int P = 5, SELF = 0, F = P - 1;
    const int64_t TERM = 7, LAST0 = 1000;

    rg_table_t *table = nullptr;
    if (rg_table_create(0, G, P, SELF, 1, &table) != 0) { fprintf(stderr, "rg_table_create: %s\n", rg_last_error(nullptr)); return 1; }
    // ---- contexts, their state, the byte streams -------------------------------------------------------------------------
    const KryoBodyCodec codec({{"10.0.0.1", 7001}, {"10.0.0.2", 7002}, {"10.0.0.3", 7003}, {"10.0.0.4", 7004}, {"10.0.0.5", 7005}});
    ContextIndex index(G);
    std::vector<std::string> ids(G);
    std::vector<uint8_t> leads(G);
    std::vector<uint32_t> conn_of(G);
    for (uint32_t g = 0; g < G; g++) {
        char b[48];
        snprintf(b, sizeof b, "orders/partition-%05u", g);
        ids[g] = b;
        if (!index.insert(ids[g].data(), ids[g].size(), g)) return 2;
        const uint64_t h = (uint64_t)g * 0x9E3779B97F4A7C15ull;
        leads[g] = (h >> 32) % 5 == 0;
        conn_of[g] = (uint32_t)((h >> 40) % C);                  // a Follower's leader sits behind this connection; peer slot = 1 + conn % 4
    }
    {
        std::vector<int64_t> term(G, TERM), elected_term(G, 0), commit(G, LAST0), eidx(G, 0), eterm(G, 0), first(G, 1), last(G, LAST0);
        std::vector<int64_t> run_start((size_t)G * RUNS), run_term((size_t)G * RUNS);
        std::vector<int32_t> voted(G), role(G), leader(G), votes(G, 1);
        std::vector<uint8_t> td(G, 0), prepared(G);
        std::vector<uint32_t> repoch(G, 3), elected_epoch(G, 0), run_count(G, (uint32_t)RUNS), run_offset(G);
        std::vector<int64_t> pe((size_t)G * F, 0), pn((size_t)G * F, LAST0 + 1), pm((size_t)G * F, LAST0);
        std::vector<int32_t> pr((size_t)G * F, 0);
        std::vector<uint8_t> pp((size_t)G * F, 0);
        for (uint32_t g = 0; g < G; g++) {
            run_offset[g] = g * (uint32_t)RUNS;
            for (int k = 0; k < RUNS; k++) { run_start[(size_t)g * RUNS + k] = 1 + k * (LAST0 / RUNS); run_term[(size_t)g * RUNS + k] = TERM - RUNS + 1 + k; }
            role[g] = leads[g] ? RG_LEADER : RG_FOLLOWER;
            voted[g] = leads[g] ? SELF : 1 + (int32_t)(conn_of[g] % 4);
            leader[g] = leads[g] ? RG_NO_NODE : voted[g];
            prepared[g] = leads[g];
        }
        rg_group_state_t st{};
        st.current_term = term.data(); st.voted_for = voted.data(); st.role = role.data(); st.current_leader = leader.data();
        st.timeout_detected = td.data(); st.repl_prepared = prepared.data(); st.role_epoch = repoch.data(); st.votes = votes.data();
        st.elected_epoch = elected_epoch.data(); st.elected_term = elected_term.data(); st.commit_index = commit.data();
        st.epoch_index = eidx.data(); st.epoch_term = eterm.data(); st.first_index = first.data(); st.last_index = last.data();
        st.run_count = run_count.data(); st.run_offset = run_offset.data(); st.run_start = run_start.data(); st.run_term = run_term.data();
        st.peer_last_epoch = pe.data(); st.peer_next_index = pn.data(); st.peer_match_index = pm.data(); st.peer_rejection = pr.data();
        st.peer_pending = pp.data();
        RG(rg_load_state(table, 0, G, &st));
    }
    std::vector<std::string> stream(C);
    struct Put { uint32_t conn; int32_t seq; uint32_t gid; Pending p; };
    std::vector<Put> puts;
    std::vector<int32_t> seq(C, 0);
    std::vector<int64_t> want_last(G, LAST0);
    uint64_t x = 0x9E3779B97F4A7C15ull, rows = 0, requests = 0;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (uint32_t r = 0; r < ROUNDS; r++)
        for (uint32_t g = 0; g < G; g++) {
            Frame f;
            f.head = make_scope(M_APPEND_ENTRIES, ids[g]);
            uint32_t conn;
            if (!leads[g]) {
                conn = conn_of[g];
                Request q;
                q.term = TERM; q.node = 1 + (int32_t)(conn % 4); q.x = want_last[g]; q.y = TERM; q.leader_commit = q.x;
                const uint64_t e = rnd() % 4;
                q.entry_terms.assign(e == 3 ? 4 : e, TERM);
                if (OLD_PCT > 0 && (int)(rnd() % 100) < OLD_PCT) {   // a heartbeat probing an entry of the oldest run: consistent, but below the cached runs
                    q.x = 2; q.y = TERM - RUNS + 1; q.leader_commit = 0;
                    q.entry_terms.clear();
                }
                want_last[g] += (int64_t)q.entry_terms.size();
                f.type = ENQ;
                codec.encode_request(M_APPEND_ENTRIES, q, f.body);
                requests++;
            } else {
                conn = (uint32_t)(rnd() % C);
                f.type = ACK;
                codec.encode_response(Response{TERM, true}, f.body);
                puts.push_back(Put{conn, seq[conn], g, Pending{3, 0, LAST0}});     // role epoch 3, epoch.index 0, lastIndex sent = matchIndex: a heartbeat's ack
            }
            f.sequence = seq[conn]++;
            encode_frame(f, false, stream[conn]);
            rows++;
        }
    size_t bytes = 0;
    for (const std::string &s : stream) bytes += s.size();

    // ---- page-locked banks and outcome buffers ----------------------------------------------------------------------------
    const size_t cells = (size_t)G * R;
    const uint64_t TERMS_CAP = 1 << 16;
    Ingress::Buffers bank[2];
    rg_outcome_packed_t out[2];
    for (int i = 0; i < 2; i++) {
        void *p;
        RG(rg_host_alloc(table, cells * sizeof(rg_ev_head_t), &p)); bank[i].head = (rg_ev_head_t *)p;
        RG(rg_host_alloc(table, cells * sizeof(rg_ev_quad32_t), &p)); bank[i].abcd = (rg_ev_quad32_t *)p;
        RG(rg_host_alloc(table, TERMS_CAP * sizeof(int32_t), &p)); bank[i].entry_terms = (int32_t *)p;
        bank[i].entry_cap = TERMS_CAP;
        RG(rg_host_alloc(table, cells * sizeof(rg_reply_t), &p)); out[i].reply = (rg_reply_t *)p;
        RG(rg_host_alloc(table, cells * sizeof(rg_logfx_t), &p)); out[i].logfx = (rg_logfx_t *)p;
        RG(rg_host_alloc(table, cells * sizeof(rg_persist_t), &p)); out[i].persist = (rg_persist_t *)p;
        RG(rg_host_alloc(table, 2 * sizeof(uint32_t), &p)); out[i].counts = (uint32_t *)p;
        out[i].logfx_cap = out[i].persist_cap = (uint32_t)cells;
    }
    Ingress ing(G, R, C, codec, index, bank[0], bank[1], 1u << 22);
    for (uint32_t c = 0; c < C; c++) ing.set_peer(c, 1 + (int32_t)(c % 4));
    for (const Put &p : puts) ing.pending(p.conn).put(p.seq, M_APPEND_ENTRIES, p.gid, p.p);
    std::unique_ptr<raftgpu::host::StableStore> store;
    if (journal) { remove(journal); store.reset(new raftgpu::host::StableStore(journal)); }

    // ---- run -------------------------------------------------------------------------------------------------------------------
    const size_t CH = 64 * 1024;
    std::atomic<unsigned> reading{READERS};
    const double t0 = now_s();
    std::vector<std::thread> readers;
    for (unsigned t = 0; t < READERS; t++)
        readers.emplace_back([&, t] {
            pin(t);
            std::vector<size_t> at(C, 0);
            for (bool more = true; more;) {                      // its connections in turn, one read each: rows of all of them arrive interleaved
                more = false;
                for (uint32_t c = t; c < C; c += READERS) {
                    if (at[c] >= stream[c].size()) continue;
                    const size_t n = std::min(CH, stream[c].size() - at[c]);
                    if (ing.feed(c, reinterpret_cast<const uint8_t *>(stream[c].data()) + at[c], n) < 0) abort();
                    at[c] += n;
                    more = true;
                }
            }
            reading--;
        });
    uint64_t decided = 0, replied = 0, succeeded = 0, need_host = 0, frames = 0, out_bytes = 0, batches = 0, persisted = 0, wide = 0;
    PipelineLog hostlog;
    hostlog.table = table; hostlog.term = TERM; hostlog.last0 = LAST0; hostlog.runs = RUNS;
    double t_repair = 0;
    double t_submit = 0, t_emit = 0, t_recycle = 0, t_seal = 0, t_scan = 0, t_persist = 0;
    for (;;) {
        const bool readers_done = reading.load() == 0;
        double a = now_s();
        const SealedBatch &b = ing.seal();
        t_seal += now_s() - a;
        if (b.rows == 0) {
            ing.recycle(b);
            if (readers_done && ing.held() == 0) break;
            std::this_thread::yield();
            continue;
        }
        wide += b.wide.size();
        const int k = ing.bank_of(b);
        a = now_s();
        RG(rg_submit_async_packed(table, &b.batch, &out[k]));
        const int w = rg_submit_wait(table);
        if (w < 0) { fprintf(stderr, "rg_submit_wait: %s\n", rg_last_error(table)); return 1; }
        t_submit += now_s() - a;
        a = now_s();
        if (OLD_PCT > 0 && repair_need_host(b, out[k].reply, out[k].logfx, true, hostlog) < 0) { fprintf(stderr, "repair failed: %s\n", rg_last_error(table)); return 1; }
        t_repair += now_s() - a;
        a = now_s();
        const size_t n = (size_t)b.batch.rounds * G;
        std::vector<raftgpu::host::StableStore::Record> dirty;
        size_t pi = 0;
        for (size_t i = 0; i < n; i++) {
            if (RG_HDR_KIND(b.batch.head[i].hdr) == RG_EV_NONE) continue;
            const uint32_t fl = out[k].reply[i].flags;
            decided++;
            replied += (fl & RG_F_REPLIED) != 0;
            succeeded += (fl & RG_F_SUCCESS) != 0;
            need_host += RG_F_STATUS(fl) == RG_NEED_HOST || RG_F_STATUS(fl) == RG_SKIPPED_AFTER_NEED_HOST;
            if (fl & RG_F_PERSIST) {                                   // (a repaired row's persist item reached the host through applied(): none in this workload)
                if (pi < out[k].counts[1]) { const rg_persist_t &p = out[k].persist[pi]; dirty.push_back({(uint32_t)(i % G), p.term, p.voted_for}); }
                pi++;
            }
        }
        t_scan += now_s() - a;
        a = now_s();
        if (store && !dirty.empty()) { store->persist(dirty); persisted += dirty.size(); }      // N3: before any reply of this batch leaves
        t_persist += now_s() - a;
        a = now_s();
        {
            std::vector<std::thread> th;
            std::vector<size_t> made(EMITTERS, 0), ob(EMITTERS, 0);
            for (unsigned t = 0; t < EMITTERS; t++)
                th.emplace_back([&, t] {
                    pin(READERS + t);
                    std::vector<std::string> o(C);
                    made[t] = ing.emit(b, out[k].reply, o, n * t / EMITTERS, n * (t + 1) / EMITTERS);
                    for (const std::string &s : o) ob[t] += s.size();
                });
            for (std::thread &t : th) t.join();
            for (unsigned t = 0; t < EMITTERS; t++) { frames += made[t]; out_bytes += ob[t]; }
        }
        t_emit += now_s() - a;
        a = now_s();
        ing.recycle(b);
        t_recycle += now_s() - a;
        batches++;
    }
    for (std::thread &t : readers) t.join();
    const double s = now_s() - t0;

    // ---- verify ----------------------------------------------------------------------------------------------------------------
    std::vector<int64_t> term(G), elected_term(G), commit(G), eidx(G), eterm(G), first(G), last(G), run_start((size_t)G * RG_TERM_RUNS), run_term((size_t)G * RG_TERM_RUNS);
    std::vector<int32_t> voted(G), role(G), leader(G), votes(G), pr((size_t)G * F);
    std::vector<uint8_t> td(G), prepared(G), pp((size_t)G * F);
    std::vector<uint32_t> repoch(G), elected_epoch(G), run_count(G), run_offset(G);
    std::vector<int64_t> pe((size_t)G * F), pn((size_t)G * F), pm((size_t)G * F);
    rg_group_state_t st{};
    st.current_term = term.data(); st.voted_for = voted.data(); st.role = role.data(); st.current_leader = leader.data();
    st.timeout_detected = td.data(); st.repl_prepared = prepared.data(); st.role_epoch = repoch.data(); st.votes = votes.data();
    st.elected_epoch = elected_epoch.data(); st.elected_term = elected_term.data(); st.commit_index = commit.data();
    st.epoch_index = eidx.data(); st.epoch_term = eterm.data(); st.first_index = first.data(); st.last_index = last.data();
    st.run_count = run_count.data(); st.run_offset = run_offset.data(); st.run_start = run_start.data(); st.run_term = run_term.data();
    st.peer_last_epoch = pe.data(); st.peer_next_index = pn.data(); st.peer_match_index = pm.data(); st.peer_rejection = pr.data();
    st.peer_pending = pp.data();
    RG(rg_read_state(table, 0, G, &st));
    uint64_t wrong = 0;
    for (uint32_t g = 0; g < G; g++) wrong += last[g] != want_last[g] || term[g] != TERM || role[g] != (leads[g] ? RG_LEADER : RG_FOLLOWER);
    const bool ok = decided == rows && replied == requests && succeeded == requests && frames == requests && need_host == 0 && wrong == 0 && wide == 0 &&
                    ing.refused() == 0 && (OLD_PCT == 0 || hostlog.repaired > 0);
    printf("ingress pipeline ok=%d: %u groups, %llu rows (%llu requests) in %.1f MB of frames on %u connections, %u readers, %u emitters, <= %u rounds per batch\n",
           (int)ok, G, (unsigned long long)rows, (unsigned long long)requests, bytes / 1e6, C, READERS, EMITTERS, R);
    printf("  socket bytes -> decisions -> response bytes: %.3f s = %.3e rows/s end to end; %llu batches (%.0f rows each), %llu response frames (%.1f MB), "
           "%llu groups wrong, %llu rows needed the host, %llu persisted\n", s, rows / s, (unsigned long long)batches, batches ? (double)decided / batches : 0.0,
           (unsigned long long)frames, out_bytes / 1e6, (unsigned long long)wrong, (unsigned long long)need_host, (unsigned long long)persisted);
    printf("  flush thread: seal %.3f s, submit + wait %.3f s, repair %.3f s (%llu rows in %llu sparse submits), reply scan %.3f s, persist %.3f s, emit %.3f s, "
           "recycle %.3f s (of %.3f s)\n", t_seal, t_submit, t_repair, (unsigned long long)hostlog.repaired, (unsigned long long)hostlog.submits, t_scan, t_persist, t_emit,
           t_recycle, s);
    rg_table_destroy(table);
    return ok ? 0 : 1;
}
