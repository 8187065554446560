// ingress_cluster_flow.cpp — BASELINE configs[0] (the reference's 3-node demo) on the WIRE path only: three nodes, each a table + an Ingress +
// an IngressFlusher + MemoryLogs, that exchange nothing but bytes of the reference's wire protocol. Start-up timeouts, PreVote, RequestVote,
// a leader per group, client commands, replication (rg_replicate -> Ingress::encode_sends), acks, commit — every decision by the device code
// (host emulation of the kernels, lane-serial mode: IngressFlusher's wide_kernel switch), every host reaction through IngressFlusher::on_row.
// A leader is cut off half-way (its bytes are dropped both ways) and comes back: the others elect a new leader, the old one steps down.
// Checked at every tick: election safety (one leader per group and term), committed entries never change and agree across the nodes; at the
// end: every group committed commands on all three nodes, the logs are identical up to the smallest commit index.
// usage: ingress_cluster_flow [groups=6] [ticks=500] [compact | wide] [journal path prefix] [partition every N ticks | 0] [seed]     (compact: rg_submit32 — the GPU, or the wavefront mode of the emulation)
// TEST INFRASTRUCTURE (tests/test_devemu_cpu.py, tests/test_ingress_gpu.py). prints "ingress cluster ok=1"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "ingress_flusher.hpp"

using namespace rafting::wire;
using raftgpu::host::Entry;
using raftgpu::host::MemoryLog;

static const int P = 3;
static const int64_t TICK_MS = 50, HEARTBEAT_MS = 300, ELECTION_MS = 900, NEVER = INT64_MAX / 2;

struct Node {
    int slot = 0;
    rg_table_t *table = nullptr;
    std::unique_ptr<ContextIndex> index;
    std::vector<rg_ev_head_t> head[2];
    std::vector<rg_ev_quad32_t> abcd[2];
    std::vector<int32_t> terms[2];
    std::unique_ptr<Ingress> ing;
    std::unique_ptr<raftgpu::host::StableStore> store;       // N3: the node's journal of (term, votedFor)
    std::unique_ptr<IngressFlusher> flusher;
    std::vector<std::unique_ptr<MemoryLog>> logs;
    std::vector<int64_t> deadline;
    std::vector<int> role;
    std::vector<uint32_t> role_epoch;
    std::vector<uint8_t> want_prevote, want_reqvote, want_replicate;
    std::string outbox[P];                 // bytes for peer p, delivered next tick
};

int main(int argc, char **argv)
{
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 6;
    const int TICKS = argc > 2 ? atoi(argv[2]) : 500;
    const bool compact = argc > 3 && std::string(argv[3]) == "compact";
    const int EVERY = argc > 5 ? atoi(argv[5]) : 0;                          // > 0: a partition every so many ticks instead of the single one      // decide the batches with rg_submit32 (the GPU, or the wavefront emulation)
    const KryoBodyCodec codec({{"10.4.0.1", 7401}, {"10.4.0.2", 7402}, {"10.4.0.3", 7403}});
    std::mt19937_64 rng(argc > 6 ? (uint64_t)atoll(argv[6]) : 20240922ull);
    std::vector<std::string> ids(G);
    for (uint32_t g = 0; g < G; g++) ids[g] = "file/" + std::to_string(g);
    Node nodes[P];
    const uint32_t R = 8, LOCAL = P;       // connection p = the peer in slot p (the own slot stays unused), connection P = the node's own rows
    for (int n = 0; n < P; n++) {
        Node &nd = nodes[n];
        nd.slot = n;
        if (rg_table_create(0, G, P, (uint32_t)n, 1, &nd.table) != 0) { fprintf(stderr, "rg_table_create: %s\n", rg_last_error(nullptr)); return 1; }
        nd.index.reset(new ContextIndex(G));
        for (uint32_t g = 0; g < G; g++) nd.index->insert(ids[g].data(), ids[g].size(), g);
        const size_t cells = (size_t)G * R;
        for (int k = 0; k < 2; k++) { nd.head[k].resize(cells); nd.abcd[k].resize(cells); nd.terms[k].resize(1024); }
        nd.ing.reset(new Ingress(G, R, P + 1, codec, *nd.index, Ingress::Buffers{nd.head[0].data(), nd.abcd[0].data(), nd.terms[0].data(), 1024},
                                 Ingress::Buffers{nd.head[1].data(), nd.abcd[1].data(), nd.terms[1].data(), 1024}));
        for (int p = 0; p < P; p++) if (p != n) nd.ing->set_peer((uint32_t)p, p);
        for (uint32_t g = 0; g < G; g++) nd.logs.emplace_back(new MemoryLog);
        const std::string journal = std::string(argc > 4 ? argv[4] : "/tmp/ingress_cluster_flow") + ".node" + std::to_string(n) + ".journal";
        remove(journal.c_str());
        nd.store.reset(new raftgpu::host::StableStore(journal));
        Node *self = &nd;
        nd.flusher.reset(new IngressFlusher(nd.table, *nd.ing, codec, [self](uint32_t g) -> raftgpu::host::RaftLog & { return *self->logs[g]; },
                                            std::vector<int64_t>(G, 0), nd.store.get(), !compact));
        nd.deadline.assign(G, 0); nd.role.assign(G, RG_FOLLOWER); nd.role_epoch.assign(G, 1);
        nd.want_prevote.assign(G, 0); nd.want_reqvote.assign(G, 0); nd.want_replicate.assign(G, 0);
        for (uint32_t g = 0; g < G; g++) nd.deadline[g] = ELECTION_MS + (int64_t)(rng() % ELECTION_MS);       // RaftConfig: election timeout in [E, 2E)
    }
    std::map<std::pair<uint32_t, int64_t>, int> leader_of;        // (group, term) -> node: election safety
    std::vector<std::vector<int64_t>> committed(G);               // per group: the terms of the committed entries, as first seen
    int violations = 0;
    uint64_t commands = 0, elections = 0, step_downs = 0, sends_from_log = 0;
    int cut = -1;                                                  // the node whose bytes are dropped (both ways)
    int64_t now = 0;

    for (int tick = 0; tick < TICKS; tick++, now += TICK_MS) {
        if (EVERY == 0) {
            if (tick == TICKS / 2) {                               // cut off whoever leads group 0 right now, for 60 ticks
                for (int n = 0; n < P; n++) if (nodes[n].role[0] == RG_LEADER) cut = n;
            }
            if (tick == TICKS / 2 + 60) cut = -1;
        } else {                                                   // again and again: terms pile up, the logs grow more term runs than the device caches,
            if (tick >= 100 && (tick - 100) % EVERY == 0)          // a node that was away is probed at entries below the cached runs
                for (int n = 0; n < P; n++) if (nodes[n].role[0] == RG_LEADER) cut = n;
            if (tick >= 160 && (tick - 160) % EVERY == 0) cut = -1;
        }
        std::string inbox[P][P];                                   // [to][from]: what was written during the previous tick
        for (int n = 0; n < P; n++)
            for (int p = 0; p < P; p++) { if (n != cut && p != cut) inbox[p][n] = nodes[n].outbox[p]; nodes[n].outbox[p].clear(); }
        for (int n = 0; n < P; n++) {
            Node &nd = nodes[n];
            for (int p = 0; p < P; p++)
                if (p != n && !inbox[n][p].empty() && nd.ing->feed((uint32_t)p, reinterpret_cast<const uint8_t *>(inbox[n][p].data()), inbox[n][p].size()) < 0) {
                    fprintf(stderr, "node %d: the stream from %d broke the frame grammar\n", n, p); return 1;
                }
            for (uint32_t g = 0; g < G; g++) {
                if (now >= nd.deadline[g]) {                      // RaftRoutine.electionTimeout / keepAlive: the ticket fires once
                    nd.deadline[g] = NEVER;
                    nd.ing->add_row(LOCAL, g, rg_ev_head_t{RG_HDR_MAKE(RG_EV_TIMEOUT, 0, 0, 0), nd.role_epoch[g]}, 0, 0, 0, 0, Origin{NO_CONN, 0});
                }
                if (nd.role[g] == RG_LEADER && tick % 4 == (int)(g % 4)) {
                    nd.ing->add_row(LOCAL, g, rg_ev_head_t{RG_HDR_MAKE(RG_EV_CLIENT_APPEND, 0, 0, 1), 0}, 0, 0, 0, 0, Origin{NO_CONN, 0});
                    commands++;
                }
            }
            nd.flusher->on_row = [&](uint32_t g, const rg_ev_head_t &, const rg_reply_t &r) {
                const uint32_t f = r.flags;
                const int role_after = (int)RG_F_ROLE(f);
                if (f & RG_F_ROLE_CHANGED) {
                    if (nd.role[g] == RG_LEADER && role_after != RG_LEADER) step_downs++;
                    if (role_after == RG_LEADER && nd.role[g] != RG_LEADER) {
                        elections++;
                        const int64_t t = nd.flusher->term(g);
                        auto it = leader_of.find({g, t});
                        if (it != leader_of.end() && it->second != n) { fprintf(stderr, "tick %d group %u term %lld: two leaders (%d, %d)\n", tick, g, (long long)t, it->second, n); violations++; }
                        leader_of[{g, t}] = n;
                    }
                    nd.role[g] = role_after;
                }
                nd.role_epoch[g] = r.role_epoch;
                if (f & RG_F_RESET_TIMER) {                       // RaftRoutine.resetTimer (what rg_timers_update does on the device)
                    if (f & RG_F_TIMER_MUTED) nd.deadline[g] = NEVER;
                    else if (role_after == RG_LEADER) nd.deadline[g] = (f & RG_F_ROLE_CHANGED) ? now : now + HEARTBEAT_MS;
                    else nd.deadline[g] = now + ELECTION_MS + (int64_t)(rng() % ELECTION_MS);
                }
                switch (RG_F_EMIT(f)) {
                case RG_EMIT_PREVOTE: nd.want_prevote[g] = 1; break;
                case RG_EMIT_REQVOTE: nd.want_reqvote[g] = 1; break;
                case RG_EMIT_HEARTBEAT: nd.want_replicate[g] = 1; break;
                default: break;
                }
            };
            std::vector<std::string> out(P + 1);
            for (;;) {
                const int64_t k = nd.flusher->flush(out);
                if (k < 0) { fprintf(stderr, "node %d flush: %s\n", n, nd.flusher->error().c_str()); return 1; }
                if (k == 0) break;
            }
            for (int p = 0; p < P; p++) nd.outbox[p] += out[p];
            // what the rows asked this node to send
            std::vector<uint32_t> rep_gid;
            for (uint32_t g = 0; g < G; g++) {
                const bool pre = nd.want_prevote[g] != 0, req = nd.want_reqvote[g] != 0;
                nd.want_prevote[g] = nd.want_reqvote[g] = 0;
                if (nd.want_replicate[g] && nd.role[g] == RG_LEADER) rep_gid.push_back(g);
                nd.want_replicate[g] = 0;
                if (!pre && !req) continue;
                const auto last = nd.logs[g]->last();
                for (int p = 0; p < P; p++) {
                    if (p == n) continue;
                    Frame f;
                    f.type = ENQ;
                    const Method m = req ? M_REQUEST_VOTE : M_PRE_VOTE;    // (a row that asked for RequestVote supersedes an earlier PreVote of the same flush)
                    Request q;
                    q.term = req ? nd.flusher->term(g) : nd.flusher->term(g) + 1;       // member/Candidate.java:90-143, member/Follower.java:223-279
                    q.node = n; q.x = last ? last->index : nd.logs[g]->epoch().index; q.y = last ? last->term : nd.logs[g]->epoch().term;
                    f.sequence = nd.ing->send_sequence((uint32_t)p)++;
                    f.head = make_scope(m, ids[g]);
                    codec.encode_request(m, q, f.body);
                    encode_frame(f, false, nd.outbox[p]);
                    nd.ing->pending((uint32_t)p).put(f.sequence, m, g, Pending{nd.role_epoch[g], 0, 0});
                }
            }
            if (!rep_gid.empty()) {                               // Leader.replicateLog for the groups that asked: one launch, then frames per follower
                const uint32_t cnt = (uint32_t)rep_gid.size();
                std::vector<uint8_t> hb(cnt, 1);
                std::vector<rg_send_head_t> sh(cnt);
                std::vector<rg_send_t> sd((size_t)cnt * (P - 1));
                if (rg_replicate(nd.table, cnt, rep_gid.data(), hb.data(), nullptr, sh.data(), sd.data(), RG_MEM_HOST) != 0) { fprintf(stderr, "rg_replicate: %s\n", rg_last_error(nd.table)); return 1; }
                struct Log : Ingress::TermOf { Node *nd; int64_t term_of(uint32_t g, int64_t i) override { auto e = nd->logs[g]->get(i); return e ? e->term : 0; } } log;
                log.nd = &nd;
                for (int j = 0; j < P - 1; j++) {
                    const int peer = j < n ? j : j + 1;
                    uint32_t from_log = 0;
                    nd.ing->encode_sends((uint32_t)peer, n, cnt, rep_gid.data(), sh.data(), sd.data() + (size_t)j * cnt, log, nd.outbox[peer], &from_log);
                    sends_from_log += from_log;
                }
            }
        }
        // ---- invariants of this tick ---------------------------------------------------------------------------------------
        for (uint32_t g = 0; g < G; g++) {
            for (int n = 0; n < P; n++) {
                const MemoryLog &log = *nodes[n].logs[g];
                for (int64_t i = 1; i <= log.lastCommitted(); i++) {
                    const auto e = log.get(i);
                    if (!e) { fprintf(stderr, "tick %d group %u node %d: committed entry %lld is missing\n", tick, g, n, (long long)i); violations++; break; }
                    if ((int64_t)committed[g].size() < i) committed[g].push_back(e->term);
                    else if (committed[g][(size_t)i - 1] != e->term) { fprintf(stderr, "tick %d group %u node %d: committed entry %lld changed its term\n", tick, g, n, (long long)i); violations++; }
                }
            }
        }
        if (violations) break;
    }
    uint64_t min_commit = UINT64_MAX, refused = 0;
    for (uint32_t g = 0; g < G; g++)
        for (int n = 0; n < P; n++) min_commit = std::min<uint64_t>(min_commit, (uint64_t)nodes[n].logs[g]->lastCommitted());
    // N3: what every node's journal holds is what its table holds (the journal was written before any reply that depended on it left)
    uint64_t journal_wrong = 0, persisted = 0;
    for (int n = 0; n < P; n++) {
        std::vector<int64_t> term(G), elected_term(G), commit(G), eidx(G), eterm(G), first(G), last(G), run_start((size_t)G * RG_TERM_RUNS), run_term((size_t)G * RG_TERM_RUNS);
        std::vector<int32_t> voted(G), role(G), leader(G), votes(G), pr((size_t)G * (P - 1));
        std::vector<uint8_t> td(G), prepared(G), pp((size_t)G * (P - 1));
        std::vector<uint32_t> repoch(G), elected_epoch(G), run_count(G), run_offset(G);
        std::vector<int64_t> pe((size_t)G * (P - 1)), pn((size_t)G * (P - 1)), pm((size_t)G * (P - 1));
        rg_group_state_t st{};
        st.current_term = term.data(); st.voted_for = voted.data(); st.role = role.data(); st.current_leader = leader.data();
        st.timeout_detected = td.data(); st.repl_prepared = prepared.data(); st.role_epoch = repoch.data(); st.votes = votes.data();
        st.elected_epoch = elected_epoch.data(); st.elected_term = elected_term.data(); st.commit_index = commit.data();
        st.epoch_index = eidx.data(); st.epoch_term = eterm.data(); st.first_index = first.data(); st.last_index = last.data();
        st.run_count = run_count.data(); st.run_offset = run_offset.data(); st.run_start = run_start.data(); st.run_term = run_term.data();
        st.peer_last_epoch = pe.data(); st.peer_next_index = pn.data(); st.peer_match_index = pm.data(); st.peer_rejection = pr.data();
        st.peer_pending = pp.data();
        if (rg_read_state(nodes[n].table, 0, G, &st) != 0) { fprintf(stderr, "rg_read_state: %s\n", rg_last_error(nodes[n].table)); return 1; }
        for (uint32_t g = 0; g < G; g++) {
            int64_t jt = 0; int32_t jv = 0;
            const bool have = nodes[n].store->restore(g, &jt, &jv);
            if (!(have ? (jt == term[g] && jv == voted[g]) : (term[g] == 0 && voted[g] == RG_NO_NODE)) || last[g] != (nodes[n].logs[g]->last() ? nodes[n].logs[g]->last()->index : 0)) {
                journal_wrong++;
                fprintf(stderr, "node %d group %u: journal (%d: %lld, %d) vs table (%lld, %d)\n", n, g, (int)have, (long long)jt, jv, (long long)term[g], voted[g]);
            }
        }
        persisted += nodes[n].flusher->stats().persisted;
    }
    uint64_t rows = 0, frames = 0, repaired = 0;
    for (int n = 0; n < P; n++) { rows += nodes[n].flusher->stats().rows; frames += nodes[n].flusher->stats().frames; repaired += nodes[n].flusher->stats().repaired; refused += nodes[n].ing->refused(); }
    const bool ok = violations == 0 && min_commit >= 10 && elections >= G + 1 && step_downs >= 1 && journal_wrong == 0 && persisted > 0;
    printf("ingress cluster ok=%d: %u groups x 3 nodes, %d ticks, %llu rows decided, %llu response frames, %llu elections won, %llu step-downs, %llu client commands, "
           "smallest commit index %llu, %llu (term, votedFor) records journalled, %llu requests whose prevLogTerm came from the host's log, %llu rows repaired, %llu frames refused (responses whose request was fenced or forgotten), %d violations\n", (int)ok, G, TICKS,
           (unsigned long long)rows, (unsigned long long)frames, (unsigned long long)elections, (unsigned long long)step_downs, (unsigned long long)commands,
           (unsigned long long)min_commit, (unsigned long long)persisted, (unsigned long long)sends_from_log, (unsigned long long)repaired, (unsigned long long)refused, violations);
    for (int n = 0; n < P; n++) rg_table_destroy(nodes[n].table);
    return ok ? 0 : 1;
}
