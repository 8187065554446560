// ingress_flusher_flow.cpp — rafting_amd/host/ingress_flusher.{hpp,cpp} on the host emulation of the kernels (lane-serial mode: the batches are
// decided by the wide step kernel, IngressFlusher's wide_kernel switch): frames from a leader -> ingress -> one multi-round launch in which
// every group misses the device's cached term runs in its second row -> the repair reads the hints from real MemoryLogs (six term runs each,
// the device caches four) and applies the effects of the repaired rows one by one -> a row beyond int32 beside the batch -> the durability
// journal before any reply -> response frames. At the end every MemoryLog, the table and the responses agree with what the requests said.
// With a third argument S > 1 the same flow runs through a SHARDED ingress in front of S tables (block partition of the groups, one table per shard,
// one IngressFlusher for all of them): same requests, same expectations, every table read back.
// TEST INFRASTRUCTURE (tests/test_devemu_cpu.py). prints "ingress flusher ok=1"
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "ingress_flusher.hpp"

using namespace rafting::wire;
using raftgpu::host::Entry;
using raftgpu::host::MemoryLog;
using raftgpu::host::StableStore;

#define RG(call) do { if ((call) != 0) { fprintf(stderr, "%s failed\n", #call); return 1; } } while (0)

int main(int argc, char **argv)
{
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 70;
    const char *journal = argc > 2 ? argv[2] : "/tmp/ingress_flusher_flow.journal";
    const uint32_t S = argc > 3 ? (uint32_t)atoi(argv[3]) : 1;              // tables behind the ingress
    const uint32_t PER = (G + S - 1) / S;                                   // groups per shard (the last one may hold fewer)
    const int P = 3, SELF = 1, F = P - 1, RUNS = 6;
    const int64_t TERM = 7, PER_RUN = 10, LAST0 = RUNS * PER_RUN;          // indices 1..60, terms 2..7; the device caches the runs from index 21 on
    const uint32_t SPECIAL = 5;                                             // the group that gets ONE request whose term lies beyond int32
    std::vector<rg_table_t *> tables(S, nullptr);
    for (uint32_t s = 0; s < S; s++) {
        const uint32_t n = std::min(PER, G - s * PER);
        if (rg_table_create(0, n, P, SELF, 1, &tables[s]) != 0) { fprintf(stderr, "rg_table_create: %s\n", rg_last_error(nullptr)); return 1; }
    }
    std::vector<std::unique_ptr<MemoryLog>> logs;
    {
        std::vector<int64_t> term(G, TERM), elected_term(G, 0), commit(G, LAST0), eidx(G, 0), eterm(G, 0), first(G, 1), last(G, LAST0);
        std::vector<int64_t> run_start((size_t)G * RUNS), run_term((size_t)G * RUNS);
        std::vector<int32_t> voted(G, 0), role(G, RG_FOLLOWER), leader(G, 0), votes(G, 1), pr((size_t)G * F, 0);
        std::vector<uint8_t> td(G, 0), prepared(G, 0), pp((size_t)G * F, 0);
        std::vector<uint32_t> repoch(G, 2), elected_epoch(G, 0), run_count(G, RUNS), run_offset(G);
        std::vector<int64_t> pe((size_t)G * F, 0), pn((size_t)G * F, 0), pm((size_t)G * F, 0);
        for (uint32_t g = 0; g < G; g++) {
            run_offset[g] = g * RUNS;
            std::unique_ptr<MemoryLog> log(new MemoryLog);
            std::vector<Entry> es;
            for (int k = 0; k < RUNS; k++) {
                run_start[(size_t)g * RUNS + k] = 1 + k * PER_RUN; run_term[(size_t)g * RUNS + k] = TERM - RUNS + 1 + k;
                for (int64_t i = 0; i < PER_RUN; i++) es.push_back(Entry{1 + k * PER_RUN + i, TERM - RUNS + 1 + k});
            }
            log->append(es);
            log->markCommitted(LAST0);
            logs.push_back(std::move(log));
        }
        rg_group_state_t st{};
        st.current_term = term.data(); st.voted_for = voted.data(); st.role = role.data(); st.current_leader = leader.data();
        st.timeout_detected = td.data(); st.repl_prepared = prepared.data(); st.role_epoch = repoch.data(); st.votes = votes.data();
        st.elected_epoch = elected_epoch.data(); st.elected_term = elected_term.data(); st.commit_index = commit.data();
        st.epoch_index = eidx.data(); st.epoch_term = eterm.data(); st.first_index = first.data(); st.last_index = last.data();
        st.run_count = run_count.data(); st.run_offset = run_offset.data(); st.run_start = run_start.data(); st.run_term = run_term.data();
        st.peer_last_epoch = pe.data(); st.peer_next_index = pn.data(); st.peer_match_index = pm.data(); st.peer_rejection = pr.data();
        st.peer_pending = pp.data();
        for (uint32_t s = 0; s < S; s++) {                                    // shard s = groups [s * PER, ...): the columns are uniform, the run offsets per table
            const uint32_t n = std::min(PER, G - s * PER);
            RG(rg_load_state(tables[s], 0, n, &st));
        }
    }
    const KryoBodyCodec codec({{"10.3.0.1", 7301}, {"10.3.0.2", 7302}, {"10.3.0.3", 7303}});
    ContextIndex index(G);
    std::vector<std::string> ids(G);
    for (uint32_t g = 0; g < G; g++) { ids[g] = "ledger/" + std::to_string(g); if (!index.insert(ids[g].data(), ids[g].size(), g)) return 2; }
    const uint32_t R = 8;
    const size_t cells = (size_t)PER * S * R;
    std::vector<rg_ev_head_t> head[2] = {std::vector<rg_ev_head_t>(cells), std::vector<rg_ev_head_t>(cells)};
    std::vector<rg_ev_quad32_t> abcd[2] = {std::vector<rg_ev_quad32_t>(cells), std::vector<rg_ev_quad32_t>(cells)};
    std::vector<int32_t> terms[2] = {std::vector<int32_t>(4096), std::vector<int32_t>(4096)};
    Ingress ing(G, R, 1, codec, index, Ingress::Buffers{head[0].data(), abcd[0].data(), terms[0].data(), terms[0].size()},
                Ingress::Buffers{head[1].data(), abcd[1].data(), terms[1].data(), terms[1].size()}, 1u << 16, S);
    ing.set_peer(0, 0);
    remove(journal);
    StableStore store(journal);
    IngressFlusher flusher(tables, ing, codec, [&](uint32_t g) -> raftgpu::host::RaftLog & { return *logs[g]; }, std::vector<int64_t>(G, TERM), &store, true);

    uint64_t seen_rows = 0, timer_resets = 0;
    flusher.on_row = [&](uint32_t, const rg_ev_head_t &, const rg_reply_t &r) { seen_rows++; timer_resets += (r.flags & RG_F_RESET_TIMER) != 0; };
    // the leader's stream: per group  append 2 at the tail | heartbeat probing index 15 (term 3: below the cached runs) | append 1 | heartbeat that commits
    std::string stream;
    int32_t seq = 0;
    uint64_t requests = 0;
    auto ae = [&](uint32_t g, int64_t term, int64_t prev, int64_t prev_term, int n, int64_t commit) {
        Frame f;
        f.type = ENQ; f.sequence = seq++;
        f.head = make_scope(M_APPEND_ENTRIES, ids[g]);
        Request q;
        q.term = term; q.node = 0; q.x = prev; q.y = prev_term; q.leader_commit = commit;
        q.entry_terms.assign((size_t)n, term);
        codec.encode_request(M_APPEND_ENTRIES, q, f.body);
        encode_frame(f, false, stream);
        requests++;
    };
    for (int step = 0; step < 4; step++)
        for (uint32_t g = 0; g < G; g++) {
            if (g == SPECIAL) { if (step == 0) ae(g, ((int64_t)1 << 33) + TERM, LAST0, TERM, 1, LAST0); continue; }
            switch (step) {
            case 0: ae(g, TERM, LAST0, TERM, 2, LAST0); break;
            case 1: ae(g, TERM, 15, 3, 0, 0); break;
            case 2: ae(g, TERM, LAST0 + 2, TERM, 1, LAST0 + 2); break;
            case 3: ae(g, TERM, LAST0 + 3, TERM, 0, LAST0 + 3); break;
            }
        }
    for (size_t at = 0; at < stream.size(); at += 3000)
        if (ing.feed(0, reinterpret_cast<const uint8_t *>(stream.data()) + at, std::min<size_t>(3000, stream.size() - at)) < 0) return 3;
    std::vector<std::string> out(1);
    for (;;) {
        const int64_t n = flusher.flush(out);
        if (n < 0) { fprintf(stderr, "flush: %s\n", flusher.error().c_str()); return 1; }
        if (n == 0) break;
    }
    // ---- what must hold now ----------------------------------------------------------------------------------------------
    uint64_t wrong = 0, answers = 0, successes = 0;
    {
        FrameSplitter sp;
        std::vector<Frame> fs;
        sp.feed(reinterpret_cast<const uint8_t *>(out[0].data()), out[0].size(), fs);
        for (const Frame &f : fs) { Response r; if (f.type != ACK || !codec.decode_response(f.body, r)) wrong++; else { answers++; successes += r.success; } }
    }
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
    for (uint32_t s = 0; s < S; s++) {                                        // every table's groups into their place of the global columns
        const uint32_t n = std::min(PER, G - s * PER), at = s * PER;
        rg_group_state_t part = st;
        part.current_term += at; part.voted_for += at; part.role += at; part.current_leader += at; part.timeout_detected += at; part.repl_prepared += at;
        part.role_epoch += at; part.votes += at; part.elected_epoch += at; part.elected_term += at; part.commit_index += at; part.epoch_index += at;
        part.epoch_term += at; part.first_index += at; part.last_index += at; part.run_count += at; part.run_offset += at;
        part.run_start += (size_t)at * RG_TERM_RUNS; part.run_term += (size_t)at * RG_TERM_RUNS;
        part.peer_last_epoch += (size_t)at * F; part.peer_next_index += (size_t)at * F; part.peer_match_index += (size_t)at * F;
        part.peer_rejection += (size_t)at * F; part.peer_pending += (size_t)at * F;
        RG(rg_read_state(tables[s], 0, n, &part));
    }
    for (uint32_t g = 0; g < G; g++) {
        const int64_t want_last = g == SPECIAL ? LAST0 + 1 : LAST0 + 3, want_commit = g == SPECIAL ? LAST0 : LAST0 + 3;
        const int64_t want_term = g == SPECIAL ? ((int64_t)1 << 33) + TERM : TERM;
        const bool ok = last[g] == want_last && commit[g] == want_commit && term[g] == want_term && logs[g]->last() && logs[g]->last()->index == want_last &&
                        logs[g]->lastCommitted() == want_commit && flusher.term(g) == want_term && logs[g]->get(15) && logs[g]->get(15)->term == 3;
        if (!ok) { wrong++; fprintf(stderr, "group %u: table last %lld commit %lld term %lld, log last %lld commit %lld\n", g, (long long)last[g], (long long)commit[g],
                                    (long long)term[g], (long long)(logs[g]->last() ? logs[g]->last()->index : -1), (long long)logs[g]->lastCommitted()); }
    }
    const IngressFlusher::Stats &s = flusher.stats();
    int64_t jt = 0; int32_t jv = 0;
    const bool durable = store.restore(SPECIAL, &jt, &jv) && jt == ((int64_t)1 << 33) + TERM && jv == 0;
    const bool ok = wrong == 0 && answers == requests && successes == requests && s.repaired == (uint64_t)(G - 1) * 3 && s.wide == 1 && s.persisted == 1 && durable &&
                    s.appended == (uint64_t)(G - 1) * 3 + 1 && s.truncated == 0 && ing.refused() == 0 && ing.held() == 0 && seen_rows == requests &&
                    timer_resets == requests;      // (every AppendEntries a Follower accepts re-arms its election timer)
    printf("ingress flusher ok=%d: %llu rows in %llu batches, %llu repaired, %llu beside the batch, %llu entries appended, %llu commits, %llu persisted, "
           "%llu of %llu requests answered (%llu success), %llu wrong\n", (int)ok, (unsigned long long)s.rows, (unsigned long long)s.batches, (unsigned long long)s.repaired,
           (unsigned long long)s.wide, (unsigned long long)s.appended, (unsigned long long)s.committed, (unsigned long long)s.persisted, (unsigned long long)answers,
           (unsigned long long)requests, (unsigned long long)successes, (unsigned long long)wrong);
    for (rg_table_t *t : tables) rg_table_destroy(t);
    return ok ? 0 : 1;
}
