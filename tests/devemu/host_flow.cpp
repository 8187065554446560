// tests/devemu/host_flow.cpp — the C++ host mirror (rafting_amd/host/raft_host.cpp: RaftContext, ContextManager) driven
// through a complete protocol exchange between three nodes — start-up timeouts, PreVote, RequestVote, election, client
// commands, replication, commit on the leader and on the followers, a fenced late response, the readiness gate, the
// durability journal — with every decision taken by the product's device code on the HOST EMULATION library
// (tests/devemu/libraftgpu_emu.so, see hip/hip_runtime.h). TEST INFRASTRUCTURE: this is how `pytest -m "not gpu"`
// covers the host logic above the C-ABI; the same classes run on the GPU in build/cluster_sim. Timers are driven by
// hand here (rg_timers_expired needs wavefront ballots, which the emulation does not have). exit code 0 = all passed.
#include <unistd.h>

#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "raft_host.hpp"

using namespace raftgpu::host;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

struct NodeUnderTest {
    std::unique_ptr<StableStore> store;
    std::unique_ptr<ContextManager> mgr;
    RaftContext *ctx = nullptr;
    std::vector<std::pair<int64_t, ID>> persisted;     // StableLock.persist calls, in order
    std::vector<int64_t> committed;                    // RaftRoutine.commitState calls
};

static Outcome one(NodeUnderTest &n, int64_t now = -1)
{
    std::vector<Outcome> o = n.mgr->flush(now);        // now >= 0: the flush also folds timers and health statistics
    CHECK(o.size() == 1);
    return o.empty() ? Outcome{} : o[0];
}

int main()
{
    const int P = 3;
    std::vector<NodeUnderTest> node(P);
    for (int k = 0; k < P; k++) {
        const std::string journal = "/tmp/rg_host_flow_" + std::to_string((long)getpid()) + "_" + std::to_string(k) + ".journal";
        ::unlink(journal.c_str());
        node[k].store.reset(new StableStore(journal));
        node[k].mgr.reset(new ContextManager(0, 4, P, k, /*preVote*/ true));
        node[k].mgr->attachStableStore(node[k].store.get());
        node[k].mgr->onPersist([&node, k](RaftContext &, int64_t t, ID v) { node[k].persisted.push_back({t, v}); });
        node[k].mgr->onCommit([&node, k](RaftContext &, int64_t upTo) { node[k].committed.push_back(upTo); });
        node[k].ctx = &node[k].mgr->createContext("root");
        CHECK(node[k].ctx->role() == RG_FOLLOWER && node[k].ctx->currentTerm() == 0 && node[k].ctx->votedFor() == RG_NO_NODE);
    }

    // ---- everybody's election timer fires (cluster start): Follower.onTimeout with preVote -------------------------
    std::vector<Outcome> t(P);
    for (int k = 0; k < P; k++) {
        node[k].ctx->onTimeout();
        CHECK(node[k].mgr->pending(*node[k].ctx));
        t[k] = one(node[k]);
        CHECK(t[k].status == RG_OK && !t[k].response && t[k].emit() == RG_EMIT_PREVOTE && t[k].roleChanged() && t[k].resetTimer());
        CHECK(node[k].ctx->role() == RG_FOLLOWER && node[k].ctx->currentTerm() == 0 && t[k].roleEpoch == 2);
        CHECK(node[k].persisted.size() == 1 && node[k].persisted[0] == std::make_pair((int64_t)0, (ID)RG_NO_NODE));  // new participant object
    }
    // ---- node 0 campaigns: PreVote(term+1) to 1 and 2, both have seen their own timeout ----------------------------
    const uint32_t pre_election_epoch = t[0].roleEpoch;
    for (int k = 1; k < P; k++) {
        node[k].ctx->preVote(1, 0, 0, 0);
        const Outcome v = one(node[k]);
        CHECK(v.response && v.response->success && v.response->term == 0 && !v.roleChanged());      // no state change, replies currentTerm
        node[0].ctx->onVoteResponse(true, k, *v.response, pre_election_epoch);
        const Outcome r = one(node[0]);
        if (k == 1) {                                                                                 // 2 of 3: Candidate(term 1), RequestVote goes out
            CHECK(r.roleChanged() && r.role == RG_CANDIDATE && r.emit() == RG_EMIT_REQVOTE && node[0].ctx->currentTerm() == 1);
            CHECK(node[0].ctx->votedFor() == 0 && node[0].persisted.back() == std::make_pair((int64_t)1, (ID)0));
            t[0] = r;
        } else {
            CHECK(r.status == RG_DROPPED_STALE_ROLE && !r.roleChanged());                             // the pre-election's AsyncHead is fenced
        }
    }
    const uint32_t candidate_epoch = t[0].roleEpoch;
    // ---- RequestVote(1): both followers grant (log up to date, higher term) and persist the vote before answering ---
    Outcome leader_outcome;
    for (int k = 1; k < P; k++) {
        node[k].ctx->requestVote(1, 0, 0, 0);
        const Outcome v = one(node[k]);
        CHECK(v.response && v.response->success && v.response->term == 1 && v.roleChanged());
        CHECK(node[k].ctx->currentTerm() == 1 && node[k].ctx->votedFor() == 0 && node[k].persisted.back() == std::make_pair((int64_t)1, (ID)0));
        node[0].ctx->onVoteResponse(false, k, *v.response, candidate_epoch);
        const Outcome r = one(node[0]);
        if (k == 1) { CHECK(r.roleChanged() && r.role == RG_LEADER && node[0].ctx->role() == RG_LEADER); leader_outcome = r; }
        else CHECK(r.status == RG_OK && !r.roleChanged());                                            // Q13: late grant on the winner's head is a no-op
    }
    const uint32_t leader_epoch = leader_outcome.roleEpoch;
    CHECK(node[0].mgr->isReady(1000, 1, 100)[0] == 0);                                                // nothing heard from a follower yet
    // ---- two client commands, then Leader.replicateLog(false) ---------------------------------------------------------
    node[0].ctx->acceptCommand(2);
    const Outcome a = one(node[0]);
    CHECK(a.status == RG_OK && (a.flags & RG_F_LOG_APPEND) && a.emit() == RG_EMIT_HEARTBEAT);
    CHECK(node[0].ctx->replicatedLog().last() && node[0].ctx->replicatedLog().last()->index == 2 && node[0].ctx->replicatedLog().last()->term == 1);
    // prepareReplication ran after the FIRST newEntry (acceptCommand -> replicateLog per command), so nextIndex = 2: the first
    // AppendEntries probes with prevLog = entry 1, the empty followers reject, the leader backs off to the epoch and resends
    std::vector<SendPlan> plan;
    int64_t now = 2000;
    for (int round = 0; round < 4; round++, now += 10) {
        plan = node[0].mgr->replicateLog({node[0].ctx}, {0});
        CHECK(plan.size() == 1 && plan[0].head.is_leader && plan[0].head.term == 1 && plan[0].head.role_epoch == leader_epoch && plan[0].to.size() == 2);
        if (round == 0) CHECK(plan[0].to[0].kind == RG_SEND_APPEND && plan[0].to[0].prev_index == 1 && plan[0].to[0].prev_term == 1 && plan[0].to[0].count == 1);
        for (int k = 1; k < P; k++) {
            const rg_send_t &s = plan[0].to[k - 1];
            CHECK(s.kind == RG_SEND_APPEND);
            std::vector<Entry> entries;
            for (uint32_t e = 1; e <= s.count; e++) entries.push_back(*node[0].ctx->replicatedLog().get(s.prev_index + e));
            node[k].ctx->appendEntries(1, 0, s.prev_index, s.prev_term, entries, plan[0].head.leader_commit);
            const Outcome v = one(node[k]);
            CHECK(v.response && v.response->term == 1 && v.resetTimer());
            if (round == 0) CHECK(!v.response->success && !node[k].ctx->replicatedLog().last());      // prevLog (1, 1) is not there yet
            node[0].ctx->onAppendEntriesResponse(k, *v.response, plan[0].head.epoch_index, s.last_index, leader_epoch);
            const Outcome r = one(node[0], now);
            CHECK(r.status == RG_OK);
            if (r.flags & RG_F_COMMIT) CHECK(node[0].committed.back() == 2 && node[0].ctx->replicatedLog().lastCommitted() == 2);
        }
        if (round == 0) CHECK(node[0].mgr->isReady(now, 1, 100)[0] == 1);      // a rejection is a statSuccess too: both followers are alive
    }
    for (int k = 1; k < P; k++) CHECK(node[k].ctx->replicatedLog().last() && node[k].ctx->replicatedLog().last()->index == 2);
    CHECK(node[0].committed.size() == 1 && node[0].committed[0] == 2);         // F = 2: sorted[1] = the larger matchIndex
    std::vector<PeerProgress> pr = node[0].mgr->progress(*node[0].ctx);
    CHECK(pr.size() == 2 && pr[0].matchIndex == 2 && pr[0].nextIndex == 3 && pr[1].matchIndex == 2 && !pr[1].pendingInstallation);
    // ---- heartbeat: the followers learn the commit index -------------------------------------------------------------
    node[0].ctx->onTimeout();
    CHECK(one(node[0]).emit() == RG_EMIT_HEARTBEAT);
    plan = node[0].mgr->replicateLog({node[0].ctx}, {1});
    CHECK(plan[0].head.leader_commit == 2 && plan[0].to[0].prev_index == 2 && plan[0].to[0].prev_term == 1 && plan[0].to[0].count == 0);
    for (int k = 1; k < P; k++) {
        node[k].ctx->appendEntries(1, 0, 2, 1, {}, 2);
        const Outcome v = one(node[k]);
        CHECK(v.response && v.response->success && v.resetTimer());
        // leaderCommit 2 reached the follower either with a late round of the pump above or now — exactly once (markCommitted(==) is silent)
        CHECK(node[k].committed.size() == 1 && node[k].committed[0] == 2 && node[k].ctx->replicatedLog().lastCommitted() == 2);
    }
    // ---- a response that belongs to the candidacy arrives at the Leader: fenced -----------------------------------------
    node[0].ctx->onAppendEntriesResponse(1, {1, true}, 0, 2, candidate_epoch);
    CHECK(one(node[0]).status == RG_DROPPED_STALE_ROLE);
    // ---- RPC failures take a follower out of the readiness count, one success brings it back ---------------------------
    node[0].mgr->statFailure({{node[0].ctx, 1, true, false}, {node[0].ctx, 2, true, false}, {node[0].ctx, 1, true, false}, {node[0].ctx, 2, true, false}}, 5000);
    CHECK(node[0].mgr->isReady(9000, 1, 100)[0] == 0);                                                // recentFailure 2 > criticalPoint 1 on both
    node[0].ctx->onAppendEntriesResponse(2, {1, true}, 0, 2, leader_epoch);
    one(node[0]);                                                                                     // flush(now = -1): statistics not folded
    CHECK(node[0].mgr->isReady(9000, 1, 100)[0] == 0);
    node[0].ctx->onAppendEntriesResponse(2, {1, true}, 0, 2, leader_epoch);
    CHECK(node[0].mgr->flush(9000).size() == 1);                                                      // statSuccess clears recentFailure of peer 2
    CHECK(node[0].mgr->isReady(9001, 1, 100)[0] == 1);                                                // self + peer 2 > 2 / 2
    CHECK(node[0].mgr->isReady(9001, 1, 100000)[0] == 0);                                             // a long cool-down still remembers the failure at 5000
    // ---- a higher term in a response: the Leader steps down, votedFor = responder (Q7), persisted before anything else --
    node[0].ctx->onAppendEntriesResponse(1, {4, false}, 0, 2, leader_epoch);
    const Outcome down = one(node[0]);
    CHECK(down.roleChanged() && down.role == RG_FOLLOWER && node[0].ctx->currentTerm() == 4 && node[0].ctx->votedFor() == 1);
    CHECK(node[0].persisted.back() == std::make_pair((int64_t)4, (ID)1) && node[0].mgr->isReady(9002, 1, 100)[0] == 0);
    node[0].ctx->acceptCommand(1);
    CHECK(one(node[0]).status == RG_NOT_LEADER);
    // ---- the journal holds what every participant believes; a restarted node restores it --------------------------------
    for (int k = 0; k < P; k++) {
        int64_t term = -1; int32_t vote = -7;
        CHECK(node[k].store->restore(node[k].ctx->gid(), &term, &vote) && term == node[k].ctx->currentTerm() && vote == node[k].ctx->votedFor());
        CHECK(node[k].store->records() == node[k].persisted.size());
    }
    {
        ContextManager again(0, 4, P, 0, true);
        again.attachStableStore(node[0].store.get());
        RaftContext &c = again.createContext("root");
        CHECK(c.currentTerm() == 4 && c.votedFor() == 1 && c.role() == RG_FOLLOWER);
    }
    for (int k = 0; k < P; k++) ::unlink(("/tmp/rg_host_flow_" + std::to_string((long)getpid()) + "_" + std::to_string(k) + ".journal").c_str());
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("host flow ok: %llu rows decided\n", (unsigned long long)(node[0].mgr->rowsDecided() + node[1].mgr->rowsDecided() + node[2].mgr->rowsDecided()));
    return 0;
}
