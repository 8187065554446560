// cluster_sim.cpp — BASELINE.json configs[0] on the GPU path: the reference's own demo
// (src/test/java/.../cluster/TestNode1-3.java + cmd/FileMachine.java: three nodes, context "root", a client
// appending lines through the leader, humans killing / restarting nodes and diffing the three output files)
// as ONE deterministic in-process simulation.  Each node is a raftgpu::host::ContextManager (its own
// rg_table on the GPU) holding `groups` RaftContexts; RPCs travel through an in-memory network with one tick
// (50 ms) of delay; heartbeat 300 ms and election timeout 900 ms randomised in [E, 2E] as in raft1.xml, and the timers
// themselves live on the device (rg_timers_*, N4): the host only asks which tickets have fired.  Every decision —
// vote, append, commit, role change, what to replicate, when to time out — is taken by the HIP kernels; the host
// moves messages, keeps the RaftLog, the durability journal and the FileMachine.
//
// What is asserted (the reference checks "eventual file equality by eye", README.md:28-33; we check more):
//   * election safety: never two leaders in one term of one group;
//   * state-machine safety: the FileMachine files of all nodes agree line by line on their common prefix,
//     at every tick;
//   * liveness: commands keep committing, also after the leader is partitioned away and after it rejoins;
//   * convergence: once traffic stops, all three files are identical.
// usage: cluster_sim [groups=1] [ticks=2400 (50 ms each)] [seed=1]   exit code 0 = all invariants held
#include <unistd.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <random>
#include <set>
#include <string>

#include "raft_host.hpp"
#include "wire.hpp"

using namespace raftgpu::host;
namespace rw = rafting::wire;

enum MsgType { AE, AE_RESP, PV, PV_RESP, RV, RV_RESP, IS, IS_RESP };
struct Msg {
    int from = -1, to = -1;
    uint32_t gid;
    MsgType type;
    int64_t term = 0, x = 0, y = 0, z = 0;      // AE: prevIndex, prevTerm, leaderCommit; votes: lastIndex, lastTerm
    std::vector<Entry> entries;
    bool success = false;
    uint32_t epoch = 0;                           // role epoch of the requester when it sent the request
    int64_t epochAtSend = 0, lastSent = 0;        // Leader.replicateLog closure state echoed by the response
    int64_t sentTick = 0;                         // when the request left (echoed): pairs a response with its Async
    int32_t seq = 0;                              // wire mode: the sequence number of the Ping / Pong frame pair
};

static const int P = 3;
static const int64_t HEARTBEAT_MS = 300, ELECTION_MS = 900;     // raft1.xml:10-13: tick 300 ms, heartbeat x1, election x3

struct Group {
    RaftContext *ctx = nullptr;
    std::deque<Msg> inbox;
    bool timer_due = false;                       // the device reported this context's ticket as fired
    uint32_t timer_epoch = 0;                     // ... for the participant of this role epoch (RG_EV_TIMEOUT.aux)
    std::vector<std::string> file;                // FileMachine: "<index>:<line>"
    int64_t applied = 0;
    int inflight[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // State.requestInFlight per peer
    std::deque<int64_t> outstanding[8];           // send ticks of the AppendEntries still waiting for a response or a timeout
};

struct Node {
    int id;
    std::unique_ptr<StableStore> store;           // N3: one journal per node instead of one StableLock file per context
    std::unique_ptr<ContextManager> mgr;
    std::vector<Group> g;
    bool connected = true;
};

static std::mt19937_64 rng;
static int64_t now_tick = 0;
static std::vector<Msg> wire, wire_next;
static std::map<std::pair<uint32_t, int64_t>, int> leader_of_term;
static int violations = 0;

static const int64_t TICK_MS = 50;                // simulation step = network delay; timers keep millisecond deadlines
static const int64_t BROADCAST_TICKS = 3;         // broadcast timeout = 0.5 x 300 ms (raft1.xml:13) = 3 steps; a round trip takes 2
static const int32_t CRITICAL_POINT = 1;          // raft1.xml:30 avail-critical-point
static const int64_t COOL_DOWN_MS = 100;          // raft1.xml:31 recovery-cool-down
static void fail(const char *what, uint32_t gid) { fprintf(stderr, "INVARIANT VIOLATED (tick %lld, group %u): %s\n", (long long)now_tick, gid, what); violations++; }
static std::string line_of(const Entry &e) { return "t" + std::to_string(e.term) + "-cmd" + std::to_string(e.index); }

// ---- SIM_WIRE=1: every message crosses the "network" as an encoded frame of the reference's wire protocol ---------------
// (transport/EventCodec.java): the sender encodes (bodies in the reference's Kryo format, KryoBodyCodec; SIM_WIRE=2 takes the fixed-layout
// test codec instead), the bytes sit in a per-connection
// stream, the receiver's FrameSplitter gets them in random pieces and the decoded frame is turned back into the message.
// A response carries only (scope, sequence, term, success): what the requester needs beyond that — the role epoch it sent
// under and Leader.replicateLog's closure state — comes from ITS OWN table of pending invocations, as in AsyncService.
static bool use_wire = false;
struct Conn { std::string bytes; rw::FrameSplitter splitter; };
static Conn conn[8][8];                                       // [from][to]
struct PendingCall { uint32_t gid; MsgType type; uint32_t epoch; int64_t epochAtSend, lastSent, sentTick; };
static std::map<int32_t, PendingCall> pending[8][8];          // [requester][responder] by sequence
static int32_t next_seq[8];
static uint64_t wire_bytes = 0, wire_frames = 0;
static std::vector<std::pair<int, int>> send_order;           // the network keeps the order the frames were written in
static std::mt19937_64 chunk_rng(20240921);                   // piece sizes: their own stream, the simulation's draws stay as they are

static std::string ctx_name(uint32_t groups, uint32_t gid) { return groups == 1 ? "root" : "ctx-" + std::to_string(gid); }
static uint32_t g_groups = 1;

// the demo cluster of src/test/resources/raft{1,2,3}.xml: 127.0.0.1:6001-6003 (slots 0..2; further slots continue the ports)
static const rw::BodyCodec &body_codec()
{
    static const rw::FixedBodyCodec fixed;
    static const rw::KryoBodyCodec kryo([] { std::vector<rw::KryoBodyCodec::Node> n; for (int i = 0; i < 8; i++) n.push_back({"127.0.0.1", 6001 + i}); return n; }());
    static const bool use_fixed = getenv("SIM_WIRE") && atoi(getenv("SIM_WIRE")) == 2;
    return use_fixed ? static_cast<const rw::BodyCodec &>(fixed) : kryo;
}

static void encode_msg(Msg &m)
{
    const rw::BodyCodec &codec = body_codec();
    rw::Frame f;
    const bool request = m.type == AE || m.type == PV || m.type == RV || m.type == IS;
    const rw::Method method = (m.type == AE || m.type == AE_RESP) ? rw::M_APPEND_ENTRIES : (m.type == PV || m.type == PV_RESP) ? rw::M_PRE_VOTE
                            : (m.type == RV || m.type == RV_RESP) ? rw::M_REQUEST_VOTE : rw::M_INSTALL_SNAPSHOT;
    f.head = rw::make_scope(method, ctx_name(g_groups, m.gid));
    if (request) {
        m.seq = next_seq[m.from]++;
        pending[m.from][m.to][m.seq] = PendingCall{m.gid, m.type, m.epoch, m.epochAtSend, m.lastSent, m.sentTick};
        rw::Request q;
        q.term = m.term; q.node = m.from; q.x = m.x; q.y = m.y; q.leader_commit = m.z;
        for (const Entry &e : m.entries) q.entry_terms.push_back(e.term);
        codec.encode_request(method, q, f.body);
        f.type = rw::ENQ;
    } else {
        codec.encode_response(rw::Response{m.term, m.success}, f.body);
        f.type = rw::ACK;
    }
    f.sequence = m.seq;
    const size_t before = conn[m.from][m.to].bytes.size();
    rw::encode_frame(f, false, conn[m.from][m.to].bytes);
    wire_bytes += conn[m.from][m.to].bytes.size() - before;
    wire_frames++;
    send_order.emplace_back(m.from, m.to);
}

// one tick of the network: every connection delivers what was written to it, in arbitrary pieces
static void deliver_frames(std::vector<Msg> &out)
{
    const rw::BodyCodec &codec = body_codec();
    std::deque<std::pair<bool, Msg>> arrived[8][8];             // per connection, in order; false = decoded but dropped
    for (int from = 0; from < P; from++)
        for (int to = 0; to < P; to++) {
            Conn &c = conn[from][to];
            std::vector<rw::Frame> frames;
            size_t at = 0;
            while (at < c.bytes.size()) {
                const size_t piece = std::min<size_t>(c.bytes.size() - at, 1 + chunk_rng() % 97);
                c.splitter.feed(reinterpret_cast<const uint8_t *>(c.bytes.data()) + at, piece, frames);
                at += piece;
            }
            c.bytes.clear();
            if (c.splitter.failed()) { fail("a frame stream was rejected by the splitter", 0); return; }
            for (const rw::Frame &f : frames) {
                rw::Method method; std::string ctx;
                if (!rw::parse_scope(f.head, method, ctx)) { fail("unknown scope on the wire", 0); arrived[from][to].emplace_back(false, Msg{}); continue; }
                Msg m; m.from = from; m.to = to; m.seq = f.sequence;
                m.gid = g_groups == 1 ? 0u : (uint32_t)atoi(ctx.c_str() + 4);
                if (f.type == rw::ENQ) {
                    rw::Request q;
                    if (!codec.decode_request(method, f.body, q)) { fail("undecodable request body", m.gid); arrived[from][to].emplace_back(false, Msg{}); continue; }
                    m.type = method == rw::M_APPEND_ENTRIES ? AE : method == rw::M_PRE_VOTE ? PV : method == rw::M_REQUEST_VOTE ? RV : IS;
                    m.term = q.term; m.x = q.x; m.y = q.y; m.z = q.leader_commit;
                    for (size_t k = 0; k < q.entry_terms.size(); k++) m.entries.push_back(Entry{q.x + 1 + (int64_t)k, q.entry_terms[k]});
                } else {
                    rw::Response r;
                    if (!codec.decode_response(f.body, r)) { fail("undecodable response body", m.gid); arrived[from][to].emplace_back(false, Msg{}); continue; }
                    auto &tab = pending[to][from];                  // the requester is the node the response arrives at
                    auto it = tab.find(f.sequence);
                    if (it == tab.end()) { arrived[from][to].emplace_back(false, Msg{}); continue; }   // AsyncService: no invocation waiting under that sequence
                    const PendingCall pc = it->second;
                    tab.erase(it);
                    m.type = pc.type == AE ? AE_RESP : pc.type == PV ? PV_RESP : pc.type == RV ? RV_RESP : IS_RESP;
                    m.term = r.term; m.success = r.success;
                    m.epoch = pc.epoch; m.epochAtSend = pc.epochAtSend; m.lastSent = pc.lastSent; m.sentTick = pc.sentTick;
                }
                arrived[from][to].emplace_back(true, std::move(m));
            }
        }
    for (auto &ft : send_order) {
        auto &q = arrived[ft.first][ft.second];
        if (q.empty()) { fail("a frame was lost between encoder and splitter", 0); continue; }
        if (q.front().first) out.push_back(std::move(q.front().second));
        q.pop_front();
    }
    send_order.clear();
}

static void send(Node &n, Msg m)
{
    m.from = n.id;
    if (!n.connected) return;
    if (use_wire) encode_msg(m); else wire_next.push_back(std::move(m));
}

// Leader.replicateLog (member/Leader.java:142-245): WHAT to send is decided on the GPU (rg_replicate, one launch for
// all leader contexts of the node that emitted a heartbeat this drain); the host only reads the payload range
static void replicate(Node &n, std::vector<Group *> &leaders, const std::vector<uint8_t> &hb)
{
    if (leaders.empty()) return;
    const size_t F = P - 1;
    std::vector<RaftContext *> ctxs;
    std::vector<uint16_t> fl;
    for (Group *g : leaders) {
        ctxs.push_back(g->ctx);
        for (int peer = 0; peer < P; peer++) if (peer != n.id) fl.push_back((uint16_t)g->inflight[peer]);
    }
    std::vector<SendPlan> plans = n.mgr->replicateLog(ctxs, hb, fl);
    for (size_t i = 0; i < leaders.size(); i++) {
        Group &g = *leaders[i];
        RaftContext &c = *g.ctx;
        const SendPlan &pl = plans[i];
        if (!pl.head.is_leader) continue;
        for (size_t j = 0; j < F; j++) {
            const rg_send_t &s = pl.to[j];
            const int peer = (int)j < n.id ? (int)j : (int)j + 1;
            if (s.kind == RG_SEND_GATED || s.kind == RG_SEND_NONE) continue;
            Msg m; m.to = peer; m.gid = c.gid(); m.term = pl.head.term; m.epoch = pl.head.role_epoch; m.epochAtSend = pl.head.epoch_index;
            if (s.kind == RG_SEND_SNAPSHOT) {
                m.type = IS; m.x = pl.head.epoch_index; m.y = pl.head.epoch_term;
            } else {
                m.type = AE; m.x = s.prev_index; m.y = s.prev_term; m.z = pl.head.leader_commit; m.lastSent = s.last_index;
                for (uint32_t k = 1; k <= s.count; k++) m.entries.push_back(*c.replicatedLog().get(s.prev_index + k));
                m.sentTick = now_tick;
                g.inflight[peer]++;
                g.outstanding[peer].push_back(now_tick);
            }
            send(n, std::move(m));
        }
    }
}

static void broadcast_vote(Node &n, Group &g, bool pre, uint32_t epoch)
{
    RaftContext &c = *g.ctx;
    auto last = c.replicatedLog().last();
    const Entry l = last ? *last : c.replicatedLog().epoch();
    for (int peer = 0; peer < P; peer++) {
        if (peer == n.id) continue;
        Msg m; m.to = peer; m.gid = c.gid(); m.type = pre ? PV : RV; m.epoch = epoch;
        m.term = pre ? c.currentTerm() + 1 : c.currentTerm(); m.x = l.index; m.y = l.term;
        send(n, std::move(m));
    }
}

int main(int argc, char **argv)
{
    const uint32_t groups = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
    const int64_t ticks = argc > 2 ? atoll(argv[2]) : 2400;
    rng.seed(argc > 3 ? (uint64_t)atoll(argv[3]) : 1);
    const int trace_group = getenv("SIM_TRACE_GROUP") ? atoi(getenv("SIM_TRACE_GROUP")) : -1;
    use_wire = getenv("SIM_WIRE") && atoi(getenv("SIM_WIRE")) != 0;
    g_groups = groups;
    std::vector<Node> nodes(P);
    for (int k = 0; k < P; k++) {
        nodes[k].id = k;
        const std::string journal = "/tmp/rg_cluster_sim_" + std::to_string((long)getpid()) + "_node" + std::to_string(k) + ".journal";
        ::unlink(journal.c_str());
        nodes[k].store.reset(new StableStore(journal));
        nodes[k].mgr.reset(new ContextManager(0, groups, P, k, true));
        nodes[k].mgr->attachStableStore(nodes[k].store.get());
        nodes[k].g.resize(groups);
        for (uint32_t i = 0; i < groups; i++) {
            Group &g = nodes[k].g[i];
            g.ctx = &nodes[k].mgr->createContext(groups == 1 ? "root" : "ctx-" + std::to_string(i));
        }
        nodes[k].mgr->configureTimers(ELECTION_MS, HEARTBEAT_MS, rng());
        nodes[k].mgr->armTimers(0);
        nodes[k].mgr->onCommit([&nodes, k](RaftContext &c, int64_t upTo) {     // FileMachine.apply (cmd/FileMachine.java:62-84)
            Group &g = nodes[k].g[c.gid()];
            for (; g.applied < upTo; g.applied++) {
                auto e = c.replicatedLog().get(g.applied + 1);
                if (!e) { fail("commit beyond the stored log", c.gid()); return; }
                g.file.push_back(std::to_string(e->index) + ":" + line_of(*e));
            }
        });
    }
    const int64_t cut_at = ticks / 3, heal_at = 2 * ticks / 3, quiet_at = ticks - 200;
    int cut_node = -1;
    uint64_t commands = 0, elections = 0, rollbacks = 0, not_ready = 0, rpc_timeouts = 0;

    for (now_tick = 0; now_tick < ticks; now_tick++) {
        if (use_wire) deliver_frames(wire);              // what was sent during the previous tick arrives, frame by frame
        for (Msg &m : wire) {
            if (!(nodes[m.to].connected && nodes[m.from].connected)) continue;
            Group &g = nodes[m.to].g[m.gid];
            if (m.type == AE_RESP) {                    // completes its Async unless that already ended in a timeout / abort
                auto &o = g.outstanding[m.from];
                auto it = std::find(o.begin(), o.end(), m.sentTick);
                if (it == o.end()) continue;
                o.erase(it);
            }
            g.inbox.push_back(std::move(m));
        }
        wire.clear();
        if (now_tick == cut_at) {                       // "a human kills the leader's process"
            for (Node &n : nodes) if (n.g[0].ctx->role() == RG_LEADER) cut_node = n.id;
            if (cut_node >= 0) nodes[cut_node].connected = false;
        }
        if (now_tick == heal_at && cut_node >= 0) nodes[cut_node].connected = true;

        for (Node &n : nodes) {
          std::vector<char> commanded(groups, 0);
          // broadcast timeout = 0.5 tick (raft1.xml:13): every request still unanswered a tick later has
          // completed with a timeout error, which releases its in-flight slot (Leader.java:221,235)
          const int64_t now_ms = now_tick * TICK_MS;
          std::vector<ContextManager::RpcFailure> timed_out;
          for (Group &g : n.g)                           // requests unanswered for the broadcast timeout complete with an error:
              for (int peer = 0; peer < P; peer++) {     // the callback frees the in-flight slot and calls statFailure (Leader.java:221,235)
                  auto &o = g.outstanding[peer];
                  while (!o.empty() && o.front() + BROADCAST_TICKS <= now_tick) {
                      o.pop_front();
                      if (g.inflight[peer] > 0) g.inflight[peer]--;
                      timed_out.push_back({g.ctx, peer, true, false});
                      rpc_timeouts++;
                  }
              }
          n.mgr->statFailure(timed_out, now_ms);
          // RaftStub.process: a Leader takes commands only while isReady (command/RaftStub.java:80-87), raft1.xml:30-31
          const std::vector<uint8_t> ready = n.mgr->isReady(now_ms, CRITICAL_POINT, COOL_DOWN_MS);
          for (auto &tk : n.mgr->expiredTickets(now_ms)) {                                       // electionTimeout / keepAlive fired
              n.g[tk.first->gid()].timer_due = true;
              n.g[tk.first->gid()].timer_epoch = tk.second;
          }
          for (int sub = 0; sub < 8; sub++) {             // several EventLoop drains per tick: one row per context each
            struct Src { Group *g; Msg msg; int what; };   // what: 0 message, 1 timer, 2 client command
            std::vector<Src> src;
            for (Group &g : n.g) {
                RaftContext &c = *g.ctx;
                if (g.timer_due) {                                  // timers are urgent (EventLoop.execute(evt, true))
                    c.onTimeout(g.timer_epoch);                     // dropped if a row replaced that participant meanwhile
                    g.timer_due = false;
                    src.push_back({&g, Msg{}, 1});
                } else if (!g.inbox.empty()) {
                    Msg m = std::move(g.inbox.front()); g.inbox.pop_front();
                    switch (m.type) {
                    // Follower.installSnapshot (member/Follower.java:129-152): download + RaftLog.flush are host work; no compaction
                    // happens in this demo, so the snapshot at epoch (0,0) is empty and "installs" at once — the decision is the row's
                    case IS: c.installSnapshot(m.term, m.from, m.x, m.y, true); break;
                    case AE: c.appendEntries(m.term, m.from, m.x, m.y, m.entries, m.z); break;
                    case PV: c.preVote(m.term, m.from, m.x, m.y); break;
                    case RV: c.requestVote(m.term, m.from, m.x, m.y); break;
                    case AE_RESP:
                        if (g.inflight[m.from] > 0) g.inflight[m.from]--;
                        c.onAppendEntriesResponse(m.from, {m.term, m.success}, m.epochAtSend, m.lastSent, m.epoch);
                        break;
                    case PV_RESP: c.onVoteResponse(true, m.from, {m.term, m.success}, m.epoch); break;
                    case RV_RESP: c.onVoteResponse(false, m.from, {m.term, m.success}, m.epoch); break;
                    case IS_RESP: c.onInstallSnapshotResponse(m.from, {m.term, m.success}, m.epochAtSend, m.epoch); break;
                    }
                    src.push_back({&g, std::move(m), 0});
                } else if (c.role() == RG_LEADER && now_tick < quiet_at && !commanded[c.gid()] && (now_tick + c.gid()) % 4 == 0) {
                    commanded[c.gid()] = 1;
                    if (!ready[c.gid()]) { not_ready++; continue; }  // NotReadyException
                    c.acceptCommand(1);                             // TestNode: submit(AppendCommand(...))
                    commanded[c.gid()] = 1;
                    src.push_back({&g, Msg{}, 2});
                }
            }
            if (src.empty()) break;
            std::vector<Outcome> out = n.mgr->flush(now_ms);    // also folds RESET_TIMER / ROLE_CHANGED into the device timers
            std::vector<Group *> to_replicate;
            std::vector<uint8_t> heartbeat;                     // replicateLog(true) from onTimeout, (false) from acceptCommand
            for (size_t i = 0; i < out.size(); i++) {
                Group &g = *src[i].g;
                RaftContext &c = *g.ctx;
                const Outcome &o = out[i];
                if (o.status == RG_A_MATCH_ROLLBACK) {
                    // reachable in the reference too: pipelined AppendEntries whose 50-entry window moved DOWN after a
                    // late rejection are acknowledged after a longer one (Leadership.java:76-81 throws, the callback dies)
                    rollbacks++;
                } else if (o.status != RG_OK && o.status != RG_DROPPED_STALE_ROLE && o.status != RG_NOT_LEADER) {
                    if (o.status == RG_A_INSTALL_BEFORE_AE) { fail("a leader sent InstallSnapshot before AppendEntries", c.gid()); }
                    fprintf(stderr, "tick %lld node %d group %u: status %u\n", (long long)now_tick, n.id, c.gid(), o.status);
                    if (o.status < RG_NPE_MAJOR_NULL) fail("an AssertionError site of the reference was reached", c.gid());
                }
                if (src[i].what == 0 && o.response) {
                    const Msg &q = src[i].msg;
                    Msg r; r.to = q.from; r.gid = q.gid; r.term = o.response->term; r.success = o.response->success;
                    r.epoch = q.epoch; r.epochAtSend = q.epochAtSend; r.lastSent = q.lastSent; r.sentTick = q.sentTick; r.seq = q.seq;
                    r.type = q.type == AE ? AE_RESP : q.type == PV ? PV_RESP : q.type == IS ? IS_RESP : RV_RESP;
                    send(n, std::move(r));
                }
                if (src[i].what == 2 && o.status == RG_OK) commands++;
                if (trace_group >= 0 && (int)c.gid() == trace_group && (o.roleChanged() || o.status != RG_OK || src[i].what != 0 || (src[i].msg.type != AE && src[i].msg.type != AE_RESP)))
                    fprintf(stderr, "trace tick %lld node %d: src=%d msgtype=%d from=%d mterm=%lld -> role=%d term=%lld epoch=%u status=%u flags=%x resp=%d/%lld last=%lld commit=%lld\n",
                            (long long)now_tick, n.id, src[i].what, (int)src[i].msg.type, src[i].msg.from, (long long)src[i].msg.term, o.role,
                            (long long)c.currentTerm(), o.roleEpoch, o.status, o.flags, o.response ? (int)o.response->success : -1,
                            (long long)(o.response ? o.response->term : -1), (long long)(c.replicatedLog().last() ? c.replicatedLog().last()->index : -1),
                            (long long)c.replicatedLog().lastCommitted());
                if (o.roleChanged()) {                                          // AsyncHead.abortRequests
                    for (int &x : g.inflight) x = 0;
                    for (auto &q : g.outstanding) q.clear();
                }
                if (o.roleChanged() && o.role == RG_LEADER) {
                    elections++;
                    auto key = std::make_pair(c.gid(), c.currentTerm());
                    if (leader_of_term.count(key) && leader_of_term[key] != n.id) fail("two leaders in one term", c.gid());
                    leader_of_term[key] = n.id;
                }
                if (o.emit() == RG_EMIT_PREVOTE) broadcast_vote(n, g, true, o.roleEpoch);
                else if (o.emit() == RG_EMIT_REQVOTE) broadcast_vote(n, g, false, o.roleEpoch);
                else if (o.emit() == RG_EMIT_HEARTBEAT) { to_replicate.push_back(&g); heartbeat.push_back(src[i].what != 2); }
            }
            replicate(n, to_replicate, heartbeat);
          }
        }
        // state-machine safety, every tick
        for (uint32_t i = 0; i < groups; i++)
            for (int a = 0; a < P; a++)
                for (int b = a + 1; b < P; b++) {
                    const auto &fa = nodes[a].g[i].file, &fb = nodes[b].g[i].file;
                    for (size_t k = 0; k < std::min(fa.size(), fb.size()); k++)
                        if (fa[k] != fb[k]) { fail("FileMachine files diverge", i); k = fa.size(); }
                }
        wire.swap(wire_next);
    }

    // Safety held at every tick for every group (checked above). Liveness is statistical: the reference can
    // livelock a group for a while (a Candidate grants ANY higher-term vote — Candidate.java:69-71 — so a node
    // with a short log can keep disrupting), hence convergence is required of >= 99 % of the groups.
    size_t min_lines = SIZE_MAX, max_lines = 0, converged = 0;
    std::vector<size_t> per_group;
    for (uint32_t i = 0; i < groups; i++) {
        bool same = true;
        size_t lo = SIZE_MAX;
        for (int k = 0; k < P; k++) {
            lo = std::min(lo, nodes[k].g[i].file.size());
            max_lines = std::max(max_lines, nodes[k].g[i].file.size());
            same = same && nodes[k].g[i].file == nodes[0].g[i].file;
        }
        converged += same;
        min_lines = std::min(min_lines, lo);
        per_group.push_back(lo);
    }
    std::sort(per_group.begin(), per_group.end());
    const size_t median_lines = per_group[per_group.size() / 2];
    const bool identical = converged == groups;
    uint64_t rows = 0, hints = 0, syncs = 0, persisted = 0;
    for (Node &n : nodes) {
        rows += n.mgr->rowsDecided(); hints += n.mgr->hintsServed();
        syncs += n.store->syncs(); persisted += n.store->records();
        for (Group &g : n.g) {                           // what a restart would restore == what the participant believes
            int64_t t = 0; int32_t v = RG_NO_NODE;
            const bool have = n.store->restore(g.ctx->gid(), &t, &v);
            if (have ? (t != g.ctx->currentTerm() || v != g.ctx->votedFor()) : g.ctx->currentTerm() != 0)
                fail("journal does not hold the participant's (term, votedFor)", g.ctx->gid());
        }
    }
    if (use_wire) fprintf(stderr, "wire: %llu frames, %llu bytes through FrameSplitter / %s\n", (unsigned long long)wire_frames, (unsigned long long)wire_bytes, (getenv("SIM_WIRE") && atoi(getenv("SIM_WIRE")) == 2) ? "FixedBodyCodec" : "KryoBodyCodec");
    fprintf(stderr, "readiness gate: %llu commands refused (NotReadyException), %llu RPC timeouts\n", (unsigned long long)not_ready,
            (unsigned long long)rpc_timeouts);
    fprintf(stderr, "durability: %llu (term, votedFor) records in %llu fdatasyncs\n", (unsigned long long)persisted, (unsigned long long)syncs);
    for (uint32_t i = 0; i < groups && !identical && i < 4096; i++) {
        bool same = true;
        for (int k = 1; k < P; k++) same = same && nodes[k].g[i].file == nodes[0].g[i].file;
        if (!same) {
            fprintf(stderr, "group %u:", i);
            for (int k = 0; k < P; k++) {
                RaftContext &c = *nodes[k].g[i].ctx;
                auto l = c.replicatedLog().last();
                fprintf(stderr, "  node%d role=%d term=%lld last=%lld commit=%lld lines=%zu inbox=%zu", k, c.role(), (long long)c.currentTerm(),
                        (long long)(l ? l->index : -1), (long long)c.replicatedLog().lastCommitted(), nodes[k].g[i].file.size(), nodes[k].g[i].inbox.size());
            }
            fprintf(stderr, "\n");
        }
    }
    printf("groups=%u ticks=%lld commands_accepted=%llu elections=%llu partitioned_node=%d lines(min,max)=(%zu,%zu) "
           "files_identical=%d gpu_rows=%llu hints=%llu match_rollbacks=%llu violations=%d converged=%zu median_lines=%zu\n",
           groups, (long long)ticks, (unsigned long long)commands, (unsigned long long)elections, cut_node, min_lines, max_lines,
           (int)identical, (unsigned long long)rows, (unsigned long long)hints, (unsigned long long)rollbacks, violations, converged,
           median_lines);
    for (int k = 0; k < P; k++) ::unlink(("/tmp/rg_cluster_sim_" + std::to_string((long)getpid()) + "_node" + std::to_string(k) + ".journal").c_str());
    if (converged * 100 < (size_t)groups * 99) fail("files did not converge after traffic stopped", 0);
    if (median_lines < (size_t)(ticks / 16)) fail("too little progress", 0);
    if (elections < (groups == 1 ? 2u : groups)) fail("no (re-)election happened", 0);
    return violations ? 1 : 0;
}
