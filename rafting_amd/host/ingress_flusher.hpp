// ingress_flusher.hpp — the flush thread's loop around an Ingress as ONE call: what INTEGRATION.md §1 / §3 describe, with the host-owned
// plugins attached. seal -> decide on the table -> apply the log effects of the rows the launch applied -> repair RG_NEED_HOST rows from the
// host's RaftLog (their effects applied one by one: the next hint reads the log as the last row left it) -> the rows the compact format
// could not hold (one sparse submit each, same protocol) -> ONE durable write of every (term, votedFor) the batch changed -> only then the
// response frames -> recycle. The order inside a row is the reference handler's (truncate -> append -> persist -> commit -> reply:
// member/Follower.java:35-88, member/RaftMember.java:25, context/RaftContext.java:244-255).
//
//   RaftLog      raft_host.hpp's interface (command/RaftLog.java:72-132): one per group, owned by the caller
//   StableStore  N3: optional
// The request bodies are retained by the ingress (Ingress::retain_bodies), the entries a row appends are read from its own request.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "ingress.hpp"
#include "raft_host.hpp"

namespace rafting {
namespace wire {

class IngressFlusher {
public:
    struct Stats { uint64_t batches = 0, rows = 0, repaired = 0, wide = 0, frames = 0, persisted = 0, appended = 0, truncated = 0, committed = 0; };
    // log_of(gid) -> the group's RaftLog; term_of_group[gid] = its currentTerm (kept current from the rows' persist records: a client append
    // creates entries of that term, member/Leader.java:128-140). wide_kernel: decide the batch through rg_submit on an unpacked copy instead
    // of rg_submit32 (tests on the lane-serial emulation of the kernels, which cannot run the compact-row kernel).
    // NOTE: the constructor turns RG_OPT_REQUIRE_FENCED_TIMEOUTS on for every table it is given (the ordering contract of INTEGRATION.md section 1) and
    // throws when a table refuses; recordings of the once-per-tick paths (rg_tick_*, rg_tick2_*) on those tables must be created afterwards.
    IngressFlusher(rg_table_t *table, Ingress &ing, const KryoBodyCodec &codec, std::function<raftgpu::host::RaftLog &(uint32_t)> log_of,
                   std::vector<int64_t> term_of_group, raftgpu::host::StableStore *store = nullptr, bool wide_kernel = false);
    // An ingress in front of SEVERAL tables (Ingress's `shards`: block partition, one table per GPU — what support/EventLoopGroup.java:77-80's
    // loop-per-context becomes with a table per device; VERDICT r3 "missing" 5): tables[s] decides shard s. ONE seal, the shards' launches side by
    // side (a thread per table beyond the first: tables are independent, each has its own stream), then per shard the same apply -> repair
    // order, the rows beside the batch on the table their group lives in, ONE durable write for all shards, only then any response frame.
    // log_of / term_of_group / on_row speak GLOBAL group ids (shard s, table group g = first_gid(s) + g).
    IngressFlusher(std::vector<rg_table_t *> tables, Ingress &ing, const KryoBodyCodec &codec, std::function<raftgpu::host::RaftLog &(uint32_t)> log_of,
                   std::vector<int64_t> term_of_group, raftgpu::host::StableStore *store = nullptr, bool wide_kernel = false);
    // Called for every decided row, in the order the rows were decided, AFTER the batch's effects were applied and its (term, votedFor)
    // records are durable (so what it sends never runs ahead of the disk, member/RaftMember.java:25): the host's reaction to what the handler
    // did — RG_F_RESET_TIMER / RG_F_TIMER_MUTED (re-arm the election timer), RG_F_ROLE_CHANGED (abort the old role's invocations), RG_F_EMIT
    // (broadcast PreVote / RequestVote, start replicating: rg_replicate + Ingress::encode_sends), reply.role_epoch (the tag of RPCs sent from
    // now on) — INTEGRATION.md §3 step 5.
    std::function<void(uint32_t gid, const rg_ev_head_t &head, const rg_reply_t &reply)> on_row;
    // One batch. out[conn] receives the response frames. Returns the rows decided (0: nothing was waiting), -1 on a table error (message in error()).
    int64_t flush(std::vector<std::string> &out);
    const Stats &stats() const { return st_; }
    const std::string &error() const { return err_; }
    int64_t term(uint32_t gid) const { return term_[gid]; }

private:
    struct Host;
    struct Reaction { uint32_t gid; rg_ev_head_t head; rg_reply_t reply; };
    std::vector<Reaction> reactions_;                            // on_row calls of the batch in flight, made after its durable write
    void apply(uint32_t gid, rg_ev_head_t head, int64_t a, int64_t b, const char *body, size_t body_len, const rg_reply_t &rep, const rg_logfx_t &lfx,
               const rg_persist_t &per, std::vector<raftgpu::host::StableStore::Record> &dirty);
    int submit_shard(const SealedBatch &b, uint32_t s);
    std::vector<rg_table_t *> tables_;                           // [shards]
    Ingress &ing_;
    const KryoBodyCodec &codec_;
    std::function<raftgpu::host::RaftLog &(uint32_t)> log_of_;
    std::vector<int64_t> term_;
    raftgpu::host::StableStore *store_;
    bool wide_kernel_;
    Stats st_;
    std::string err_;
    std::vector<std::vector<rg_reply_t>> rep_;                   // [shards][rounds * count of the shard]
    std::vector<std::vector<rg_logfx_t>> lfx_;
    std::vector<std::vector<rg_persist_t>> per_;
};

}  // namespace wire
}  // namespace rafting
