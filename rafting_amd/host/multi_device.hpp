// multi_device.hpp — one node's contexts spread over several GPUs.
//
// Raft groups share nothing (context/ContextManager.java:41,112-120: a ConcurrentHashMap<String, RaftContext> is all that
// connects them), so the reference's single ContextManager becomes N independent shards here: each GPU gets its own table,
// its own stream and its own FEEDER THREAD, and no data ever crosses devices (no RCCL, north_star). What is shared is the
// routing key: contextId -> global gid (creation order) -> device = gid / ceil(capacity / N) (block partition).
//   createContext   routes to the shard that owns the next gid; the RaftContext it returns enqueues into that shard by itself
//   flushAll(now)   fan-out: every shard with queued rows drains on its own feeder thread — one rg_submit per device, all
//                   devices at once; fan-in: returns when the last one is done. Outcomes come back per shard, ticket order.
// A shard is a plain ContextManager: timers, health, replicateLog etc. are reached through shard(k).
// Threads: the routing table (contextId -> shard, gid) is guarded; createContext / getContext / shardOf / globalGid may be called from any
// thread while another thread runs flushAll — the reference's map is a ConcurrentHashMap (context/ContextManager.java:41) and its
// contexts are created while event loops drain. createContext on shard k waits for a drain of shard k that is running (one shard is one
// C-ABI handle: include/raftgpu.h "not re-entrant per handle"). What stays with the caller, as with ContextManager itself: the rows of ONE
// shard are queued by one thread at a time and not while that shard drains.
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>

#include "raft_host.hpp"

namespace raftgpu {
namespace host {

class MultiDeviceManager {
  public:
    // devices[k] = HIP device ordinal of shard k (the same ordinal may appear twice: two tables on one GPU, used by the tests)
    MultiDeviceManager(const std::vector<int> &devices, uint32_t maxContexts, uint32_t clusterSize, ID self, bool preVote);
    ~MultiDeviceManager();
    MultiDeviceManager(const MultiDeviceManager &) = delete;
    MultiDeviceManager &operator=(const MultiDeviceManager &) = delete;

    RaftContext &createContext(const std::string &id, int64_t restoreTerm = 0, ID restoreBallot = RG_NO_NODE);
    RaftContext *getContext(const std::string &id);
    size_t shards() const { return shards_.size(); }
    ContextManager &shard(size_t k) { return *shards_[k]->mgr; }
    size_t shardOf(const std::string &id) const;              // which shard (= device slot) owns the context
    uint32_t globalGid(const std::string &id) const;          // creation index: gid of the context in a single-table layout

    // [shard][ticket of that shard]; shards without queued rows contribute an empty vector
    std::vector<std::vector<Outcome>> flushAll(int64_t now = -1);

  private:
    struct Shard {
        std::unique_ptr<ContextManager> mgr;
        std::thread feeder;
        std::mutex m;
        std::mutex work;                 // held around everything that touches `mgr`'s table: a drain, a createContext
        std::condition_variable cv;
        bool go = false, done = false, quit = false;
        int64_t now = -1;
        std::vector<Outcome> out;
        std::exception_ptr error;
    };
    void feed(Shard &s);
    std::vector<std::unique_ptr<Shard>> shards_;
    uint32_t per_shard_, capacity_, created_ = 0;
    mutable std::mutex route_m_;                                   // guards created_ and where_
    std::map<std::string, std::pair<size_t, uint32_t>> where_;     // id -> (shard, global gid)
};

}  // namespace host
}  // namespace raftgpu
