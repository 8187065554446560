// stable_store.hpp — N3: the durability barrier of the decision path.
//
// The reference persists (currentTerm, votedFor) with ONE fsync per role conversion per group:
// RaftMember.<init> -> StableLock.persist (member/RaftMember.java:25, support/StableLock.java:69-91), a 28-byte
// header in one file per context. With thousands of groups deciding in one flush that is thousands of fsyncs.
// StableStore keeps ONE append-only journal per node: ContextManager::flush() hands it every (gid, term, votedFor)
// the batch marked RG_F_PERSIST and it issues one write + one fdatasync for all of them — still BEFORE any response
// of that flush is released, which is the ordering the reference guarantees.
//
// Record (24 B, little endian): gid u32 | votedFor i32 | term i64 | seq u32 | crc32 u32 (over the first 20 bytes).
// restore() replays the journal and stops at the first record whose checksum fails (a torn tail after a crash).
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace raftgpu {
namespace host {

class StableStore {
  public:
    struct Record {
        uint32_t gid;
        int64_t term;
        int32_t votedFor;
    };
    explicit StableStore(const std::string &path);
    ~StableStore();
    StableStore(const StableStore &) = delete;
    StableStore &operator=(const StableStore &) = delete;

    void persist(const std::vector<Record> &batch);     // one write + one fdatasync; returns after the data is durable
    bool restore(uint32_t gid, int64_t *term, int32_t *votedFor) const;   // StableLock.restore for one context
    void compact();                                     // rewrite the journal with the latest record of every group
    uint64_t syncs() const { return syncs_; }
    uint64_t records() const { return records_; }
    size_t groups() const { return latest_.size(); }

  private:
    void replay();
    std::string path_;
    int fd_ = -1;
    uint32_t seq_ = 0;
    uint64_t syncs_ = 0, records_ = 0;
    bool broken_ = false;          // a failed batch could not be removed from the file again
    std::unordered_map<uint32_t, Record> latest_;
};

}  // namespace host
}  // namespace raftgpu
