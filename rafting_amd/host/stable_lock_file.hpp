// stable_lock_file.hpp — reading (and writing back) the reference's per-context StableLock file, so that a node can be SWITCHED OVER without losing what it
// has promised: (currentTerm, votedFor) and the snapshot milestone (missing #6).
//
// support/StableLock.java:47-91: a RandomAccessFile —
//     offset  0  long  milestone.index     (persist(Snapshot): lastIncludeIndex)          java.io.DataOutput: BIG-endian
//     offset  8  long  milestone.term
//     offset 16  long  currentTerm         (persist(term, candidate))
//     offset 24  int   length of what follows
//     offset 28  bytes Serialization.writeObject(candidate): Kryo 4.0.2 writeClassAndObject of a NodeID{hostname, port}, or of null (one byte 0)
// A fresh file is 28 zero bytes (term 0, no vote). The file may be LONGER than 28 + length: persist() never truncates, an older, longer id leaves its tail.
// The Kryo bytes are read with the same rules as the RPC bodies (kryo_body.cpp) and are, like them, unverified against a JVM (no JVM in this image).
#pragma once

#include <cstdint>
#include <string>

namespace raftgpu {
namespace host {

struct StableLockImage {
    int64_t milestone_index = 0, milestone_term = 0, term = 0;
    bool has_vote = false;           // votedFor != null
    std::string vote_host;
    int32_t vote_port = 0;
};

// false + *err: the file does not have that shape (too short, a length beyond the file, bytes that are not a NodeID)
bool read_stable_lock(const std::string &path, StableLockImage *out, std::string *err);
// the way back (a node that returns to the reference; tests): the same layout, fsync'ed
bool write_stable_lock(const std::string &path, const StableLockImage &img, std::string *err);

}  // namespace host
}  // namespace raftgpu
