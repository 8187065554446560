// wire.hpp — N2: the reference's wire frames, split and decoded straight into event rows (host C++, no GPU involved).
//
// What feeds the decision kernels in a deployment are the RPC frames of transport/EventCodec.java. Their grammar is
// first-party and simple:
//      SOH | TYPE | {SEQ:int32} | STX | HEADLEN:int32 | HEAD | BODYLEN:int32 | {BODY} | ETX [| EOT]      (big-endian)
//   TYPE   ENQ 0x05 = PingEvent (request), ACK 0x06 = PongEvent (response) — both carry SEQ; SYN 0x16, MW 0x95, PM 0x9E are
//          the string events of the hand-shake / snapshot channel (no SEQ, no body)            transport/EventCodec.java:28-40,239-249
//   HEAD   UTF-8, at most 128 bytes: "<method>:<contextId>", method = appendEntries | preVote | requestVote | installSnapshot
//                                                                                              transport/NettyNode.java:54-73,93-107
//   BODY   at most 64 MiB, Kryo-serialised Object[] (request parameters) / RaftResponse (transport/EventCodec.java:186-191,270-279)
//   EOT    after a frame = the sender ends the framed protocol on this connection; everything after it is passed through
//          undecoded (transport/EventCodec.java:283-299: "transparent")
// FrameSplitter is that grammar as a streaming state machine (bytes arrive in arbitrary pieces). The BODY is the one
// third-party part (kryo 4.0.2, absent here): BodyCodec is the plug; FixedBodyCodec is a fixed-layout stand-in (big-endian
// longs) used by the tests and the cluster simulation, to be replaced by a Kryo reader in the JVM deployment.
// RowWriter turns decoded messages into rows of an rg_batch_t (include/raftgpu.h) — the structure of arrays rg_submit takes —
// resolving "<contextId>" to a group id and a response's (scope, sequence) to what the host remembered about the request.
//
// tests/test_wire_cpu.py checks FrameSplitter byte for byte against the reference's own FrameDecoder / FrameEncoder, compiled
// from transport/EventCodec.java by tools/make_ref.py (oracle/_ref/libref_wire.so).
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/raftgpu.h"

namespace rafting {
namespace wire {

enum : uint8_t { NUL = 0x00, SOH = 0x01, STX = 0x02, ETX = 0x03, EOT = 0x04, ENQ = 0x05, ACK = 0x06, SYN = 0x16, MW = 0x95, PM = 0x9E };
constexpr int32_t MAX_HEAD_SIZE = 128;             // transport/EventCodec.java:25
constexpr int32_t MAX_BODY_SIZE = 1 << 26;         // :26

struct Frame {
    uint8_t type = NUL;
    int32_t sequence = 0;          // ENQ / ACK only
    std::string head;
    std::string body;              // raw bytes (what Serialization.readObject would consume)
};

// A frame that still lives in the splitter's buffer: valid until the next call into the splitter (no copies on the hot path).
struct FrameView {
    uint8_t type = NUL;
    int32_t sequence = 0;
    const char *head = nullptr; size_t head_len = 0;
    const char *body = nullptr; size_t body_len = 0;
};

// Streaming decoder. feed() consumes what it can and appends complete frames; bytes of an incomplete frame are kept.
// After a protocol violation the connection is dead (the reference closes the channel, :330-334): failed() stays true and
// further input is ignored. After an EOT the rest of the stream is handed to passthrough() untouched.
class FrameSplitter {
public:
    // returns the number of frames appended
    size_t feed(const uint8_t *data, size_t n, std::vector<Frame> &out);
    // the same state machine handing every complete frame to `sink` as a view into the internal buffer (nothing is copied or allocated
    // per frame); feed() above is this plus a copy
    // `drained` (optional) is called once after the last frame of this call and BEFORE the buffer is compacted: a sink that queues views
    // (they stay valid until then) finishes its work there
    size_t feed_views(const uint8_t *data, size_t n, const std::function<void(const FrameView &)> &sink, const std::function<void()> &drained = nullptr);
    bool failed() const { return failed_; }
    const std::string &error() const { return error_; }
    bool transparent() const { return transparent_; }
    std::string &passthrough() { return passthrough_; }
    size_t buffered() const { return buf_.size() - pos_; }
    size_t held() const { return buf_.size(); }          // bytes kept in memory, consumed ones included: bounded by one frame + 64 KiB (test_wire_cpu.py)

private:
    bool need(size_t n) const { return buf_.size() - pos_ >= n; }
    int32_t be32(size_t at) const;
    void fail(const std::string &why);
    bool open_ = false, has_seq_ = false;  // a frame is being read / its sequence number is still to come
    Frame cur_;                            // type / sequence of the frame being read (head and body stay in buf_ until it is complete)
    size_t head_at_ = 0;
    int32_t head_len_ = -1, body_len_ = -1;
    std::string buf_;
    size_t pos_ = 0;
    bool failed_ = false, transparent_ = false;
    std::string error_, passthrough_;
};

// FrameEncoder.encode (transport/EventCodec.java:171-196)
void encode_frame(const Frame &f, bool ending, std::string &out);

// "<method>:<contextId>" (transport/NettyNode.java:54-73,93-107)
enum Method { M_NONE = 0, M_APPEND_ENTRIES, M_PRE_VOTE, M_REQUEST_VOTE, M_INSTALL_SNAPSHOT };
bool parse_scope(const std::string &head, Method &method, std::string &context_id);
bool parse_scope(const char *head, size_t len, Method &method, std::string &context_id);
bool parse_scope(const char *head, size_t len, Method &method, const char *&context_id, size_t &id_len);    // the id as a span of `head`
std::string make_scope(Method method, const std::string &context_id);

// ---- bodies ---------------------------------------------------------------------------------------------------------
struct Request {                   // RaftService method parameters (RaftService.java:22-61)
    int64_t term = 0;
    int32_t node = RG_NO_NODE;     // leaderId / candidateId as a peer slot
    int64_t x = 0, y = 0;          // prevLogIndex, prevLogTerm | lastLogIndex, lastLogTerm | lastIncludedIndex, lastIncludedTerm
    int64_t leader_commit = 0;     // appendEntries only
    std::vector<int64_t> entry_terms;   // appendEntries only: term of entries[k], index prevLogIndex + 1 + k
};
struct Response { int64_t term = 0; bool success = false; };   // RaftResponse.java:8-24

class BodyCodec {
public:
    virtual ~BodyCodec() {}
    virtual bool decode_request(Method m, const char *body, size_t len, Request &out) const = 0;   // `out` may be reused: every field is set
    virtual bool decode_response(const char *body, size_t len, Response &out) const = 0;
    bool decode_request(Method m, const std::string &body, Request &out) const { return decode_request(m, body.data(), body.size(), out); }
    bool decode_response(const std::string &body, Response &out) const { return decode_response(body.data(), body.size(), out); }
    virtual void encode_request(Method m, const Request &in, std::string &body) const = 0;
    virtual void encode_response(const Response &in, std::string &body) const = 0;
};

// Stand-in for the Kryo body: big-endian fixed layout.
//   appendEntries   i64 term, i32 leader, i64 prevLogIndex, i64 prevLogTerm, i64 leaderCommit, i32 n, n x i64 entry term
//   votes / snapshot i64 term, i32 node, i64 x, i64 y
//   response        i64 term, u8 success
class FixedBodyCodec : public BodyCodec {
public:
    using BodyCodec::decode_request;
    using BodyCodec::decode_response;
    bool decode_request(Method m, const char *body, size_t len, Request &out) const override;
    bool decode_response(const char *body, size_t len, Response &out) const override;
    void encode_request(Method m, const Request &in, std::string &body) const override;
    void encode_response(const Response &in, std::string &body) const override;
};

// The body as the reference writes it: Kryo 4.0.2 (pom.xml:25-29), `kryo.writeClassAndObject` on a default-configured Kryo
// (support/serial/Serialization.java:21-26: `new Kryo()` + an instantiator strategy, nothing registered, references on), of
//   request   Object[]{Long term, NodeID id, Long x, Long y [, RaftLog.Entry[] entries, Long leaderCommit]}   transport/NettyNode.java:54-73
//   response  RaftResponse{boolean success, long term}                                                          RaftResponse.java:8-24
// with NodeID{String hostname, int port} (transport/event/NodeID.java:8-17) and entries of class RocksEntry{byte[] data, long index,
// long term} (command/storage/RocksEntry.java:5-16; data = the stored value, whose first 8 bytes are the term, storage/RocksLog.java:82-89).
// The byte format is restated in wire.cpp (KryoBodyCodec) from Kryo's published sources, rule by rule with the method each rule comes
// from. Kryo itself is a JVM library that is absent here: the vectors in tests/golden/kryo_bodies.json are built from the same rules by
// an independent Python encoder and are UNVERIFIED AGAINST A JVM — INTEGRATION.md shows the one-test check a maintainer with a JDK runs.
// Peer slots <-> NodeID: the cluster list of the XML config (transport/NettyCluster.java:32-50), given to the constructor.
class KryoBodyCodec : public BodyCodec {
public:
    struct Node { std::string hostname; int32_t port; };
    explicit KryoBodyCodec(std::vector<Node> nodes);
    using BodyCodec::decode_request;
    using BodyCodec::decode_response;
    bool decode_request(Method m, const char *body, size_t len, Request &out) const override;
    bool decode_response(const char *body, size_t len, Response &out) const override;
    void encode_request(Method m, const Request &in, std::string &body) const override;
    void encode_response(const Response &in, std::string &body) const override;
    // decode_request = a fast path for the one byte shape the reference's encoder produces, else this: the reader of everything the format allows
    // (public so that tests can hold the two to the same answers)
    bool decode_request_general(Method m, const char *body, size_t len, Request &out) const;
    // The follower's write path: every entry of an appendEntries body as (index, term, stored value) — RocksEntry.data is the value RaftLog.append
    // puts under the index (command/storage/RocksLog.java:169-198; its first 8 bytes are the term, :82-89). The spans point into `body`.
    // false: not an appendEntries body of this cluster (entries visited before the defect was found have been reported).
    typedef std::function<void(int64_t index, int64_t term, const char *data, size_t n)> EntryVisitor;
    bool entries(const char *body, size_t len, const EntryVisitor &visit) const;
    // A NodeID on its own, as Serialization.writeObject(candidate) leaves it in a StableLock file (support/StableLock.java:69-80): kryo.writeClassAndObject
    // of a NodeID, or of null (ONE byte, 0). No node table involved: host name and port as they stand in the bytes.
    static bool decode_node_object(const char *bytes, size_t len, Node &out, bool &is_null);
    static void encode_node_object(const Node *node_or_null, std::string &out);
    // a request's entries carry payload bytes the decision rows never see; index_of_first = prevLogIndex + 1 (what Leader.replicateLog ships)
private:
    bool read_request(Method m, const char *body, size_t len, Request &out, const EntryVisitor *visit) const;
    std::vector<Node> nodes_;
    std::vector<std::string> node_bytes_;     // hostname + port of each node as the encoder writes them (the decoder's fast path compares bytes)
};

// ---- frames -> rows -------------------------------------------------------------------------------------------------
// What the host remembered when it SENT a request (the closure state of the reference's callbacks, member/Leader.java:174-188,
// 218-237; AsyncService keeps invocations by (scope, sequence), transport/NettyNode.java:89-91).
struct Pending {
    uint32_t role_epoch = 0;       // rg_reply_t.role_epoch / rg_send_head_t.role_epoch at send time
    int64_t epoch_at_send = 0;     // appendEntries / installSnapshot: RaftLog.epoch().index
    int64_t last_index_sent = 0;   // appendEntries: rg_send_t.last_index
};

// Appends rows to caller-owned structure-of-arrays buffers (page-locked memory from rg_host_alloc in a deployment).
class RowWriter {
public:
    RowWriter(rg_ev_head_t *head, rg_ev_pair_t *ab, rg_ev_pair_t *cd, uint32_t *gid, int64_t *entry_terms, size_t max_rows, size_t max_terms)
        : head_(head), ab_(ab), cd_(cd), gid_(gid), terms_(entry_terms), max_rows_(max_rows), max_terms_(max_terms) {}
    // peer = the slot of the node at the other end of the connection the frame came from
    // returns false when the frame is not a raft RPC of a known context, or the buffers are full (nothing is written then)
    bool add(const Frame &f, int32_t peer, const BodyCodec &codec, const std::function<bool(const std::string &, uint32_t &)> &gid_of,
             const std::function<bool(const std::string &, int32_t, Pending &)> &pending_of);
    bool add(const FrameView &f, int32_t peer, const BodyCodec &codec, const std::function<bool(const std::string &, uint32_t &)> &gid_of,
             const std::function<bool(const std::string &, int32_t, Pending &)> &pending_of);
    size_t rows() const { return rows_; }
    size_t terms() const { return nterms_; }
    void clear() { rows_ = 0; nterms_ = 0; }

private:
    rg_ev_head_t *head_; rg_ev_pair_t *ab_, *cd_; uint32_t *gid_; int64_t *terms_;
    size_t max_rows_, max_terms_, rows_ = 0, nterms_ = 0;
    Request q_;                            // scratch reused from row to row (no allocation once its vector has grown)
    std::string scope_, ctx_;
};

}  // namespace wire
}  // namespace rafting
