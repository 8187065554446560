// wire.cpp — see wire.hpp. Frame grammar: transport/EventCodec.java:169-335; scopes: transport/NettyNode.java:54-158.
#include "wire.hpp"

#include <cstring>

namespace rafting {
namespace wire {

// ---- FrameSplitter --------------------------------------------------------------------------------------------------
int32_t FrameSplitter::be32(size_t at) const
{
    const uint8_t *p = reinterpret_cast<const uint8_t *>(buf_.data()) + at;
    return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
}

void FrameSplitter::fail(const std::string &why)
{
    failed_ = true;                      // the reference logs, drops its state and closes the channel (:330-334)
    error_ = why;
    buf_.clear();
    pos_ = 0;
}

// The field-by-field progress below follows FrameDecoder.decode (:300-327) literally — one pass over "type? sequence? head
// length? then head+body length OR body" per loop turn, each guarded by "enough bytes?" — because the reference's behaviour on
// degenerate input depends on it: a NUL type byte leaves `type == NUL`, so the type is read AGAIN on the next turn if the turn
// ended for lack of bytes, but the same turn goes on to demand STX from the byte after it when five more bytes are there.
size_t FrameSplitter::feed(const uint8_t *data, size_t n, std::vector<Frame> &out)
{
    return feed_views(data, n, [&out](const FrameView &v) {
        Frame f;
        f.type = v.type; f.sequence = v.sequence;
        f.head.assign(v.head, v.head_len);
        f.body.assign(v.body, v.body_len);
        out.push_back(std::move(f));
    });
}

size_t FrameSplitter::feed_views(const uint8_t *data, size_t n, const std::function<void(const FrameView &)> &sink, const std::function<void()> &drained)
{
    if (failed_) return 0;
    if (transparent_) { passthrough_.append(reinterpret_cast<const char *>(data), n); return 0; }
    buf_.append(reinterpret_cast<const char *>(data), n);
    size_t made = 0;
    while (need(1)) {
        if (!open_) {
            const uint8_t b = (uint8_t)buf_[pos_++];
            if (b == EOT) {              // end of the framed protocol: the rest of the stream is not ours (:296-298)
                if (drained) drained();  // (queued views point into buf_, which goes now)
                transparent_ = true;
                passthrough_.append(buf_, pos_, std::string::npos);
                buf_.clear(); pos_ = 0;
                return made;
            }
            if (b != SOH) { if (drained) drained(); fail("frame does not start with SOH"); return made; }
            cur_.type = NUL; cur_.sequence = 0;
            head_len_ = body_len_ = -1;
            has_seq_ = false;
            open_ = true;
            continue;
        }
        if (cur_.type == NUL) {
            if (!need(1)) break;
            cur_.type = (uint8_t)buf_[pos_++];
            if (cur_.type == ENQ || cur_.type == ACK) has_seq_ = true;       // only Ping / Pong carry a sequence (:239-244)
        }
        if (has_seq_) {
            if (!need(4)) break;
            cur_.sequence = be32(pos_);
            pos_ += 4;
            has_seq_ = false;
        }
        if (head_len_ == -1) {
            if (!need(5)) break;
            if ((uint8_t)buf_[pos_] != STX) { if (drained) drained(); fail("STX expected"); return made; }
            const int32_t len = be32(pos_ + 1);
            pos_ += 5;
            if (len < 0 || len > MAX_HEAD_SIZE) { if (drained) drained(); fail("illegal head length"); return made; }
            head_len_ = len;
        }
        if (body_len_ == -1) {
            if (!need((size_t)head_len_ + 4)) break;
            head_at_ = pos_;                                                  // the head stays in buf_ (compaction rebases this offset)
            const int32_t len = be32(pos_ + (size_t)head_len_);
            pos_ += (size_t)head_len_ + 4;
            if (len < 0 || len > MAX_BODY_SIZE) { if (drained) drained(); fail("illegal body length"); return made; }
            body_len_ = len;
        } else {
            if (!need((size_t)body_len_ + 1)) break;
            FrameView v;
            v.type = cur_.type; v.sequence = cur_.sequence;
            v.head = buf_.data() + head_at_; v.head_len = (size_t)head_len_;
            v.body = buf_.data() + pos_; v.body_len = (size_t)body_len_;
            pos_ += (size_t)body_len_;
            if ((uint8_t)buf_[pos_++] != ETX) { if (drained) drained(); fail("ETX expected"); return made; }
            sink(v);
            made++;
            open_ = false;
        }
    }
    if (drained) drained();
    // Drop what has been consumed. A read normally ENDS inside a frame (the loop has already eaten the next SOH), so waiting for a
    // moment between frames would let a busy connection grow buf_ without bound: while a frame is open everything before
    // the bytes it still needs goes — its head once that has been located (head_at_ is rebased), else the parse position.
    const size_t keep_from = (open_ && body_len_ != -1) ? head_at_ : pos_;
    if (keep_from > (1u << 16) && keep_from * 2 > buf_.size()) {
        buf_.erase(0, keep_from);
        pos_ -= keep_from;
        if (open_ && body_len_ != -1) head_at_ = 0;
    }
    return made;
}

static void put32(std::string &o, int32_t v)
{
    const uint32_t u = (uint32_t)v;
    o.push_back((char)(u >> 24)); o.push_back((char)(u >> 16)); o.push_back((char)(u >> 8)); o.push_back((char)u);
}
static void put64(std::string &o, int64_t v) { put32(o, (int32_t)((uint64_t)v >> 32)); put32(o, (int32_t)(uint64_t)v); }

void encode_frame(const Frame &f, bool ending, std::string &out)
{
    char h[11];                                                       // SOH | TYPE | {SEQ} | STX | HEADLEN
    size_t n = 0;
    auto be32 = [&](int32_t v) { const uint32_t u = (uint32_t)v; h[n++] = (char)(u >> 24); h[n++] = (char)(u >> 16); h[n++] = (char)(u >> 8); h[n++] = (char)u; };
    h[n++] = (char)SOH;
    h[n++] = (char)f.type;
    if (f.type == ENQ || f.type == ACK) be32(f.sequence);
    h[n++] = (char)STX;
    be32((int32_t)f.head.size());
    out.append(h, n);
    out += f.head;
    n = 0;
    be32((int32_t)f.body.size());
    out.append(h, n);
    out += f.body;
    out.push_back((char)ETX);
    if (ending) out.push_back((char)EOT);
}

// ---- scopes ---------------------------------------------------------------------------------------------------------
static const char *const METHOD_NAME[] = {"", "appendEntries", "preVote", "requestVote", "installSnapshot"};

bool parse_scope(const char *head, size_t len, Method &method, std::string &context_id)
{
    for (int m = M_APPEND_ENTRIES; m <= M_INSTALL_SNAPSHOT; m++) {     // the order NettyNode.parseContextId tests them in
        const size_t n = strlen(METHOD_NAME[m]);
        if (len >= n && memcmp(head, METHOD_NAME[m], n) == 0) {          // scope.startsWith(name): the separator itself is not checked
            method = (Method)m;
            if (len > n) context_id.assign(head + n + 1, len - n - 1); else context_id.clear();
            return true;
        }
    }
    return false;
}
bool parse_scope(const char *head, size_t len, Method &method, const char *&context_id, size_t &id_len)
{
    for (int m = M_APPEND_ENTRIES; m <= M_INSTALL_SNAPSHOT; m++) {
        const size_t n = strlen(METHOD_NAME[m]);
        if (len >= n && memcmp(head, METHOD_NAME[m], n) == 0) {
            method = (Method)m;
            if (len > n) { context_id = head + n + 1; id_len = len - n - 1; } else { context_id = head + len; id_len = 0; }
            return true;
        }
    }
    return false;
}
bool parse_scope(const std::string &head, Method &method, std::string &context_id) { return parse_scope(head.data(), head.size(), method, context_id); }

std::string make_scope(Method method, const std::string &context_id) { return std::string(METHOD_NAME[method]) + ":" + context_id; }

// ---- FixedBodyCodec -------------------------------------------------------------------------------------------------
namespace {
struct Reader {
    const char *s; size_t n; size_t at = 0; bool ok = true;
    Reader(const char *p, size_t len) : s(p), n(len) {}
    size_t size() const { return n; }
    int32_t i32() { if (at + 4 > n) { ok = false; return 0; } uint32_t v; memcpy(&v, s + at, 4); at += 4; return (int32_t)__builtin_bswap32(v); }
    int64_t i64() { if (at + 8 > n) { ok = false; at = n; return 0; } uint64_t v; memcpy(&v, s + at, 8); at += 8; return (int64_t)__builtin_bswap64(v); }
    uint8_t u8() { if (at + 1 > n) { ok = false; return 0; } return (uint8_t)s[at++]; }
};
}  // namespace

bool FixedBodyCodec::decode_request(Method m, const char *body, size_t len, Request &out) const
{
    Reader r(body, len);
    out.leader_commit = 0;
    out.entry_terms.clear();                                           // (keeps its capacity: `out` is reused from frame to frame)
    out.term = r.i64(); out.node = r.i32(); out.x = r.i64(); out.y = r.i64();
    if (m == M_APPEND_ENTRIES) {
        out.leader_commit = r.i64();
        const int32_t n = r.i32();
        if (!r.ok || n < 0 || (size_t)n > (len - r.at) / 8) return false;
        out.entry_terms.resize((size_t)n);
        for (int32_t k = 0; k < n; k++) out.entry_terms[(size_t)k] = r.i64();
    }
    return r.ok && r.at == len;
}

bool FixedBodyCodec::decode_response(const char *body, size_t len, Response &out) const
{
    Reader r(body, len);
    out.term = r.i64();
    out.success = r.u8() != 0;
    return r.ok && r.at == len;
}

void FixedBodyCodec::encode_request(Method m, const Request &in, std::string &body) const
{
    put64(body, in.term); put32(body, in.node); put64(body, in.x); put64(body, in.y);
    if (m == M_APPEND_ENTRIES) {
        put64(body, in.leader_commit);
        put32(body, (int32_t)in.entry_terms.size());
        for (int64_t t : in.entry_terms) put64(body, t);
    }
}

void FixedBodyCodec::encode_response(const Response &in, std::string &body) const
{
    put64(body, in.term);
    body.push_back(in.success ? 1 : 0);
}

// ---- RowWriter --------------------------------------------------------------------------------------------------------
bool RowWriter::add(const Frame &f, int32_t peer, const BodyCodec &codec, const std::function<bool(const std::string &, uint32_t &)> &gid_of,
                    const std::function<bool(const std::string &, int32_t, Pending &)> &pending_of)
{
    FrameView v;
    v.type = f.type; v.sequence = f.sequence;
    v.head = f.head.data(); v.head_len = f.head.size();
    v.body = f.body.data(); v.body_len = f.body.size();
    return add(v, peer, codec, gid_of, pending_of);
}

bool RowWriter::add(const FrameView &f, int32_t peer, const BodyCodec &codec, const std::function<bool(const std::string &, uint32_t &)> &gid_of,
                    const std::function<bool(const std::string &, int32_t, Pending &)> &pending_of)
{
    if (f.type != ENQ && f.type != ACK) return false;                  // hand-shake / snapshot channel events are not decisions
    Method m;
    std::string &ctx = ctx_;
    uint32_t gid = 0;
    if (!parse_scope(f.head, f.head_len, m, ctx) || !gid_of(ctx, gid) || rows_ >= max_rows_) return false;
    rg_ev_head_t h{0, 0};
    rg_ev_pair_t ab{0, 0}, cd{0, 0};
    size_t add_terms = 0;
    if (f.type == ENQ) {                                               // a request: NettyNode.prepareLocalInvocation (:109-158)
        Request &q = q_;
        if (!codec.decode_request(m, f.body, f.body_len, q) || q.node < 0 || q.node > 15) return false;
        switch (m) {
        case M_APPEND_ENTRIES:
            if (q.entry_terms.size() > RG_MAX_AE_ENTRIES || nterms_ + q.entry_terms.size() > max_terms_) return false;
            h.hdr = RG_HDR_MAKE(RG_EV_AE_REQ, q.node, 0, q.entry_terms.size());
            h.aux = (uint32_t)nterms_;
            ab = {q.term, q.x}; cd = {q.y, q.leader_commit};
            add_terms = q.entry_terms.size();
            for (size_t k = 0; k < add_terms; k++) terms_[nterms_ + k] = q.entry_terms[k];
            break;
        case M_PRE_VOTE:     h.hdr = RG_HDR_MAKE(RG_EV_PV_REQ, q.node, 0, 0); ab = {q.term, q.x}; cd = {q.y, 0}; break;
        case M_REQUEST_VOTE: h.hdr = RG_HDR_MAKE(RG_EV_RV_REQ, q.node, 0, 0); ab = {q.term, q.x}; cd = {q.y, 0}; break;
        case M_INSTALL_SNAPSHOT:
            // flag = what RaftContext.installSnapshot() RETURNED (member/Follower.java:146-148): a frame cannot know that. The row is
            // written with flag 0 ("not installed"); the host submits it only after its download finished and sets RG_HDR flag bit 8 then
            // (a row submitted as decoded must not answer success for a snapshot nobody holds).
            h.hdr = RG_HDR_MAKE(RG_EV_IS_REQ, q.node, 0, 0); ab = {q.term, q.x}; cd = {q.y, 0}; break;
        default: return false;
        }
    } else {                                                           // a response: AsyncService.Invocation by (scope, sequence)
        Response r;
        Pending p;
        scope_.assign(f.head, f.head_len);
        if (!codec.decode_response(f.body, f.body_len, r) || !pending_of(scope_, f.sequence, p) || peer < 0 || peer > 15) return false;
        switch (m) {
        case M_APPEND_ENTRIES:
            h.hdr = RG_HDR_MAKE(RG_EV_AE_ACK, peer, r.success, 0); h.aux = p.role_epoch;
            ab = {r.term, p.epoch_at_send}; cd = {p.last_index_sent, 0};
            break;
        case M_INSTALL_SNAPSHOT:
            h.hdr = RG_HDR_MAKE(RG_EV_IS_ACK, peer, r.success, 0); h.aux = p.role_epoch;
            ab = {r.term, p.epoch_at_send};
            break;
        case M_PRE_VOTE:     h.hdr = RG_HDR_MAKE(RG_EV_PV_REPLY, peer, r.success, 0); h.aux = p.role_epoch; ab = {r.term, 0}; break;
        case M_REQUEST_VOTE: h.hdr = RG_HDR_MAKE(RG_EV_RV_REPLY, peer, r.success, 0); h.aux = p.role_epoch; ab = {r.term, 0}; break;
        default: return false;
        }
    }
    head_[rows_] = h; ab_[rows_] = ab; cd_[rows_] = cd;
    if (gid_) gid_[rows_] = gid;
    rows_++;
    nterms_ += add_terms;
    return true;
}

}  // namespace wire
}  // namespace rafting
