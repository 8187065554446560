// wire_capi.cpp — the C-ABI of include/raftwire.h over wire.hpp
#include "../../include/raftwire.h"

#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "wire.hpp"

using namespace rafting::wire;

struct rw_splitter {
    FrameSplitter sp;
    std::deque<Frame> q;
    Frame last;
};

extern "C" {

rw_splitter_t *rw_splitter_new(void) { return new rw_splitter(); }
void rw_splitter_free(rw_splitter_t *s) { delete s; }

int rw_splitter_feed(rw_splitter_t *s, const uint8_t *data, size_t n)
{
    if (!s) return -1;
    std::vector<Frame> out;
    s->sp.feed(data, n, out);
    for (Frame &f : out) s->q.push_back(std::move(f));
    return s->sp.failed() ? -1 : (int)s->q.size();
}

int rw_splitter_pop(rw_splitter_t *s, uint8_t *type, int32_t *sequence, const char **head, size_t *head_len, const uint8_t **body, size_t *body_len)
{
    if (!s || s->q.empty()) return 0;
    s->last = std::move(s->q.front());
    s->q.pop_front();
    *type = s->last.type; *sequence = s->last.sequence;
    *head = s->last.head.data(); *head_len = s->last.head.size();
    *body = reinterpret_cast<const uint8_t *>(s->last.body.data()); *body_len = s->last.body.size();
    return 1;
}

int rw_splitter_failed(const rw_splitter_t *s) { return s && s->sp.failed(); }
size_t rw_splitter_held(const rw_splitter_t *s) { return s ? s->sp.held() : 0; }
int rw_splitter_transparent(const rw_splitter_t *s) { return s && s->sp.transparent(); }
size_t rw_splitter_passthrough(rw_splitter_t *s, const uint8_t **data)
{
    *data = reinterpret_cast<const uint8_t *>(s->sp.passthrough().data());
    return s->sp.passthrough().size();
}

static size_t emit(const std::string &o, uint8_t *out, size_t cap)
{
    if (o.size() > cap) return 0;
    memcpy(out, o.data(), o.size());
    return o.size();
}

size_t rw_encode_frame(uint8_t type, int32_t sequence, const char *head, size_t head_len, const uint8_t *body, size_t body_len, int ending,
                       uint8_t *out, size_t cap)
{
    Frame f;
    f.type = type; f.sequence = sequence; f.head.assign(head, head_len);
    f.body.assign(reinterpret_cast<const char *>(body), body_len);
    std::string o;
    encode_frame(f, ending != 0, o);
    return emit(o, out, cap);
}

size_t rw_fixed_request(int method, int64_t term, int32_t node, int64_t x, int64_t y, int64_t leader_commit, const int64_t *entry_terms, uint32_t n,
                        uint8_t *out, size_t cap)
{
    Request q;
    q.term = term; q.node = node; q.x = x; q.y = y; q.leader_commit = leader_commit;
    q.entry_terms.assign(entry_terms, entry_terms + n);
    std::string o;
    FixedBodyCodec().encode_request((Method)method, q, o);
    return emit(o, out, cap);
}

size_t rw_fixed_response(int64_t term, int success, uint8_t *out, size_t cap)
{
    std::string o;
    FixedBodyCodec().encode_response(Response{term, success != 0}, o);
    return emit(o, out, cap);
}

// ---- Kryo-format bodies (KryoBodyCodec): nodes = "host:port,host:port,..." in slot order ------------------------------------------
static std::vector<KryoBodyCodec::Node> parse_nodes(const char *nodes)
{
    std::vector<KryoBodyCodec::Node> v;
    std::string s(nodes ? nodes : "");
    size_t at = 0;
    while (at < s.size()) {
        size_t comma = s.find(',', at);
        if (comma == std::string::npos) comma = s.size();
        const std::string one = s.substr(at, comma - at);
        const size_t colon = one.rfind(':');
        if (colon != std::string::npos) v.push_back({one.substr(0, colon), atoi(one.c_str() + colon + 1)});
        at = comma + 1;
    }
    return v;
}

size_t rw_kryo_request(const char *nodes, int method, int64_t term, int32_t node, int64_t x, int64_t y, int64_t leader_commit, const int64_t *entry_terms,
                       uint32_t n, uint8_t *out, size_t cap)
{
    Request q;
    q.term = term; q.node = node; q.x = x; q.y = y; q.leader_commit = leader_commit;
    q.entry_terms.assign(entry_terms, entry_terms + n);
    std::string o;
    KryoBodyCodec(parse_nodes(nodes)).encode_request((Method)method, q, o);
    return emit(o, out, cap);
}

size_t rw_kryo_response(int64_t term, int success, uint8_t *out, size_t cap)
{
    std::string o;
    KryoBodyCodec({}).encode_response(Response{term, success != 0}, o);
    return emit(o, out, cap);
}

int rw_kryo_decode_request(const char *nodes, int method, const uint8_t *body, size_t len, int64_t *term, int32_t *node, int64_t *x, int64_t *y,
                           int64_t *leader_commit, int64_t *entry_terms, uint32_t max_terms, uint32_t *n_terms)
{
    Request q;
    if (!KryoBodyCodec(parse_nodes(nodes)).decode_request((Method)method, reinterpret_cast<const char *>(body), len, q)) return 0;
    if (q.entry_terms.size() > max_terms) return 0;
    *term = q.term; *node = q.node; *x = q.x; *y = q.y; *leader_commit = q.leader_commit;
    *n_terms = (uint32_t)q.entry_terms.size();
    for (size_t k = 0; k < q.entry_terms.size(); k++) entry_terms[k] = q.entry_terms[k];
    return 1;
}

int rw_kryo_decode_response(const uint8_t *body, size_t len, int64_t *term, int *success)
{
    Response r;
    if (!KryoBodyCodec({}).decode_response(reinterpret_cast<const char *>(body), len, r)) return 0;
    *term = r.term; *success = r.success ? 1 : 0;
    return 1;
}

int rw_rows_add_frame(uint8_t type, int32_t sequence, const char *head, size_t head_len, const uint8_t *body, size_t body_len, int32_t peer,
                      const char *const *ctx_ids, uint32_t n_ctx, uint32_t pending_role_epoch, int64_t pending_epoch_at_send,
                      int64_t pending_last_index_sent, rg_ev_head_t *head_out, rg_ev_pair_t *ab, rg_ev_pair_t *cd, uint32_t *gid,
                      int64_t *entry_terms, size_t max_rows, size_t max_terms, size_t *rows, size_t *terms)
{
    Frame f;
    f.type = type; f.sequence = sequence; f.head.assign(head, head_len);
    f.body.assign(reinterpret_cast<const char *>(body), body_len);
    RowWriter w(head_out + *rows, ab + *rows, cd + *rows, gid ? gid + *rows : nullptr, entry_terms + *terms, max_rows - *rows, max_terms - *terms);
    FixedBodyCodec codec;
    const bool ok = w.add(f, peer, codec,
        [&](const std::string &ctx, uint32_t &g) { for (uint32_t i = 0; i < n_ctx; i++) if (ctx == ctx_ids[i]) { g = i; return true; } return false; },
        [&](const std::string &, int32_t, Pending &p) { p.role_epoch = pending_role_epoch; p.epoch_at_send = pending_epoch_at_send; p.last_index_sent = pending_last_index_sent; return true; });
    if (!ok) return 0;
    head_out[*rows].aux += (type == ENQ && (head_out[*rows].hdr & 0xFu) == RG_EV_AE_REQ) ? (uint32_t)*terms : 0u;   // entry offsets are batch-wide
    *rows += 1;
    *terms += w.terms();
    return 1;
}

}  // extern "C"
