// wire_capi.cpp — the C-ABI of include/raftwire.h over wire.hpp
#include "../../include/raftwire.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <deque>
#include <string>
#include <vector>

#include "ingress.hpp"
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

int rw_kryo_entry(const char *nodes, const uint8_t *body, size_t len, uint32_t k, int64_t *index, int64_t *term, const uint8_t **data, size_t *n)
{
    uint32_t at = 0;
    bool found = false;
    const bool ok = KryoBodyCodec(parse_nodes(nodes)).entries(reinterpret_cast<const char *>(body), len, [&](int64_t i, int64_t t, const char *d, size_t dn) {
        if (at++ == k) { *index = i; *term = t; *data = reinterpret_cast<const uint8_t *>(d); *n = dn; found = true; }
    });
    return ok && found;
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

// ---- rw_ingress_* ---------------------------------------------------------------------------------------------------------------------
}  // extern "C"

struct rw_ingress {
    std::unique_ptr<BodyCodec> codec;
    ContextIndex index;
    std::unique_ptr<Ingress> in;
    const SealedBatch *sealed[2] = {nullptr, nullptr};
    uint32_t conns, groups;
    rw_ingress(uint32_t groups_) : index(groups_), conns(0), groups(groups_) {}
};

extern "C" {

rw_ingress_t *rw_ingress_new(uint32_t groups, uint32_t max_rounds, uint32_t conns, const char *nodes, rg_ev_head_t *head0, rg_ev_quad32_t *abcd0,
                             int32_t *entry_terms0, rg_ev_head_t *head1, rg_ev_quad32_t *abcd1, int32_t *entry_terms1, uint64_t entry_cap, uint32_t shards)
{
    if (!groups || !max_rounds || !conns || !shards || shards > groups || !head0 || !abcd0 || !head1 || !abcd1 || (entry_cap && (!entry_terms0 || !entry_terms1))) return nullptr;
    rw_ingress *g = new rw_ingress(groups);
    if (nodes) g->codec.reset(new KryoBodyCodec(parse_nodes(nodes))); else g->codec.reset(new FixedBodyCodec());
    g->conns = conns;
    g->in.reset(new Ingress(groups, max_rounds, conns, *g->codec, g->index, Ingress::Buffers{head0, abcd0, entry_terms0, entry_cap},
                            Ingress::Buffers{head1, abcd1, entry_terms1, entry_cap}, 1u << 16, shards));
    return g;
}
void rw_ingress_free(rw_ingress_t *g) { delete g; }
int rw_ingress_add_context(rw_ingress_t *g, const char *id, size_t len, uint32_t gid) { return g && g->index.insert(id, len, gid); }
int rw_ingress_remove_context(rw_ingress_t *g, const char *id, size_t len) { return g && g->index.erase(id, len) < g->groups; }
int rw_ingress_set_peer(rw_ingress_t *g, uint32_t conn, int32_t peer_slot)
{
    if (!g || conn >= g->conns) return 0;
    g->in->set_peer(conn, peer_slot);
    return 1;
}
int rw_ingress_sent(rw_ingress_t *g, uint32_t conn, int32_t sequence, int method, uint32_t gid, uint32_t role_epoch, int64_t epoch_at_send,
                    int64_t last_index_sent)
{
    if (!g || conn >= g->conns || method < M_APPEND_ENTRIES || method > M_INSTALL_SNAPSHOT) return 0;
    Pending p;
    p.role_epoch = role_epoch; p.epoch_at_send = epoch_at_send; p.last_index_sent = last_index_sent;
    g->in->pending(conn).put(sequence, (Method)method, gid, p);
    return 1;
}
int rw_ingress_retain_bodies(rw_ingress_t *g, int on)
{
    if (!g) return 0;
    g->in->retain_bodies(on != 0);
    return 1;
}
int rw_ingress_body(const rw_ingress_t *g, int bank, uint32_t shard, uint64_t cell, const uint8_t **body, size_t *len)
{
    if (!g || bank < 0 || bank > 1 || !g->sealed[bank] || !body || !len) return 0;
    const char *p = g->in->body(*g->sealed[bank], shard, (size_t)cell, *len);
    *body = reinterpret_cast<const uint8_t *>(p);
    return p != nullptr;
}
int rw_ingress_reset_conn(rw_ingress_t *g, uint32_t conn)
{
    if (!g || conn >= g->conns) return 0;
    g->in->reset_connection(conn);
    return 1;
}
int rw_ingress_feed(rw_ingress_t *g, uint32_t conn, const uint8_t *data, size_t n) { return (!g || conn >= g->conns) ? -1 : g->in->feed(conn, data, n); }
}  // extern "C"
namespace {
struct CTermOf : Ingress::TermOf {
    int64_t (*fn)(void *, uint32_t, int64_t);
    void *user;
    int64_t term_of(uint32_t gid, int64_t index) override { return fn(user, gid, index); }
};
}  // namespace
extern "C" {
size_t rw_ingress_encode_sends(rw_ingress_t *g, uint32_t conn, int32_t self_slot, uint32_t count, const uint32_t *gid, const rg_send_head_t *head,
                               const rg_send_t *send_j, int64_t (*term_of)(void *user, uint32_t gid, int64_t index), void *user, uint8_t *out, size_t cap,
                               uint32_t *frames, uint32_t *need_host)
{
    if (!g || conn >= g->conns || !head || !send_j || !term_of || !frames || !need_host) return 0;
    CTermOf log;
    log.fn = term_of; log.user = user;
    std::string o;
    const int32_t seq0 = g->in->send_sequence(conn);
    *frames = (uint32_t)g->in->encode_sends(conn, self_slot, count, gid, head, send_j, log, o, need_host);
    if (o.size() > cap) { g->in->send_sequence(conn) = seq0; *frames = 0; return o.size(); }   // (the records filed are filed again, identically, by the retry)
    if (!o.empty()) memcpy(out, o.data(), o.size());
    return o.size();
}
int rw_ingress_add_row(rw_ingress_t *g, uint32_t conn, uint32_t gid, uint32_t hdr, uint32_t aux, int64_t a, int64_t b, int64_t c, int64_t d,
                       uint32_t reply_conn, int32_t reply_sequence)
{
    if (!g || conn >= g->conns || (reply_conn != NO_CONN && reply_conn >= g->conns)) return 0;
    g->in->add_row(conn, gid, rg_ev_head_t{hdr, aux}, a, b, c, d, Origin{reply_conn, reply_sequence});
    return 1;
}
int rw_ingress_seal(rw_ingress_t *g, rg_batch32_t *batch, uint64_t *rows, uint32_t *wide)
{
    if (!g) return -1;
    const SealedBatch *sp;
    const size_t erased_before = g->index.retired();
    try { sp = &g->in->seal(); } catch (const std::logic_error &) { return -1; }      // the batch sealed before has not been recycled
    const SealedBatch &s = *sp;
    g->index.reclaim(erased_before);                                 // (no lookup that began before those removals is still running)
    const int bank = g->in->bank_of(s);
    g->sealed[bank] = &s;
    *batch = s.batch; *rows = s.rows; *wide = (uint32_t)s.wide.size();
    return bank;
}
int rw_ingress_shard(const rw_ingress_t *g, int bank, uint32_t shard, rg_batch32_t *batch, uint64_t *events, uint32_t *first_gid)
{
    if (!g || bank < 0 || bank > 1 || !g->sealed[bank] || shard >= g->sealed[bank]->shard.size()) return 0;
    const SealedShard &s = g->sealed[bank]->shard[shard];
    *batch = s.batch; *events = s.events; *first_gid = s.first_gid;
    return 1;
}
int rw_ingress_wide_row(const rw_ingress_t *g, int bank, uint32_t i, uint32_t *gid, rg_ev_head_t *head, int64_t abcd[4], int64_t *entry_terms, uint32_t max_terms,
                        uint32_t *reply_conn, int32_t *reply_sequence)
{
    if (!g || bank < 0 || bank > 1 || !g->sealed[bank] || i >= g->sealed[bank]->wide.size()) return -1;
    const HeldRow &h = g->sealed[bank]->wide[i];
    if (h.terms.size() > max_terms) return -1;
    *reply_conn = h.from.conn; *reply_sequence = h.from.sequence;
    *gid = h.gid; *head = h.head; abcd[0] = h.a; abcd[1] = h.b; abcd[2] = h.c; abcd[3] = h.d;
    for (size_t k = 0; k < h.terms.size(); k++) entry_terms[k] = h.terms[k];
    return (int)h.terms.size();
}
size_t rw_ingress_emit_wide(const rw_ingress_t *g, int bank, uint32_t i, const rg_reply_t *reply, uint32_t *conn, uint8_t *out, size_t cap)
{
    if (!g || bank < 0 || bank > 1 || !g->sealed[bank] || i >= g->sealed[bank]->wide.size() || !reply || !conn) return 0;
    std::string o;
    *conn = g->in->emit_wide(g->sealed[bank]->wide[i], *reply, o);
    if (*conn == NO_CONN || o.size() > cap) return 0;
    memcpy(out, o.data(), o.size());
    return o.size();
}
int rw_ingress_origin(const rw_ingress_t *g, int bank, uint32_t shard, uint64_t cell, uint32_t *conn, int32_t *sequence)
{
    if (!g || bank < 0 || bank > 1 || !g->sealed[bank] || shard >= g->sealed[bank]->shard.size()) return 0;
    const SealedShard &s = g->sealed[bank]->shard[shard];
    if (cell >= (uint64_t)s.batch.rounds * s.batch.count || RG_HDR_KIND(s.batch.head[cell].hdr) == RG_EV_NONE || s.origin[cell].conn == NO_CONN) return 0;
    *conn = s.origin[cell].conn; *sequence = s.origin[cell].sequence;
    return 1;
}
size_t rw_ingress_emit(const rw_ingress_t *g, int bank, uint32_t shard, const rg_reply_t *reply, uint64_t cell_begin, uint64_t cell_end, uint32_t conn, uint8_t *out,
                       size_t cap)
{
    if (!g || bank < 0 || bank > 1 || !g->sealed[bank] || conn >= g->conns || shard >= g->sealed[bank]->shard.size()) return 0;
    std::vector<std::string> o(g->conns);
    g->in->emit(*g->sealed[bank], reply, o, (size_t)cell_begin, (size_t)cell_end, conn, shard);
    if (!o[conn].empty() && o[conn].size() <= cap) memcpy(out, o[conn].data(), o[conn].size());
    return o[conn].size();
}
}  // extern "C"
namespace {
struct CRepairHost : RepairHost {
    const rw_repair_host_t &h;
    explicit CRepairHost(const rw_repair_host_t &host) : h(host) {}
    int64_t term_at(uint32_t gid, int64_t index) override { return h.term_at(h.user, gid, index); }
    int64_t conflict(uint32_t gid, int64_t first, const int64_t *terms, uint32_t n) override { return h.conflict(h.user, gid, first, terms, n); }
    int64_t epoch_index(uint32_t gid) override { return h.epoch_index(h.user, gid); }
    int submit(const rg_batch_t &in, const rg_outcome_t &out) override { return h.submit(h.user, &in, &out); }
    void applied(uint32_t gid, size_t cell, const rg_reply_t &r, const rg_logfx_t &l, const rg_persist_t &p) override { h.applied(h.user, gid, cell, &r, &l, &p); }
};
}  // namespace
extern "C" {
int64_t rw_ingress_repair(const rw_ingress_t *g, int bank, uint32_t shard, rg_reply_t *reply, const rg_logfx_t *logfx, int packed, const rw_repair_host_t *host)
{
    if (!g || bank < 0 || bank > 1 || !g->sealed[bank] || shard >= g->sealed[bank]->shard.size() || !reply || !logfx || !host || !host->term_at || !host->conflict || !host->epoch_index ||
        !host->submit || !host->applied) return -1;
    CRepairHost h(*host);
    return repair_need_host(*g->sealed[bank], reply, logfx, packed != 0, h, shard);
}
int rw_ingress_recycle(rw_ingress_t *g, int bank)
{
    if (!g || bank < 0 || bank > 1 || !g->sealed[bank]) return 0;
    g->in->recycle(*g->sealed[bank]);
    g->sealed[bank] = nullptr;
    return 1;
}
uint64_t rw_ingress_refused(const rw_ingress_t *g) { return g ? g->in->refused() : 0; }
uint64_t rw_ingress_held(const rw_ingress_t *g) { return g ? g->in->held() : 0; }
uint64_t rw_ingress_held_on(const rw_ingress_t *g, uint32_t conn) { return (g && conn < g->conns) ? g->in->held_on(conn) : 0; }

}  // extern "C"
