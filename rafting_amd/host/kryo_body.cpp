// kryo_body.cpp — KryoBodyCodec (wire.hpp): the RPC bodies of the reference in the byte format of Kryo 4.0.2.
//
// Kryo is a third-party JVM library (com.esotericsoftware:kryo:4.0.2, pom.xml:25-29) that is NOT vendored in /root/reference and cannot
// run here. This file restates the part of its format the reference's RPC bodies use, from Kryo's published sources (class / method named
// at every rule). STATUS: unverified against a JVM — see tests/golden/kryo_bodies.json and INTEGRATION.md §"Checking the Kryo format".
//
// How the reference configures it (support/serial/Serialization.java:21-26): `new Kryo()` with an instantiator strategy — i.e. every default:
// registrationRequired = false, references = true, default serializer FieldSerializer, ten pre-registered types with ids 0..9
// (Kryo.<init>: int, String, float, boolean, byte, char, short, long, double, void — a wrapper class resolves to its primitive's
// registration, DefaultClassResolver.register). Bodies are written by kryo.writeClassAndObject (Serialization.java:47-62, 98-111).
//
// Rules used (all from the Kryo 4.0.2 sources):
//  R1 varint       Output.writeVarInt(value, optimizePositive): 7 bits per byte, low group first, bit 7 = "more"; optimizePositive = false
//                  zig-zags first ((v << 1) ^ (v >> 31)). Output.writeVarLong the same on 64 bits, up to 9 bytes (the 9th holds 8 bits).
//  R2 class        DefaultClassResolver.writeClass: null -> varint 0; registered type -> varint id + 2; unregistered -> varint 1 (NAME + 2), then
//                  varint nameId, and — the FIRST time this class appears in the object graph — its Class.getName() as a string (writeName).
//                  nameIds count up from 0 per top-level write (Kryo.reset -> ClassResolver.reset).
//  R3 string       Output.writeString: null -> 0x80; "" -> 0x81; 2..63 chars, all ASCII -> the chars, bit 7 set on the LAST one (writeAscii);
//                  anything else -> writeUtf8Length(charCount + 1) (first byte carries 6 bits with bit 7 set and bit 6 = "more", later bytes 7
//                  bits + "more") followed by the chars as UTF-8 (1 byte for c <= 0x7F, 2 for <= 0x7FF, else 3).
//  R4 references   Kryo.writeReferenceOrNull (references = true): for an object whose class MapReferenceResolver.useReferences accepts — every
//                  class except the eight primitive wrappers (Util.isWrapperClass) — varint 1 (NOT_NULL) in front of its first occurrence, varint
//                  id + 2 for a repeated one (ids count first occurrences from 0); where null is possible (writeObjectOrNull) null is varint 0.
//                  Wrappers (Long ...) get no marker under writeClassAndObject.
//  R5 Long         DefaultSerializers.LongSerializer.write: writeLong(value, false) = zig-zag varlong.
//  R6 Object[]     DefaultArraySerializers.ObjectArraySerializer.write: varint length + 1 (optimizePositive); the component type of Object[] and
//                  of RaftLog.Entry[] is not final, so every element is written with kryo.writeClassAndObject (R2 + R4 + its serializer).
//  R7 byte[]       DefaultArraySerializers.ByteArraySerializer.write: varint length + 1, then the bytes.
//  R8 fields       FieldSerializer: the non-static, non-transient fields of the class (and its superclasses), sorted by NAME
//                  (FieldSerializer.rebuildCachedFields: Collections.sort(cachedFields, this) — compare = field name); a primitive field is
//                  written raw — boolean 1 byte, int writeInt(v, false), long writeLong(v, false) (zig-zag varints, varIntsEnabled default) — an
//                  object field whose declared class is final (String, byte[]) by kryo.writeObjectOrNull with that class's serializer (R4
//                  marker with null possible, then the value; no class), any other by writeClassAndObject.
//       NodeID      {hostname: String, port: int}            RocksEntry {data: byte[], index: long, term: long}
//       RaftResponse{success: boolean, term: long}
// Decoding reads exactly this shape; a back-reference (R4, id + 2) where a value is needed is answered with `false` — nothing in these bodies
// repeats an object (a request holds one NodeID, every entry its own byte[]).
#include <cstring>

#include "wire.hpp"

namespace rafting {
namespace wire {

namespace {

const char *const N_OBJECT_ARRAY = "[Ljava.lang.Object;";
const char *const N_NODE_ID = "io.lubricant.consensus.raft.transport.event.NodeID";
const char *const N_ENTRY_ARRAY = "[Lio.lubricant.consensus.raft.command.RaftLog$Entry;";
const char *const N_ROCKS_ENTRY = "io.lubricant.consensus.raft.command.storage.RocksEntry";
const char *const N_RESPONSE = "io.lubricant.consensus.raft.RaftResponse";
constexpr uint32_t ID_LONG = 7;                 // Kryo.<init>: register(long.class) is the eighth call

struct Out {
    std::string &o;
    std::vector<const char *> names;            // classes already written by name (identity = pointer of the constants above)
    explicit Out(std::string &out) : o(out) {}
    void u8(uint8_t b) { o.push_back((char)b); }
    void varint(uint32_t v) { while (v >> 7) { u8((uint8_t)(v & 0x7F) | 0x80); v >>= 7; } u8((uint8_t)v); }                 // R1, optimizePositive
    void varint_zz(int32_t v) { varint(((uint32_t)v << 1) ^ (uint32_t)(v >> 31)); }
    void varlong_zz(int64_t v)                                                                                              // R1 / R5
    {
        uint64_t u = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
        for (int i = 0; i < 8 && (u >> 7); i++) { u8((uint8_t)(u & 0x7F) | 0x80); u >>= 7; }
        u8((uint8_t)u);                         // the ninth byte, if it comes to that, carries the remaining 8 bits
    }
    void string(const std::string &s)                                                                                       // R3
    {
        const size_t n = s.size();
        if (n == 0) { u8(0x81); return; }
        bool ascii = n > 1 && n < 64;
        for (size_t i = 0; ascii && i < n; i++) ascii = (uint8_t)s[i] <= 127;
        if (ascii) { o.append(s); o.back() = (char)((uint8_t)o.back() | 0x80); return; }
        // (callers only pass ASCII host names and class names; a non-ASCII or long string takes the length-prefixed form, chars <= 0x7F)
        uint32_t v = (uint32_t)n + 1;
        if (v >> 6 == 0) { u8((uint8_t)(v | 0x80)); }
        else { u8((uint8_t)((v & 0x3F) | 0x40 | 0x80)); v >>= 6; while (v >> 7) { u8((uint8_t)(v & 0x7F) | 0x80); v >>= 7; } u8((uint8_t)v); }
        o.append(s);
    }
    void class_by_name(const char *name)                                                                                    // R2
    {
        varint(1);
        for (size_t i = 0; i < names.size(); i++) if (names[i] == name) { varint((uint32_t)i); return; }
        varint((uint32_t)names.size());
        names.push_back(name);
        string(name);
    }
    void boxed_long(int64_t v) { varint(ID_LONG + 2); varlong_zz(v); }                                                      // R2 + R5 (no R4 marker)
};

enum Klass { K_NULL = 0, K_OTHER, K_OBJECT_ARRAY, K_NODE_ID, K_ENTRY_ARRAY, K_ROCKS_ENTRY, K_RESPONSE, K_REGISTERED };

// No allocation on the decode path: class names and the host name are compared where they lie in the body.
struct In {
    const uint8_t *p, *end;
    bool ok = true;
    Klass names[8];                                          // nameId -> class, in order of first appearance (R2)
    uint32_t n_names = 0, registered = 0;
    explicit In(const char *s, size_t n) : p(reinterpret_cast<const uint8_t *>(s)), end(p + n) {}
    uint8_t u8() { if (p >= end) { ok = false; return 0; } return *p++; }
    uint32_t varint()
    {
        uint32_t v = 0;
        for (int shift = 0; shift < 35; shift += 7) { const uint8_t b = u8(); v |= (uint32_t)(b & 0x7F) << shift; if (!(b & 0x80)) return v; }
        ok = false; return 0;
    }
    int32_t varint_zz() { const uint32_t u = varint(); return (int32_t)((u >> 1) ^ (~(u & 1) + 1)); }
    int64_t varlong_zz()
    {
        uint64_t u = 0;
        int shift = 0;
        for (int i = 0; i < 8; i++, shift += 7) { const uint8_t b = u8(); u |= (uint64_t)(b & 0x7F) << shift; if (!(b & 0x80)) goto done; }
        u |= (uint64_t)u8() << 56;
    done:
        return (int64_t)((u >> 1) ^ (~(u & 1) + 1));
    }
    // R3 (never null here): the characters of the string as a span of the body. ASCII form: `len` bytes, the last one with bit 7 set
    // (`ascii` = true); length-prefixed form: `len` bytes of UTF-8
    bool string_span(const uint8_t *&at, size_t &len, bool &ascii)
    {
        if (p >= end) { ok = false; return false; }
        const uint8_t b = *p;
        if (!(b & 0x80)) {
            at = p;
            while (p < end && !(*p & 0x80)) p++;
            if (p >= end) { ok = false; return false; }
            p++;
            len = (size_t)(p - at); ascii = true;
            return true;
        }
        p++;
        uint32_t v = b & 0x3F;                               // writeUtf8Length
        if (b & 0x40) { int shift = 6; while (true) { const uint8_t c = u8(); if (!ok) return false; v |= (uint32_t)(c & 0x7F) << shift; if (!(c & 0x80)) break; shift += 7; if (shift > 34) { ok = false; return false; } } }
        if (v == 0) { ok = false; return false; }            // null
        at = p;
        for (uint32_t i = 1; i < v; i++) {                   // v - 1 chars as UTF-8
            const uint8_t c = u8();
            if (!ok) return false;
            if (c >= 0xE0) { u8(); u8(); } else if (c >= 0xC0) u8();
        }
        len = (size_t)(p - at); ascii = false;
        return ok;
    }
    static bool same(const uint8_t *at, size_t len, bool ascii, const char *want, size_t wlen)
    {
        if (len != wlen || len == 0) return false;
        if (!ascii) return memcmp(at, want, len) == 0;
        return memcmp(at, want, len - 1) == 0 && (uint8_t)(at[len - 1] & 0x7F) == (uint8_t)want[len - 1];
    }
    // R2: which class follows
    Klass klass()
    {
        const uint32_t c = varint();
        if (c == 0) return K_NULL;
        if (c != 1) { registered = c - 2; return K_REGISTERED; }
        const uint32_t id = varint();
        if (id < n_names) return names[id];
        const uint8_t *at; size_t len; bool ascii;
        if (id != n_names || n_names == 8 || !string_span(at, len, ascii)) { ok = false; return K_OTHER; }
        Klass k = K_OTHER;
#define RG_KRYO_NAME(cst, val) if (same(at, len, ascii, cst, sizeof(cst) - 1)) k = val
        RG_KRYO_NAME("[Ljava.lang.Object;", K_OBJECT_ARRAY);
        else RG_KRYO_NAME("io.lubricant.consensus.raft.transport.event.NodeID", K_NODE_ID);
        else RG_KRYO_NAME("[Lio.lubricant.consensus.raft.command.RaftLog$Entry;", K_ENTRY_ARRAY);
        else RG_KRYO_NAME("io.lubricant.consensus.raft.command.storage.RocksEntry", K_ROCKS_ENTRY);
        else RG_KRYO_NAME("io.lubricant.consensus.raft.RaftResponse", K_RESPONSE);
#undef RG_KRYO_NAME
        names[n_names++] = k;
        return k;
    }
    // fast path: the next bytes are exactly `tpl` (the body is left where it was when they are not)
    bool expect(const std::string &tpl)
    {
        if ((size_t)(end - p) < tpl.size() || memcmp(p, tpl.data(), tpl.size()) != 0) return false;
        p += tpl.size();
        return true;
    }
    bool expect_long(int64_t &v) { if (p >= end || *p != ID_LONG + 2) return false; p++; v = varlong_zz(); return ok; }
    bool first_occurrence() { if (varint() != 1) { ok = false; return false; } return ok; }                                  // R4: NOT_NULL, not a back-reference
    bool boxed_long(int64_t &v) { if (klass() != K_REGISTERED || registered != ID_LONG) { ok = false; return false; } v = varlong_zz(); return ok; }
};

}  // namespace

void KryoBodyCodec::encode_request(Method m, const Request &in, std::string &body) const
{
    Out w(body);
    const bool ae = m == M_APPEND_ENTRIES;
    w.class_by_name(N_OBJECT_ARRAY); w.varint(1);                                  // Object[] (R2, R4)
    w.varint((ae ? 6u : 4u) + 1u);                                                 // R6
    w.boxed_long(in.term);
    const Node none{"", 0};
    const Node &id = (in.node >= 0 && (size_t)in.node < nodes_.size()) ? nodes_[(size_t)in.node] : none;
    w.class_by_name(N_NODE_ID); w.varint(1);                                       // NodeID: fields by name — hostname, port (R8)
    w.varint(1); w.string(id.hostname);                                            //   String field: writeObjectOrNull (R4 marker, R3)
    w.varint_zz(id.port);
    w.boxed_long(in.x);
    w.boxed_long(in.y);
    if (ae) {
        w.class_by_name(N_ENTRY_ARRAY); w.varint(1);
        w.varint((uint32_t)in.entry_terms.size() + 1u);
        for (size_t k = 0; k < in.entry_terms.size(); k++) {
            w.class_by_name(N_ROCKS_ENTRY); w.varint(1);                           // RocksEntry: data, index, term (R8)
            w.varint(1); w.varint(8u + 1u);                                        //   byte[] field: R4 marker, R7; the stored value starts with the term
            const uint64_t t = (uint64_t)in.entry_terms[k];
            for (int s = 56; s >= 0; s -= 8) w.u8((uint8_t)(t >> s));
            w.varlong_zz((int64_t)((uint64_t)in.x + 1u + k));
            w.varlong_zz(in.entry_terms[k]);
        }
        w.boxed_long(in.leader_commit);
    }
}

void KryoBodyCodec::encode_response(const Response &in, std::string &body) const
{
    static const std::string head = [] { std::string s; Out w(s); w.class_by_name(N_RESPONSE); w.varint(1); return s; }();   // the same bytes every time
    body.append(head);
    Out w(body);
    w.u8(in.success ? 1 : 0);                                                      // success before term (R8)
    w.varlong_zz(in.term);
}

// The general reader: any body the rules above allow (class records in any order of first appearance, either string form).
bool KryoBodyCodec::decode_request_general(Method m, const char *body, size_t len, Request &out) const { return read_request(m, body, len, out, nullptr); }

// (visit: called with every entry's stored value — RocksEntry.data, the bytes RaftLog.append writes — where it lies in the body)
bool KryoBodyCodec::read_request(Method m, const char *body, size_t len, Request &out, const EntryVisitor *visit) const
{
    In r(body, len);
    out.leader_commit = 0;
    out.entry_terms.clear();
    out.node = RG_NO_NODE;
    if (r.klass() != K_OBJECT_ARRAY || !r.first_occurrence()) return false;
    const uint32_t n1 = r.varint();
    const bool ae = m == M_APPEND_ENTRIES;
    if (!r.ok || n1 != (ae ? 7u : 5u)) return false;                               // NettyNode.prepareLocalInvocation: params.length != 6 / 4 throws
    if (!r.boxed_long(out.term)) return false;
    if (r.klass() != K_NODE_ID || !r.first_occurrence()) return false;
    const uint8_t *host; size_t host_len; bool ascii;
    if (!r.first_occurrence() || !r.string_span(host, host_len, ascii)) return false;
    const int32_t port = r.varint_zz();
    for (size_t i = 0; i < nodes_.size(); i++)
        if (nodes_[i].port == port && In::same(host, host_len, ascii, nodes_[i].hostname.data(), nodes_[i].hostname.size())) out.node = (int32_t)i;
    if (!r.boxed_long(out.x) || !r.boxed_long(out.y)) return false;
    if (ae) {
        if (r.klass() != K_ENTRY_ARRAY || !r.first_occurrence()) return false;
        const uint32_t e1 = r.varint();
        if (!r.ok || e1 == 0 || (size_t)(e1 - 1) > (size_t)(r.end - r.p) || e1 - 1 > RG_MAX_ENTRIES) return false;   // (more than a row's header can count:
                                                                  // refused before a peer's 64 MiB body turns into tens of megabytes of scratch)
        for (uint32_t k = 0; k + 1 < e1; k++) {
            if (r.klass() != K_ROCKS_ENTRY || !r.first_occurrence()) return false;
            if (!r.first_occurrence()) return false;                               // data (never null: RocksLog.get builds entries from stored values)
            const uint32_t d1 = r.varint();
            if (!r.ok || d1 == 0 || (size_t)(d1 - 1) > (size_t)(r.end - r.p)) return false;
            const uint8_t *data = r.p;
            r.p += d1 - 1;                                                         // the payload never reaches a decision row
            const int64_t index = r.varlong_zz(), term = r.varlong_zz();
            if (!r.ok || index != (int64_t)((uint64_t)out.x + 1u + k)) return false;   // the C-ABI's rows carry implicit indices
            out.entry_terms.push_back(term);
            if (visit) (*visit)(index, term, reinterpret_cast<const char *>(data), d1 - 1);
        }
        if (!r.boxed_long(out.leader_commit)) return false;
    }
    return r.ok && r.p == r.end && out.node != RG_NO_NODE;
}

bool KryoBodyCodec::decode_node_object(const char *bytes, size_t len, Node &out, bool &is_null)
{
    In r(bytes, len);
    is_null = false;
    const Klass k = r.klass();
    if (r.ok && k == K_NULL) { is_null = true; return r.p == r.end; }
    if (!r.ok || k != K_NODE_ID || !r.first_occurrence()) return false;
    const uint8_t *host; size_t host_len; bool ascii;
    if (!r.first_occurrence() || !r.string_span(host, host_len, ascii)) return false;
    out.port = r.varint_zz();
    out.hostname.assign(reinterpret_cast<const char *>(host), host_len);
    if (ascii && host_len) out.hostname[host_len - 1] = (char)(out.hostname[host_len - 1] & 0x7F);      // (R3: the last character carries the end mark)
    return r.ok && r.p == r.end;
}

void KryoBodyCodec::encode_node_object(const Node *node, std::string &out)
{
    Out w(out);
    if (!node) { w.varint(0); return; }                                            // Kryo.NULL
    w.class_by_name(N_NODE_ID); w.varint(1);
    w.varint(1); w.string(node->hostname);
    w.varint_zz(node->port);
}

static bool decode_response_general(const char *body, size_t len, Response &out)
{
    In r(body, len);
    if (r.klass() != K_RESPONSE || !r.first_occurrence()) return false;
    out.success = r.u8() != 0;
    out.term = r.varlong_zz();
    return r.ok && r.p == r.end;
}

// ---- the fast path -----------------------------------------------------------------------------------------------------------------------
// What the reference's own encoder produces has ONE shape per method: the class records appear in a fixed order (Object[] = name 0, NodeID = 1,
// Entry[] = 2, RocksEntry = 3; RaftResponse = 0), names in the ASCII form. Those runs of constant bytes are built once with the writer above
// and compared with memcmp; only the varints between them are parsed. Any deviation — another order, the length-prefixed string form, a
// back-reference — leaves the fast path without a verdict and the general reader decides from the start of the body: the fast path accepts
// a subset of what the general reader accepts, with the same result (tests/test_kryo_cpu.py runs both on every vector and on random bodies).
namespace {

struct Templates {
    std::string head_ae, head_rq, node, entries, entry_first, entry_next, response;
    Templates()
    {
        { std::string s; Out w(s); w.class_by_name(N_OBJECT_ARRAY); w.varint(1); const size_t a = s.size(); w.varint(7); head_ae = s;
          s.resize(a); w.varint(5); head_rq = s; }
        std::string s;
        Out w(s);
        w.class_by_name(N_OBJECT_ARRAY);                     // (name ids as they stand after the head)
        size_t at = s.size();
        auto cut = [&](std::string &dst) { dst.assign(s, at, std::string::npos); at = s.size(); };
        w.class_by_name(N_NODE_ID); w.varint(1); w.varint(1); cut(node);                 // class, first occurrence, hostname's NOT_NULL marker
        w.class_by_name(N_ENTRY_ARRAY); w.varint(1); cut(entries);
        w.class_by_name(N_ROCKS_ENTRY); w.varint(1); w.varint(1); cut(entry_first);      // class (by name), first occurrence, data's NOT_NULL marker
        w.class_by_name(N_ROCKS_ENTRY); w.varint(1); w.varint(1); cut(entry_next);       // class (by id), ...
        std::string r; Out wr(r); wr.class_by_name(N_RESPONSE); wr.varint(1); response = r;
    }
};
const Templates &templates() { static const Templates t; return t; }

// 1 = decoded, 0 = refused, -1 = not the encoder's shape (ask the general reader)
int decode_request_fast(const std::vector<std::string> &node_bytes, Method m, const char *body, size_t len, Request &out)
{
    const Templates &T = templates();
    const bool ae = m == M_APPEND_ENTRIES;
    In r(body, len);
    out.leader_commit = 0;
    out.entry_terms.clear();
    out.node = RG_NO_NODE;
    if (!r.expect(ae ? T.head_ae : T.head_rq) || !r.expect_long(out.term) || !r.expect(T.node)) return -1;
    for (size_t i = 0; i < node_bytes.size(); i++)
        if (r.expect(node_bytes[i])) { out.node = (int32_t)i; break; }
    if (out.node == RG_NO_NODE) return -1;                   // an unknown node, or a name in the other string form: the general reader says which
    if (!r.expect_long(out.x) || !r.expect_long(out.y)) return -1;
    if (ae) {
        if (!r.expect(T.entries)) return -1;
        const uint32_t e1 = r.varint();
        if (!r.ok || e1 == 0 || (size_t)(e1 - 1) > (size_t)(r.end - r.p)) return -1;
        if (e1 - 1 > RG_MAX_ENTRIES) return 0;
        for (uint32_t k = 0; k + 1 < e1; k++) {
            if (!r.expect(k == 0 ? T.entry_first : T.entry_next)) return -1;
            const uint32_t d1 = r.varint();
            if (!r.ok || d1 == 0 || (size_t)(d1 - 1) > (size_t)(r.end - r.p)) return -1;
            r.p += d1 - 1;
            const int64_t index = r.varlong_zz(), term = r.varlong_zz();
            if (!r.ok) return -1;
            if (index != (int64_t)((uint64_t)out.x + 1u + k)) return 0;
            out.entry_terms.push_back(term);
        }
        if (!r.expect_long(out.leader_commit)) return -1;
    }
    return (r.ok && r.p == r.end) ? 1 : -1;
}

}  // namespace

bool KryoBodyCodec::entries(const char *body, size_t len, const EntryVisitor &visit) const
{
    Request scratch;
    return read_request(M_APPEND_ENTRIES, body, len, scratch, &visit);
}

KryoBodyCodec::KryoBodyCodec(std::vector<Node> nodes) : nodes_(std::move(nodes))
{
    for (const Node &n : nodes_) {                           // hostname (R3) + port (zig-zag varint) as the encoder writes them
        std::string s;
        Out w(s);
        w.string(n.hostname);
        w.varint_zz(n.port);
        node_bytes_.push_back(s);
    }
}

bool KryoBodyCodec::decode_request(Method m, const char *body, size_t len, Request &out) const
{
    const int fast = decode_request_fast(node_bytes_, m, body, len, out);
    return fast >= 0 ? fast == 1 : decode_request_general(m, body, len, out);
}

bool KryoBodyCodec::decode_response(const char *body, size_t len, Response &out) const
{
    In r(body, len);
    if (r.expect(templates().response) && r.p < r.end) {
        out.success = *r.p++ != 0;
        out.term = r.varlong_zz();
        if (r.ok && r.p == r.end) return true;
    }
    return decode_response_general(body, len, out);
}

}  // namespace wire
}  // namespace rafting
