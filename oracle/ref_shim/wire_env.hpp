// wire_env.hpp — Netty / Kryo stand-ins for the mechanically translated EventCodec frame classes (oracle/_ref/gen/wire_*.hpp).
// TEST INFRASTRUCTURE, hand-written, not reference text.  ByteBuf is a growable big-endian buffer with reader / writer indices
// (io.netty.buffer.ByteBuf semantics for the dozen methods the codec calls); Serialization reads / writes the body as raw
// bytes (the real one is Kryo, third party).
#pragma once
#include "jrt.hpp"

struct Class : virtual Object { const char *name; Class(const char *n) : name(n) {} };
#include "gen/wire_fwd.hpp"

template <class... A> JString jconcat(const A &... a) { std::string o; int dummy[] = {0, (jrt::fmt_arg(o, a), 0)...}; (void)dummy; return JString(o); }
namespace jrt { inline void fmt_arg(std::string &out, const char *v) { out += v; } inline void fmt_arg(std::string &out, jbyte v) { out += std::to_string((int)v); } }

struct SerializeException : Exception { SerializeException() {} SerializeException(const char *m) : Exception(m) {} const char *kind() const override { return "SerializeException"; } };
struct DecoderException : RuntimeException {
    DecoderException(const JString &m) : RuntimeException(m.s) {}
    DecoderException(const char *m) : RuntimeException(m) {}
    DecoderException(const char *m, const Throwable &) : RuntimeException(m) {}
    const char *kind() const override { return "DecoderException"; }
};
struct StandardCharsets { static constexpr int UTF_8 = 0; };
struct CharSequence : virtual Object { std::string s; JString toString() { return JString(s); } };

struct ByteBuf : virtual Object {
    std::string b; size_t r = 0, w = 0, mark_w = 0;
    jboolean isReadable() { return w > r; }
    jint readableBytes() { return (jint)(w - r); }
    void need(size_t n) { if (w - r < n) throw IndexOutOfBoundsException("readerIndex + length exceeds writerIndex"); }
    jbyte readByte() { need(1); return (jbyte)b[r++]; }
    jint readInt() { need(4); uint32_t v = 0; for (int k = 0; k < 4; k++) v = (v << 8) | (uint8_t)b[r++]; return (jint)v; }
    Ref<CharSequence> readCharSequence(jint len, int) { need((size_t)len); Ref<CharSequence> c = jnew<CharSequence>(); c->s = b.substr(r, (size_t)len); r += (size_t)len; return c; }
    void ensure(size_t n) { if (b.size() < w + n) b.resize(w + n); }
    void writeByte(jint v) { ensure(1); b[w++] = (char)v; }
    void writeInt(jint v) { ensure(4); uint32_t u = (uint32_t)v; b[w++] = (char)(u >> 24); b[w++] = (char)(u >> 16); b[w++] = (char)(u >> 8); b[w++] = (char)u; }
    jint writeCharSequence(const JString &s, int) { ensure(s.s.size()); memcpy(&b[w], s.s.data(), s.s.size()); w += s.s.size(); return (jint)s.s.size(); }
    jint writerIndex() { return (jint)w; }
    void setInt(jint at, jint v) { uint32_t u = (uint32_t)v; b[(size_t)at] = (char)(u >> 24); b[(size_t)at + 1] = (char)(u >> 16); b[(size_t)at + 2] = (char)(u >> 8); b[(size_t)at + 3] = (char)u; }
    void markWriterIndex() { mark_w = w; }
    void resetWriterIndex() { w = mark_w; }
    void append(const uint8_t *p, size_t n) { ensure(n); memcpy(&b[w], p, n); w += n; }
    void discardReadBytes() { b.erase(0, r); w -= r; r = 0; }
};
struct RawBody : virtual Object { std::string bytes; };
struct Serialization {
    static Ref<Object> readObject(Ref<ByteBuf> buf, jint len) { buf->need((size_t)len); Ref<RawBody> o = jnew<RawBody>(); o->bytes = buf->b.substr(buf->r, (size_t)len); buf->r += (size_t)len; return o; }
    static void writeObject(Ref<Object> o, Ref<ByteBuf> buf) { RawBody *rb = dynamic_cast<RawBody *>(o.get()); buf->ensure(rb->bytes.size()); memcpy(&buf->b[buf->w], rb->bytes.data(), rb->bytes.size()); buf->w += rb->bytes.size(); }
};
struct ChannelHandlerContext : virtual Object {
    bool closed = false; std::string passthrough;
    void fireChannelRead(Ref<ByteBuf> buf) { passthrough.append(buf->b, buf->r, buf->w - buf->r); buf->r = buf->w; }
    void close() { closed = true; }
};
static Ref<Logger> &logger_ref() { static Ref<Logger> l = jnew<Logger>(); return l; }
#define logger (logger_ref())

#include "gen/wire_decls.hpp"
#include "gen/wire_defs.hpp"
Ref<Class> EventCodec_class = jnew<Class>("EventCodec"), EventFrame_class = jnew<Class>("EventFrame"),
           FrameEncoder_class = jnew<Class>("FrameEncoder"), FrameDecoder_class = jnew<Class>("FrameDecoder");
