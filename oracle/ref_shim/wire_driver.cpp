// wire_driver.cpp — C API over the translated EventCodec.FrameDecoder / FrameEncoder (oracle/_ref/libref_wire.so). TEST INFRASTRUCTURE.
// The decoder is driven the way io.netty.handler.codec.ByteToMessageDecoder drives it: incoming bytes are appended to a
// cumulation buffer and decode() is called while it makes progress (callDecode); a closed channel ignores further input.
#include "wire_env.hpp"

struct refwire_decoder {
    Ref<FrameDecoder> dec = jnew<FrameDecoder>();
    Ref<ByteBuf> cumulation = jnew<ByteBuf>();
    Ref<ChannelHandlerContext> ctx = jnew<ChannelHandlerContext>();
    Ref<List<Object>> frames = jnew<ArrayList<Object>>();
    size_t popped = 0;
    std::string last_head, last_body;
};

extern "C" {

refwire_decoder *refwire_decoder_new() { return new refwire_decoder(); }
void refwire_decoder_free(refwire_decoder *d) { delete d; }

/* returns the number of complete frames queued, or -1 once the channel has been closed by a decoding error */
int refwire_decoder_feed(refwire_decoder *d, const uint8_t *data, size_t n)
{
    if (d->ctx->closed) return -1;
    d->cumulation->append(data, n);
    while (d->cumulation->isReadable() && !d->ctx->closed) {            // ByteToMessageDecoder.callDecode
        const jint out_before = d->frames->size(), in_before = d->cumulation->readableBytes();
        d->dec->decode(d->ctx, d->cumulation, d->frames);
        if (out_before == d->frames->size() && in_before == d->cumulation->readableBytes()) break;
    }
    d->cumulation->discardReadBytes();
    return d->ctx->closed ? -1 : (int)(d->frames->size() - (jint)d->popped);
}

int refwire_decoder_pop(refwire_decoder *d, uint8_t *type, int32_t *sequence, const char **head, size_t *head_len, const uint8_t **body, size_t *body_len)
{
    if ((jint)d->popped >= d->frames->size()) return 0;
    Ref<EventFrame> f = jcast<EventFrame>(d->frames->get((jint)d->popped++));
    *type = (uint8_t)f->type; *sequence = f->sequence;
    d->last_head = f->head.s;
    RawBody *rb = dynamic_cast<RawBody *>(f->body.get());
    d->last_body = rb ? rb->bytes : std::string();
    *head = d->last_head.data(); *head_len = d->last_head.size();
    *body = reinterpret_cast<const uint8_t *>(d->last_body.data()); *body_len = d->last_body.size();
    return 1;
}
int refwire_decoder_closed(refwire_decoder *d) { return d->ctx->closed; }
int refwire_decoder_transparent(refwire_decoder *d) { return d->dec->transparent; }
size_t refwire_decoder_passthrough(refwire_decoder *d, const uint8_t **data)
{
    *data = reinterpret_cast<const uint8_t *>(d->ctx->passthrough.data());
    return d->ctx->passthrough.size();
}

size_t refwire_encode(uint8_t type, int has_sequence, int32_t sequence, const char *head, size_t head_len, const uint8_t *body, size_t body_len,
                      int has_body, int ending, uint8_t *out, size_t cap)
{
    Ref<EventFrame> f = jnew<EventFrame>();
    f->type = (jbyte)type; f->hasSequence = has_sequence != 0; f->sequence = sequence; f->isEnding = ending != 0;
    f->head = JString(std::string(head, head_len));
    if (has_body) { Ref<RawBody> rb = jnew<RawBody>(); rb->bytes.assign(reinterpret_cast<const char *>(body), body_len); f->body = rb; }
    Ref<ByteBuf> buf = jnew<ByteBuf>();
    jnew<FrameEncoder>()->encode(nullptr, f, buf);
    if (buf->w > cap) return 0;
    memcpy(out, buf->b.data(), buf->w);
    return buf->w;
}

}  // extern "C"
