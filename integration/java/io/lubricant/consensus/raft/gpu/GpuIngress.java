package io.lubricant.consensus.raft.gpu;

/**
 * libraftwire.so's ingress (include/raftwire.h, rw_ingress_*): the bytes of all peer channels -> the dense [round][group] batch the step
 * kernel takes -> response frames. Replaces EventCodec.FrameDecoder + NettyCluster.on(PingEvent / PongEvent) for the raft channels
 * (transport/EventCodec.java:169-335, transport/NettyCluster.java:59-105) with one handler that hands the raw bytes down: no Java object per RPC.
 * The flusher thread's half (seal / emit / repair) is reached the same way; INTEGRATION.md section 1 holds the loop.
 */
public final class GpuIngress {
    private GpuIngress() {}

    /** bytes of connection `conn` as they arrive (address = ByteBuf.memoryAddress() + readerIndex()); < 0: the frame grammar was violated, close the channel */
    static native int feed(long ingress, int conn, long address, int length);
    /** a row that does not come off the wire: RG_EV_TIMEOUT (aux = the fired ticket's role epoch), RG_EV_CLIENT_APPEND, RG_EV_LOG_FLUSH, a released RG_EV_IS_REQ */
    static native int addRow(long ingress, int conn, int gid, int hdr, int aux, long a, long b, long c, long d, int replyConn, int replySequence);
    static native int recycle(long ingress, int bank);
    static native long held(long ingress);
}
