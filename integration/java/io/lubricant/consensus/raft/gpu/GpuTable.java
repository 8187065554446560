package io.lubricant.consensus.raft.gpu;

import java.nio.ByteBuffer;

/**
 * libraftgpu.so (include/raftgpu.h) as seen from the JVM: one object = one rg_table_t = the raft groups one GPU decides.
 * Every native method is ONE call of the C-ABI (integration/jni/raftgpu_jni.c); buffers are DIRECT ByteBuffers in native byte order laid out
 * exactly as the header says (head 8 B, ab / cd 16 B, abcd 16 B per row; reply / logfx / persist 16 B; rg_out32_t / rg_persist32_t 16 B).
 * Every buffer is checked by the shim before the library sees its address: a heap (non-direct) or short ByteBuffer is an IllegalArgumentException,
 * never a native out-of-bounds access, and never a silently "absent" column.
 * A table is not re-entrant: one flusher thread per table, as one ContextLoop thread per context before (support/EventLoopGroup.java:77-80).
 */
public final class GpuTable implements AutoCloseable {

    static { System.loadLibrary("raftgpu_jni"); }

    public static final int ABI = 5;
    public static final int OPT_REQUIRE_FENCED_TIMEOUTS = 1;

    final long handle;
    public final int groups, cluster, selfSlot;

    public GpuTable(int device, int groups, int cluster, int selfSlot, boolean preVote) {
        if (abiVersion() != ABI) throw new IllegalStateException("libraftgpu ABI " + abiVersion() + ", binding " + ABI);
        this.handle = create(device, groups, cluster, selfSlot, preVote);
        this.groups = groups; this.cluster = cluster; this.selfSlot = selfSlot;
        // a timeout row names the participant whose ticket fired — what RaftRoutine.electionTimeout checks on the timer thread (context/RaftRoutine.java:65-77)
        option(handle, OPT_REQUIRE_FENCED_TIMEOUTS, 1);
    }

    @Override public void close() { destroy(handle); }

    public String lastError() { return lastError(handle); }

    // ---- life cycle ----------------------------------------------------------------------------------------------------------------------
    static native int abiVersion();
    private static native long create(int device, int groups, int cluster, int selfSlot, boolean preVote);
    private static native void destroy(long handle);
    static native String lastError(long handle);
    static native int option(long handle, int option, int value);
    /** page-locked memory for batch and outcome columns (a ByteBuffer.allocateDirect block cannot be pinned afterwards) */
    static native ByteBuffer hostAlloc(long handle, long bytes);
    static native int hostFree(long handle, ByteBuffer buffer);

    // ---- state: RaftContext.initialize (context/RaftContext.java:96-104) for `count` groups at once ------------------------------------
    /** columns: the 24 arrays of rg_group_state_t in declaration order */
    static native int loadState(long handle, int first, int count, ByteBuffer[] columns);
    static native int readState(long handle, int first, int count, ByteBuffer[] columns);

    // ---- the drain (support/EventLoopGroup.java:32-46) -----------------------------------------------------------------------------------
    static native int submit(long handle, int rounds, int count, ByteBuffer gid, ByteBuffer head, ByteBuffer ab, ByteBuffer cd,
                             ByteBuffer entryTerms, long entryCount, ByteBuffer hint, ByteBuffer reply, ByteBuffer logfx, ByteBuffer persist);
    static native int submit32(long handle, int rounds, int count, ByteBuffer gid, ByteBuffer head, ByteBuffer abcd,
                               ByteBuffer entryTerms, long entryCount, ByteBuffer reply, ByteBuffer logfx, ByteBuffer persist);
    static native int submit32c(long handle, int rounds, int count, ByteBuffer head, ByteBuffer abcd, ByteBuffer entryTerms, long entryCount,
                                ByteBuffer row, ByteBuffer persist32, ByteBuffer reply, ByteBuffer logfx, ByteBuffer persist);
    static native int unpack32(int rounds, int count, ByteBuffer row, ByteBuffer persist32, ByteBuffer wideReply, ByteBuffer wideLogfx,
                               ByteBuffer widePersist, ByteBuffer roleEpoch, ByteBuffer reply, ByteBuffer logfx, ByteBuffer persist);
    static native int submitAsyncPacked(long handle, int rounds, int count, ByteBuffer gid, ByteBuffer head, ByteBuffer abcd,
                                        ByteBuffer entryTerms, long entryCount, ByteBuffer reply, ByteBuffer logfx, int logfxCap,
                                        ByteBuffer persist, int persistCap, ByteBuffer counts);
    static native int submitWait(long handle);
    static native int sync(long handle);

    // ---- N1 / N4 ---------------------------------------------------------------------------------------------------------------------------
    static native int replicate(long handle, int count, ByteBuffer gid, ByteBuffer heartbeat, ByteBuffer inFlight, ByteBuffer head, ByteBuffer send);
    static native int timersConfigure(long handle, long electionMs, long heartbeatMs, long seed);
    static native int timersArm(long handle, long now);
    static native int timersUpdate(long handle, int rounds, int count, ByteBuffer gid, ByteBuffer reply, ByteBuffer now);
    /** ABI 5: the same from the compact outcome rows of a dense batch (submit32c's row / persist32 columns), no unpacking in between */
    static native int timersUpdate32(long handle, int rounds, ByteBuffer row, ByteBuffer persist32, ByteBuffer now);
    static native int healthUpdate32(long handle, int rounds, ByteBuffer head, ByteBuffer row, ByteBuffer now);
    /** -> number of expired groups (or < 0); the epochs go into the aux field of the RG_EV_TIMEOUT rows */
    static native int timersExpired(long handle, long now, ByteBuffer outGid, ByteBuffer outEpoch, int capacity);
    static native int healthUpdate(long handle, int rounds, int count, ByteBuffer gid, ByteBuffer head, ByteBuffer reply, ByteBuffer now);
    static native int ready(long handle, long now, int criticalPoint, long coolDownMs, ByteBuffer ready);

    // ---- the device-resident tick (ABI 5): one recorded launch = decisions, timers, follower health, fired tickets, send table, readiness -------------------
    /** Every buffer comes from {@link #hostAlloc} (page-locked, device-addressable) and stays bound to the tick until {@link #tick2Destroy}; heartbeat, inFlight,
     *  the expired* columns, sendHead / send and ready may be null (that step is then not recorded). Refill head / abcd / now, {@link #tick2Launch},
     *  {@link #tick2Wait}, read row / persist32 / the lists. -> tick handle (0: failed, IllegalStateException carries the reason) */
    static native long tick2Create(long handle, int rounds, ByteBuffer head, ByteBuffer abcd, ByteBuffer entryTerms, long entryCapacity, ByteBuffer now,
                                   ByteBuffer heartbeat, ByteBuffer inFlight, int criticalPoint, long coolDownMs, ByteBuffer row, ByteBuffer persist32,
                                   ByteBuffer expiredGid, ByteBuffer expiredEpoch, ByteBuffer expiredCount, int expiredCapacity, ByteBuffer sendHead,
                                   ByteBuffer send, ByteBuffer ready);
    static native int tick2Launch(long tick);
    static native int tick2Wait(long tick);
    static native int tick2Destroy(long tick);
}
