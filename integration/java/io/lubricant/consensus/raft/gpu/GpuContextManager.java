package io.lubricant.consensus.raft.gpu;

import io.lubricant.consensus.raft.command.RaftLog;
import io.lubricant.consensus.raft.context.ContextManager;
import io.lubricant.consensus.raft.context.RaftContext;
import io.lubricant.consensus.raft.support.RaftConfig;
import io.lubricant.consensus.raft.support.StableLock;
import io.lubricant.consensus.raft.transport.RaftResponse;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayDeque;
import java.util.Map;
import java.util.concurrent.CompletableFuture;
import java.util.concurrent.ConcurrentHashMap;

/**
 * ContextManager whose contexts are decided by libraftgpu.so instead of by a RaftParticipant object on a ContextLoop thread.
 *
 * What stays the reference's: the contextId -> RaftContext map and its life cycle (context/ContextManager.java:41,112-171), RaftLog (RocksLog),
 * StableLock, RaftMachine, SnapshotArchive, NettyCluster. What changes is WHO DECIDES: the calls that reach RaftContext.participant()
 * (context/RaftContext.java:169) — NettyCluster.on(PingEvent) for appendEntries / preVote / requestVote (transport/NettyCluster.java:59-90),
 * the response callbacks of Leader.replicateLog / Candidate.startElection / Follower.prepareElection (member/Leader.java:174-188,218-237,
 * member/Candidate.java:121-134, member/Follower.java:258-270), RaftRoutine's timer callbacks (context/RaftRoutine.java:53-77) — become ROWS
 * queued here; one flusher thread per GPU drains them with ONE rg_submit per tick and applies every reply row in the reference handler's own
 * order (INTEGRATION.md section 3): log effects -> persist -> commit -> timers / sends -> the RaftResponse.
 *
 * Groups are block-partitioned over the node's GPUs (gid / groupsPerDevice): contexts share nothing (context/ContextManager.java:41,112-120),
 * so there is no traffic between tables.
 *
 * UNTESTED HERE (no JDK in the build image). The same flow — queue, sparse submit, hint protocol for RG_NEED_HOST, effects in handler order —
 * is implemented and tested in C++ (rafting_amd/host/raft_host.cpp: ContextManager::flush; tests/devemu/host_flow.cpp).
 */
public class GpuContextManager extends ContextManager {

    /** one queued RaftParticipant call / callback / timeout: the fields of an rg_batch_t row (include/raftgpu.h) */
    static final class Row {
        int gid, hdr, aux;
        long a, b, c, d;
        long[] entryTerms;                               // AppendEntries: the term of every carried entry (payloads stay with the request)
        Object request;                                  // what the reply is released to: the PingEvent's invocation, or null for callbacks / timeouts
        CompletableFuture<RaftResponse> reply;           // completed in step 6 (requests only)
    }

    private final GpuTable[] tables;
    private final int groupsPerDevice;
    private final Map<String, Integer> gidOf = new ConcurrentHashMap<>();
    private final ArrayDeque<Row>[] pending;             // per table; a context contributes at most one row per flush (its FIFO)
    private final Thread[] flushers;
    private volatile boolean running = true;
    private int nextGid = 0;

    @SuppressWarnings("unchecked")
    public GpuContextManager(RaftConfig config, int[] devices, int groupsPerDevice) {
        super(config);
        this.groupsPerDevice = groupsPerDevice;
        this.tables = new GpuTable[devices.length];
        this.pending = new ArrayDeque[devices.length];
        this.flushers = new Thread[devices.length];
        // cluster size and own slot come from the RaftCluster at start(); RaftConfig.preVote() is support/RaftConfig.java:183-185
        for (int k = 0; k < devices.length; k++) {
            tables[k] = new GpuTable(devices[k], groupsPerDevice, clusterSizeOf(config), selfSlotOf(config), config.preVote());
            pending[k] = new ArrayDeque<>();
            final int shard = k;
            flushers[k] = new Thread(() -> flushLoop(shard), "GpuFlusher-" + k);
        }
    }

    @Override
    public synchronized RaftContext createContext(String contextId) {
        RaftContext ctx = super.createContext(contextId);            // the reference builds log, lock, machine, snapshot archive as ever
        gidOf.computeIfAbsent(contextId, id -> {
            int gid = nextGid++;
            loadGroup(gid, ctx);                                      // StableLock.restore + RaftLog.epoch / last -> one row of rg_load_state
            return gid;
        });
        return ctx;
    }

    /** NettyCluster.on(PingEvent), the response callbacks and the timer callbacks call this instead of touching a participant */
    public CompletableFuture<RaftResponse> enqueue(String contextId, Row row) {
        Integer gid = gidOf.get(contextId);
        if (gid == null) throw new IllegalStateException("no such context: " + contextId);
        row.gid = gid % groupsPerDevice;
        row.reply = new CompletableFuture<>();
        ArrayDeque<Row> q = pending[gid / groupsPerDevice];
        synchronized (q) { q.add(row); q.notify(); }
        return row.reply;
    }

    /** support/EventLoopGroup.java:32-46 for one table: swap the queue, one sparse rg_submit, apply the outcomes */
    private void flushLoop(int shard) {
        GpuTable t = tables[shard];
        ArrayDeque<Row> q = pending[shard];
        while (running) {
            Row[] batch;
            synchronized (q) {
                while (q.isEmpty() && running) { try { q.wait(1); } catch (InterruptedException e) { return; } }
                batch = takeOnePerGroup(q);                          // ascending gid, at most one row per group: the rest wait for the next flush
            }
            if (batch.length == 0) continue;
            int n = batch.length;
            ByteBuffer gid = direct(4L * n), head = direct(8L * n), ab = direct(16L * n), cd = direct(16L * n), hint = direct(16L * n);
            ByteBuffer reply = direct(16L * n), logfx = direct(16L * n), persist = direct(16L * n);
            long entries = 0;
            for (Row r : batch) entries += r.entryTerms == null ? 0 : r.entryTerms.length;
            ByteBuffer terms = entries == 0 ? null : direct(8L * entries);
            long at = 0;
            for (int i = 0; i < n; i++) {
                Row r = batch[i];
                gid.putInt(4 * i, r.gid);
                head.putInt(8 * i, r.hdr).putInt(8 * i + 4, r.entryTerms == null ? r.aux : (int) at);
                ab.putLong(16 * i, r.a).putLong(16 * i + 8, r.b);
                cd.putLong(16 * i, r.c).putLong(16 * i + 8, r.d);
                if (r.entryTerms != null) for (long term : r.entryTerms) terms.putLong((int) (8 * at++), term);
            }
            if (GpuTable.submit(t.handle, 1, n, gid, head, ab, cd, terms, entries, hint, reply, logfx, persist) != 0)
                throw new IllegalStateException(t.lastError());
            for (int i = 0; i < n; i++) {
                long respTerm = reply.getLong(16 * i);
                int flags = reply.getInt(16 * i + 8), roleEpoch = reply.getInt(16 * i + 12);
                applyOutcome(batch[i], respTerm, flags, roleEpoch, logfx.getLong(16 * i), logfx.getLong(16 * i + 8),
                             persist.getLong(16 * i), persist.getInt(16 * i + 8));
            }
        }
    }

    /** INTEGRATION.md section 3, steps 1-6, for one row */
    private void applyOutcome(Row row, long respTerm, int flags, int roleEpoch, long commitIndex, long logFrom, long term, int votedFor) {
        int status = (flags >>> 16) & 0xFF;
        if (status == 32 /* RG_NEED_HOST */) { resubmitWithHint(row, logFrom); return; }            // 1: the term of logFrom from the host's RaftLog
        RaftLog log = logOf(row.gid);
        try {
            if ((flags & (1 << 6)) != 0) log.truncate(logFrom);                                      // 2: RG_F_LOG_TRUNC (storage/RocksLog.java:219-225)
            if ((flags & (1 << 7)) != 0) appendFrom(log, row, logFrom);                              //    RG_F_LOG_APPEND (storage/RocksLog.java:169-196)
            if ((flags & (1 << 2)) != 0) lockOf(row.gid).persist(term, nodeOf(votedFor));            // 3: RG_F_PERSIST, BEFORE the reply (member/RaftMember.java:25)
            if ((flags & (1 << 5)) != 0) commit(row.gid, commitIndex);                               // 4: RG_F_COMMIT (context/RaftContext.java:244-255)
            react(row.gid, flags, roleEpoch);                                                        // 5: timers, broadcasts, abort the old role's Asyncs
            if ((flags & (1 << 1)) != 0)                                                             // 6: RG_F_REPLIED
                row.reply.complete((flags & 1) != 0 ? RaftResponse.success(respTerm) : RaftResponse.failure(respTerm));
            else
                row.reply.complete(null);                                                            // the handler died (status says where): log, send nothing
        } catch (Exception e) {
            row.reply.completeExceptionally(e);
        }
    }

    @Override
    public synchronized void close() throws Exception {
        running = false;
        for (Thread f : flushers) if (f != null) f.join();
        for (GpuTable t : tables) t.close();
        super.close();
    }

    // ---- glue to the reference's plugins: what a maintainer fills in against accessors of RaftContext (its fields are private today) ----------
    private static ByteBuffer direct(long bytes) { return ByteBuffer.allocateDirect((int) bytes).order(ByteOrder.nativeOrder()); }
    private Row[] takeOnePerGroup(ArrayDeque<Row> q) { throw new UnsupportedOperationException("sort by gid, keep the first row of every group, leave the rest queued"); }
    private void loadGroup(int gid, RaftContext ctx) { throw new UnsupportedOperationException("GpuTable.loadState from StableLock.restore() and RaftLog.epoch() / last()"); }
    private void resubmitWithHint(Row row, long index) { throw new UnsupportedOperationException("hint = (log.get(index).term | -1, conflict index | 0); RG_HDR_HINT_BIT; enqueue again"); }
    private void appendFrom(RaftLog log, Row row, long logFrom) { throw new UnsupportedOperationException("log.append(entries of row.request with index >= logFrom)"); }
    private void commit(int gid, long index) { throw new UnsupportedOperationException("RaftContext.commitLog: log.markCommitted + routine.commitState"); }
    private void react(int gid, int flags, int roleEpoch) { throw new UnsupportedOperationException("RG_F_RESET_TIMER / RG_F_EMIT / RG_F_ROLE_CHANGED"); }
    private RaftLog logOf(int gid) { throw new UnsupportedOperationException(); }
    private StableLock lockOf(int gid) { throw new UnsupportedOperationException(); }
    private io.lubricant.consensus.raft.transport.RaftCluster.ID nodeOf(int slot) { throw new UnsupportedOperationException(); }
    private static int clusterSizeOf(RaftConfig config) { throw new UnsupportedOperationException("RaftCluster.size() (transport/NettyCluster.java:129-132)"); }
    private static int selfSlotOf(RaftConfig config) { throw new UnsupportedOperationException("index of the local NodeID in the sorted cluster"); }
}
