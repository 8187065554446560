package io.lubricant.consensus.raft.gpu;

import io.lubricant.consensus.raft.RaftParticipant;
import io.lubricant.consensus.raft.RaftResponse;
import io.lubricant.consensus.raft.RaftService;
import io.lubricant.consensus.raft.command.RaftLog;
import io.lubricant.consensus.raft.command.RaftLog.Entry;
import io.lubricant.consensus.raft.command.RaftStub.Command;
import io.lubricant.consensus.raft.command.spi.MachineProvider;
import io.lubricant.consensus.raft.command.spi.StateLoader;
import io.lubricant.consensus.raft.context.ContextManager;
import io.lubricant.consensus.raft.context.DecisionEngine;
import io.lubricant.consensus.raft.context.RaftContext;
import io.lubricant.consensus.raft.support.EventLoop.ContextEventLoop;
import io.lubricant.consensus.raft.support.Promise;
import io.lubricant.consensus.raft.support.RaftConfig;
import io.lubricant.consensus.raft.support.StableLock.Persistence;
import io.lubricant.consensus.raft.support.anomaly.NotLeaderException;
import io.lubricant.consensus.raft.support.anomaly.NotReadyException;
import io.lubricant.consensus.raft.transport.RaftCluster;
import io.lubricant.consensus.raft.transport.RaftCluster.ID;
import io.lubricant.consensus.raft.transport.rpc.Async;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayDeque;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.Comparator;
import java.util.List;
import java.util.Map;
import java.util.concurrent.ConcurrentHashMap;
import java.util.concurrent.Executors;
import java.util.concurrent.ScheduledExecutorService;
import java.util.concurrent.TimeUnit;
import java.util.function.Consumer;

/**
 * ContextManager whose contexts are decided by libraftgpu.so instead of by a RaftParticipant object on a ContextLoop thread.
 *
 * Needs integration/patches/0001-decision-engine-seam.patch applied to the reference (the DecisionEngine interface, NettyCluster.on(PingEvent) and
 * RaftStub.process handing over to it, ContextManager.bootstrap as an overridable step of buildContext, RaftContext.attach / decisionEngine() /
 * logFlushed and its two callers in RaftRoutine). tests/test_java_binding_cpu.py applies the patch to a copy of the reference's sources and resolves every
 * import and every call this file makes on a reference type against the patched tree (class exists, method exists with that many parameters).
 *
 * What stays the reference's: the contextId -> RaftContext map and its life cycle (context/ContextManager.java:41,112-171), RaftLog (RocksLog),
 * StableLock, RaftMachine, SnapshotArchive, NettyCluster. What changes is WHO DECIDES: the calls that reach RaftContext.participant()
 * (context/RaftContext.java:169) — NettyCluster.on(PingEvent) for appendEntries / preVote / requestVote / installSnapshot
 * (transport/NettyCluster.java:59-90), the response callbacks of Leader.replicateLog / Candidate.startElection / Follower.prepareElection
 * (member/Leader.java:174-188,218-237, member/Candidate.java:121-134, member/Follower.java:258-270), RaftRoutine's timer callbacks
 * (context/RaftRoutine.java:53-77), RaftStub.process (command/RaftStub.java:79-91) — become ROWS queued here; one flusher thread per GPU drains them
 * with ONE rg_submit per tick and applies every reply row in the reference handler's own order (INTEGRATION.md section 3):
 * log effects -> persist -> commit -> timers / sends -> the RaftResponse.
 *
 * Groups are block-partitioned over the node's GPUs (gid / groupsPerDevice): contexts share nothing (context/ContextManager.java:41,112-120),
 * so there is no traffic between tables.
 *
 * NOT COMPILED HERE (no JDK in the build image). The same flow — queue, sparse submit, hint protocol for RG_NEED_HOST, effects in handler order —
 * is implemented and tested in C++ (rafting_amd/host/raft_host.cpp: ContextManager::flush; tests/devemu/host_flow.cpp).
 */
public class GpuContextManager extends ContextManager implements DecisionEngine {

    // ---- include/raftgpu.h ------------------------------------------------------------------------------------------------------------
    static final int EV_AE_REQ = 1, EV_AE_ACK = 2, EV_IS_ACK = 3, EV_RV_REQ = 4, EV_PV_REQ = 5, EV_RV_REPLY = 6, EV_PV_REPLY = 7,
                     EV_TIMEOUT = 8, EV_CLIENT_APPEND = 9, EV_LOG_FLUSH = 10, EV_IS_REQ = 11;
    static final int F_SUCCESS = 1, F_REPLIED = 1 << 1, F_PERSIST = 1 << 2, F_ROLE_CHANGED = 1 << 3, F_RESET_TIMER = 1 << 4, F_COMMIT = 1 << 5,
                     F_LOG_TRUNC = 1 << 6, F_LOG_APPEND = 1 << 7, F_TIMER_MUTED = 1 << 12;
    static final int EMIT_PREVOTE = 1, EMIT_REQVOTE = 2, EMIT_HEARTBEAT = 3;
    static final int ROLE_FOLLOWER = 0, ROLE_LEADER = 2;
    static final int STATUS_NEED_HOST = 32, HDR_HINT_BIT = 1 << 9, NO_NODE = -1, TERM_RUNS = 4;
    static final int SEND_APPEND = 1, SEND_SNAPSHOT = 2, SEND_NEED_HOST = 4;

    static int hdr(int kind, int slot, boolean flag, int n) { return (kind & 15) | ((slot & 15) << 4) | (flag ? 1 << 8 : 0) | (n << 12); }

    /** one queued RaftParticipant call / callback / timeout / command: the fields of an rg_batch_t row */
    static final class Row {
        int hdr, aux;
        long a, b, c, d;
        long hintX, hintY;
        Entry[] entries;                                 // AppendEntries: what RaftLog.append gets from log_from on (payloads never cross the boundary)
        Consumer<RaftResponse> reply;                    // requests: where the RaftResponse goes (step 6)
        Command command;                                 // CLIENT_APPEND: what RaftContext.acceptCommand stores once the table said LOG_APPEND
        Promise promise;
        int kind() { return hdr & 15; }
    }

    /** what the host keeps per context beside the table's row: the plugins' owner, the RPC fence of the current participant, its timer */
    static final class Group {
        final int gid;                                   // dense id on this node; table = gid / groupsPerDevice, row = gid % groupsPerDevice
        final RaftContext ctx;
        final ArrayDeque<Row> queue = new ArrayDeque<>();// FIFO of the context: at most its first row goes into a flush (support/EventLoopGroup.java:32-46)
        volatile long term;                              // mirror of the last persist row (what participant().currentTerm() / votedFor() answer)
        volatile ID votedFor;
        volatile int role = ROLE_FOLLOWER, roleEpoch = 1;
        Async.AsyncHead fence = Async.head();            // requests of the current participant (Leader.replication / Candidate.election / Follower.qualifier)
        java.util.concurrent.ScheduledFuture<?> timer;
        final short[] inFlight;                          // Leadership.State.requestInFlight per follower
        Group(int gid, RaftContext ctx, int followers) { this.gid = gid; this.ctx = ctx; this.inFlight = new short[followers]; }
    }

    private final int[] devices;
    private final int groupsPerDevice;
    private GpuTable[] tables;
    private ID[] slots;                                  // peer slot -> node id: all nodes sort the same id strings, so slots agree cluster-wide
    private int selfSlot;
    private final Map<String, Group> groups = new ConcurrentHashMap<>();
    private final List<Group> byGid = new ArrayList<>();
    private final Object[] wake;                         // per table: the flusher sleeps here
    private Thread[] flushers;
    private final ScheduledExecutorService timers = Executors.newScheduledThreadPool(1);
    private volatile boolean running = true;

    public GpuContextManager(RaftConfig config, int[] devices, int groupsPerDevice) {
        super(config);
        this.devices = devices.clone();
        this.groupsPerDevice = groupsPerDevice;
        this.wake = new Object[devices.length];
        for (int k = 0; k < devices.length; k++) wake[k] = new Object();
    }

    /** context/ContextManager.java:50-56, plus the tables: cluster size and own slot are known from here on */
    @Override
    public void start(RaftCluster raftCluster, StateLoader stateLoader, MachineProvider machineProvider) throws Exception {
        List<ID> ids = new ArrayList<>(raftCluster.remoteIDs());
        ids.add(raftCluster.localID());
        ids.sort(Comparator.comparing(Object::toString));
        slots = ids.toArray(new ID[0]);
        selfSlot = ids.indexOf(raftCluster.localID());
        tables = new GpuTable[devices.length];
        flushers = new Thread[devices.length];
        for (int k = 0; k < devices.length; k++) {
            tables[k] = new GpuTable(devices[k], groupsPerDevice, raftCluster.size(), selfSlot, config().preVote());
            final int shard = k;
            flushers[k] = new Thread(() -> flushLoop(shard), "GpuFlusher-" + k);
        }
        super.start(raftCluster, stateLoader, machineProvider);
        for (Thread f : flushers) f.start();
    }

    // ---- DecisionEngine ---------------------------------------------------------------------------------------------------------------

    /** the seam of ContextManager.buildContext: bind the loop, restore (term, votedFor), load ONE row of the table instead of `new Follower(...)` */
    @Override
    protected void bootstrap(RaftContext context, ContextEventLoop loop) throws Exception {
        Persistence restored = context.attach(loop, this);
        Group g;
        synchronized (byGid) {
            if (byGid.size() >= devices.length * groupsPerDevice) throw new IllegalStateException("tables are full");
            g = new Group(byGid.size(), context, slots.length - 1);
            byGid.add(g);
        }
        g.term = restored.term;
        g.votedFor = restored.ballot;
        loadGroup(g);
        groups.put(context.ctxID(), g);
        armTimer(g, g.roleEpoch, false);
    }

    @Override
    public void onRequest(RaftContext context, String method, Object[] p, Consumer<RaftResponse> reply) {
        Row row = new Row();
        row.reply = reply;
        row.a = ((Number) p[0]).longValue();
        int from = slotOf((ID) p[1]);
        if ("appendEntries".equals(method)) {            // transport/NettyNode.java:110-123
            Entry[] entries = (Entry[]) p[4];
            row.entries = entries == null ? new Entry[0] : entries;
            row.b = ((Number) p[2]).longValue(); row.c = ((Number) p[3]).longValue(); row.d = ((Number) p[5]).longValue();
            row.hdr = hdr(EV_AE_REQ, from, false, row.entries.length);
        } else if ("installSnapshot".equals(method)) {   // transport/NettyNode.java:147-156; the download is RaftContext.installSnapshot's (it wants the
            row.b = ((Number) p[2]).longValue(); row.c = ((Number) p[3]).longValue();                    // context's loop), the row carries its verdict
            final ID leader = (ID) p[1];
            final Group target = groups.get(context.ctxID());
            context.eventLoop().execute(() -> {
                boolean installed;
                try { installed = context.installSnapshot(leader, row.b, row.c); } catch (Exception e) { installed = false; }
                row.hdr = hdr(EV_IS_REQ, from, installed, 0);
                enqueue(target, row, false);
            });
            return;
        } else {                                         // preVote / requestVote: transport/NettyNode.java:125-145
            row.b = ((Number) p[2]).longValue(); row.c = ((Number) p[3]).longValue();
            row.hdr = hdr("preVote".equals(method) ? EV_PV_REQ : EV_RV_REQ, from, false, 0);
        }
        enqueue(groups.get(context.ctxID()), row, false);
    }

    @Override
    public void onCommand(RaftContext context, Command command, Promise promise) {
        Row row = new Row();
        row.hdr = hdr(EV_CLIENT_APPEND, 0, false, 1);
        row.command = command;
        row.promise = promise;
        enqueue(groups.get(context.ctxID()), row, false);
    }

    /** the device's cache of the stored key window follows the epoch (RG_EV_LOG_FLUSH: a = index, b = term) */
    @Override
    public void onLogFlush(RaftContext context, long index, long term) {
        Row row = new Row();
        row.hdr = hdr(EV_LOG_FLUSH, 0, false, 0);
        row.a = index; row.b = term;
        enqueue(groups.get(context.ctxID()), row, false);
    }

    @Override
    public RaftParticipant view(RaftContext context) {
        final Group g = groups.get(context.ctxID());
        return new RaftParticipant() {                   // what RaftContext.participant() is asked outside the handlers: the term and the vote
            @Override public long currentTerm() { return g.term; }
            @Override public ID votedFor() { return g.votedFor; }
            @Override public RaftResponse appendEntries(long term, ID leaderId, long prevLogIndex, long prevLogTerm, Entry[] entries, long leaderCommit) { throw direct(); }
            @Override public RaftResponse preVote(long term, ID candidateId, long lastLogIndex, long lastLogTerm) { throw direct(); }
            @Override public RaftResponse requestVote(long term, ID candidateId, long lastLogIndex, long lastLogTerm) { throw direct(); }
            @Override public RaftResponse installSnapshot(long term, ID leaderId, long lastIncludedIndex, long lastIncludedTerm) { throw direct(); }
            private IllegalStateException direct() { return new IllegalStateException("decided by the table: go through DecisionEngine.onRequest"); }
        };
    }

    // ---- the queue ----------------------------------------------------------------------------------------------------------------------

    /** urgent = a response callback or a conversion-bearing row: ahead of the requests the context dequeues later (support/EventLoop.java:87-101) */
    private void enqueue(Group g, Row row, boolean urgent) {
        if (g == null || !running) return;
        synchronized (g.queue) { if (urgent) g.queue.addFirst(row); else g.queue.addLast(row); }
        Object w = wake[g.gid / groupsPerDevice];
        synchronized (w) { w.notify(); }
    }

    /** support/EventLoopGroup.java:32-46 for one table: the first row of every context that has one, ascending gid, one sparse rg_submit */
    private void flushLoop(int shard) {
        GpuTable t = tables[shard];
        while (running) {
            List<Group> who = new ArrayList<>();
            List<Row> rows = new ArrayList<>();
            synchronized (byGid) {
                for (int i = shard * groupsPerDevice; i < Math.min(byGid.size(), (shard + 1) * groupsPerDevice); i++) {
                    Group g = byGid.get(i);
                    Row r;
                    synchronized (g.queue) { r = g.queue.pollFirst(); }
                    if (r != null) { who.add(g); rows.add(r); }
                }
            }
            if (rows.isEmpty()) {
                synchronized (wake[shard]) { try { wake[shard].wait(1); } catch (InterruptedException e) { return; } }
                continue;
            }
            submitAndApply(t, who, rows);
        }
    }

    private void submitAndApply(GpuTable t, List<Group> who, List<Row> rows) {
        int n = rows.size();
        ByteBuffer gid = direct(4L * n), head = direct(8L * n), ab = direct(16L * n), cd = direct(16L * n), hint = direct(16L * n);
        ByteBuffer reply = direct(16L * n), logfx = direct(16L * n), persist = direct(16L * n);
        long entryCount = 0;
        for (Row r : rows) entryCount += r.entries == null ? 0 : r.entries.length;
        ByteBuffer terms = entryCount == 0 ? null : direct(8L * entryCount);
        long at = 0;
        for (int i = 0; i < n; i++) {
            Row r = rows.get(i);
            gid.putInt(4 * i, who.get(i).gid % groupsPerDevice);
            head.putInt(8 * i, r.hdr).putInt(8 * i + 4, r.entries != null && r.entries.length > 0 ? (int) at : r.aux);
            ab.putLong(16 * i, r.a).putLong(16 * i + 8, r.b);
            cd.putLong(16 * i, r.c).putLong(16 * i + 8, r.d);
            hint.putLong(16 * i, r.hintX).putLong(16 * i + 8, r.hintY);
            if (r.entries != null) for (Entry e : r.entries) terms.putLong((int) (8 * at++), e.term());
        }
        if (GpuTable.submit(t.handle, 1, n, gid, head, ab, cd, terms, entryCount, hint, reply, logfx, persist) != 0)
            throw new IllegalStateException(t.lastError());
        // Every row's effects run ON ITS CONTEXT'S LOOP THREAD — RaftLog, RaftContext.commitLog / acceptCommand are single-threaded per context and assert
        // it (context/RaftContext.java:222-224,245-247) — and the flush waits for all of them: flush k's effects are complete before flush k + 1 is decided.
        final List<Send> sends = java.util.Collections.synchronizedList(new ArrayList<Send>());
        final java.util.concurrent.CountDownLatch applied = new java.util.concurrent.CountDownLatch(n);
        for (int i = 0; i < n; i++) {
            final Group g = who.get(i);
            final Row row = rows.get(i);
            final long respTerm = reply.getLong(16 * i), commitIndex = logfx.getLong(16 * i), logFrom = logfx.getLong(16 * i + 8), term = persist.getLong(16 * i);
            final int flags = reply.getInt(16 * i + 8), roleEpoch = reply.getInt(16 * i + 12), votedFor = persist.getInt(16 * i + 8);
            g.ctx.eventLoop().enforce(() -> {
                try { applyOutcome(g, row, respTerm, flags, roleEpoch, commitIndex, logFrom, term, votedFor, sends); } finally { applied.countDown(); }
            });
        }
        try { applied.await(); } catch (InterruptedException e) { Thread.currentThread().interrupt(); return; }
        if (!sends.isEmpty()) replicate(t, sends);
    }

    /** a leader whose row asked for the send side: Leader.onTimeout (heartbeat limits) or an accepted command */
    static final class Send {
        final Group g; final boolean heartbeat;
        Send(Group g, boolean heartbeat) { this.g = g; this.heartbeat = heartbeat; }
    }

    /** INTEGRATION.md section 3, steps 1-6, for one row */
    private void applyOutcome(Group g, Row row, long respTerm, int flags, int roleEpoch, long commitIndex, long logFrom, long term, int votedFor, List<Send> sends) {
        final RaftContext ctx = g.ctx;
        final int status = (flags >>> 16) & 0xFF, kind = row.kind();
        try {
            if (status == STATUS_NEED_HOST) { resubmitWithHint(g, row, logFrom); return; }              // 1: the term the table's run cache does not hold
            RaftLog log = ctx.replicatedLog();
            if ((flags & F_LOG_TRUNC) != 0) log.truncate(logFrom);                                       // 2: storage/RocksLog.java:219-225
            if ((flags & F_LOG_APPEND) != 0) {
                if (kind == EV_CLIENT_APPEND) {                                                          //    member/Leader.java:128-140: newEntry + the promise
                    ctx.acceptCommand(g.term, row.command, row.promise);
                } else {
                    log.append(entriesFrom(row.entries, logFrom));                                       //    storage/RocksLog.java:169-196
                }
            }
            if ((flags & F_PERSIST) != 0) {                                                              // 3: BEFORE the reply (member/RaftMember.java:25)
                ID ballot = votedFor == NO_NODE ? null : slots[votedFor];
                ctx.stableStorage().persist(term, ballot);
                g.term = term; g.votedFor = ballot;
            }
            if ((flags & F_COMMIT) != 0) ctx.commitLog(commitIndex, kind == EV_AE_REQ);                  // 4: context/RaftContext.java:244-255 (Follower: passive; Leader.tryCommit: not)
            // 5: timers, the old participant's requests, what the new one broadcasts
            final int newRole = (flags >>> 10) & 3;
            if ((flags & F_ROLE_CHANGED) != 0) {
                g.fence.abortRequests();                                                                  //    onFencing (context/RaftRoutine.java:192-197)
                g.fence = Async.head();
                Arrays.fill(g.inFlight, (short) 0);
                if (newRole == ROLE_LEADER || g.role == ROLE_LEADER) ctx.abortPromise();                 //    member/Leader.java:27,115
            }
            g.role = newRole;
            g.roleEpoch = roleEpoch;
            if ((flags & F_RESET_TIMER) != 0) armTimer(g, roleEpoch, (flags & F_TIMER_MUTED) != 0);
            int emit = (flags >>> 8) & 3;
            if (emit == EMIT_PREVOTE) broadcastVote(g, true, g.term + 1, roleEpoch);                     //    member/Follower.java:223-279
            else if (emit == EMIT_REQVOTE) broadcastVote(g, false, g.term, roleEpoch);                   //    member/Candidate.java:90-143
            else if (emit == EMIT_HEARTBEAT) sends.add(new Send(g, true));                               //    member/Leader.java:120-126
            if (kind == EV_CLIENT_APPEND) {
                if ((flags & F_LOG_APPEND) != 0) sends.add(new Send(g, false));                          //    member/Leader.java:135
                else row.promise.completeExceptionally(newRole == ROLE_LEADER ? new NotReadyException() : new NotLeaderException(view(ctx)));
            }
            if (row.reply != null && (flags & F_REPLIED) != 0)                                           // 6: transport/NettyCluster.java:79-81
                row.reply.accept(RaftResponse.reply(respTerm, (flags & F_SUCCESS) != 0));
            // (a request row without F_REPLIED: the reference's handler died with an AssertionError — status says where — and sent nothing)
        } catch (Exception e) {
            if (row.promise != null) row.promise.completeExceptionally(e);
        }
    }

    private static Entry[] entriesFrom(Entry[] entries, long logFrom) {
        int k = 0;
        while (k < entries.length && entries[k].index() < logFrom) k++;
        return Arrays.copyOfRange(entries, k, entries.length);
    }

    /** step 1: hint = (term of the index the table asked for | -1, index of the first conflicting entry | 0); the row goes back to the head of its queue */
    private void resubmitWithHint(Group g, Row row, long index) throws Exception {
        RaftLog log = g.ctx.replicatedLog();
        Entry at = log.get(index);
        row.hintX = at == null ? -1 : at.term();
        row.hintY = 0;
        if (row.kind() == EV_AE_REQ && row.entries.length > 0) {
            Entry conflict = log.conflict(row.entries);                                                  // storage/RocksLog.java:199-216
            row.hintY = conflict == null ? 0 : conflict.index();
        }
        row.hdr |= HDR_HINT_BIT;
        enqueue(g, row, true);
    }

    // ---- state in: StableLock.restore + RaftLog -> one row of rg_load_state (context/RaftContext.java:96-104) ----------------------------

    private void loadGroup(Group g) throws Exception {
        RaftLog log = g.ctx.replicatedLog();
        Entry epoch = log.epoch(), last = log.last();
        int f = slots.length - 1;
        long[] runStart = new long[TERM_RUNS], runTerm = new long[TERM_RUNS];
        int runs = 0;
        long first = epoch.index() + 1;
        if (last != null) {                                                                              // the newest TERM_RUNS maximal equal-term runs, newest first
            long i = last.index(), term = last.term(), start = i;
            for (; i > epoch.index() && runs < TERM_RUNS; i--) {
                Entry e = log.get(i);
                if (e == null) break;
                if (e.term() != term) { runStart[runs] = start; runTerm[runs] = term; runs++; term = e.term(); }
                start = i;
            }
            if (runs < TERM_RUNS) { runStart[runs] = start; runTerm[runs] = term; runs++; }
            if (log.get(epoch.index()) != null) first = epoch.index();                                   // the key equal to the epoch survives deleteRange (RocksLog.java:235)
        }
        ByteBuffer[] col = new ByteBuffer[24];
        col[0] = direct(8).putLong(0, g.term);                                   // current_term
        col[1] = direct(4).putInt(0, slotOf(g.votedFor));                        // voted_for
        col[2] = direct(4).putInt(0, ROLE_FOLLOWER);                             // role: RaftContext.initialize switches to Follower (context/RaftContext.java:104)
        col[3] = direct(4).putInt(0, NO_NODE);                                   // current_leader
        col[4] = direct(1); col[5] = direct(1);                                  // timeout_detected, repl_prepared
        col[6] = direct(4).putInt(0, 1);                                         // role_epoch
        col[7] = direct(4).putInt(0, 1);                                         // votes
        col[8] = direct(4); col[9] = direct(8);                                  // elected_epoch, elected_term
        col[10] = direct(8).putLong(0, log.lastCommitted());                     // commit_index
        col[11] = direct(8).putLong(0, epoch.index());
        col[12] = direct(8).putLong(0, epoch.term());
        col[13] = direct(8).putLong(0, first);
        col[14] = direct(8).putLong(0, last == null ? 0 : last.index());
        col[15] = direct(4).putInt(0, runs);                                     // run_count (0 <=> log empty)
        col[16] = direct(4);                                                     // run_offset
        col[17] = direct(8L * TERM_RUNS); col[18] = direct(8L * TERM_RUNS);
        for (int k = 0; k < runs; k++) {                                         // ascending
            col[17].putLong(8 * k, runStart[runs - 1 - k]);
            col[18].putLong(8 * k, runTerm[runs - 1 - k]);
        }
        col[19] = direct(8L * f); col[20] = direct(8L * f); col[21] = direct(8L * f); col[22] = direct(4L * f); col[23] = direct(f);
        GpuTable t = tables[g.gid / groupsPerDevice];
        if (GpuTable.loadState(t.handle, g.gid % groupsPerDevice, 1, col) != 0) throw new IllegalStateException(t.lastError());
    }

    // ---- timers: RaftRoutine.resetTimer / electionTimeout / keepAlive (context/RaftRoutine.java:53-130) on ONE scheduler thread ------------
    // (the device-side wheel — GpuTable.timersUpdate / timersExpired — replaces this for large tables: INTEGRATION.md section 1)

    private void armTimer(Group g, int roleEpoch, boolean muted) {
        if (g.timer != null) g.timer.cancel(false);
        g.timer = null;
        if (muted) return;                                                                              // resetTimer(…, true): Long.MAX_VALUE
        long delay = g.role == ROLE_LEADER ? config().heartbeatInterval() : config().electionTimeout();     // (the draw from [E, 2E] is RaftConfig's own: support/RaftConfig.java:187-190)
        g.timer = timers.schedule(() -> {
            Row row = new Row();
            row.hdr = hdr(EV_TIMEOUT, 0, false, 0);
            row.aux = roleEpoch;                                                                         // the fence: context/RaftRoutine.java:70
            enqueue(g, row, false);
        }, delay, TimeUnit.MILLISECONDS);
    }

    // ---- the send side ------------------------------------------------------------------------------------------------------------------------

    /** Follower.prepareElection / Candidate.startElection: the broadcast and its callbacks (member/Follower.java:243-275, member/Candidate.java:110-139) */
    private void broadcastVote(Group g, boolean pre, long term, int roleEpoch) throws Exception {
        RaftContext ctx = g.ctx;
        Entry last = ctx.replicatedLog().last();
        if (last == null) last = ctx.replicatedLog().epoch();
        final long lastIndex = last.index(), lastTerm = last.term(), timeout = ctx.envConfig().broadcastTimeout();
        final Async.AsyncHead fence = g.fence;
        if (fence.isAborted()) return;
        for (ID id : ctx.cluster().remoteIDs()) {
            RaftService service = ctx.cluster().remoteService(id, ctx.ctxID());
            if (service == null) continue;
            Async<RaftResponse> response = pre ? service.preVote(term, ctx.nodeID(), lastIndex, lastTerm) : service.requestVote(term, ctx.nodeID(), lastIndex, lastTerm);
            final int slot = slotOf(id);
            response.on(fence, timeout, (result, error, canceled) -> {
                if (canceled || error != null || result == null) return;
                Row row = new Row();
                row.hdr = hdr(pre ? EV_PV_REPLY : EV_RV_REPLY, slot, result.success(), 0);
                row.aux = roleEpoch;
                row.a = result.term();
                enqueue(g, row, true);
            });
        }
    }

    /** Leader.replicateLog (member/Leader.java:142-245) for the leaders of this flush: ONE rg_replicate says what every follower gets */
    private void replicate(GpuTable t, List<Send> leaders) {
        // rg_replicate wants strictly ascending group ids; a group appears once (a heartbeat and an accepted command in one flush cannot both be its row)
        Send[] order = leaders.toArray(new Send[0]);
        Arrays.sort(order, Comparator.comparingInt((Send x) -> x.g.gid));
        int n = order.length, f = slots.length - 1;
        ByteBuffer gid = direct(4L * n), hb = direct(n), inFlight = direct(2L * f * n), head = direct(48L * n), send = direct(32L * f * n);
        for (int r = 0; r < n; r++) {
            Group g = order[r].g;
            gid.putInt(4 * r, g.gid % groupsPerDevice);
            hb.put(r, (byte) (order[r].heartbeat ? 1 : 0));
            for (int j = 0; j < f; j++) inFlight.putShort(2 * (j * n + r), g.inFlight[j]);
        }
        if (GpuTable.replicate(t.handle, n, gid, hb, inFlight, head, send) != 0) throw new IllegalStateException(t.lastError());
        for (int r = 0; r < n; r++) {
            Group g = order[r].g;
            if (head.getInt(48 * r + 36) == 0) continue;                                                 // is_leader
            final long term = head.getLong(48 * r), leaderCommit = head.getLong(48 * r + 8), epochIndex = head.getLong(48 * r + 16), epochTerm = head.getLong(48 * r + 24);
            final int roleEpoch = head.getInt(48 * r + 32);
            for (int j = 0; j < f; j++) {
                int at = 32 * (j * n + r);
                try {
                    ship(g, j, send.getInt(at + 28), term, leaderCommit, epochIndex, epochTerm, roleEpoch, send.getLong(at), send.getLong(at + 8), send.getLong(at + 16), send.getInt(at + 24));
                } catch (Exception e) { /* RaftCtx invoke appendEntries failed: member/Leader.java:240-242 */ }
            }
        }
    }

    private void ship(Group g, int j, int kind, long term, long leaderCommit, long epochIndex, long epochTerm, int roleEpoch,
                      long prevIndex, long prevTerm, long lastIndex, int count) throws Exception {
        final RaftContext ctx = g.ctx;
        final int slot = j < selfSlot ? j : j + 1;
        final ID id = slots[slot];
        RaftService service = ctx.cluster().remoteService(id, ctx.ctxID());
        if (service == null || (kind != SEND_APPEND && kind != SEND_SNAPSHOT && kind != SEND_NEED_HOST)) return;
        final long timeout = ctx.envConfig().broadcastTimeout();
        final Async.AsyncHead fence = g.fence;
        if (kind == SEND_NEED_HOST) {                                                                    // the term of prev_index lies below the cached runs
            Entry prev = ctx.replicatedLog().get(prevIndex);
            if (prev == null) return;
            prevTerm = prev.term();
            kind = SEND_APPEND;
        }
        final boolean snapshot = kind == SEND_SNAPSHOT;
        Async<RaftResponse> response;
        if (snapshot) {
            response = service.installSnapshot(term, ctx.nodeID(), epochIndex, epochTerm);
        } else {
            Entry[] entries = count == 0 ? new Entry[0] : ctx.replicatedLog().batch(prevIndex + 1, count);
            response = service.appendEntries(term, ctx.nodeID(), prevIndex, prevTerm, entries, leaderCommit);
        }
        g.inFlight[j]++;
        response.on(fence, timeout, (result, error, canceled) -> {
            g.inFlight[j]--;
            if (canceled || error != null || result == null) return;                                    // statFailure: GpuTable health columns, not a decision
            Row row = new Row();
            row.hdr = hdr(snapshot ? EV_IS_ACK : EV_AE_ACK, slot, result.success(), 0);
            row.aux = roleEpoch;
            row.a = result.term(); row.b = epochIndex; row.c = lastIndex;
            enqueue(g, row, true);
        });
    }

    // ---- life cycle ---------------------------------------------------------------------------------------------------------------------------

    @Override
    public synchronized void close() throws Exception {
        running = false;
        timers.shutdownNow();
        if (flushers != null) for (Thread f : flushers) { f.interrupt(); f.join(); }
        if (tables != null) for (GpuTable t : tables) t.close();
        super.close();
    }

    private int slotOf(ID id) {
        if (id == null) return NO_NODE;
        for (int s = 0; s < slots.length; s++) if (slots[s].equals(id)) return s;
        throw new IllegalArgumentException("not a member of this cluster: " + id);
    }

    private static ByteBuffer direct(long bytes) { return ByteBuffer.allocateDirect((int) Math.max(bytes, 1)).order(ByteOrder.nativeOrder()); }

}
