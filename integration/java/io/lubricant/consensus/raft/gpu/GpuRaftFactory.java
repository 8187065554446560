package io.lubricant.consensus.raft.gpu;

import io.lubricant.consensus.raft.context.ContextManager;
import io.lubricant.consensus.raft.support.RaftConfig;
import io.lubricant.consensus.raft.support.RaftFactory;

/**
 * The whole substitution is one factory method: RaftFactory.resumeContext() is where the reference makes its ContextManager
 * (support/RaftFactory.java:22-24); everything else — StateLoader, NettyCluster, MachineProvider, bootstrap() — stays the reference's.
 * An application that extended RaftFactory before extends GpuRaftFactory instead and keeps its restartMachine().
 *
 * UNTESTED HERE: this image has no JDK. The C side of every native method below is type-checked and exercised (integration/README.md).
 */
public abstract class GpuRaftFactory extends RaftFactory {

    /** HIP device ordinals this node decides on; groups are block-partitioned over them (one table, stream and flusher thread per GPU, no collective) */
    protected int[] devices() { return new int[]{0}; }

    /** capacity of one table: RaftContexts per GPU (the reference has no bound; a table is allocated once: 272 B per group) */
    protected int groupsPerDevice() { return 1 << 17; }

    @Override
    public ContextManager resumeContext(RaftConfig config) throws Exception {
        return new GpuContextManager(config, devices(), groupsPerDevice());
    }
}
