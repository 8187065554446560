package io.lubricant.consensus.raft;

import io.lubricant.consensus.raft.command.RaftLog;
import io.lubricant.consensus.raft.command.storage.RocksEntry;
import io.lubricant.consensus.raft.support.serial.Serialization;
import io.lubricant.consensus.raft.RaftResponse;
import io.lubricant.consensus.raft.transport.event.NodeID;

/**
 * Drop into src/test/java of the reference and run ONCE with its own build (mvn -Dtest=KryoVectorsTest test): prints the bytes Kryo 4.0.2 really
 * writes for the RPC bodies (support/serial/Serialization.java:21-62, kryo.writeClassAndObject) — what tests/golden/kryo_bodies.json of the
 * MI355X build must contain. That file was produced WITHOUT a JVM (tests/kryo_ref.py restates Kryo's rules R1-R8); this test is how the pin
 * is made. A line that differs names the rule to fix in rafting_amd/host/kryo_body.cpp and tests/kryo_ref.py (INTEGRATION.md, "Checking the
 * Kryo format"); most likely candidates: R4 (reference tracking of String / byte[]) and R8 (variable-length long fields).
 */
public class KryoVectorsTest {

    static String hex(byte[] b) {
        StringBuilder s = new StringBuilder();
        for (byte x : b) s.append(String.format("%02x", x));
        return s.toString();
    }

    @org.junit.Test
    public void vectors() throws Exception {
        NodeID n0 = new NodeID("127.0.0.1", 6001), n1 = new NodeID("127.0.0.1", 6002);
        RaftLog.Entry[] two = {
            new RocksEntry(7, 42, java.nio.ByteBuffer.allocate(8).putLong(7).array()),
            new RocksEntry(7, 43, java.nio.ByteBuffer.allocate(8).putLong(7).array()) };
        System.out.println("heartbeat            " + hex(Serialization.writeObject(new Object[]{7L, n0, 41L, 7L, new RaftLog.Entry[0], 40L})));
        System.out.println("two_entries_one_term " + hex(Serialization.writeObject(new Object[]{7L, n1, 41L, 7L, two, 41L})));
        System.out.println("requestVote          " + hex(Serialization.writeObject(new Object[]{8L, n1, 41L, 7L})));
        System.out.println("response_success     " + hex(Serialization.writeObject(RaftResponse.success(7))));
        System.out.println("response_failure     " + hex(Serialization.writeObject(RaftResponse.failure(9))));
    }
}
