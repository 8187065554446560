"""An independent restatement (Python) of the slice of Kryo 4.0.2's byte format the reference's RPC bodies use — test infrastructure.

Written from the same published rules as rafting_amd/host/kryo_body.cpp (Output.writeVarInt / writeVarLong / writeString,
DefaultClassResolver.writeClass / writeName, Kryo.writeReferenceOrNull, ObjectArraySerializer, ByteArraySerializer, FieldSerializer's
fields-by-name order) but sharing no code with it: tests/test_kryo_cpu.py requires the C++ codec to produce these bytes and to read them
back. NOT VERIFIED AGAINST A JVM: Kryo is a Java library that is absent from this image; INTEGRATION.md shows the JUnit check."""

OBJECT_ARRAY = "[Ljava.lang.Object;"
NODE_ID = "io.lubricant.consensus.raft.transport.event.NodeID"
ENTRY_ARRAY = "[Lio.lubricant.consensus.raft.command.RaftLog$Entry;"
ROCKS_ENTRY = "io.lubricant.consensus.raft.command.storage.RocksEntry"
RESPONSE = "io.lubricant.consensus.raft.RaftResponse"
ID_LONG = 7


def varint(v):
    assert 0 <= v < 1 << 32
    out = bytearray()
    while v >> 7:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def zigzag32(v):
    return ((v << 1) ^ (v >> 31)) & 0xFFFFFFFF


def varlong_zz(v):
    u = ((v << 1) ^ (v >> 63)) & 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    for _ in range(8):
        if u >> 7 == 0:
            break
        out.append((u & 0x7F) | 0x80)
        u >>= 7
    out.append(u & 0xFF)
    return bytes(out)


def string(s):
    if s == "":
        return b"\x81"
    raw = s.encode("utf-8")
    if 1 < len(s) < 64 and all(ord(c) <= 127 for c in s):
        return raw[:-1] + bytes([raw[-1] | 0x80])
    v = len(s) + 1
    if v >> 6 == 0:
        head = bytes([v | 0x80])
    else:
        head = bytearray([(v & 0x3F) | 0x40 | 0x80])
        v >>= 6
        while v >> 7:
            head.append((v & 0x7F) | 0x80)
            v >>= 7
        head.append(v)
        head = bytes(head)
    return head + raw


class Writer:
    def __init__(self):
        self.out, self.names = bytearray(), []

    def klass(self, name):
        self.out += varint(1)
        if name in self.names:
            self.out += varint(self.names.index(name))
        else:
            self.out += varint(len(self.names))
            self.names.append(name)
            self.out += string(name)

    def long(self, v):
        self.out += varint(ID_LONG + 2) + varlong_zz(v)


def request(nodes, method_is_append, term, node, x, y, leader_commit=0, entry_terms=()):
    w = Writer()
    w.klass(OBJECT_ARRAY)
    w.out += varint(1) + varint((6 if method_is_append else 4) + 1)
    w.long(term)
    host, port = nodes[node]
    w.klass(NODE_ID)
    w.out += varint(1) + varint(1) + string(host) + varint(zigzag32(port))
    w.long(x)
    w.long(y)
    if method_is_append:
        w.klass(ENTRY_ARRAY)
        w.out += varint(1) + varint(len(entry_terms) + 1)
        for k, t in enumerate(entry_terms):
            w.klass(ROCKS_ENTRY)
            w.out += varint(1) + varint(1) + varint(8 + 1) + (t & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "big")
            w.out += varlong_zz(x + 1 + k) + varlong_zz(t)
        w.long(leader_commit)
    return bytes(w.out)


def response(term, success):
    w = Writer()
    w.klass(RESPONSE)
    w.out += varint(1) + bytes([1 if success else 0]) + varlong_zz(term)
    return bytes(w.out)
