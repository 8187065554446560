"""ctypes binding of build/libraftwire.so (include/raftwire.h): the host-side wire decoder and the ingress that lays decoded RPCs out as the
multi-round compact batches the step kernel takes (rafting_amd/host/ingress.hpp). Host C++ only — nothing here touches a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "build", "libraftwire.so")
M_APPEND_ENTRIES, M_PRE_VOTE, M_REQUEST_VOTE, M_INSTALL_SNAPSHOT = 1, 2, 3, 4
METHOD_NAME = {1: b"appendEntries", 2: b"preVote", 3: b"requestVote", 4: b"installSnapshot"}
ENQ, ACK = 0x05, 0x06
NO_CONN = 0xFFFFFFFF
_sz, _vp, _u32, _i32, _i64, _u64 = C.c_size_t, C.c_void_p, C.c_uint32, C.c_int32, C.c_int64, C.c_uint64
_lib = None


def lib():
    global _lib
    if _lib is None:
        # (a no-op when the library is newer than its sources; a stale library would disagree with include/raftwire.h and this binding)
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rafting_amd", "host"), os.path.join("..", "..", "build", "libraftwire.so")], check=True)
        L = C.CDLL(LIB_PATH)
        L.rw_encode_frame.restype = _sz
        L.rw_encode_frame.argtypes = [C.c_uint8, _i32, C.c_char_p, _sz, C.c_char_p, _sz, C.c_int, C.c_char_p, _sz]
        L.rw_kryo_request.restype = _sz
        L.rw_kryo_request.argtypes = [C.c_char_p, C.c_int, _i64, _i32, _i64, _i64, _i64, _vp, _u32, C.c_char_p, _sz]
        L.rw_kryo_response.restype = _sz
        L.rw_kryo_response.argtypes = [_i64, C.c_int, C.c_char_p, _sz]
        L.rw_kryo_decode_response.argtypes = [C.c_char_p, _sz, C.POINTER(_i64), C.POINTER(C.c_int)]
        L.rw_splitter_new.restype = _vp
        L.rw_splitter_free.argtypes = [_vp]
        L.rw_splitter_feed.argtypes = [_vp, C.c_char_p, _sz]
        L.rw_splitter_pop.argtypes = [_vp, C.POINTER(C.c_uint8), C.POINTER(_i32), C.POINTER(C.c_char_p), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz)]
        L.rw_ingress_new.restype = _vp
        L.rw_ingress_new.argtypes = [_u32, _u32, _u32, C.c_char_p, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u32]
        L.rw_ingress_free.argtypes = [_vp]
        L.rw_ingress_add_context.argtypes = [_vp, C.c_char_p, _sz, _u32]
        L.rw_ingress_set_peer.argtypes = [_vp, _u32, _i32]
        L.rw_ingress_remove_context.argtypes = [_vp, C.c_char_p, _sz]
        L.rw_ingress_sent.argtypes = [_vp, _u32, _i32, C.c_int, _u32, _u32, _i64, _i64]
        L.rw_ingress_feed.argtypes = [_vp, _u32, C.c_char_p, _sz]
        L.rw_ingress_reset_conn.argtypes = [_vp, _u32]
        L.rw_ingress_retain_bodies.argtypes = [_vp, C.c_int]
        L.rw_ingress_body.argtypes = [_vp, C.c_int, _u32, _u64, C.POINTER(_vp), C.POINTER(_sz)]
        L.rw_kryo_decode_request.argtypes = [C.c_char_p, C.c_int, C.c_char_p, _sz, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), _vp, _u32, C.POINTER(_u32)]
        L.rw_ingress_add_row.argtypes = [_vp, _u32, _u32, _u32, _u32, _i64, _i64, _i64, _i64, _u32, _i32]
        L.rw_ingress_encode_sends.restype = _sz
        L.rw_ingress_encode_sends.argtypes = [_vp, _u32, _i32, _u32, _vp, _vp, _vp, TERM_OF, _vp, C.c_char_p, _sz, C.POINTER(_u32), C.POINTER(_u32)]
        L.rw_ingress_seal.argtypes = [_vp, C.POINTER(abi.CBatch32), C.POINTER(_u64), C.POINTER(_u32)]
        L.rw_ingress_wide_row.argtypes = [_vp, C.c_int, _u32, C.POINTER(_u32), _vp, _vp, _vp, _u32, C.POINTER(_u32), C.POINTER(_i32)]
        L.rw_ingress_origin.argtypes = [_vp, C.c_int, _u32, _u64, C.POINTER(_u32), C.POINTER(_i32)]
        L.rw_ingress_shard.argtypes = [_vp, C.c_int, _u32, C.POINTER(abi.CBatch32), C.POINTER(_u64), C.POINTER(_u32)]
        L.rw_ingress_emit.restype = _sz
        L.rw_ingress_emit.argtypes = [_vp, C.c_int, _u32, _vp, _u64, _u64, _u32, C.c_char_p, _sz]
        L.rw_ingress_repair.restype = _i64
        L.rw_ingress_repair.argtypes = [_vp, C.c_int, _u32, _vp, _vp, C.c_int, C.POINTER(RepairHost)]
        L.rw_ingress_emit_wide.restype = _sz
        L.rw_ingress_emit_wide.argtypes = [_vp, C.c_int, _u32, _vp, C.POINTER(_u32), C.c_char_p, _sz]
        L.rw_ingress_recycle.argtypes = [_vp, C.c_int]
        L.rw_ingress_refused.restype = _u64
        L.rw_ingress_refused.argtypes = [_vp]
        L.rw_ingress_held.restype = _u64
        L.rw_ingress_held.argtypes = [_vp]
        L.rw_ingress_held_on.restype = _u64
        L.rw_ingress_held_on.argtypes = [_vp, _u32]
        _lib = L
    return _lib


TERM_AT = C.CFUNCTYPE(_i64, _vp, _u32, _i64)
TERM_OF = TERM_AT
CONFLICT = C.CFUNCTYPE(_i64, _vp, _u32, _i64, C.POINTER(_i64), _u32)
EPOCH_INDEX = C.CFUNCTYPE(_i64, _vp, _u32)
SUBMIT = C.CFUNCTYPE(C.c_int, _vp, C.POINTER(abi.CBatch), C.POINTER(abi.COutcome))
APPLIED = C.CFUNCTYPE(None, _vp, _u32, _u64, _vp, _vp, _vp)


class RepairHost(C.Structure):          # rw_repair_host_t
    _fields_ = [("user", _vp), ("term_at", TERM_AT), ("conflict", CONFLICT), ("epoch_index", EPOCH_INDEX), ("submit", SUBMIT), ("applied", APPLIED)]


def nodes_arg(nodes):
    return ",".join("%s:%d" % n for n in nodes).encode()


def frame(ftype, sequence, head, body):
    out = C.create_string_buffer(len(head) + len(body) + 32)
    n = lib().rw_encode_frame(ftype, sequence, head, len(head), body, len(body), 0, out, len(out))
    assert n
    return out.raw[:n]


def request_frame(nodes, method, ctx, sequence, term, node, x, y, leader_commit=0, entry_terms=()):
    arr = np.ascontiguousarray(entry_terms, dtype=np.int64)
    out = C.create_string_buffer(512 + 64 * len(arr))
    n = lib().rw_kryo_request(nodes, method, term, node, x, y, leader_commit, arr.ctypes.data, len(arr), out, len(out))
    assert n
    return frame(ENQ, sequence, METHOD_NAME[method] + b":" + ctx, out.raw[:n])


def response_frame(method, ctx, sequence, term, success):
    out = C.create_string_buffer(128)
    n = lib().rw_kryo_response(term, int(success), out, len(out))
    assert n
    return frame(ACK, sequence, METHOD_NAME[method] + b":" + ctx, out.raw[:n])


def split_frames(data):
    """[(type, sequence, head, body)] of a byte stream (the library's own splitter)"""
    L = lib()
    s = L.rw_splitter_new()
    try:
        assert L.rw_splitter_feed(s, data, len(data)) >= 0
        out = []
        t, q, h, hl, b, bl = C.c_uint8(), _i32(), C.c_char_p(), _sz(), _vp(), _sz()
        while L.rw_splitter_pop(s, C.byref(t), C.byref(q), C.byref(h), C.byref(hl), C.byref(b), C.byref(bl)):
            out.append((t.value, q.value, C.string_at(h, hl.value), C.string_at(b.value, bl.value) if bl.value else b""))
        return out
    finally:
        L.rw_splitter_free(s)


def decode_request(nodes, method, body):
    """(term, node, x, y, leader_commit, [entry terms]) of a Kryo-format request body, or None"""
    t, nd, x, y, lc, nt = _i64(), _i32(), _i64(), _i64(), _i64(), _u32()
    terms = np.zeros(abi.MAX_AE_ENTRIES + 8, dtype=np.int64)
    if not lib().rw_kryo_decode_request(nodes, method, body, len(body), C.byref(t), C.byref(nd), C.byref(x), C.byref(y), C.byref(lc), terms.ctypes.data, len(terms), C.byref(nt)):
        return None
    return t.value, nd.value, x.value, y.value, lc.value, [int(v) for v in terms[:nt.value]]


def decode_response(body):
    t, s = _i64(), C.c_int()
    assert lib().rw_kryo_decode_response(body, len(body), C.byref(t), C.byref(s)) == 1
    return t.value, bool(s.value)


class Sealed:
    """One sealed batch: an abi.Batch32 over the bank's memory (shard 0), its bank, the number of event rows (all shards), the rows kept out of
    the compact format, and per shard (batch32, events, first_gid)."""

    def __init__(self, bank, batch32, rows, wide, shards):
        self.bank, self.batch, self.rows, self.wide, self.shards = bank, batch32, rows, wide, shards


class Ingress:
    """rw_ingress_* with numpy-owned banks (a deployment hands over page-locked memory from rg_host_alloc instead)."""

    def __init__(self, groups, max_rounds, conns, nodes=None, entry_cap=1 << 16, shards=1):
        self.groups, self.max_rounds, self.conns, self.nshards = groups, max_rounds, conns, shards
        cells = groups * max_rounds
        self.head = [np.empty(cells, dtype=abi.HEAD_DT) for _ in range(2)]
        self.abcd = [np.empty(cells, dtype=abi.QUAD32_DT) for _ in range(2)]
        self.terms = [np.empty(max(entry_cap, 1), dtype=np.int32) for _ in range(2)]
        self._entry_cap = entry_cap
        self._h = lib().rw_ingress_new(groups, max_rounds, conns, None if nodes is None else nodes_arg(nodes),
                                       self.head[0].ctypes.data, self.abcd[0].ctypes.data, self.terms[0].ctypes.data,
                                       self.head[1].ctypes.data, self.abcd[1].ctypes.data, self.terms[1].ctypes.data, entry_cap, shards)
        if not self._h:
            raise ValueError("rw_ingress_new refused its arguments")

    def close(self):
        if self._h:
            lib().rw_ingress_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def add_context(self, ctx, gid):
        return bool(lib().rw_ingress_add_context(self._h, ctx, len(ctx), gid))

    def remove_context(self, ctx):
        return bool(lib().rw_ingress_remove_context(self._h, ctx, len(ctx)))

    def set_peer(self, conn, slot):
        assert lib().rw_ingress_set_peer(self._h, conn, slot)

    def sent(self, conn, sequence, method, gid, role_epoch, epoch_at_send=0, last_index_sent=0):
        assert lib().rw_ingress_sent(self._h, conn, sequence, method, gid, role_epoch, epoch_at_send, last_index_sent)

    def retain_bodies(self, on=True):
        assert lib().rw_ingress_retain_bodies(self._h, int(on))

    def body(self, bank, cell, shard=0):
        p, n = _vp(), _sz()
        return C.string_at(p.value, n.value) if lib().rw_ingress_body(self._h, bank, shard, cell, C.byref(p), C.byref(n)) else None

    def reset_conn(self, conn):
        assert lib().rw_ingress_reset_conn(self._h, conn)

    def feed(self, conn, data):
        return lib().rw_ingress_feed(self._h, conn, data, len(data))

    def add_row(self, conn, gid, hdr, aux=0, a=0, b=0, c=0, d=0, reply_conn=NO_CONN, reply_sequence=0):
        assert lib().rw_ingress_add_row(self._h, conn, gid, hdr, aux, a, b, c, d, reply_conn, reply_sequence)

    def encode_sends(self, conn, self_slot, head, send_j, term_of, gid=None):
        """rw_ingress_encode_sends: (bytes for `conn`, frames, rows that still need the host). head: SEND_HEAD_DT[count], send_j: SEND_DT[count]"""
        head, send_j = np.ascontiguousarray(head), np.ascontiguousarray(send_j)
        gid = None if gid is None else np.ascontiguousarray(gid, dtype=np.uint32)
        cb = TERM_OF(lambda _u, g, i: term_of(g, i))
        frames, need = _u32(), _u32()
        args = (self._h, conn, self_slot, len(head), None if gid is None else gid.ctypes.data, head.ctypes.data, send_j.ctypes.data, cb, None)
        size = lib().rw_ingress_encode_sends(*args, None, 0, C.byref(frames), C.byref(need))
        out = C.create_string_buffer(max(size, 1))
        assert lib().rw_ingress_encode_sends(*args, out, size, C.byref(frames), C.byref(need)) == size
        return out.raw[:size], frames.value, need.value

    def seal(self):
        cb, rows, nwide = abi.CBatch32(), _u64(), _u32()
        bank = lib().rw_ingress_seal(self._h, C.byref(cb), C.byref(rows), C.byref(nwide))
        if bank not in (0, 1):
            raise RuntimeError("rw_ingress_seal: the batch sealed before this one has not been recycled")
        shards = []
        per = (self.groups + self.nshards - 1) // self.nshards
        term_cap = len(self.terms[bank]) // self.nshards if self.nshards > 1 else len(self.terms[bank])
        k = 0
        while True:
            sb, ev, first = abi.CBatch32(), _u64(), _u32()
            if not lib().rw_ingress_shard(self._h, bank, k, C.byref(sb), C.byref(ev), C.byref(first)):
                break
            off, cells = first.value * self.max_rounds, sb.rounds * sb.count
            toff = k * (self._entry_cap // self.nshards)
            shards.append((abi.Batch32(sb.rounds, sb.count, None, self.head[bank][off:off + cells], self.abcd[bank][off:off + cells],
                                       self.terms[bank][toff:], sb.entry_count), ev.value, first.value))
            k += 1
        b32 = shards[0][0]
        wide = []
        for i in range(nwide.value):
            gid, head, q, terms = _u32(), np.zeros(1, dtype=abi.HEAD_DT), np.zeros(4, dtype=np.int64), np.zeros(abi.MAX_AE_ENTRIES, dtype=np.int64)
            rc, rs = _u32(), _i32()
            n = lib().rw_ingress_wide_row(self._h, bank, i, C.byref(gid), head.ctypes.data, q.ctypes.data, terms.ctypes.data, len(terms), C.byref(rc), C.byref(rs))
            assert n >= 0
            wide.append((gid.value, int(head["hdr"][0]), int(head["aux"][0]), [int(v) for v in q], [int(v) for v in terms[:n]],
                         None if rc.value == NO_CONN else (rc.value, rs.value)))
        return Sealed(bank, b32, rows.value, wide, shards)

    def origin(self, bank, cell, shard=0):
        c, s = _u32(), _i32()
        return (c.value, s.value) if lib().rw_ingress_origin(self._h, bank, shard, cell, C.byref(c), C.byref(s)) else None

    def emit(self, bank, reply, conn, cell_begin=0, cell_end=(1 << 62), shard=0):
        reply = np.ascontiguousarray(reply)
        need = lib().rw_ingress_emit(self._h, bank, shard, reply.ctypes.data, cell_begin, cell_end, conn, None, 0)
        out = C.create_string_buffer(max(need, 1))
        assert lib().rw_ingress_emit(self._h, bank, shard, reply.ctypes.data, cell_begin, cell_end, conn, out, need) == need
        return out.raw[:need]

    def repair(self, bank, reply, logfx, packed, term_at, conflict, epoch_index, submit, applied, shard=0):
        """rw_ingress_repair with Python callables: term_at(gid, index), conflict(gid, first_index, [terms]), epoch_index(gid) -> int;
        submit(CBatch*, COutcome*) -> 0; applied(gid, cell, reply row, logfx row, persist row as numpy records). reply is patched in place."""
        assert reply.flags["C_CONTIGUOUS"] and reply.dtype == abi.REPLY_DT
        logfx = np.ascontiguousarray(logfx)

        def _applied(_u, gid, cell, r, l, p):
            applied(gid, cell, np.frombuffer(C.string_at(r, 16), dtype=abi.REPLY_DT)[0], np.frombuffer(C.string_at(l, 16), dtype=abi.LOGFX_DT)[0],
                    np.frombuffer(C.string_at(p, 16), dtype=abi.PERSIST_DT)[0])

        host = RepairHost(None, TERM_AT(lambda _u, g, i: term_at(g, i)), CONFLICT(lambda _u, g, f, t, n: conflict(g, f, [t[k] for k in range(n)])),
                          EPOCH_INDEX(lambda _u, g: epoch_index(g)), SUBMIT(lambda _u, b, o: submit(b, o)), APPLIED(_applied))
        return lib().rw_ingress_repair(self._h, bank, shard, reply.ctypes.data, logfx.ctypes.data, int(packed), C.byref(host))

    def emit_wide(self, bank, i, reply_row):
        """(conn, frame bytes) of wide row i's response, or None"""
        row = np.ascontiguousarray(np.array([reply_row], dtype=abi.REPLY_DT))
        conn, out = _u32(), C.create_string_buffer(512)
        n = lib().rw_ingress_emit_wide(self._h, bank, i, row.ctypes.data, C.byref(conn), out, len(out))
        return (conn.value, out.raw[:n]) if n else None

    def recycle(self, bank):
        assert lib().rw_ingress_recycle(self._h, bank)

    def refused(self):
        return lib().rw_ingress_refused(self._h)

    def held(self):
        return lib().rw_ingress_held(self._h)

    def held_on(self, conn):
        return lib().rw_ingress_held_on(self._h, conn)


def unpack32(b32):
    """abi.Batch32 -> abi.Batch (what rg_submit / the oracle take): the inverse of rg_batch32_pack, for checks."""
    rows = b32.rounds * b32.count
    wide = abi.Batch(b32.rounds, b32.count, gid=b32.gid)
    if rows == 0:
        return wide
    head = b32.head[:rows]
    wide.ab["x"], wide.ab["y"] = b32.abcd["a"][:rows], b32.abcd["b"][:rows]
    wide.cd["x"], wide.cd["y"] = b32.abcd["c"][:rows], b32.abcd["d"][:rows]
    hdr = head["hdr"].copy()
    aux = head["aux"].copy()
    kind, n = hdr & 0xF, hdr >> 12
    terms, off = [], 0
    for i in np.flatnonzero((kind == abi.EV_AE_REQ) & (n > 0)):
        k = int(n[i])
        if hdr[i] & abi.HDR_SAME_TERM:
            terms.extend([int(aux[i])] * k)
        else:
            terms.extend(int(v) for v in b32.entry_terms[int(aux[i]) : int(aux[i]) + k])
        aux[i] = off
        off += k
    wide.head["hdr"] = hdr & ~np.uint32(abi.HDR_SAME_TERM)
    wide.head["aux"] = aux
    wide.entry_terms = np.array(terms if terms else [0], dtype=np.int64)
    wide.entry_count = off
    return wide
