"""N2, first half: the host-side wire decoder (rafting_amd/host/wire.cpp, C-ABI include/raftwire.h) against the reference's own
frame codec — EventCodec.FrameDecoder / FrameEncoder (transport/EventCodec.java:169-335), translated mechanically by
tools/make_ref.py and compiled (oracle/_ref/libref_wire.so).  Byte streams are fed to both in arbitrary pieces: the same
frames must come out, the same streams must be rejected, an EOT must switch both to pass-through at the same byte."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from rafting_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIRE_LIB = os.path.join(ROOT, "build", "libraftwire.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_wire.so")
ENQ, ACK, SYN, MW, PM, SOH, STX, ETX, EOT = 0x05, 0x06, 0x16, 0x95, 0x9E, 1, 2, 3, 4
u8p, sz = C.POINTER(C.c_uint8), C.c_size_t


@pytest.fixture(scope="module")
def wire():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rafting_amd", "host"), os.path.join("..", "..", "build", "libraftwire.so")], check=True)
    L = C.CDLL(WIRE_LIB)
    L.rw_splitter_new.restype = C.c_void_p
    L.rw_splitter_free.argtypes = [C.c_void_p]
    L.rw_splitter_feed.argtypes = [C.c_void_p, C.c_char_p, sz]
    L.rw_splitter_pop.argtypes = [C.c_void_p, u8p, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.POINTER(sz), C.POINTER(C.c_void_p), C.POINTER(sz)]
    L.rw_splitter_failed.argtypes = [C.c_void_p]
    L.rw_splitter_transparent.argtypes = [C.c_void_p]
    L.rw_splitter_passthrough.restype = sz
    L.rw_splitter_passthrough.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.rw_encode_frame.restype = sz
    L.rw_encode_frame.argtypes = [C.c_uint8, C.c_int32, C.c_char_p, sz, C.c_char_p, sz, C.c_int, C.c_char_p, sz]
    L.rw_fixed_request.restype = sz
    L.rw_fixed_request.argtypes = [C.c_int, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_uint32, C.c_char_p, sz]
    L.rw_fixed_response.restype = sz
    L.rw_fixed_response.argtypes = [C.c_int64, C.c_int, C.c_char_p, sz]
    L.rw_rows_add_frame.argtypes = [C.c_uint8, C.c_int32, C.c_char_p, sz, C.c_char_p, sz, C.c_int32, C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32,
                                    C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, sz, C.POINTER(sz), C.POINTER(sz)]
    return L


@pytest.fixture(scope="module")
def ref():
    from tests import ref_lib
    if not ref_lib.available() or not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libref_wire.so needs the reference checkout to be built")
    L = C.CDLL(REF_LIB)
    L.refwire_decoder_new.restype = C.c_void_p
    L.refwire_decoder_free.argtypes = [C.c_void_p]
    L.refwire_decoder_feed.argtypes = [C.c_void_p, C.c_char_p, sz]
    L.refwire_decoder_pop.argtypes = [C.c_void_p, u8p, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.POINTER(sz), C.POINTER(C.c_void_p), C.POINTER(sz)]
    L.refwire_decoder_closed.argtypes = [C.c_void_p]
    L.refwire_decoder_transparent.argtypes = [C.c_void_p]
    L.refwire_decoder_passthrough.restype = sz
    L.refwire_decoder_passthrough.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.refwire_encode.restype = sz
    L.refwire_encode.argtypes = [C.c_uint8, C.c_int, C.c_int32, C.c_char_p, sz, C.c_char_p, sz, C.c_int, C.c_int, C.c_char_p, sz]
    return L


def _drain(lib, h, pop):
    out = []
    t, s, hp, hl, bp, bl = C.c_uint8(), C.c_int32(), C.c_char_p(), sz(), C.c_void_p(), sz()
    while pop(h, C.byref(t), C.byref(s), C.byref(hp), C.byref(hl), C.byref(bp), C.byref(bl)):
        out.append((t.value, s.value, C.string_at(hp, hl.value), C.string_at(bp, bl.value) if bl.value else b""))
    return out


def _through(new, feed, pop, failed, transparent, passthrough, free, lib, stream, cuts):
    h = new()
    frames, at = [], 0
    for c in list(cuts) + [len(stream)]:
        feed(h, stream[at:c], c - at)
        at = c
        frames += _drain(lib, h, pop)
    p = C.c_void_p()
    n = passthrough(h, C.byref(p))
    res = (frames, bool(failed(h)), bool(transparent(h)), C.string_at(p, n) if n else b"")
    free(h)
    return res


def product(wire, stream, cuts):
    return _through(wire.rw_splitter_new, wire.rw_splitter_feed, wire.rw_splitter_pop, wire.rw_splitter_failed, wire.rw_splitter_transparent,
                    wire.rw_splitter_passthrough, wire.rw_splitter_free, wire, stream, cuts)


def reference(ref, stream, cuts):
    return _through(ref.refwire_decoder_new, ref.refwire_decoder_feed, ref.refwire_decoder_pop, ref.refwire_decoder_closed,
                    ref.refwire_decoder_transparent, ref.refwire_decoder_passthrough, ref.refwire_decoder_free, ref, stream, cuts)


def _frame(rng):
    t = rng.choice([ENQ, ENQ, ACK, ACK, SYN, MW, PM, 0x07])
    head = ("%s:%s" % (rng.choice(["appendEntries", "preVote", "requestVote", "installSnapshot", "other"]), "ctx%d" % rng.randrange(50))).encode()
    if rng.random() < 0.1:
        head = bytes(rng.randrange(32, 127) for _ in range(rng.choice([0, 1, 127, 128])))
    body = bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 0, 9, 41, 300]))) if t in (ENQ, ACK) else b""
    return t, rng.randrange(-2**31, 2**31), head, body


def test_encoder_matches_the_reference_encoder(wire, ref):
    rng = random.Random(1)
    out1, out2 = C.create_string_buffer(4096), C.create_string_buffer(4096)
    for _ in range(3000):
        t, seq, head, body = _frame(rng)
        ending = int(rng.random() < 0.1)
        n1 = wire.rw_encode_frame(t, seq, head, len(head), body, len(body), ending, out1, 4096)
        n2 = ref.refwire_encode(t, int(t in (ENQ, ACK)), seq, head, len(head), body, len(body), int(len(body) > 0), ending, out2, 4096)
        assert n1 == n2 > 0 and out1.raw[:n1] == out2.raw[:n2]


def test_splitter_matches_the_reference_decoder_on_valid_streams(wire, ref):
    rng = random.Random(2)
    buf = C.create_string_buffer(4096)
    for trial in range(400):
        stream, sent = b"", []
        for _ in range(rng.randrange(1, 12)):
            t, seq, head, body = _frame(rng)
            n = wire.rw_encode_frame(t, seq, head, len(head), body, len(body), 0, buf, 4096)
            stream += buf.raw[:n]
            sent.append((t, seq if t in (ENQ, ACK) else 0, head, body))
        if rng.random() < 0.3:                                      # the sender ends the framed protocol: EOT, then anything
            stream += bytes([EOT]) + bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 40)))
        cuts = sorted(rng.sample(range(1, len(stream)), min(len(stream) - 1, rng.randrange(0, 30))))
        got, exp = product(wire, stream, cuts), reference(ref, stream, cuts)
        assert got == exp, trial
        assert got[0] == sent and not got[1]


def test_splitter_matches_the_reference_decoder_on_corrupted_streams(wire, ref):
    """flip / drop / insert bytes: both must deliver the same frames before the damage and agree on whether the stream died"""
    rng = random.Random(3)
    buf = C.create_string_buffer(4096)
    died = alive = 0
    for trial in range(3000):
        stream = b""
        for _ in range(rng.randrange(1, 6)):
            t, seq, head, body = _frame(rng)
            n = wire.rw_encode_frame(t, seq, head, len(head), body, len(body), int(rng.random() < 0.05), buf, 4096)
            stream += buf.raw[:n]
        s = bytearray(stream)
        for _ in range(rng.randrange(1, 4)):
            k = rng.randrange(len(s))
            op = rng.random()
            if op < 0.5:
                s[k] = rng.choice([0, SOH, STX, ETX, EOT, ENQ, ACK, 0xFF, rng.getrandbits(8)])
            elif op < 0.75:
                del s[k]
            else:
                s.insert(k, rng.choice([0, SOH, STX, ETX, EOT, 0x80, rng.getrandbits(8)]))
        stream = bytes(s)
        cuts = sorted(rng.sample(range(1, len(stream)), min(len(stream) - 1, rng.randrange(0, 12)))) if len(stream) > 1 else []
        got, exp = product(wire, stream, cuts), reference(ref, stream, cuts)
        assert got == exp, (trial, stream.hex())
        died += got[1]
        alive += not got[1]
    assert died > 300 and alive > 300


def test_splitter_memory_stays_bounded_on_a_connection_that_never_pauses_between_frames(wire):
    """reads normally end INSIDE a frame (the next SOH has already been eaten), so compacting the buffer only between frames let
    a pipelined peer grow it without bound. 300 000 frames in chunks that never align with a frame boundary: every frame comes out, and
    the bytes held stay below one maximal chunk + frame + the 64 KiB compaction threshold."""
    wire.rw_splitter_held.restype = sz
    wire.rw_splitter_held.argtypes = [C.c_void_p]
    out = C.create_string_buffer(4096)
    body = bytes(range(41))
    n = wire.rw_encode_frame(ENQ, 7, b"appendEntries:ctx1", 18, body, len(body), 0, out, 4096)
    one = out.raw[:n]
    stream = one * 3000                                     # 3000 frames per pass, 100 passes
    h = wire.rw_splitter_new()
    t, s, hp, hl, bp, bl = C.c_uint8(), C.c_int32(), C.c_char_p(), sz(), C.c_void_p(), sz()
    total, peak, carry = 0, 0, b""
    chunk = 4093                                            # coprime with the frame length: a chunk never ends on a frame boundary twice in a row
    for _ in range(100):
        data = carry + stream
        cut = (len(data) // chunk) * chunk
        for at in range(0, cut, chunk):
            assert wire.rw_splitter_feed(h, data[at:at + chunk], chunk) >= 0
            while wire.rw_splitter_pop(h, C.byref(t), C.byref(s), C.byref(hp), C.byref(hl), C.byref(bp), C.byref(bl)):
                total += 1
            peak = max(peak, wire.rw_splitter_held(h))
        carry = data[cut:]
    assert total >= 300000 - 100 and not wire.rw_splitter_failed(h)
    assert peak < (1 << 17) + 2 * chunk + len(one), peak
    wire.rw_splitter_free(h)


def test_limits_and_quirks_of_the_grammar(wire, ref):
    def both(stream):
        got, exp = product(wire, stream, []), reference(ref, stream, [])
        assert got == exp, stream.hex()
        return got
    i32 = lambda v: int(v).to_bytes(4, "big", signed=True)        # noqa: E731
    ok = bytes([SOH, SYN, STX]) + i32(128) + b"h" * 128 + i32(0) + bytes([ETX])
    assert both(ok)[0] == [(SYN, 0, b"h" * 128, b"")]                                  # MAX_HEAD_SIZE = 128 (:25) is allowed ...
    assert both(bytes([SOH, SYN, STX]) + i32(129) + b"h" * 129 + i32(0) + bytes([ETX]))[1]      # ... 129 is not
    assert both(bytes([SOH, SYN, STX]) + i32(-1))[1]
    assert both(bytes([SOH, ENQ]) + i32(7) + bytes([STX]) + i32(1) + b"x" + i32((1 << 26) + 1))[1]   # MAX_BODY_SIZE = 64 MiB (:26)
    # a NUL type byte leaves `type == NUL` (:301-305): with five more bytes at hand the same turn demands STX from the next byte and the
    # channel dies; when the turn ends for lack of bytes, the type is simply read again — the reference's behaviour, chunk-dependent as it is
    nul = bytes([SOH, 0, MW, STX]) + i32(1) + b"x" + i32(0) + bytes([ETX])
    assert both(nul)[1]
    for cut in (2, 4):
        got, exp = product(wire, nul, [cut]), reference(ref, nul, [cut])
        assert got == exp and got[0] == [(MW, 0, b"x", b"")] and not got[1]
    assert both(bytes([SOH, ENQ]) + i32(-5) + bytes([STX]) + i32(0) + i32(0) + bytes([ETX]))[0] == [(ENQ, -5, b"", b"")]
    assert both(bytes([0x10]))[1] and both(bytes([SOH, SYN, ETX]) + i32(0))[1]         # no SOH / no STX (judged once five bytes are there)
    assert not both(bytes([SOH, SYN, ETX]))[1]
    assert both(bytes([SOH, SYN, STX]) + i32(0) + i32(0) + bytes([STX]))[1]            # no ETX
    r = both(ok + bytes([EOT]) + b"raw bytes of the snapshot channel")
    assert r[2] and r[3] == b"raw bytes of the snapshot channel" and len(r[0]) == 1    # EOT: pass-through from here on (:283-299)


def test_frames_become_rows(wire):
    """frames -> rows of an rg_batch_t (NettyNode.parseContextId / prepareLocalInvocation + the response invocations)"""
    ctx = [b"root", b"@raft", b"other"]
    arr = (C.c_char_p * len(ctx))(*ctx)
    head, ab, cd = np.zeros(16, abi.HEAD_DT), np.zeros(16, abi.PAIR_DT), np.zeros(16, abi.PAIR_DT)
    gid, terms = np.zeros(16, np.uint32), np.zeros(64, np.int64)
    rows, nterms = sz(0), sz(0)
    body = C.create_string_buffer(1024)

    def add(t, seq, scope, payload, peer, pend=(0, 0, 0)):
        return wire.rw_rows_add_frame(t, seq, scope, len(scope), payload, len(payload), peer, arr, len(ctx), pend[0], pend[1], pend[2],
                                      head.ctypes.data, ab.ctypes.data, cd.ctypes.data, gid.ctypes.data, terms.ctypes.data, 16, 64,
                                      C.byref(rows), C.byref(nterms))
    et = np.array([7, 7, 8], dtype=np.int64)
    n = wire.rw_fixed_request(1, 8, 2, 100, 7, 99, et.ctypes.data, 3, body, 1024)
    assert add(ENQ, 1, b"appendEntries:@raft", body.raw[:n], 2) == 1
    n = wire.rw_fixed_request(3, 9, 1, 103, 8, 0, None, 0, body, 1024)
    assert add(ENQ, 2, b"requestVote:root", body.raw[:n], 1) == 1
    n = wire.rw_fixed_response(9, 1, body, 1024)
    assert add(ACK, 3, b"appendEntries:other", body.raw[:n], 4, pend=(6, 50, 120)) == 1
    assert add(ACK, 4, b"preVote:root", body.raw[:n], 3, pend=(11, 0, 0)) == 1
    n = wire.rw_fixed_request(1, 8, 2, 200, 7, 99, et.ctypes.data, 2, body, 1024)
    assert add(ENQ, 5, b"appendEntries:root", body.raw[:n], 2) == 1
    assert add(ENQ, 6, b"appendEntries:unknown", body.raw[:n], 2) == 0            # no such context (NettyCluster.java:69-73)
    assert add(SYN, 0, b"hello", b"", 2) == 0 and add(ENQ, 7, b"obtainSnapshot:root", body.raw[:n], 2) == 0
    assert add(ENQ, 8, b"appendEntries:root", body.raw[:n - 3], 2) == 0           # truncated body
    assert (rows.value, nterms.value) == (5, 5)
    kinds = (head["hdr"][:5] & 0xF).tolist()
    assert kinds == [abi.EV_AE_REQ, abi.EV_RV_REQ, abi.EV_AE_ACK, abi.EV_PV_REPLY, abi.EV_AE_REQ]
    assert gid[:5].tolist() == [1, 0, 2, 0, 0]
    assert ((head["hdr"][:5] >> 4) & 0xF).tolist() == [2, 1, 4, 3, 2]                # leader / candidate / responder slot
    assert (int(ab["x"][0]), int(ab["y"][0]), int(cd["x"][0]), int(cd["y"][0])) == (8, 100, 7, 99)
    assert (int(head["hdr"][0]) >> 12, int(head["aux"][0]), terms[:3].tolist()) == (3, 0, [7, 7, 8])
    assert (int(head["hdr"][4]) >> 12, int(head["aux"][4]), terms[3:5].tolist()) == (2, 3, [7, 7])
    assert (int(head["aux"][2]), int(ab["x"][2]), int(ab["y"][2]), int(cd["x"][2]), (int(head["hdr"][2]) >> 8) & 1) == (6, 9, 50, 120, 1)
    assert (int(head["aux"][3]), int(ab["x"][3])) == (11, 9)


def test_zero_copy_stream_path_decodes_every_frame(tmp_path):
    """FrameSplitter::feed_views + RowWriter::add(FrameView): the allocation-free path a gateway would run (rafting_amd/host/wire_bench.cpp
    checks that every frame of a 200 000-frame stream, fed in 64 KiB reads, comes out as a row)."""
    exe = str(tmp_path / "wire_bench")
    host = os.path.join(ROOT, "rafting_amd", "host")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", exe, os.path.join(host, "wire_bench.cpp"), os.path.join(host, "wire.cpp"), os.path.join(host, "kryo_body.cpp")], check=True)
    p = subprocess.run([exe, "200000"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "split+rows 200000 rows" in p.stdout, p.stdout + p.stderr


def test_decoders_survive_hostile_bytes_under_sanitizers(tmp_path):
    """tests/native/wire_fuzz.cpp with ASan + UBSan: random bytes, mutated valid bodies and truncations through both body codecs, garbage through
    the frame splitter, damaged frame streams through the ingress (cells, term array and held rows stay in bounds and accounted for) — a peer
    must not be able to crash the host or make it read out of bounds (the Kryo-format reader parses lengths and class names that come off the wire)."""
    host = os.path.join(ROOT, "rafting_amd", "host")
    exe = str(tmp_path / "wire_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-I" + host,
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "wire_fuzz.cpp"), os.path.join(host, "wire.cpp"),
                        os.path.join(host, "kryo_body.cpp"), os.path.join(host, "ingress.cpp"), "-pthread", "-o", exe], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("no sanitizer runtime: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    p = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "fuzz ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
