"""ctypes binding of rafting_amd/libraftgpu.so (include/raftgpu.h) — the product path.

There is no CPU fallback: if the shared library is missing, or no MI355X is visible, construction
raises.  The library is loaded by absolute path from inside the package so the GPU-side tooling can
see that the in-tree native code is what ran.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RG_LIB") or os.path.join(_HERE, "libraftgpu.so")   # RG_LIB: experiment builds only
_LIB = None


class EngineError(RuntimeError):
    pass


def library_sha16():
    """first 16 hex digits of the SHA-256 of the loaded libraftgpu.so: ties a profile to the build it describes"""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def exported_symbols():
    """Every entry point include/raftgpu.h declares."""
    return [
        "rg_abi_version", "rg_table_create", "rg_table_destroy", "rg_last_error", "rg_table_groups",
        "rg_table_cluster", "rg_table_option", "rg_load_state", "rg_read_state", "rg_submit", "rg_submit32", "rg_submit32c", "rg_outcome32_unpack", "rg_outcome32_unpack_rel", "rg_index_base_set", "rg_index_base_get", "rg_batch32_pack_rel", "rg_batch32_pack", "rg_submit_async", "rg_submit_async_packed", "rg_submit_wait", "rg_tick_create", "rg_tick_launch", "rg_tick_wait", "rg_tick_destroy", "rg_tick2_create", "rg_tick2_launch", "rg_tick2_wait", "rg_tick2_destroy", "rg_timers_update32", "rg_health_update32", "rg_sync", "rg_step_kernel", "rg_host_alloc", "rg_host_free", "rg_dev_alloc",
        "rg_dev_free", "rg_copy_to_device", "rg_copy_to_host", "rg_stream", "rg_replicate", "rg_timers_configure", "rg_timers_update",
        "rg_timers_expired", "rg_timers_expired_epochs", "rg_timers_arm", "rg_timers_read", "rg_health_update", "rg_health_failure", "rg_ready", "rg_health_read",
        "rg_timing_enable",
        "rg_timing_read", "rg_timing_begin", "rg_timing_end", "rg_counters_read", "rg_wide_body_workgroups", "rg_copy_bandwidth",
    ]


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime. The PyTorch wheel bundles its own libamdhip64.so (same SONAME as
    /opt/rocm's); if libraftgpu.so pulled in the system copy first and torch its bundled copy later, the second
    runtime finds no device ("no ROCm-capable device is detected"). Loading torch's copy first — by path,
    without importing torch — makes both resolve to the same library whatever the import order. A host with no
    torch installed (the JNI deployment) simply uses the system runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                "%s is missing — build it with `make -C rafting_amd/csrc` (or __graft_entry__.build()); "
                "there is no CPU fallback for the decision path" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        if hasattr(L, "rg_is_host_emulation") and os.environ.get("RG_ALLOW_HOST_EMULATION") != "1":
            raise EngineError("%s is the test-only host emulation of the kernels (tests/devemu), not libraftgpu.so; "
                              "the decision path has no CPU fallback" % LIB_PATH)
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.rg_abi_version.restype = i32
        L.rg_table_create.argtypes = [i32, u32, u32, u32, i32, C.POINTER(vp)]
        L.rg_table_destroy.argtypes = [vp]
        L.rg_last_error.restype = C.c_char_p
        L.rg_last_error.argtypes = [vp]
        L.rg_table_groups.restype = u32
        L.rg_table_groups.argtypes = [vp]
        L.rg_table_cluster.restype = u32
        L.rg_table_cluster.argtypes = [vp]
        L.rg_table_option.argtypes = [vp, i32, i32]
        L.rg_load_state.argtypes = [vp, u32, u32, C.POINTER(abi.CGroupState)]
        L.rg_read_state.argtypes = [vp, u32, u32, C.POINTER(abi.CGroupState)]
        L.rg_submit.argtypes = [vp, C.POINTER(abi.CBatch), C.POINTER(abi.COutcome), i32]
        L.rg_submit32.argtypes = [vp, C.POINTER(abi.CBatch32), C.POINTER(abi.COutcome), i32]
        if os.environ.get("RG_LIB"):           # experiment builds of an older source tree (same-box A/Bs) may lack the newest entry points: bind what is there
            class _Tolerant:
                def __init__(self, lib_):
                    object.__setattr__(self, "_lib", lib_)

                def __getattr__(self, name):
                    try:
                        return getattr(self._lib, name)
                    except AttributeError:
                        class _Missing:          # (assigning argtypes to it is harmless; calling it is not)
                            def __call__(self, *a):
                                raise EngineError("%s is not exported by %s" % (name, LIB_PATH))
                        return _Missing()
            real, L = L, _Tolerant(L)
        L.rg_submit32c.argtypes = [vp, C.POINTER(abi.CBatch32), C.POINTER(abi.COutcome32), i32]
        L.rg_outcome32_unpack.argtypes = [C.POINTER(abi.COutcome32), u32, u32, vp, C.POINTER(abi.COutcome)]
        L.rg_outcome32_unpack_rel.argtypes = [C.POINTER(abi.COutcome32), u32, u32, vp, vp, C.POINTER(abi.COutcome)]
        L.rg_index_base_set.argtypes = [vp, u32, u32, vp]
        L.rg_index_base_get.argtypes = [vp, u32, u32, vp]
        L.rg_batch32_pack_rel.restype = C.c_int64
        L.rg_batch32_pack_rel.argtypes = [C.POINTER(abi.CBatch), vp, vp, vp, vp]
        L.rg_batch32_pack.restype = C.c_int64
        L.rg_batch32_pack.argtypes = [C.POINTER(abi.CBatch), vp, vp, vp]
        L.rg_submit_async.argtypes = [vp, C.POINTER(abi.CBatch), C.POINTER(abi.COutcome)]
        L.rg_submit_wait.argtypes = [vp]
        L.rg_submit_async_packed.argtypes = [vp, vp, vp]
        L.rg_sync.argtypes = [vp]
        L.rg_tick_create.argtypes = [vp, vp, vp, C.POINTER(vp)]
        L.rg_tick_launch.argtypes = [vp]
        L.rg_tick_wait.argtypes = [vp]
        L.rg_tick_destroy.argtypes = [vp]
        L.rg_tick2_create.argtypes = [vp, C.POINTER(abi.CTick2Io), C.POINTER(vp)]
        L.rg_tick2_launch.argtypes = [vp]
        L.rg_tick2_wait.argtypes = [vp]
        L.rg_tick2_destroy.argtypes = [vp]
        L.rg_timers_update32.argtypes = [vp, u32, vp, vp, vp, i32]
        L.rg_health_update32.argtypes = [vp, u32, vp, vp, vp, i32]
        L.rg_replicate.argtypes = [vp, u32, vp, vp, vp, vp, vp, i32]
        L.rg_step_kernel.restype = C.c_char_p
        L.rg_step_kernel.argtypes = [vp, u32]
        L.rg_timers_configure.argtypes = [vp, C.c_int64, C.c_int64, C.c_uint64]
        L.rg_timers_update.argtypes = [vp, u32, u32, vp, vp, vp, i32]
        L.rg_timers_expired.argtypes = [vp, C.c_int64, vp, u32, C.POINTER(u32), i32]
        L.rg_timers_expired_epochs.argtypes = [vp, C.c_int64, vp, vp, u32, C.POINTER(u32), i32]
        L.rg_timers_arm.argtypes = [vp, C.c_int64]
        L.rg_timers_read.argtypes = [vp, u32, u32, vp]
        L.rg_health_update.argtypes = [vp, u32, u32, vp, vp, vp, vp, i32]
        L.rg_health_failure.argtypes = [vp, u32, vp, vp, vp, C.c_int64]
        L.rg_ready.argtypes = [vp, C.c_int64, i32, C.c_int64, vp, i32]
        L.rg_health_read.argtypes = [vp, u32, u32, vp, vp, vp]
        L.rg_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.rg_host_free.argtypes = [vp, vp]
        L.rg_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.rg_dev_free.argtypes = [vp, vp]
        L.rg_copy_to_device.argtypes = [vp, vp, vp, C.c_size_t]
        L.rg_copy_to_host.argtypes = [vp, vp, vp, C.c_size_t]
        L.rg_stream.restype = vp
        L.rg_stream.argtypes = [vp]
        L.rg_timing_enable.argtypes = [vp, i32]
        L.rg_timing_read.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double), i32]
        L.rg_timing_begin.argtypes = [vp]
        L.rg_timing_end.argtypes = [vp, C.POINTER(C.c_double)]
        L.rg_counters_read.argtypes = [vp, C.POINTER(C.c_uint64), i32]
        L.rg_wide_body_workgroups.argtypes = [vp, C.POINTER(C.c_uint64), i32]
        L.rg_copy_bandwidth.argtypes = [vp, C.c_size_t, i32, C.POINTER(C.c_double)]
        if L.rg_abi_version() != abi.ABI_VERSION:
            raise EngineError("libraftgpu.so ABI %d != binding ABI %d" % (L.rg_abi_version(), abi.ABI_VERSION))
        _LIB = L
    return _LIB


def _replicate(call, groups, cluster, gid, heartbeat, in_flight):
    F = cluster - 1
    gid = None if gid is None else np.ascontiguousarray(gid, dtype=np.uint32)
    count = groups if gid is None else len(gid)
    hb = None if heartbeat is None else np.ascontiguousarray(np.broadcast_to(heartbeat, (count,)), dtype=np.uint8)
    # callers pass [count, F]; the wire is follower-major [F, count]
    fl = None if in_flight is None else np.ascontiguousarray(np.asarray(in_flight, dtype=np.uint16).reshape(count, F).T).reshape(count * F)
    head = np.zeros(count, dtype=abi.SEND_HEAD_DT)
    send = np.zeros(count * F, dtype=abi.SEND_DT)
    ptr = lambda a: None if a is None else a.ctypes.data   # noqa: E731
    call(count, ptr(gid), ptr(hb), ptr(fl), head.ctypes.data, send.ctypes.data)
    return head, np.ascontiguousarray(send.reshape(F, count).T)


def pack32(batch, index_base=None):
    """abi.Batch -> abi.Batch32 through the library's host-side packer (rg_batch32_pack / _rel): AppendEntries rows whose entries share one term
    carry it in the row (RG_HDR_SAME_TERM). index_base: int64 per GROUP OF THE TABLE — log indices then travel relative to their group's base
    (include/raftgpu.h, "the index base of the compact formats"). Raises when the batch cannot travel in the compact format (hints, a value
    outside [0, 2^31), an index at or below its base)."""
    rows = batch.rounds * batch.count
    head = np.zeros(rows, dtype=abi.HEAD_DT)
    abcd = np.zeros(rows, dtype=abi.QUAD32_DT)
    terms = np.zeros(max(batch.entry_count, 1), dtype=np.int32)
    b = batch.as_struct()
    if index_base is None:
        n = lib().rg_batch32_pack(C.byref(b), head.ctypes.data, abcd.ctypes.data, terms.ctypes.data)
    else:
        ib = np.ascontiguousarray(index_base, dtype=np.int64)
        n = lib().rg_batch32_pack_rel(C.byref(b), ib.ctypes.data, head.ctypes.data, abcd.ctypes.data, terms.ctypes.data)
    if n < 0:
        raise EngineError("rg_batch32_pack: %d (%s)" % (n, {-1: "missing column", -2: "the batch carries hints", -3: "a value outside [0, 2^31)",
                                                             -4: "entry_terms needed"}.get(n, "?")))
    return abi.Batch32(batch.rounds, batch.count, batch.gid, head, abcd, terms, n)


def pinned_like(table, array):
    """A page-locked copy of `array` (rg_host_alloc) as a numpy view; keep the returned owner alive, free with .free()."""
    a = np.ascontiguousarray(array)
    p = C.c_void_p()
    table._check(lib().rg_host_alloc(table._h, max(a.nbytes, 16), C.byref(p)))
    buf = (C.c_char * max(a.nbytes, 16)).from_address(p.value)
    view = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
    view[...] = a

    class _Owner:
        ptr = p.value

        def free(self):
            if self.ptr:
                lib().rg_host_free(table._h, self.ptr)
                self.ptr = None
    return view, _Owner()


class PackedBatch:
    """An abi.Batch re-laid in the compact transfer formats of rg_submit_async_packed, in page-locked memory: int32 event fields up,
    dense replies + packed logfx / persist lists down. unpack() rebuilds a dense abi.Outcome (unflagged rows zero) for comparison."""

    def __init__(self, table, batch, logfx_cap=None, persist_cap=None, entry_cap=0):
        assert abi.batch_fits_32(batch), "a value outside [0, 2^31) or a hint column: use submit_async for this batch"
        rows = batch.rounds * batch.count
        self.rows, self.table, self._owners = rows, table, []
        b32 = pack32(batch)
        self.head, self.abcd = self._pin(b32.head), self._pin(b32.abcd)
        et = np.zeros(max(entry_cap, b32.entry_count, 1), dtype=np.int32)      # entry_cap: room for the entry terms of later batches of the same shape (engine.Tick)
        et[:b32.entry_count] = b32.entry_terms[:b32.entry_count]
        self.entry_terms = self._pin(et)
        self.gid = None if batch.gid is None else self._pin(batch.gid)
        self.logfx_cap = rows if logfx_cap is None else logfx_cap
        self.persist_cap = rows if persist_cap is None else persist_cap
        self.reply = self._pin(np.zeros(rows, dtype=abi.REPLY_DT))
        self.logfx = self._pin(np.zeros(max(self.logfx_cap, 1), dtype=abi.LOGFX_DT))
        self.persist = self._pin(np.zeros(max(self.persist_cap, 1), dtype=abi.PERSIST_DT))
        self.counts = self._pin(np.zeros(2, dtype=np.uint32))
        b = abi.CBatch32()
        b.rounds, b.count = batch.rounds, batch.count
        b.gid = None if self.gid is None else self.gid.ctypes.data
        b.head, b.abcd = self.head.ctypes.data, self.abcd.ctypes.data
        b.entry_terms = self.entry_terms.ctypes.data if (b32.entry_count or entry_cap) else None
        b.entry_count = b32.entry_count
        o = abi.COutcomePacked()
        o.reply, o.logfx, o.persist, o.counts = self.reply.ctypes.data, self.logfx.ctypes.data, self.persist.ctypes.data, self.counts.ctypes.data
        o.logfx_cap, o.persist_cap = self.logfx_cap, self.persist_cap
        self.c_in, self.c_out = b, o

    def _pin(self, a):
        view, own = pinned_like(self.table, a)
        self._owners.append(own)
        return view

    @property
    def bytes_up(self):
        return self.head.nbytes + self.abcd.nbytes + (self.entry_terms.nbytes if self.c_in.entry_count else 0) + (0 if self.gid is None else self.gid.nbytes)

    @property
    def bytes_down(self):
        return self.reply.nbytes + int(self.counts[0]) * abi.LOGFX_DT.itemsize + int(self.counts[1]) * abi.PERSIST_DT.itemsize + self.counts.nbytes

    def unpack(self):
        out = abi.Outcome(self.rows)
        out.reply[:] = self.reply
        nl, npers = int(self.counts[0]), int(self.counts[1])
        ml, mp = abi.has_logfx(self.reply["flags"]), abi.has_persist(self.reply["flags"])
        assert nl == int(ml.sum()) and npers == int(mp.sum()), "list lengths disagree with the reply marks"
        assert nl <= self.logfx_cap and npers <= self.persist_cap, "a list was truncated: raise its capacity"
        out.logfx[ml] = self.logfx[:nl]
        out.persist[mp] = self.persist[:npers]
        return out

    def free(self):
        for own in self._owners:
            own.free()
        self._owners = []


def unpack32(out32, rounds, count, role_epoch_before, index_base=None):
    """abi.Outcome32 -> abi.Outcome through the library's host-side rg_outcome32_unpack(_rel). `role_epoch_before`: the groups' role epochs before the
    batch (GroupState.role_epoch); index_base: the groups' index bases (None: all 0); returns (outcome, role epochs after the batch)."""
    out = abi.Outcome(rounds * count)
    ep = np.ascontiguousarray(role_epoch_before, dtype=np.uint32).copy()
    assert len(ep) == count and out32.rows == rounds * count
    i, o = out32.as_struct(), out.as_struct()
    ib = None if index_base is None else np.ascontiguousarray(index_base, dtype=np.int64)
    assert ib is None or len(ib) == count
    rc = lib().rg_outcome32_unpack_rel(C.byref(i), rounds, count, ep.ctypes.data, None if ib is None else ib.ctypes.data, C.byref(o))
    if rc:
        raise EngineError("rg_outcome32_unpack: %d (%s)" % (rc, {-1: "missing column", -3: "a row is flagged RG_F_WIDE_VALUES but there are no overflow columns",
                                                                 -4: "the compact and the wide copy of a row disagree"}.get(rc, "?")))
    return out, ep


class Tick:
    """A prepared once-per-tick submission (rg_tick_create): the PackedBatch's page-locked buffers, a fixed shape, one HIP graph. refill(batch) writes
    the next tick's rows into the same buffers; launch() / wait() replay the graph; the outcome is read through the PackedBatch (unpack())."""

    def __init__(self, table, pb):
        self.table, self.pb = table, pb
        h = C.c_void_p()
        pb.c_in.entry_count = len(pb.entry_terms) if pb.c_in.entry_terms else 0       # the array's capacity: all of it travels every tick
        table._check(lib().rg_tick_create(table._h, C.byref(pb.c_in), C.byref(pb.c_out), C.byref(h)))
        self._h = h

    def refill(self, batch):
        b32 = batch if isinstance(batch, abi.Batch32) else pack32(batch)
        assert (b32.rounds, b32.count) == (self.pb.c_in.rounds, self.pb.c_in.count) and b32.entry_count <= self.pb.c_in.entry_count
        self.pb.head[:], self.pb.abcd[:] = b32.head, b32.abcd
        self.pb.entry_terms[:b32.entry_count] = b32.entry_terms[:b32.entry_count]
        if b32.gid is not None:
            self.pb.gid[:] = b32.gid

    def launch(self):
        self.table._check(lib().rg_tick_launch(self._h))

    def wait(self):
        self.table._check(lib().rg_tick_wait(self._h))

    def close(self):
        if self._h:
            lib().rg_tick_destroy(self._h)
            self._h = None


class Tick2:
    """The device-resident tick (rg_tick2_create): compact rows in, compact outcome rows out, the batch folded into the timers and the followers' health,
    the fired tickets listed, the leaders' send table and the readiness gate — ONE HIP graph, replayed with launch() / wait(). Every column lives in
    page-locked host memory here (the device reads and writes it over the link) so that a test can fill and read it with numpy; a deployment puts what
    only the device consumes into HBM (rg_dev_alloc).  refill(batch, now[, heartbeat, in_flight]) writes the next tick's rows and clocks."""

    def __init__(self, table, rounds, entry_cap=0, expired_cap=None, send=True, ready=True, critical_point=0, cool_down_ms=0, device_resident=False):
        G, F = table.groups, table.cluster - 1
        self.table, self.rounds, self.G, self.F = table, rounds, G, F
        rows = rounds * G
        self._pins = []
        self._devs = []

        def col(dtype, n):
            a, p = pinned_like(table, np.zeros(max(n, 1), dtype=dtype))
            self._pins.append(p)
            return a

        def big(dtype, n):
            # (device_resident: what only the device reads or writes stays in HBM; the tick is then measured without the link)
            if not device_resident:
                return col(dtype, n)
            b = DeviceBuffer.from_host(table, np.zeros(max(n, 1), dtype=dtype))
            self._devs.append(b)
            return b
        self.head, self.abcd = big(abi.HEAD_DT, rows), big(abi.QUAD32_DT, rows)
        self.entry_terms = big(np.int32, entry_cap) if entry_cap else None
        self.now = col(np.int64, rounds)
        self.heartbeat, self.in_flight = big(np.uint8, G), big(np.uint16, F * G)      # (read by every lane of the send kernel: over the link they cost it tens of microseconds)
        self.row, self.persist32 = big(abi.OUT32_DT, rows), big(abi.PERSIST32_DT, rows)
        cap = G if expired_cap is None else expired_cap
        self.expired_cap = cap
        self.expired_gid, self.expired_epoch, self.expired_count = (col(np.uint32, cap), col(np.uint32, cap), col(np.uint32, 1)) if cap else (None, None, None)
        self.send_head, self.send = (big(abi.SEND_HEAD_DT, G), big(abi.SEND_DT, F * G)) if send else (None, None)
        self.ready = big(np.uint8, G) if ready else None
        addr = lambda a: None if a is None else (a.ptr if isinstance(a, DeviceBuffer) else a.ctypes.data)   # noqa: E731
        io = abi.CTick2Io()
        io.rounds, io.head, io.abcd, io.entry_terms, io.entry_capacity = rounds, addr(self.head), addr(self.abcd), addr(self.entry_terms), entry_cap
        io.now, io.heartbeat, io.in_flight = addr(self.now), addr(self.heartbeat), addr(self.in_flight)
        io.critical_point, io.cool_down_ms = critical_point, cool_down_ms
        io.row, io.persist32 = addr(self.row), addr(self.persist32)
        io.expired_gid, io.expired_epoch, io.expired_count, io.expired_capacity = addr(self.expired_gid), addr(self.expired_epoch), addr(self.expired_count), cap
        io.send_head, io.send, io.ready = addr(self.send_head), addr(self.send), addr(self.ready)
        self.io = io
        h = C.c_void_p()
        table._check(lib().rg_tick2_create(table._h, C.byref(io), C.byref(h)))
        self._h = h

    def _put(self, dst, src):
        if isinstance(dst, DeviceBuffer):
            a = np.ascontiguousarray(src)
            self.table._check(lib().rg_copy_to_device(self.table._h, dst.ptr, a.ctypes.data, a.nbytes))
        else:
            dst[: len(src)] = src

    def _get(self, src, dtype, n):
        return src.to_host(dtype, n) if isinstance(src, DeviceBuffer) else np.array(src[:n], copy=True)

    def refill(self, batch, now, heartbeat=None, in_flight=None, index_base=None):
        b32 = batch if isinstance(batch, abi.Batch32) else pack32(batch, index_base)
        assert (b32.rounds, b32.count) == (self.rounds, self.G) and b32.gid is None and b32.entry_count <= self.io.entry_capacity
        self._put(self.head, b32.head)
        self._put(self.abcd, b32.abcd)
        if b32.entry_count:
            self._put(self.entry_terms, b32.entry_terms[: b32.entry_count])
        self.now[:] = np.asarray(now, dtype=np.int64)
        if heartbeat is not None or not isinstance(self.heartbeat, DeviceBuffer):
            self._put(self.heartbeat, np.zeros(self.G, np.uint8) if heartbeat is None else np.ascontiguousarray(heartbeat, dtype=np.uint8))
        if in_flight is not None or not isinstance(self.in_flight, DeviceBuffer):
            self._put(self.in_flight, np.zeros(self.F * self.G, np.uint16) if in_flight is None else np.ascontiguousarray(np.asarray(in_flight, dtype=np.uint16).reshape(-1)))

    def launch(self):
        self.table._check(lib().rg_tick2_launch(self._h))

    def wait(self):
        self.table._check(lib().rg_tick2_wait(self._h))

    def outcome32(self):
        rows = self.rounds * self.G
        out = abi.Outcome32(rows, wide=False)
        out.row, out.persist = self._get(self.row, abi.OUT32_DT, rows), self._get(self.persist32, abi.PERSIST32_DT, rows)
        return out

    def expired(self):
        """-> (gids, role epochs, total) of the tickets that fired by now[rounds - 1]"""
        n = int(self.expired_count[0])
        k = min(n, self.expired_cap)
        return np.array(self.expired_gid[:k]), np.array(self.expired_epoch[:k]), n

    def sends(self):
        return self._get(self.send_head, abi.SEND_HEAD_DT, self.G), self._get(self.send, abi.SEND_DT, self.F * self.G).reshape(self.F, self.G).T.copy()

    def readiness(self):
        return self._get(self.ready, np.uint8, self.G)

    def close(self):
        if self._h:
            lib().rg_tick2_destroy(self._h)
            self._h = None
        for p in self._pins:
            p.free()
        for b in self._devs:
            b.free()
        self._pins, self._devs = [], []


class DeviceBuffer:
    """A block of HBM owned through rg_dev_alloc."""

    def __init__(self, table, nbytes):
        self.table, self.nbytes = table, nbytes
        p = C.c_void_p()
        table._check(lib().rg_dev_alloc(table._h, nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, table, array):
        a = np.ascontiguousarray(array)
        buf = cls(table, max(a.nbytes, 16))
        if a.nbytes:
            table._check(lib().rg_copy_to_device(table._h, buf.ptr, a.ctypes.data, a.nbytes))
        return buf

    def to_host(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.table._check(lib().rg_copy_to_host(self.table._h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().rg_dev_free(self.table._h, self.ptr)
            self.ptr = None


class DeviceBatch:
    """An rg_batch_t + rg_outcome_t resident in HBM (RG_MEM_DEVICE): built once, submitted many times."""

    def __init__(self, table, batch):
        self.table, self.rounds, self.count = table, batch.rounds, batch.count
        rows = batch.rounds * batch.count
        self.rows = rows
        mk = lambda a: DeviceBuffer.from_host(table, a)   # noqa: E731
        self.gid = None if batch.gid is None else mk(batch.gid)
        self.head, self.ab, self.cd = mk(batch.head), mk(batch.ab), mk(batch.cd)
        self.entry_count = batch.entry_count
        self.entry_terms = mk(batch.entry_terms[: batch.entry_count]) if batch.entry_count else None
        self.hint = None if batch.hint is None else mk(batch.hint)
        self.reply = DeviceBuffer(table, rows * abi.REPLY_DT.itemsize)
        self.logfx = DeviceBuffer(table, rows * abi.LOGFX_DT.itemsize)
        self.persist = DeviceBuffer(table, rows * abi.PERSIST_DT.itemsize)
        b = abi.CBatch()
        b.rounds, b.count = batch.rounds, batch.count
        b.gid = self.gid.ptr if self.gid else None
        b.head, b.ab, b.cd = self.head.ptr, self.ab.ptr, self.cd.ptr
        b.entry_terms = self.entry_terms.ptr if self.entry_terms else None
        b.entry_count = self.entry_count
        b.hint = self.hint.ptr if self.hint else None
        o = abi.COutcome()
        o.reply, o.logfx, o.persist = self.reply.ptr, self.logfx.ptr, self.persist.ptr
        self.c_batch, self.c_out = b, o

    def outcome(self, fill_from=None):
        out = abi.Outcome(self.rows)
        out.reply = self.reply.to_host(abi.REPLY_DT, self.rows)
        out.logfx = self.logfx.to_host(abi.LOGFX_DT, self.rows)
        out.persist = self.persist.to_host(abi.PERSIST_DT, self.rows)
        return out

    def free(self):
        for b in (self.gid, self.head, self.ab, self.cd, self.entry_terms, self.hint, self.reply, self.logfx, self.persist):
            if b is not None:
                b.free()


class DeviceBatch32:
    """An rg_batch32_t + rg_outcome_t resident in HBM (rg_submit32, RG_MEM_DEVICE): built once from an abi.Batch, submitted many times.
    compact=True: the outcome is an rg_outcome32_t instead (rg_submit32c: one 16-byte row per event + persist rows; `wide` adds the overflow
    columns) — outcome32() reads it back, outcome(role_epoch_before) unpacks it into the wide image."""

    def __init__(self, table, batch, compact=False, wide=True):
        b32 = batch if isinstance(batch, abi.Batch32) else pack32(batch)
        self.table, self.rounds, self.count = table, b32.rounds, b32.count
        rows = b32.rounds * b32.count
        self.rows = rows
        mk = lambda a: DeviceBuffer.from_host(table, a)   # noqa: E731
        self.gid = None if b32.gid is None else mk(b32.gid)
        self.head, self.abcd = mk(b32.head), mk(b32.abcd)
        self.entry_count = b32.entry_count
        self.entry_terms = mk(b32.entry_terms[: b32.entry_count]) if b32.entry_count else None
        self.compact = compact
        self.reply = self.logfx = self.persist = self.row32 = self.persist32 = None
        if not compact or wide:
            self.reply = DeviceBuffer(table, rows * abi.REPLY_DT.itemsize)
            self.logfx = DeviceBuffer(table, rows * abi.LOGFX_DT.itemsize)
            self.persist = DeviceBuffer(table, rows * abi.PERSIST_DT.itemsize)
        b = abi.CBatch32()
        b.rounds, b.count = b32.rounds, b32.count
        b.gid = self.gid.ptr if self.gid else None
        b.head, b.abcd = self.head.ptr, self.abcd.ptr
        b.entry_terms = self.entry_terms.ptr if self.entry_terms else None
        b.entry_count = self.entry_count
        o = abi.COutcome()
        if self.reply is not None:
            o.reply, o.logfx, o.persist = self.reply.ptr, self.logfx.ptr, self.persist.ptr
        if compact:
            self.row32 = DeviceBuffer(table, rows * abi.OUT32_DT.itemsize)
            self.persist32 = DeviceBuffer(table, rows * abi.PERSIST32_DT.itemsize)
            o32 = abi.COutcome32()
            o32.row, o32.persist, o32.wide = self.row32.ptr, self.persist32.ptr, o
            o = o32
        self.c_batch, self.c_out = b, o
        self.bytes_in = b32.head.nbytes + b32.abcd.nbytes + 4 * b32.entry_count

    def outcome32(self):
        assert self.compact
        out = abi.Outcome32(self.rows, wide=self.reply is not None)
        out.row = self.row32.to_host(abi.OUT32_DT, self.rows)
        out.persist = self.persist32.to_host(abi.PERSIST32_DT, self.rows)
        if out.wide is not None:
            out.wide.reply = self.reply.to_host(abi.REPLY_DT, self.rows)
            out.wide.logfx = self.logfx.to_host(abi.LOGFX_DT, self.rows)
            out.wide.persist = self.persist.to_host(abi.PERSIST_DT, self.rows)
        return out

    def outcome(self, role_epoch_before=None):
        if self.compact:
            assert role_epoch_before is not None, "compact outcome rows carry the role epoch only where it changes: pass the epochs before the batch"
            return unpack32(self.outcome32(), self.rounds, self.count, role_epoch_before)[0]
        out = abi.Outcome(self.rows)
        out.reply = self.reply.to_host(abi.REPLY_DT, self.rows)
        out.logfx = self.logfx.to_host(abi.LOGFX_DT, self.rows)
        out.persist = self.persist.to_host(abi.PERSIST_DT, self.rows)
        return out

    def free(self):
        for b in (self.gid, self.head, self.abcd, self.entry_terms, self.reply, self.logfx, self.persist, self.row32, self.persist32):
            if b is not None:
                b.free()


class Table:
    """G raft groups resident on one MI355X. Mirrors ContextManager for the decision path: the host
    keeps RaftLog / StableLock / timers / Netty, this object answers what every RaftParticipant would
    have decided."""

    def __init__(self, groups, cluster, self_slot=0, pre_vote=True, device=0):
        self.groups, self.cluster, self.self_slot, self.pre_vote, self.device = groups, cluster, self_slot, pre_vote, device
        h = C.c_void_p()
        rc = lib().rg_table_create(device, groups, cluster, self_slot, int(pre_vote), C.byref(h))
        if rc:
            raise EngineError("rg_table_create: %s" % lib().rg_last_error(None).decode())
        self._h = h

    def _check(self, rc):
        if rc:
            raise EngineError("libraftgpu rc=%d: %s" % (rc, lib().rg_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            lib().rg_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_index_base(self, base, first=0):
        """rg_index_base_set: the index base of the compact formats for groups [first, first + len(base))"""
        b = np.ascontiguousarray(base, dtype=np.int64)
        self._check(lib().rg_index_base_set(self._h, first, len(b), b.ctypes.data))

    def index_base(self, first=0, count=None):
        count = self.groups - first if count is None else count
        b = np.zeros(count, dtype=np.int64)
        self._check(lib().rg_index_base_get(self._h, first, count, b.ctypes.data))
        return b

    def set_option(self, option, value):
        """rg_table_option, e.g. (abi.OPT_REQUIRE_FENCED_TIMEOUTS, 1)"""
        self._check(lib().rg_table_option(self._h, option, int(value)))

    # state ---------------------------------------------------------------------------------------
    def load_state(self, state, first=0):
        s = state.as_struct()
        self._check(lib().rg_load_state(self._h, first, state.count, C.byref(s)))

    def read_state(self, first=0, count=None):
        count = self.groups - first if count is None else count
        st = abi.GroupState(count, self.cluster)
        s = st.as_struct()
        self._check(lib().rg_read_state(self._h, first, count, C.byref(s)))
        return st

    # hot path ------------------------------------------------------------------------------------
    def submit(self, batch, out=None, fill=0):
        """Host-buffer submission (RG_MEM_HOST): staged over PCIe, synchronous."""
        out = abi.Outcome(batch.rounds * batch.count, fill) if out is None else out
        b, o = batch.as_struct(), out.as_struct()
        self._check(lib().rg_submit(self._h, C.byref(b), C.byref(o), abi.MEM_HOST))
        return out

    def submit32(self, batch, out=None, fill=0):
        """Host-buffer submission of COMPACT rows (rg_submit32, RG_MEM_HOST); `batch` is an abi.Batch32 (or an abi.Batch, packed here)."""
        b32 = batch if isinstance(batch, abi.Batch32) else pack32(batch)
        out = abi.Outcome(b32.rounds * b32.count, fill) if out is None else out
        b, o = b32.as_struct(), out.as_struct()
        self._check(lib().rg_submit32(self._h, C.byref(b), C.byref(o), abi.MEM_HOST))
        return out

    def submit32c(self, batch, out32=None, fill=0, wide=True):
        """Host-buffer submission of compact rows with COMPACT OUTCOME rows (rg_submit32c, RG_MEM_HOST) -> abi.Outcome32"""
        b32 = batch if isinstance(batch, abi.Batch32) else pack32(batch)
        out32 = abi.Outcome32(b32.rounds * b32.count, fill, wide=wide) if out32 is None else out32
        b, o = b32.as_struct(), out32.as_struct()
        self._check(lib().rg_submit32c(self._h, C.byref(b), C.byref(o), abi.MEM_HOST))
        return out32

    def submit32c_unpacked(self, batch, role_epoch_before, fill=0):
        """rg_submit32c, then rg_outcome32_unpack: the wide image of the compact outcome rows (what rg_submit32 would have returned)"""
        b32 = batch if isinstance(batch, abi.Batch32) else pack32(batch)
        return unpack32(self.submit32c(b32, fill=fill), b32.rounds, b32.count, role_epoch_before)[0]

    def submit_async(self, batch, out):
        """Pipelined host-buffer submission (rg_submit_async): returns at once; `batch` and `out` must stay alive and untouched
        until submit_wait() has returned for them. Keeps the ctypes structs alive itself."""
        b, o = batch.as_struct(), out.as_struct()
        self._inflight = getattr(self, "_inflight", [])
        if len(self._inflight) >= abi.PIPELINE_DEPTH:      # the library waits for the oldest batch itself before taking a new one
            self._inflight.pop(0)
        self._inflight.append((b, o, batch, out))
        self._check(lib().rg_submit_async(self._h, C.byref(b), C.byref(o)))
        return out

    def submit_async_packed(self, pb):
        """Pipelined submission with the compact transfer formats (rg_submit_async_packed); `pb` is a PackedBatch, whose page-locked
        buffers stay valid until submit_wait() has returned for it."""
        self._inflight = getattr(self, "_inflight", [])
        if len(self._inflight) >= abi.PIPELINE_DEPTH:
            self._inflight.pop(0)
        self._inflight.append((pb.c_in, pb.c_out, pb, pb))
        self._check(lib().rg_submit_async_packed(self._h, C.byref(pb.c_in), C.byref(pb.c_out)))
        return pb

    def submit_wait(self):
        """Blocks until the oldest batch in flight has landed; returns its Outcome (None when nothing was in flight)."""
        rc = lib().rg_submit_wait(self._h)
        if rc == 1:                                         # nothing in flight (any other entry point drains the pipeline too)
            self._inflight = []
            return None
        self._check(rc)
        return self._inflight.pop(0)[3]

    def submit_device(self, dbatch):
        """HBM-resident submission (RG_MEM_DEVICE): asynchronous on the table's stream. DeviceBatch -> rg_submit, DeviceBatch32 -> rg_submit32."""
        if isinstance(dbatch, DeviceBatch32) and dbatch.compact:
            self._check(lib().rg_submit32c(self._h, C.byref(dbatch.c_batch), C.byref(dbatch.c_out), abi.MEM_DEVICE))
        elif isinstance(dbatch, DeviceBatch32):
            self._check(lib().rg_submit32(self._h, C.byref(dbatch.c_batch), C.byref(dbatch.c_out), abi.MEM_DEVICE))
        else:
            self._check(lib().rg_submit(self._h, C.byref(dbatch.c_batch), C.byref(dbatch.c_out), abi.MEM_DEVICE))

    def sync(self):
        self._check(lib().rg_sync(self._h))

    def step_kernel(self, count=None):
        """Name of the step kernel that decides a batch with `count` rows per round (default: every group)."""
        return lib().rg_step_kernel(self._h, self.groups if count is None else count).decode()

    # N4 timers ------------------------------------------------------------------------------------
    def timers_configure(self, election_ms, heartbeat_ms, seed=0):
        self._check(lib().rg_timers_configure(self._h, election_ms, heartbeat_ms, seed))

    def timers_update(self, batch_rounds, batch_count, reply, now, gid=None):
        now = np.ascontiguousarray(now, dtype=np.int64)
        gid = None if gid is None else np.ascontiguousarray(gid, dtype=np.uint32)
        reply = np.ascontiguousarray(reply)
        assert len(now) == batch_rounds and len(reply) == batch_rounds * batch_count
        self._check(lib().rg_timers_update(self._h, batch_rounds, batch_count, None if gid is None else gid.ctypes.data,
                                           reply.ctypes.data, now.ctypes.data, abi.MEM_HOST))

    def timers_update32(self, rounds, out32, now):
        """rg_timers_update from the COMPACT outcome rows of a dense batch (abi.Outcome32: .row, .persist)"""
        now = np.ascontiguousarray(now, dtype=np.int64)
        row, per = np.ascontiguousarray(out32.row), np.ascontiguousarray(out32.persist)
        assert len(now) == rounds and len(row) == rounds * self.groups == len(per)
        self._check(lib().rg_timers_update32(self._h, rounds, row.ctypes.data, per.ctypes.data, now.ctypes.data, abi.MEM_HOST))

    def health_update32(self, batch, out32, now):
        now = np.ascontiguousarray(now, dtype=np.int64)
        row = np.ascontiguousarray(out32.row)
        assert batch.gid is None and len(now) == batch.rounds and len(row) == batch.rounds * self.groups
        self._check(lib().rg_health_update32(self._h, batch.rounds, batch.head.ctypes.data, row.ctypes.data, now.ctypes.data, abi.MEM_HOST))

    def timers_arm(self, now):
        self._check(lib().rg_timers_arm(self._h, now))

    def timers_expired(self, now, capacity=None):
        capacity = self.groups if capacity is None else capacity
        out = np.zeros(max(capacity, 1), dtype=np.uint32)
        n = C.c_uint32()
        self._check(lib().rg_timers_expired(self._h, now, out.ctypes.data, capacity, C.byref(n), abi.MEM_HOST))
        return out[: min(n.value, capacity)], n.value

    def timers_expired_epochs(self, now, capacity=None):
        """-> (gids, role epochs of the participants whose tickets fired, total): the epochs go into RG_EV_TIMEOUT.aux"""
        capacity = self.groups if capacity is None else capacity
        out, ep = np.zeros(max(capacity, 1), dtype=np.uint32), np.zeros(max(capacity, 1), dtype=np.uint32)
        n = C.c_uint32()
        self._check(lib().rg_timers_expired_epochs(self._h, now, out.ctypes.data, ep.ctypes.data, capacity, C.byref(n), abi.MEM_HOST))
        k = min(n.value, capacity)
        return out[:k], ep[:k], n.value

    def timers_read(self, first=0, count=None):
        count = self.groups - first if count is None else count
        out = np.zeros(count, dtype=np.int64)
        self._check(lib().rg_timers_read(self._h, first, count, out.ctypes.data))
        return out

    # N4b health / readiness -------------------------------------------------------------------------
    def health_update(self, batch, reply, now):
        """Fold a finished batch (its event heads + reply rows) into Leadership.State's statistics."""
        now = np.ascontiguousarray(now, dtype=np.int64)
        reply = np.ascontiguousarray(reply)
        assert len(now) == batch.rounds and len(reply) == batch.rounds * batch.count
        self._check(lib().rg_health_update(self._h, batch.rounds, batch.count, None if batch.gid is None else batch.gid.ctypes.data,
                                           batch.head.ctypes.data, reply.ctypes.data, now.ctypes.data, abi.MEM_HOST))

    def submit_timed(self, batch, now, fill=0):
        """submit + health_update: what a flush at wall-clock `now` (one value per round) does."""
        out = self.submit(batch, fill=fill)
        self.health_update(batch, out.reply, now)
        return out

    def submit_and_update_timers(self, batch, now, fill=0):
        """one drain + the timer pass over its reply rows: rg_submit, then rg_timers_update with the round clocks `now`."""
        out = self.submit(batch, fill=fill)
        self.timers_update(batch.rounds, batch.count, out.reply, now, gid=batch.gid)
        return out

    def health_failure(self, gid, slot, flags, now):
        """State.statFailure(now, unreachable=flags&1, reject=flags&2) per (group, peer slot) row."""
        gid = np.ascontiguousarray(gid, dtype=np.uint32)
        slot = np.ascontiguousarray(slot, dtype=np.uint8)
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        assert len(gid) == len(slot) == len(flags)
        self._check(lib().rg_health_failure(self._h, len(gid), gid.ctypes.data, slot.ctypes.data, flags.ctypes.data, now))

    def ready(self, now, critical_point, cool_down_ms):
        """Leader.isReady for every group (uint8[groups])."""
        out = np.zeros(self.groups, dtype=np.uint8)
        self._check(lib().rg_ready(self._h, now, critical_point, cool_down_ms, out.ctypes.data, abi.MEM_HOST))
        return out

    def health_read(self, first=0, count=None):
        count = self.groups - first if count is None else count
        F = self.cluster - 1
        ok, fl, rc = np.zeros((count, F), np.int64), np.zeros((count, F), np.int64), np.zeros((count, F), np.int32)
        self._check(lib().rg_health_read(self._h, first, count, ok.ctypes.data, fl.ctypes.data, rc.ctypes.data))
        return ok, fl, rc

    def replicate(self, gid=None, heartbeat=None, in_flight=None):
        """Leader.replicateLog for `gid` (None = every group): returns (head[count], send[count, F])."""
        return _replicate(lambda *a: self._check(lib().rg_replicate(self._h, *a, abi.MEM_HOST)), self.groups, self.cluster,
                          gid, heartbeat, in_flight)

    # measurement ---------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._check(lib().rg_timing_enable(self._h, int(on)))

    def timing_read(self, reset=True):
        n, ms = C.c_uint64(), C.c_double()
        self._check(lib().rg_timing_read(self._h, C.byref(n), C.byref(ms), int(reset)))
        return n.value, ms.value

    def timing_begin(self):
        self._check(lib().rg_timing_begin(self._h))

    def timing_end(self):
        ms = C.c_double()
        self._check(lib().rg_timing_end(self._h, C.byref(ms)))
        return ms.value

    def counters(self, reset=False):
        arr = (C.c_uint64 * abi.NUM_COUNTERS)()
        self._check(lib().rg_counters_read(self._h, arr, int(reset)))
        return list(arr)

    def wide_body_workgroups(self, reset=False):
        """workgroups of compact-row launches decided by the 64-bit body since the last reset (rg_wide_body_workgroups)"""
        n = C.c_uint64()
        self._check(lib().rg_wide_body_workgroups(self._h, C.byref(n), int(reset)))
        return n.value

    def copy_bandwidth(self, nbytes=1 << 30, iters=10):
        g = C.c_double()
        self._check(lib().rg_copy_bandwidth(self._h, nbytes, iters, C.byref(g)))
        return g.value
