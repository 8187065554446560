"""ctypes binding of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Nothing under rafting_amd/ may import this module; it exists so tests (and bench.py's cpu_baseline
leg, and __graft_entry__.smoke) can check the HIP path against the restated reference logic.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from rafting_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "raft_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        L = C.CDLL(path)
        L.orc_table_create.restype = C.c_void_p
        L.orc_table_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_table_destroy.argtypes = [C.c_void_p]
        L.orc_table_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_load_state.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(abi.CGroupState)]
        L.orc_read_state.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(abi.CGroupState)]
        L.orc_submit.argtypes = [C.c_void_p, C.POINTER(abi.CBatch), C.POINTER(abi.COutcome)]
        L.orc_submit_threads.restype = C.c_double
        L.orc_submit_threads.argtypes = [C.c_void_p, C.POINTER(abi.CBatch), C.POINTER(abi.COutcome), C.c_int]
        L.orc_rejection_step.restype = C.c_int64
        L.orc_rejection_step.argtypes = [C.c_int32]
        L.orc_major_indices.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_replicate.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
        L.orc_timers_configure.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64]
        L.orc_timers_update.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_timers_arm.argtypes = [C.c_void_p, C.c_int64]
        L.orc_timers_expired.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_timers_expired_epochs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_timers_read.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_health_clock.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_health_failure.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_ready.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]
        L.orc_health_read.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_log_term.restype = C.c_int
        L.orc_log_term.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_log_conflict.restype = C.c_int64
        L.orc_log_conflict.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_uint32, C.c_void_p]
        L.orc_update_index.restype = C.c_int
        L.orc_update_index.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_int64, C.c_int64, C.c_int, C.c_int]
        L.orc_update_index_batch.argtypes = [C.c_uint32] + [C.c_void_p] * 8
        L.orc_is_better_batch.argtypes = [C.c_uint32] + [C.c_void_p] * 7
        L.orc_major_indices_batch.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_is_better.restype = C.c_int
        L.orc_is_better.argtypes = [C.c_int, C.c_int64, C.c_int32, C.c_int, C.c_int64, C.c_int32]
        _LIB = L
    return _LIB


class OracleTable:
    """Same surface as rafting_amd.engine.Table, backed by the CPU oracle."""

    def __init__(self, groups, cluster, self_slot=0, pre_vote=True):
        self.groups, self.cluster, self.self_slot, self.pre_vote = groups, cluster, self_slot, pre_vote
        self._h = lib().orc_table_create(groups, cluster, self_slot, int(pre_vote))
        if not self._h:
            raise ValueError("orc_table_create rejected the arguments")

    def close(self):
        if self._h:
            lib().orc_table_destroy(self._h)
            self._h = None

    def set_option(self, option, value):
        if lib().orc_table_option(self._h, option, int(value)):
            raise ValueError("orc_table_option(%d) refused" % option)

    def __del__(self):
        self.close()

    def load_state(self, state, first=0):
        s = state.as_struct()
        rc = lib().orc_load_state(self._h, first, state.count, C.byref(s))
        if rc:
            raise ValueError("orc_load_state failed: %d" % rc)

    def read_state(self, first=0, count=None):
        count = self.groups - first if count is None else count
        st = abi.GroupState(count, self.cluster)
        s = st.as_struct()
        rc = lib().orc_read_state(self._h, first, count, C.byref(s))
        if rc:
            raise ValueError("orc_read_state failed: %d" % rc)
        return st

    def submit(self, batch, out=None, fill=0, now=None):
        """now: one wall-clock value per round -> Leadership.State statistics are kept for this submit."""
        out = abi.Outcome(batch.rounds * batch.count, fill) if out is None else out
        b, o = batch.as_struct(), out.as_struct()
        clock = None
        if now is not None:
            clock = np.ascontiguousarray(now, dtype=np.int64)
            assert len(clock) == batch.rounds
            assert lib().orc_health_clock(self._h, clock.ctypes.data) == 0
        try:
            rc = lib().orc_submit(self._h, C.byref(b), C.byref(o))
        finally:
            if clock is not None:
                lib().orc_health_clock(self._h, None)
        if rc:
            raise ValueError("orc_submit failed: %d" % rc)
        return out

    def submit_timed(self, batch, now, fill=0):
        return self.submit(batch, fill=fill, now=now)

    def submit_and_update_timers(self, batch, now, fill=0):
        """one drain + the timer pass over its reply rows (rg_submit, then rg_timers_update)"""
        out = self.submit(batch, fill=fill)
        self.timers_update(batch.rounds, batch.count, out.reply, now, gid=getattr(batch, "gid", None))
        return out

    def health_failure(self, gid, slot, flags, now):
        gid = np.ascontiguousarray(gid, dtype=np.uint32)
        slot = np.ascontiguousarray(slot, dtype=np.uint8)
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        assert lib().orc_health_failure(self._h, len(gid), gid.ctypes.data, slot.ctypes.data, flags.ctypes.data, now) == 0

    def ready(self, now, critical_point, cool_down_ms):
        out = np.zeros(self.groups, dtype=np.uint8)
        assert lib().orc_ready(self._h, now, critical_point, cool_down_ms, out.ctypes.data) == 0
        return out

    def health_read(self, first=0, count=None):
        count = self.groups - first if count is None else count
        F = self.cluster - 1
        ok, fl, rc = np.zeros((count, F), np.int64), np.zeros((count, F), np.int64), np.zeros((count, F), np.int32)
        assert lib().orc_health_read(self._h, first, count, ok.ctypes.data, fl.ctypes.data, rc.ctypes.data) == 0
        return ok, fl, rc

    def timers_configure(self, election_ms, heartbeat_ms, seed=0):
        assert lib().orc_timers_configure(self._h, election_ms, heartbeat_ms, seed) == 0

    def timers_update(self, batch_rounds, batch_count, reply, now, gid=None):
        now = np.ascontiguousarray(now, dtype=np.int64)
        gid = None if gid is None else np.ascontiguousarray(gid, dtype=np.uint32)
        reply = np.ascontiguousarray(reply)
        assert lib().orc_timers_update(self._h, batch_rounds, batch_count, None if gid is None else gid.ctypes.data,
                                       reply.ctypes.data, now.ctypes.data) == 0

    def timers_arm(self, now):
        assert lib().orc_timers_arm(self._h, now) == 0

    def timers_expired(self, now, capacity=None):
        capacity = self.groups if capacity is None else capacity
        out = np.zeros(max(capacity, 1), dtype=np.uint32)
        n = C.c_uint32()
        assert lib().orc_timers_expired(self._h, now, out.ctypes.data, capacity, C.byref(n)) == 0
        return out[: min(n.value, capacity)], n.value

    def timers_expired_epochs(self, now, capacity=None):
        capacity = self.groups if capacity is None else capacity
        out, ep = np.zeros(max(capacity, 1), dtype=np.uint32), np.zeros(max(capacity, 1), dtype=np.uint32)
        n = C.c_uint32()
        assert lib().orc_timers_expired_epochs(self._h, now, out.ctypes.data, ep.ctypes.data, capacity, C.byref(n)) == 0
        k = min(n.value, capacity)
        return out[:k], ep[:k], n.value

    def timers_read(self, first=0, count=None):
        count = self.groups - first if count is None else count
        out = np.zeros(count, dtype=np.int64)
        assert lib().orc_timers_read(self._h, first, count, out.ctypes.data) == 0
        return out

    def replicate(self, gid=None, heartbeat=None, in_flight=None):
        from rafting_amd.engine import _replicate

        def call(*a):
            if lib().orc_replicate(self._h, *a):
                raise ValueError("orc_replicate failed")
        return _replicate(call, self.groups, self.cluster, gid, heartbeat, in_flight)

    def log_term(self, gid, index):
        """RaftLog.get(index).term() of the lossless log, None when the key does not exist."""
        t = C.c_int64()
        return int(t.value) if lib().orc_log_term(self._h, gid, index, C.byref(t)) else None

    def log_conflict(self, gid, e0, terms):
        a = np.ascontiguousarray(terms, dtype=np.int64)
        return int(lib().orc_log_conflict(self._h, gid, e0, len(a), a.ctypes.data))

    def submit_threads(self, batch, threads, out=None):
        out = abi.Outcome(batch.rounds * batch.count) if out is None else out
        b, o = batch.as_struct(), out.as_struct()
        secs = lib().orc_submit_threads(self._h, C.byref(b), C.byref(o), threads)
        if secs < 0:
            raise ValueError("orc_submit_threads failed")
        return secs, out


def rejection_step(r):
    return int(lib().orc_rejection_step(int(r)))


def major_indices(match):
    m = np.ascontiguousarray(match, dtype=np.int64)
    out = np.zeros(2, dtype=np.int64)
    lib().orc_major_indices(m.ctypes.data, len(m), out.ctypes.data)
    return int(out[0]), int(out[1])


def is_better(new, cur):
    """new/cur = (role, term, ballot). Returns True/False or the negative RG_A_* code."""
    r = lib().orc_is_better(new[0], new[1], new[2], cur[0], cur[1], cur[2])
    return r if r < 0 else bool(r)


def update_index(last_epoch, next_index, match_index, rejection, pending, epoch, index, success, snapshot):
    st = np.array([last_epoch, next_index, match_index], dtype=np.int64)
    rej, pen = C.c_int32(rejection), C.c_uint8(pending)
    rc = lib().orc_update_index(st.ctypes.data, C.byref(rej), C.byref(pen), epoch, index, int(success), int(snapshot))
    return rc, (int(st[0]), int(st[1]), int(st[2]), int(rej.value), int(pen.value))
