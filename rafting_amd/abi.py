"""ctypes/numpy mirror of include/raftgpu.h — wire structs, constants and buffer holders.

Pure layout code (no library is loaded here): both the product binding (rafting_amd.engine) and the
test-only oracle binding (tests/oracle_lib.py) describe their buffers with these types, so the two
sides are fed byte-identical inputs.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 5
MIN_CLUSTER, MAX_CLUSTER, MAX_COMPACT_CLUSTER = 2, 15, 7
TERM_RUNS = 4
NO_NODE = -1

FOLLOWER, CANDIDATE, LEADER = 0, 1, 2

EV_NONE, EV_AE_REQ, EV_AE_ACK, EV_IS_ACK, EV_RV_REQ, EV_PV_REQ = 0, 1, 2, 3, 4, 5
EV_RV_REPLY, EV_PV_REPLY, EV_TIMEOUT, EV_CLIENT_APPEND, EV_LOG_FLUSH, EV_IS_REQ = 6, 7, 8, 9, 10, 11
MAX_AE_ENTRIES = 200
PIPELINE_DEPTH = 2
HDR_HINT_BIT = 1 << 9
HDR_SAME_TERM = 1 << 10     # rg_batch32_t rows only: every carried entry of the AppendEntries row has the term held in aux

F_SUCCESS, F_REPLIED, F_PERSIST, F_ROLE_CHANGED = 1 << 0, 1 << 1, 1 << 2, 1 << 3
F_RESET_TIMER, F_COMMIT, F_LOG_TRUNC, F_LOG_APPEND = 1 << 4, 1 << 5, 1 << 6, 1 << 7
F_EMIT_SHIFT, F_EMIT_MASK = 8, 3 << 8
EMIT_NONE, EMIT_PREVOTE, EMIT_REQVOTE, EMIT_HEARTBEAT = 0, 1, 2, 3
F_ROLE_SHIFT, F_ROLE_MASK = 10, 3 << 10
F_TIMER_MUTED = 1 << 12
F_WIDE_VALUES = 1 << 13    # rg_out32_t rows only: the row's numbers are low 32 bits, the full rows are in the overflow columns
F_STATUS_SHIFT = 16

OK = 0
A_TWO_LEADERS, A_PREV_ZERO_MISMATCH, A_EPOCH_TERM_MISMATCH, A_IMPOSSIBLE_LOG = 1, 2, 3, 4
A_COMMIT_ROLLBACK, A_LOG_NOT_CONTINUOUS, A_LEADER_SELF_AE, A_SAME_TERM_LEADER = 5, 6, 7, 8
A_LEADER_NOT_SELF_VOTE, A_CAND_SELF_RV, A_CAND_NOT_SELF_VOTE, A_LEADER_UNCHANGED = 9, 10, 11, 12
A_CAND_BALLOT, A_MATCH_ROLLBACK, A_IMPOSSIBLE_REPLICATION, NPE_MAJOR_NULL = 13, 14, 15, 16
DROPPED_STALE_ROLE, NOT_LEADER, FLUSH_OUT_OF_BOUNDS = 17, 18, 19
A_INSTALL_BEFORE_AE, A_NO_DOWNGRADE = 20, 21
NEED_HOST, SKIPPED_AFTER_NEED_HOST, BAD_EVENT, UNSUPPORTED_LOG_STATE = 32, 33, 34, 35

MEM_HOST, MEM_DEVICE = 0, 1
OPT_REQUIRE_FENCED_TIMEOUTS = 1
NUM_COUNTERS = 8

HEAD_DT = np.dtype([("hdr", "<u4"), ("aux", "<u4")])
PAIR_DT = np.dtype([("x", "<i8"), ("y", "<i8")])
REPLY_DT = np.dtype([("resp_term", "<i8"), ("flags", "<u4"), ("role_epoch", "<u4")])
LOGFX_DT = np.dtype([("commit_index", "<i8"), ("log_from", "<i8")])
PERSIST_DT = np.dtype([("term", "<i8"), ("voted_for", "<i4"), ("role", "<i4")])
SEND_HEAD_DT = np.dtype([("term", "<i8"), ("leader_commit", "<i8"), ("epoch_index", "<i8"), ("epoch_term", "<i8"),
                         ("role_epoch", "<u4"), ("is_leader", "<u4"), ("reserved", "<u8")])
SEND_DT = np.dtype([("prev_index", "<i8"), ("prev_term", "<i8"), ("last_index", "<i8"), ("count", "<u4"), ("kind", "<u4")])
SEND_NONE, SEND_APPEND, SEND_SNAPSHOT, SEND_GATED, SEND_NEED_HOST = 0, 1, 2, 3, 4
REPLICATE_LIMIT, IN_FLIGHT_LIMIT = 50, 20
assert SEND_HEAD_DT.itemsize == 48 and SEND_DT.itemsize == 32
assert HEAD_DT.itemsize == 8 and PAIR_DT.itemsize == 16 and REPLY_DT.itemsize == 16
assert LOGFX_DT.itemsize == 16 and PERSIST_DT.itemsize == 16


def hdr_make(kind, slot=0, flag=0, n=0):
    """RG_HDR_MAKE; works on scalars and numpy arrays."""
    return (
        (np.uint32(kind) & np.uint32(0xF))
        | ((np.asarray(slot).astype(np.uint32) & np.uint32(0xF)) << np.uint32(4))
        | ((np.asarray(flag).astype(np.uint32) & np.uint32(1)) << np.uint32(8))
        | (np.asarray(n).astype(np.uint32) << np.uint32(12))
    ).astype(np.uint32)


def flags_status(flags):
    return (np.asarray(flags) >> F_STATUS_SHIFT) & 0xFF


def flags_role(flags):
    return (np.asarray(flags) & F_ROLE_MASK) >> F_ROLE_SHIFT


def flags_emit(flags):
    return (np.asarray(flags) & F_EMIT_MASK) >> F_EMIT_SHIFT


class CBatch(C.Structure):
    _fields_ = [
        ("rounds", C.c_uint32),
        ("count", C.c_uint32),
        ("gid", C.c_void_p),
        ("head", C.c_void_p),
        ("ab", C.c_void_p),
        ("cd", C.c_void_p),
        ("entry_terms", C.c_void_p),
        ("entry_count", C.c_uint64),
        ("hint", C.c_void_p),
    ]


class COutcome(C.Structure):
    _fields_ = [("reply", C.c_void_p), ("logfx", C.c_void_p), ("persist", C.c_void_p)]


class CBatch32(C.Structure):           # rg_batch32_t
    _fields_ = [("rounds", C.c_uint32), ("count", C.c_uint32), ("gid", C.c_void_p), ("head", C.c_void_p), ("abcd", C.c_void_p),
                ("entry_terms", C.c_void_p), ("entry_count", C.c_uint64)]


class COutcome32(C.Structure):         # rg_outcome32_t
    _fields_ = [("row", C.c_void_p), ("persist", C.c_void_p), ("wide", COutcome)]


class COutcomePacked(C.Structure):     # rg_outcome_packed_t
    _fields_ = [("reply", C.c_void_p), ("logfx", C.c_void_p), ("persist", C.c_void_p), ("counts", C.c_void_p),
                ("logfx_cap", C.c_uint32), ("persist_cap", C.c_uint32)]


class CTick2Io(C.Structure):           # rg_tick2_io_t
    _fields_ = [("rounds", C.c_uint32), ("head", C.c_void_p), ("abcd", C.c_void_p), ("entry_terms", C.c_void_p), ("entry_capacity", C.c_uint64),
                ("now", C.c_void_p), ("heartbeat", C.c_void_p), ("in_flight", C.c_void_p), ("critical_point", C.c_int32), ("cool_down_ms", C.c_int64),
                ("row", C.c_void_p), ("persist32", C.c_void_p), ("expired_gid", C.c_void_p), ("expired_epoch", C.c_void_p), ("expired_count", C.c_void_p),
                ("expired_capacity", C.c_uint32), ("send_head", C.c_void_p), ("send", C.c_void_p), ("ready", C.c_void_p)]


_STATE_FIELDS = [
    ("current_term", np.int64, 1),
    ("voted_for", np.int32, 1),
    ("role", np.int32, 1),
    ("current_leader", np.int32, 1),
    ("timeout_detected", np.uint8, 1),
    ("repl_prepared", np.uint8, 1),
    ("role_epoch", np.uint32, 1),
    ("votes", np.int32, 1),
    ("elected_epoch", np.uint32, 1),
    ("elected_term", np.int64, 1),
    ("commit_index", np.int64, 1),
    ("epoch_index", np.int64, 1),
    ("epoch_term", np.int64, 1),
    ("first_index", np.int64, 1),
    ("last_index", np.int64, 1),
    ("run_count", np.uint32, 1),
    ("run_offset", np.uint32, 1),
    ("run_start", np.int64, "runs"),
    ("run_term", np.int64, "runs"),
    ("peer_last_epoch", np.int64, "peers"),
    ("peer_next_index", np.int64, "peers"),
    ("peer_match_index", np.int64, "peers"),
    ("peer_rejection", np.int32, "peers"),
    ("peer_pending", np.uint8, "peers"),
]


class CGroupState(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in _STATE_FIELDS]


def _ptr(a):
    return None if a is None else a.ctypes.data


class GroupState:
    """Host SoA image of `count` groups (rg_group_state_t). Fresh groups are what
    RaftContext.initialize produces: Follower, term 0, no vote, empty log, epoch (0,0)."""

    def __init__(self, count, cluster, runs_total=None):
        self.count, self.cluster, self.followers = count, cluster, cluster - 1
        nruns = count * TERM_RUNS if runs_total is None else runs_total
        for name, dt, shape in _STATE_FIELDS:
            n = count if shape == 1 else (nruns if shape == "runs" else count * self.followers)
            setattr(self, name, np.zeros(n, dtype=dt))
        self.voted_for[:] = NO_NODE
        self.current_leader[:] = NO_NODE
        self.role_epoch[:] = 1
        self.votes[:] = 1
        self.run_offset[:] = np.arange(count, dtype=np.uint32) * TERM_RUNS if runs_total is None else 0

    def set_log(self, g, first, runs, last):
        """Give group g the log whose maximal equal-term runs are `runs` = [(start, term), ...]
        (ascending, runs[0][0] == first) and whose greatest key is `last`. Only valid on the default
        layout (TERM_RUNS slots per group), so len(runs) <= TERM_RUNS."""
        assert len(runs) <= TERM_RUNS
        self.run_count[g] = len(runs)
        self.first_index[g], self.last_index[g] = (first, last) if runs else (0, 0)
        for k, (s, t) in enumerate(runs):
            self.run_start[g * TERM_RUNS + k] = s
            self.run_term[g * TERM_RUNS + k] = t

    def peers(self, name):
        return getattr(self, "peer_" + name).reshape(self.count, self.followers)

    def as_struct(self):
        s = CGroupState()
        for name, dt, _ in _STATE_FIELDS:
            a = getattr(self, name)
            assert a.dtype == dt and a.flags["C_CONTIGUOUS"], name
            setattr(s, name, _ptr(a))
        return s

    def fields(self):
        return [name for name, _, _ in _STATE_FIELDS]


class Batch:
    """Host image of an rg_batch_t: `rounds` x `count` rows."""

    def __init__(self, rounds, count, gid=None, max_entries=0, hints=False):
        self.rounds, self.count = rounds, count
        n = rounds * count
        self.gid = None if gid is None else np.ascontiguousarray(gid, dtype=np.uint32)
        self.head = np.zeros(n, dtype=HEAD_DT)
        self.ab = np.zeros(n, dtype=PAIR_DT)
        self.cd = np.zeros(n, dtype=PAIR_DT)
        self.entry_terms = np.zeros(max_entries, dtype=np.int64)
        self.entry_count = 0
        self.hint = np.zeros(n, dtype=PAIR_DT) if hints else None

    def row(self, r, i):
        return r * self.count + i

    def put(self, r, i, kind, slot=0, flag=0, a=0, b=0, c=0, d=0, aux=0, entries=None, n=0):
        """Write one event row; `entries` (AE_REQ) is the list of entry terms, `n` the command
        count of a CLIENT_APPEND."""
        row = self.row(r, i)
        if entries is not None and len(entries):
            n = len(entries)
            if self.entry_count + n > len(self.entry_terms):
                grown = np.zeros(max(2 * len(self.entry_terms), self.entry_count + n, 64), dtype=np.int64)
                grown[: self.entry_count] = self.entry_terms[: self.entry_count]
                self.entry_terms = grown
            aux = self.entry_count
            self.entry_terms[self.entry_count : self.entry_count + n] = entries
            self.entry_count += n
        self.head[row] = (int(hdr_make(kind, slot, flag, n)), int(aux) & 0xFFFFFFFF)
        self.ab[row] = (a, b)
        self.cd[row] = (c, d)
        return row

    def set_hint(self, row, x, y):
        assert self.hint is not None
        self.hint[row] = (x, y)
        self.head["hdr"][row] |= HDR_HINT_BIT

    def as_struct(self):
        b = CBatch()
        b.rounds, b.count = self.rounds, self.count
        b.gid = _ptr(self.gid)
        b.head, b.ab, b.cd = _ptr(self.head), _ptr(self.ab), _ptr(self.cd)
        b.entry_terms = _ptr(self.entry_terms) if self.entry_count else None
        b.entry_count = self.entry_count
        b.hint = _ptr(self.hint)
        return b


QUAD32_DT = np.dtype([("a", np.int32), ("b", np.int32), ("c", np.int32), ("d", np.int32)])
NARROW_EVENT_LIMIT = 1 << 31


def batch_fits_32(batch):
    """every a, b, c, d and entry term of the batch is in [0, 2^31): the batch may travel as an rg_batch32_t"""
    cols = [batch.ab["x"], batch.ab["y"], batch.cd["x"], batch.cd["y"], batch.entry_terms[: batch.entry_count]]
    return batch.hint is None and all(len(c) == 0 or (int(c.min()) >= 0 and int(c.max()) < NARROW_EVENT_LIMIT) for c in cols)


class Batch32:
    """Host image of an rg_batch32_t (compact rows): head + int32 a, b, c, d + int32 entry terms of the rows without RG_HDR_SAME_TERM.
    Built from a Batch by rafting_amd.engine.pack32 (the library's rg_batch32_pack) or tests' own packers."""

    def __init__(self, rounds, count, gid, head, abcd, entry_terms, entry_count):
        self.rounds, self.count, self.gid = rounds, count, gid
        self.head, self.abcd = head, abcd
        self.entry_terms, self.entry_count = entry_terms, int(entry_count)

    def as_struct(self):
        b = CBatch32()
        b.rounds, b.count = self.rounds, self.count
        b.gid = _ptr(self.gid)
        b.head, b.abcd = _ptr(self.head), _ptr(self.abcd)
        b.entry_terms = _ptr(self.entry_terms) if self.entry_count else None
        b.entry_count = self.entry_count
        return b


OUT32_DT = np.dtype([("resp_term", "<i4"), ("flags", "<u4"), ("commit_index", "<i4"), ("log_from", "<i4")])
PERSIST32_DT = np.dtype([("term", "<i4"), ("voted_for", "<i4"), ("role_epoch", "<u4"), ("role", "<i4")])
assert OUT32_DT.itemsize == 16 and PERSIST32_DT.itemsize == 16


class Outcome32:
    """Host image of an rg_outcome32_t (compact outcome rows, ABI 4): one rg_out32_t per event, an rg_persist32_t where the row carries
    RG_F_PERSIST, and optionally the wide overflow columns for rows flagged RG_F_WIDE_VALUES."""

    def __init__(self, rows, fill=0, wide=True):
        self.rows = rows
        self.row = np.zeros(rows, dtype=OUT32_DT)
        self.persist = np.zeros(rows, dtype=PERSIST32_DT)
        self.wide = Outcome(rows, fill) if wide else None
        if fill:
            self.row.view(np.uint8)[:] = fill
            self.persist.view(np.uint8)[:] = fill

    def as_struct(self):
        o = COutcome32()
        o.row, o.persist = _ptr(self.row), _ptr(self.persist)
        if self.wide is not None:
            o.wide = self.wide.as_struct()
        return o


def has_logfx(flags):
    """rows whose reply says a logfx item exists (include/raftgpu.h rg_logfx_t)"""
    flags = np.asarray(flags)
    return ((flags & (F_COMMIT | F_LOG_APPEND | F_LOG_TRUNC)) != 0) | (flags_status(flags) == NEED_HOST)


def has_persist(flags):
    return (np.asarray(flags) & F_PERSIST) != 0


class Outcome:
    """Host image of an rg_outcome_t. Buffers are pre-filled with a sentinel so tests can see which
    conditional fields a side left untouched."""

    def __init__(self, rows, fill=0):
        self.reply = np.zeros(rows, dtype=REPLY_DT)
        self.logfx = np.zeros(rows, dtype=LOGFX_DT)
        self.persist = np.zeros(rows, dtype=PERSIST_DT)
        if fill:
            self.reply.view(np.uint8)[:] = fill
            self.logfx.view(np.uint8)[:] = fill
            self.persist.view(np.uint8)[:] = fill

    def as_struct(self):
        o = COutcome()
        o.reply, o.logfx, o.persist = _ptr(self.reply), _ptr(self.logfx), _ptr(self.persist)
        return o

    # decoded views -------------------------------------------------------------------------
    @property
    def status(self):
        return flags_status(self.reply["flags"])

    @property
    def success(self):
        return (self.reply["flags"] & F_SUCCESS) != 0

    @property
    def replied(self):
        return (self.reply["flags"] & F_REPLIED) != 0
