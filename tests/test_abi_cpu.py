"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol that
include/raftgpu.h declares, agrees with the Python mirror on layouts/constants, and refuses to run
without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rafting_amd import abi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "raftgpu.h")).read()


def test_library_exports_every_declared_symbol():
    declared = set(re.findall(r"\b(rg_[a-z_0-9]+)\s*\(", HEADER))
    declared -= {"rg_table_t"}
    assert declared == set(engine.exported_symbols()), declared ^ set(engine.exported_symbols())
    L = C.CDLL(engine.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert L.rg_abi_version() == abi.ABI_VERSION


def test_python_mirror_matches_header_constants():
    def define(name):
        m = re.search(r"#define\s+%s\s+\(?\s*(-?\d+)(?:u)?\s*(?:<<\s*(\d+))?\)?" % name, HEADER)
        assert m, name
        return int(m.group(1)) << int(m.group(2) or 0)
    assert define("RG_ABI_VERSION") == abi.ABI_VERSION
    assert define("RG_TERM_RUNS") == abi.TERM_RUNS
    assert (define("RG_MIN_CLUSTER"), define("RG_MAX_CLUSTER")) == (abi.MIN_CLUSTER, abi.MAX_CLUSTER)
    for flag in ("SUCCESS", "REPLIED", "PERSIST", "ROLE_CHANGED", "RESET_TIMER", "COMMIT", "LOG_TRUNC", "LOG_APPEND"):
        assert define("RG_F_" + flag) == getattr(abi, "F_" + flag), flag
    enums = dict(re.findall(r"^\s*(RG_[A-Z_0-9]+)\s*=\s*(\d+)\s*,?\s*(?:/\*.*)?$", HEADER, re.M))
    assert len(enums) >= 35
    for name, val in enums.items():
        short = name[3:]
        py = getattr(abi, short, None)
        if py is None and short.startswith("EV_"):
            py = getattr(abi, short)
        assert py == int(val), name


def test_wire_struct_sizes():
    assert abi.HEAD_DT.itemsize == 8 and abi.PAIR_DT.itemsize == 16
    assert abi.REPLY_DT.itemsize == abi.LOGFX_DT.itemsize == abi.PERSIST_DT.itemsize == 16
    assert C.sizeof(abi.CBatch) == 64 and C.sizeof(abi.COutcome) == 24
    assert C.sizeof(abi.CGroupState) == 8 * 24
    h = int(abi.hdr_make(abi.EV_AE_REQ, slot=5, flag=1, n=1234))
    assert (h & 0xF, (h >> 4) & 0xF, (h >> 8) & 1, h >> 12) == (abi.EV_AE_REQ, 5, 1, 1234)


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return not os.path.exists("/dev/kfd")


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_no_silent_cpu_fallback():
    with pytest.raises(engine.EngineError, match="no HIP device|failed"):
        engine.Table(4, 3)
    h = C.c_void_p()
    assert engine.lib().rg_table_create(0, 4, 3, 0, 1, C.byref(h)) != 0
    assert b"HIP" in engine.lib().rg_last_error(None)


def test_argument_validation_needs_no_gpu():
    h = C.c_void_p()
    L = engine.lib()
    assert L.rg_table_create(0, 0, 3, 0, 1, C.byref(h)) == -1
    assert L.rg_table_create(0, 4, 16, 0, 1, C.byref(h)) == -1                      # (2 .. 15 nodes since ABI 5)
    assert L.rg_table_create(0, 4, 3, 3, 1, C.byref(h)) == -1
    assert b"self_slot" in L.rg_last_error(None)


def test_wire_library_exports_every_declared_symbol():
    """include/raftwire.h (the host-side wire decoder and the ingress): every declared entry point is exported by build/libraftwire.so,
    and the library exports no rw_* symbol the header does not declare."""
    import subprocess
    from rafting_amd import wirelib
    header = open(os.path.join(ROOT, "include", "raftwire.h")).read()
    declared = set(re.findall(r"\b(rw_[a-z_0-9]+)\s*\(", header)) - {"rw_splitter_t", "rw_ingress_t", "rw_repair_host_t"}
    L = wirelib.lib()
    for name in declared:
        assert hasattr(L, name), name
    nm = subprocess.run(["nm", "-D", "--defined-only", wirelib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith("rw_")}
    assert exported == declared, exported ^ declared
