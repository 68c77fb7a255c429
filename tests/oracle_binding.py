"""Loads the CPU oracle (oracle/_build/libswim_oracle.so) for the tests.  TEST-ONLY: the
product package never imports this."""
import ctypes as C
import os
import subprocess

from swim_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "libswim_oracle.so")

_cached = None


class OMsg(C.Structure):
    """swimoracle_msg_t"""
    _fields_ = [("type", C.c_uint8), ("seq_no", C.c_uint32), ("node", C.c_uint32),
                ("target", C.c_uint32), ("incarnation", C.c_uint32), ("dead_from", C.c_uint32),
                ("to", C.c_uint32), ("broadcast", C.c_uint8)]


MSG_PING, MSG_INDIRECT_PING, MSG_ACK, MSG_SUSPECT, MSG_ALIVE, MSG_DEAD = range(6)


def build():
    src = os.path.join(ORACLE_DIR, "swim_oracle.c")
    if (not os.path.exists(LIB)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB)):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def load():
    global _cached
    if _cached is None:
        build()
        lib = C.CDLL(LIB)
        ns = _abi.bind(lib, "swimoracle_")
        H = C.c_void_p
        lib.swimoracle_process.restype = C.c_int
        lib.swimoracle_process.argtypes = [H, C.c_uint32, C.c_uint32, C.POINTER(OMsg), C.c_int,
                                           C.POINTER(OMsg), C.c_size_t, C.POINTER(C.c_size_t)]
        lib.swimoracle_reference_rule.restype = C.c_uint32
        lib.swimoracle_reference_rule.argtypes = [C.c_uint32, C.c_uint32]
        lib.swimoracle_merge_rule.restype = C.c_uint32
        lib.swimoracle_merge_rule.argtypes = [C.c_uint32, C.c_uint32]
        lib.swimoracle_remove_dead_nodes.restype = C.c_size_t
        lib.swimoracle_remove_dead_nodes.argtypes = [C.POINTER(_abi.ViewEntry), C.c_size_t]
        lib.swimoracle_hash.restype = C.c_uint32
        lib.swimoracle_hash.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.swimoracle_set_shuffle.restype = C.c_int
        lib.swimoracle_set_shuffle.argtypes = [H, C.c_uint64]
        lib.swimoracle_set_literal_rule.restype = C.c_int
        lib.swimoracle_set_literal_rule.argtypes = [H, C.c_int]
        lib.swimoracle_d13_hits.restype = C.c_uint64
        lib.swimoracle_d13_hits.argtypes = [H]
        lib.swimoracle_set_threads.restype = C.c_int
        lib.swimoracle_set_threads.argtypes = [H, C.c_uint32]
        _cached = ns
    return _cached


def set_threads(sim, n):
    """Step this oracle handle with n member-range threads (results do not depend on n)."""
    rc = load().lib.swimoracle_set_threads(sim._h, n)
    assert rc == 0, rc


def process(sim, self_id, sender, msg, literal_d8=False):
    """swimoracle_process in capture mode -> list of OMsg copies."""
    ns = load()
    out = (OMsg * 8)()
    n = C.c_size_t()
    rc = ns.lib.swimoracle_process(sim._h, self_id, sender, C.byref(msg), 1 if literal_d8 else 0, out, 8, C.byref(n))
    assert rc == 0, rc
    res = []
    for k in range(n.value):
        m = OMsg()
        C.memmove(C.byref(m), C.byref(out[k]), C.sizeof(OMsg))
        res.append(m)
    return res
