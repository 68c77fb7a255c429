"""Builds and loads tests/hostemu/_build/libswimsim_hostemu.so: the PRODUCT's kernel sources
(swim_amd/csrc) compiled by g++ against a stand-in <hip/hip_runtime.h> (tests/hostemu/hip/).
TEST-ONLY: lets the kernels' logic be parity-checked against the oracle on a machine with no GPU.
The product package never imports this; libswimsim.so itself has no CPU path."""
import ctypes as C
import fcntl
import os
import subprocess

from swim_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hostemu")
CSRC = os.path.join(ROOT, "swim_amd", "csrc")
LIB = os.path.join(EMU, "_build", "libswimsim_hostemu.so")

_cached = None
# SWIM_EMU_EXTRA="tag:DEF1=1,DEF2" (test knob): EVERY emulation build of the session gets these defines on top of its own, under
# library names of their own -- the whole CPU suite against an A/B build of the kernels (e.g. the other view-cell layout)
_EXTRA = os.environ.get("SWIM_EMU_EXTRA", "")
_XTAG, _XDEFS = (_EXTRA.split(":", 1)[0], _EXTRA.split(":", 1)[1].split(",")) if ":" in _EXTRA else ("", [])


def build(lib=None, defines=()):
    lib = lib or LIB
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp"))]
    srcs += [os.path.join(EMU, "hip", "hip_runtime.h")] + [os.path.join(ROOT, "include", f) for f in ("swimsim.h", "swimwire.h", "swimbridge.h")]
    fresh = lambda: os.path.exists(lib) and all(os.path.getmtime(s) <= os.path.getmtime(lib) for s in srcs)
    if fresh():
        return
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    # several test processes may want the same library at once (xdist, the soak): one builds, into a
    # name of its own, and renames -- nobody ever maps a half-written file
    with open(lib + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if fresh():
            return
        tmp = "%s.%d.tmp" % (lib, os.getpid())
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", EMU,
                               "-Wno-unused-function", "-Wl,-Bsymbolic", *["-D" + d for d in defines], "-o", tmp, os.path.join(CSRC, "swimsim.hip"),
                               os.path.join(CSRC, "swim_wire.cpp"), os.path.join(CSRC, "swim_bridge.cpp")])
        os.replace(tmp, lib)


def load():
    global _cached
    if _cached is None:
        lib = LIB if not _XTAG else os.path.join(EMU, "_build", "libswimsim_hostemu_x%s.so" % _XTAG)
        build(lib, _XDEFS)
        _cached = _abi.bind(C.CDLL(lib), "swimsim_")
    return _cached


def load_variant(tag, defines):
    """A second build of the same sources with compile-time knobs changed (e.g. a tiny examination
    list so that the overflow path runs), for stress tests."""
    lib = os.path.join(EMU, "_build", "libswimsim_hostemu_%s%s.so" % (tag, ("_x" + _XTAG) if _XTAG else ""))
    build(lib, list(defines) + _XDEFS)
    return _abi.bind(C.CDLL(lib), "swimsim_")
