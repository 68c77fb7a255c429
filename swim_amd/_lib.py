"""Loads the product library libswimsim.so (HIP / gfx950).  No fallback: if the
library is missing or a symbol of include/swimsim.h is absent this raises."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libswimsim.so")

_cached = None

KERNEL_SOURCES = ("swim_kernels.h", "swim_sparse.h", "swim_device.h", "swimsim.hip")


def kernel_sources_sha():
    """sha256 (first 16 hex digits) of the kernel sources the library is built from: what profiles/traffic.json is keyed
    by, so that a PMC figure is never quoted for kernels it was not measured on (bench.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(_HERE, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def load_variant(tag):
    """A test build of the same sources with a compile-time knob shrunk (__graft_entry__.VARIANTS)."""
    path = os.path.join(_HERE, "csrc", "libswimsim_%s.so" % tag)
    if not os.path.exists(path):
        raise ImportError("%s not built: run __graft_entry__.build()" % path)
    return _abi.bind(C.CDLL(path, mode=C.RTLD_LOCAL), "swimsim_")


def load():
    """Return the bound ABI namespace of libswimsim.so (prefix swimsim_)."""
    global _cached
    if _cached is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libswimsim.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _cached = _abi.bind(lib, "swimsim_")
    return _cached
