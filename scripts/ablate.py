"""What each class of memory operations costs the two tick kernels (MI355X): libswimsim_abl.so (-DSWIM_ABLATE) leaves a
class out when its bit is set (swim_device.h ABL_*), so the launch time with and without it can be compared.  The
cluster is warmed up with the full kernels; only the timed ticks run ablated (their results are wrong by construction
and are not looked at).  usage: ablate.py [lib]   env: WARM, TICKS, MEMBERS, LOSS, GC"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads, _abi
here = os.path.dirname(os.path.abspath(__file__))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "swim_amd", "csrc", "libswimsim_abl.so")
WARM = int(os.environ.get("WARM", 150)); TICKS = int(os.environ.get("TICKS", 12)); N = int(os.environ.get("MEMBERS", 1 << 20))
lib = C.CDLL(os.path.abspath(path)); abi = _abi.bind(lib, "swimsim_")
lib.swimsim_debug_ablate.argtypes = [C.c_void_p, C.c_uint32]
CASES = [("full kernels", 0), ("probe: no push atomics", 1), ("probe: no pk gathers of the targets", 2), ("probe: no mb gathers", 4),
         ("probe: no atomics, no pk, no mb", 7), ("merge: no V stores", 16), ("merge: no V loads", 32), ("merge: no V loads, no V stores", 48),
         ("merge: no own-line loads", 64), ("merge: no line store", 128), ("merge: no event digest", 256), ("merge: no find_rid", 512),
         ("merge: no group_put / kill_slot", 1024), ("merge: no pk / inmask / trow stores", 2048), ("merge: no deadlines", 8192),
         ("merge: no delivered rumours", 16384), ("merge: no V, no lines, no state stores", 48 | 64 | 128 | 2048)]
for name, mask in CASES:
    sc, crashes, _ = workloads.saturated(N, WARM + TICKS, loss_ppm=int(os.environ.get("LOSS", 0)))
    if os.environ.get("GC"):
        sc.gcTicks = _abi.GC_AUTO
    s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes)
    s.step(WARM)
    lib.swimsim_debug_ablate(s._h, mask)
    s.kernelTimingEnable(True)
    try:
        s.step(TICKS)
    except Exception as e:      # a capacity flag raised by wrong results: the timing stands
        print("  (%s)" % e)
    kt = s.kernelTiming()
    print(json.dumps({"case": name, "mask": mask, "probe_us": round(kt["probe_ms"] * 1e3 / max(1, kt["ticks"]), 1),
                      "merge_us": round(kt["merge_ms"] * 1e3 / max(1, kt["ticks"]), 1)}), flush=True)
    lib.swimsim_debug_ablate(s._h, 0)
    s.close()
