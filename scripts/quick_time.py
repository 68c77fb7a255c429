"""Quick per-kernel timing of the HIP tick at 1M members, saturated regime (not the bench contract).
usage: quick_time.py [lib.so ...]   -- each library variant is timed with the library's own HIP events.
env: WARM, TICKS, MEMBERS, REGIME (saturated|quiescent), SCHEME=robust, LOSS (ppm), GC=1, STRICT=1 (strict_reference_rules)"""
import json, os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads, _lib, _abi
WARM = int(os.environ.get('WARM', 150)); TICKS = int(os.environ.get('TICKS', 200))
N = int(os.environ.get('MEMBERS', 1 << 20)); REGIME = os.environ.get('REGIME', 'saturated')
libs = sys.argv[1:] or [_lib.LIB_PATH]
for path in libs:
    abi = _abi.bind(C.CDLL(os.path.abspath(path)), "swimsim_")
    mk = workloads.saturated if REGIME == 'saturated' else workloads.quiescent
    sc, crashes, _ = mk(N, WARM + TICKS, loss_ppm=int(os.environ.get('LOSS', 0))) if REGIME == 'saturated' else mk(N, WARM + TICKS)
    if os.environ.get('GC'):
        sc.gcTicks = _abi.GC_AUTO
    sc.targetScheme = 1 if os.environ.get('SCHEME') == 'robust' else 0
    sc.strictReferenceRules = bool(os.environ.get('STRICT'))
    s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes)
    s.step(WARM)
    s.kernelTimingEnable(True)
    raw0 = (C.c_uint64 * _abi.CTR_COUNT)(); abi.counters(s._h, raw0, _abi.CTR_COUNT)
    t0 = time.time(); s.step(TICKS); dt = time.time() - t0
    raw1 = (C.c_uint64 * _abi.CTR_COUNT)(); abi.counters(s._h, raw1, _abi.CTR_COUNT)
    kt = s.kernelTiming()
    print(json.dumps({"lib": os.path.basename(path), "regime": REGIME, "us_per_tick": round(dt / TICKS * 1e6, 1),
                      "probe_us": round(kt["probe_ms"] * 1e3 / kt["ticks"], 1),
                      "merge_us": round(kt["merge_ms"] * 1e3 / kt["ticks"], 1),
                      "Gmt_per_s": round(TICKS * N / dt / 1e9, 3),
                      "examined_per_mt": round((raw1[14] - raw0[14]) / TICKS / N, 3),
                      "changes_per_mt": round((raw1[7] - raw0[7]) / TICKS / N, 3),
                      "digest": "%016x" % s.digest()}), flush=True)
    s.close()
