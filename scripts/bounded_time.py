"""Bounded member maps (view_cap): probe / merge kernel time per tick from the library's HIP events, at config 5's load
(30 % loss unless LOSS=).  usage: bounded_time.py <members> <cap> [<members> <cap> ...]   TICKS= WARM= LIB=<library path>"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Config, Sim, SimConfig, _abi, _lib

LOSS = int(os.environ.get("LOSS", 300000)); TICKS = int(os.environ.get("TICKS", 20)); WARM = int(os.environ.get("WARM", 12))
abi = _abi.bind(C.CDLL(os.environ["LIB"]), "swimsim_") if os.environ.get("LIB") else _lib.load()
args = [int(x) for x in sys.argv[1:]]
for n, cap in zip(args[0::2], args[1::2]):
    sc = SimConfig(cfg=Config(numToGossip=int(os.environ.get("K", 3))), nMembers=n, seed=1, lossPpm=LOSS, eventMask=0x10, viewCap=cap)
    s = Sim.create(abi, sc)
    s.step(WARM)
    c0 = s.counters()
    s.kernelTimingEnable(True)
    t0 = time.time(); s.step(TICKS); dt = time.time() - t0
    kt = s.kernelTiming(); c1 = s.counters()
    mt = float(n) * TICKS
    ts = (C.c_uint64 * 9)(); abi.table_stats(s._h, ts, 9)
    print(json.dumps({"members": n, "view_cap": cap, "grids": [int(ts[7]), int(ts[8])], "loss_ppm": LOSS, "ms_per_tick_wall": round(dt / TICKS * 1e3, 3),
                      "probe_us": round(kt["probe_ms"] / kt["ticks"] * 1e3, 1), "merge_us": round(kt["merge_ms"] / kt["ticks"] * 1e3, 1),
                      "member_ticks_per_s": round(mt / dt), "payloads": round((c1["payloads"] - c0["payloads"]) / mt, 2),
                      "changes": round((c1["changes"] - c0["changes"]) / mt, 2), "evicted": round((c1["evicted"] - c0["evicted"]) / mt, 2),
                      "digest": "%016x" % s.digest()}), flush=True)
    s.close()
