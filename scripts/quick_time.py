"""Quick wall-clock probe of the HIP tick at 1M members (not the bench contract).
usage: quick_time.py [per_mille ...]   (0 = one crash, n = n/1000 of members crash over 1000 ticks)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads, _lib
abi = _lib.load()
regimes = [int(x) for x in sys.argv[1:]] or [0, 1]
WARM = int(os.environ.get('WARM', 150)); TICKS = int(os.environ.get('TICKS', 400))
for per_mille in regimes:
    sc, crashes, _ = workloads.config3(crash_per_mille=per_mille, t0=10, t1=1010)
    t0 = time.time(); s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes); t_create = time.time() - t0
    s.step(WARM)
    c0 = s.counters()
    t0 = time.time(); s.step(TICKS); dt = time.time() - t0
    c = s.counters()
    print(json.dumps({"per_mille": per_mille, "crashes": len(crashes), "create_s": round(t_create, 2),
                      "ticks_per_s": round(TICKS / dt, 1), "Gmember_ticks_per_s": round(TICKS * sc.nMembers / dt / 1e9, 3),
                      "us_per_tick": round(dt / TICKS * 1e6, 1),
                      "changes_per_mt": round((c["changes"] - c0["changes"]) / TICKS / sc.nMembers, 3),
                      "payloads_per_mt": round((c["payloads"] - c0["payloads"]) / TICKS / sc.nMembers, 3)}))
    s.close()
