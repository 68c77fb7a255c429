"""SURVEY 8(d) config 3(s) read literally: 1 % of 1 048 576 members crash, spread over 1 100 ticks (~9.5
crashes per tick, ~20 new rumours per tick).  That overloads an 8-slot piggyback queue (the protocol, not
the implementation); this script measures what the tick costs there and how complete dissemination is."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads, _lib
abi = _lib.load()
n = 1 << 20
WARM, TICKS = 150, 100
crashes = workloads.hashed_crashes(n, 1, 10, 1000, 10, 1110)
from swim_amd import Config, SimConfig
sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=1, maxSubjects=16384)
s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes)
s.step(WARM); c0 = s.counters(); s.kernelTimingEnable(True)
t0 = time.time(); s.step(TICKS); dt = time.time() - t0
c1 = s.counters(); kt = s.kernelTiming()
mt = float(n) * TICKS
print(json.dumps({"regime": "literal 1 % over 1100 ticks", "crashes": len(crashes), "us_per_tick": round(dt / TICKS * 1e6, 1),
                  "Gmt_per_s": round(mt / dt / 1e9, 3), "probe_us": round(kt["probe_ms"] * 1e3 / kt["ticks"], 1),
                  "merge_us": round(kt["merge_ms"] * 1e3 / kt["ticks"], 1),
                  "d": round((c1["payloads"] - c0["payloads"]) / mt, 2), "r": round((c1["changes"] - c0["changes"]) / mt, 2),
                  "false_suspects": c1["false_suspects"]}))
