"""Long and churny runs on one MI355X: bounded view rows under settling.
(a) 1 048 576 members, ~1 crash per tick for 100 000 ticks with room for 512 subjects (the run goes through
    ~100 000): rows are reclaimed and reused, the high-water mark stays bounded;
(b) BASELINE config 5's churn on a 2 097 152-member single-GPU slice: {0, 0.1} % of the members crash and come
    back (50 ticks later) per 100 ticks, 400 ticks each (1 % per 100 ticks is beyond a dense view: ~50 000 rows).
usage (GPU box): python scripts/long_run.py [ticks_a]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Config, Sim, SimConfig, _abi, _lib, workloads
abi = _lib.load()
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n = 1 << 20
sc, crashes, _ = workloads.saturated(n, ticks, seed=7, t0=0)
sc.maxSubjects = 512; sc.gcTicks = _abi.GC_AUTO
s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes)
t0 = time.time(); done = 0
while done < ticks:
    k = min(10000, ticks - done); s.step(k); done += k
    st = s.tableStats(); c = s.counters()
    print(json.dumps({"run": "a", "tick": done, "wall_s": round(time.time() - t0, 1), "crashes_scheduled": len(crashes), "settled": c["settled"],
                      "false_suspects": c["false_suspects"], **{k2: int(v) for k2, v in st.items()}}), flush=True)
s.close()
n = 1 << 21
for per_mille_per_100 in (0, 1):
    T = 400
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=9, maxSubjects=8000, gcTicks=_abi.GC_AUTO, eventMask=1)
    s = Sim.create(abi, sc)
    churn = workloads.hashed_crashes(n, 9, per_mille_per_100 * T // 100, 1000, 5, T - 60) if per_mille_per_100 else []
    for (t, m) in churn:
        s.crash(m, t); s.scheduleFault(t + 50, m, True)
    t0 = time.time(); s.step(T); dt = time.time() - t0
    st = s.tableStats(); c = s.counters()
    print(json.dumps({"run": "b", "members": n, "churn_percent_per_100_ticks": per_mille_per_100 / 10.0, "ticks": T, "ms_per_tick": round(dt / T * 1e3, 3),
                      "crash_rejoin_pairs": len(churn), "settled": c["settled"], "refutes": c["refutes"], "false_suspects": c["false_suspects"],
                      **{k2: int(v) for k2, v in st.items()}}), flush=True)
    s.close()
