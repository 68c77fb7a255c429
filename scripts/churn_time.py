"""BASELINE config 5's churn on single-GPU slices (settling on): C per mille of the members crash and come back 50 ticks
later per 100 ticks.  usage (GPU box): churn_time.py [members per_mille_per_100 [join_pull]] ...   default: the three cases of DESIGN.md
ORACLE=<max members>: cases up to that size are replayed on the CPU oracle (32 threads) and compared (digest, counters)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Config, Sim, SimConfig, _abi, _lib, workloads
abi = _lib.load()
cases = [(1 << 21, 1, 0), (1 << 21, 1, 1), (1 << 18, 10, 0)]
if len(sys.argv) > 1:
    a = [int(x) for x in sys.argv[1:]]; cases = [(a[k], a[k + 1], a[k + 2]) for k in range(0, len(a), 3)]
T = int(os.environ.get("TICKS", 400))
for n, pm, jp in cases:
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=9, maxSubjects=int(os.environ.get("ROWS", 12000)), gcTicks=_abi.GC_AUTO, eventMask=1, joinPull=jp)
    s = Sim.create(abi, sc)
    churn = workloads.hashed_crashes(n, 9, pm * T // 100, 1000, 5, T - 60)
    for (t, m) in churn:
        s.crash(m, t); s.scheduleFault(t + 50, m, True)
    s.step(100)
    t0 = time.time(); s.step(T - 100); dt = time.time() - t0
    st = s.tableStats(); c = s.counters()
    extra = {}
    if n <= int(os.environ.get("ORACLE", 0)):
        from tests import oracle_binding
        o = Sim.create(oracle_binding.load(), sc)
        oracle_binding.set_threads(o, min(32, os.cpu_count() or 1))
        for (t, m) in churn:
            o.crash(m, t); o.scheduleFault(t + 50, m, True)
        t1 = time.time(); o.step(T); od = time.time() - t1
        co = o.counters(); co.pop("events_dropped"); cg = dict(c); cg.pop("events_dropped")
        extra = {"verified_vs_oracle": bool(o.digest() == s.digest() and co == cg), "oracle_ms_per_tick": round(od / T * 1e3, 2), "oracle_threads": min(32, os.cpu_count() or 1)}
        o.close()
    print(json.dumps({**extra, "members": n, "churn_percent_per_100_ticks": pm / 10.0, "join_pull": jp, "ticks_timed": T - 100, "ms_per_tick": round(dt / (T - 100) * 1e3, 3),
                      "crash_rejoin_pairs": len(churn), "rejoins_per_tick": round(len(churn) / (T - 65.0), 1), "settled": c["settled"], "refutes": c["refutes"],
                      "rows_high_water": int(st["rows_high_water"]), "subjects_live": int(st["subjects_live"]), "digest": "%016x" % s.digest()}), flush=True)
    s.close()
