"""A/B timing of library variants on one GPU box (MI355X): all variants hold the SAME cluster (same schedule, same
digests), are warmed up together and then stepped in turn, CHUNK ticks at a time, for ROUNDS rounds -- so that clock
drift, the phase of the workload and the placement of a fresh allocation hit every variant alike.  Reports the median
(and min) over the rounds of each tick kernel's HIP-event time per tick.
usage: ab_time.py lib.so [lib.so ...]    env: WARM, CHUNK, ROUNDS, MEMBERS, LOSS (ppm), GC=1, SCHEME=robust, P, CPT (crashes per tick: 9.5 = BASELINE.md
row 3(s) as written, with GC=1 MAXSUBJ=8192), MAXSUBJ"""
import json, os, statistics, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads, _abi
WARM = int(os.environ.get('WARM', 150)); CHUNK = int(os.environ.get('CHUNK', 40)); ROUNDS = int(os.environ.get('ROUNDS', 7))
N = int(os.environ.get('MEMBERS', 1 << 20)); LOSS = int(os.environ.get('LOSS', 0)); P = int(os.environ.get('P', 3))
CPT = float(os.environ.get('CPT', 1.0)); MAXSUBJ = int(os.environ.get('MAXSUBJ', 0))
sims = []
envs = {}                                           # "lib.so@VAR=VAL[,VAR=VAL]": environment knobs the library reads per call, set around this variant's steps
for spec in sys.argv[1:]:
    path, _, ev = spec.partition("@")
    envs[spec] = dict(kv.split("=", 1) for kv in ev.split(",")) if ev else {}
    os.environ.update(envs[spec])
    abi = _abi.bind(C.CDLL(os.path.abspath(path)), "swimsim_")
    sc, crashes, _ = workloads.saturated(N, WARM + CHUNK * ROUNDS, loss_ppm=LOSS, num_to_gossip=P, crashes_per_tick=CPT, t0=0)
    if MAXSUBJ:
        sc.maxSubjects = MAXSUBJ
    if os.environ.get('GC'):
        sc.gcTicks = _abi.GC_AUTO
    sc.targetScheme = 1 if os.environ.get('SCHEME') == 'robust' else 0
    s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes)
    s.step(WARM)
    for k_ in envs[spec]: os.environ.pop(k_, None)
    sims.append((os.path.basename(spec), s, {"probe": [], "merge": [], "wall": []}))
for r in range(ROUNDS):
    order = sims if r % 2 == 0 else sims[::-1]
    for name, s, acc in order:
        os.environ.update(envs.get(name, {}) or next((v for k_, v in envs.items() if os.path.basename(k_) == name), {}))
        s.kernelTimingEnable(True)
        t0 = time.time(); s.step(CHUNK); dt = time.time() - t0
        for v_ in envs.values():
            for k_ in v_: os.environ.pop(k_, None)
        kt = s.kernelTiming()
        acc["probe"].append(kt["probe_ms"] * 1e3 / kt["ticks"]); acc["merge"].append(kt["merge_ms"] * 1e3 / kt["ticks"])
        acc["wall"].append(dt / CHUNK * 1e6)
for name, s, acc in sims:
    print(json.dumps({"lib": name, "probe_us": round(statistics.median(acc["probe"]), 1), "merge_us": round(statistics.median(acc["merge"]), 1),
                      "wall_us": round(statistics.median(acc["wall"]), 1), "merge_min": round(min(acc["merge"]), 1), "probe_min": round(min(acc["probe"]), 1),
                      "merge_all": [round(x, 1) for x in acc["merge"]], "digest": "%016x" % s.digest()}), flush=True)
    s.close()
