"""BASELINE config 4 -- 4 194 304 members over 4 shards of 1 048 576 -- as far as ONE GPU can show it: the sharded cluster
(4 handles on one device stepped by swimsim_cluster_step: the exchange inside the library), the same population on one unsharded
handle, and the CPU oracle (32 threads) step the same saturated workload; digests and counters must agree.  Timing on one
GPU says what the sharded path COSTS (the shards run one after the other), not how it scales: no multi-GPU number here.
usage (GPU box): config4_one_gpu.py        env: MEMBERS (default 4194304), SHARDS (4), WARM (100), TICKS (40), ORACLE=0 to skip"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swim_amd import Sim, workloads, _lib
from swim_amd.shard import LocalFabric, ShardedSim
abi = _lib.load()
N = int(os.environ.get("MEMBERS", 1 << 22)); G = int(os.environ.get("SHARDS", 4))
WARM = int(os.environ.get("WARM", 100)); TICKS = int(os.environ.get("TICKS", 40))


def run(make, name):
    sc, crashes, _ = workloads.saturated(N, WARM + TICKS)
    sc.eventMask = 0
    s = make(sc)
    workloads.apply_crashes(s, crashes)
    s.step(WARM); torch.cuda.synchronize()
    t0 = time.time(); s.step(TICKS); torch.cuda.synchronize(); dt = time.time() - t0
    c = s.counters(); c.pop("events_dropped", None)
    out = {"what": name, "members": N, "us_per_tick": round(dt / TICKS * 1e6, 1), "member_ticks_per_s": round(N * TICKS / dt, 1),
           "digest": "%016x" % s.digest()}
    s.close()
    return out, c


res = []
res.append(run(lambda sc: ShardedSim(abi, sc, LocalFabric(G), device="cuda:0"), "%d shards of %d members on one GPU" % (G, N // G)))
res.append(run(lambda sc: Sim.create(abi, sc), "one unsharded handle"))
if os.environ.get("ORACLE", "1") != "0":
    from tests import oracle_binding

    def mk(sc):
        o = Sim.create(oracle_binding.load(), sc)
        oracle_binding.set_threads(o, min(32, os.cpu_count() or 1))
        return o
    res.append(run(mk, "CPU oracle, %d threads" % min(32, os.cpu_count() or 1)))
same = all(r[0]["digest"] == res[0][0]["digest"] and r[1] == res[0][1] for r in res)
for r, _ in res:
    print(json.dumps(r), flush=True)
print(json.dumps({"digests_and_counters_agree": same, "ticks": WARM + TICKS, "changes": res[0][1]["changes"], "payloads": res[0][1]["payloads"]}))
