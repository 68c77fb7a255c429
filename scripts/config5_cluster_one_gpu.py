"""BASELINE config 5 at FULL size -- 16 777 216 members, 30 % message loss, as 8 shards of 2 097 152 -- as far as ONE GPU can
show it: the cluster of bounded handles (8 handles on one device, LocalFabric: the exchange is device-to-device copies) and the
same population on one unsharded bounded handle step the same ticks; digests and counters must agree.  The shards run one
after the other here, so the time says what the sharded path COSTS per shard (against the unsharded handle's time / SHARDS),
not how it scales; the bytes each shard would put on xGMI per tick are reported (all-gather of 64-byte queue lines + 1-byte
member bytes, 16-byte delivery records).  No multi-GPU number here.
usage (GPU box): config5_cluster_one_gpu.py      env: MEMBERS (16777216), SHARDS (8), CAP (64), WARM (6), TICKS (8), CHURN (per mille / 100 ticks, 10)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swim_amd import Config, Sim, SimConfig, workloads, _lib
from swim_amd.shard import LocalFabric, ShardedSim
abi = _lib.load()
N = int(os.environ.get("MEMBERS", 1 << 24)); G = int(os.environ.get("SHARDS", 8)); CAP = int(os.environ.get("CAP", 64))
WARM = int(os.environ.get("WARM", 6)); TICKS = int(os.environ.get("TICKS", 8)); CHURN = int(os.environ.get("CHURN", 10))


def run(make, name):
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=N, seed=1, lossPpm=300000, eventMask=0x10, viewCap=CAP)
    s = make(sc)
    T = WARM + TICKS
    churn = workloads.hashed_crashes(N, 9, max(1, CHURN * T // 100), 1000, 1, T) if CHURN else []
    for (t, m) in churn:
        s.crash(m, t); s.scheduleFault(t + 3 + (m % 4), m, True)
    s.step(WARM); torch.cuda.synchronize()
    c0 = s.counters()
    t0 = time.time(); s.step(TICKS); torch.cuda.synchronize(); dt = time.time() - t0
    c = s.counters(); c.pop("events_dropped", None)
    out = {"what": name, "members": N, "view_cap": CAP, "ms_per_tick": round(dt / TICKS * 1e3, 2), "member_ticks_per_s": round(N * TICKS / dt),
           "payloads_per_member_tick": round((c["payloads"] - c0["payloads"]) / float(N * TICKS), 2), "digest": "%016x" % s.digest()}
    if isinstance(s, ShardedSim):
        out["host_phase_breakdown_us"] = s.phaseBreakdown()
        per = N // G
        deliveries = (c["payloads"] - c0["payloads"]) / float(TICKS) / G           # per shard and tick
        out["per_shard_per_tick_on_the_wire_MB"] = {"all_gather_in": round((N - per) * 65 / 1e6, 1),
                                                    "delivery_records_out (<=)": round(deliveries * (G - 1) / G * 16 / 1e6, 1)}
    s.close()
    return out, c


res = [run(lambda sc: ShardedSim(abi, sc, LocalFabric(G), device="cuda:0"), "%d shards of %d members on one GPU" % (G, N // G)),
       run(lambda sc: Sim.create(abi, sc), "one unsharded handle")]
same = res[0][0]["digest"] == res[1][0]["digest"] and res[0][1] == res[1][1]
for r, _ in res:
    print(json.dumps(r), flush=True)
print(json.dumps({"digests_and_counters_agree": same, "ticks": WARM + TICKS,
                  "cost_of_sharding (sharded time / unsharded time, same GPU)": round(res[0][0]["ms_per_tick"] / res[1][0]["ms_per_tick"], 2)}))
