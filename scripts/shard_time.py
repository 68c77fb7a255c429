"""Cost of the sharded path on ONE GPU: the same 1M-member saturated workload as 1 handle vs G handles
(LocalFabric: the exchange is device-to-device copies), to separate exchange-kernel + host-side overhead
from interconnect time -- with the record path and with replicated queue masks (SWIMSIM_SHARD_REPLICATED_MASKS, read at
create: both are timed in one run).  usage: shard_time.py [G ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swim_amd import Sim, workloads, _lib
from swim_amd.shard import LocalFabric, ShardedSim
abi = _lib.load()
N = int(os.environ.get("MEMBERS", 1 << 20)); WARM = int(os.environ.get("WARM", 150)); TICKS = int(os.environ.get("TICKS", 50))
for G, RM in [(g, rm) for g in ([int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]) for rm in (("0", "1") if g > 1 else ("0",))]:
    os.environ["SWIMSIM_SHARD_REPLICATED_MASKS"] = RM
    sc, crashes, _ = workloads.saturated(N, WARM + TICKS)
    s = Sim.create(abi, sc) if G == 1 else ShardedSim(abi, sc, LocalFabric(G), device="cuda:0")
    workloads.apply_crashes(s, crashes)
    s.step(WARM); torch.cuda.synchronize()
    t0 = time.time(); s.step(TICKS); torch.cuda.synchronize(); dt = time.time() - t0
    print(json.dumps({"shards_on_one_gpu": G, "replicated_masks": RM, "members": N, "us_per_tick": round(dt / TICKS * 1e6, 1),
                      "Gmt_per_s": round(N * TICKS / dt / 1e9, 3), "digest": "%016x" % s.digest()}), flush=True)
    s.close()

# where a sharded tick spends its time (host view, G = 2)
for RM in (("0", "1") if os.environ.get("PHASES", "1") == "1" else ()):
    G = 2
    os.environ["SWIMSIM_SHARD_REPLICATED_MASKS"] = RM
    sc, crashes, _ = workloads.saturated(N, WARM + TICKS)
    s = ShardedSim(abi, sc, LocalFabric(G), device="cuda:0")
    workloads.apply_crashes(s, crashes)
    s.step(WARM); torch.cuda.synchronize()
    acc = [0.0] * 5
    f, sh = s.fabric, s.shards
    for _ in range(TICKS):
        t0 = time.perf_counter(); c1 = [x.phase1() for x in sh]
        t1 = time.perf_counter()
        if sh[0].replicated:
            full = lambda x: [0 if p == x.index else x.n_local for p in range(G)]
            f.exchange(sh, (5, 6), [[full(x), full(x)] for x in sh])
        r_in = f.exchange(sh, (0,), [[c[0]] for c in c1])
        t2 = time.perf_counter(); c2 = [x.phase2(r_in[k][0]) for k, x in enumerate(sh)]
        t3 = time.perf_counter(); px = f.exchange(sh, (1, 2), [[c[1], c[2]] for c in c2])
        t4 = time.perf_counter()
        for k, x in enumerate(sh):
            x.phase3(px[k][0], px[k][1])
        t5 = time.perf_counter()
        for j, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[j] += d
    print(json.dumps({"replicated_masks": RM, "per_tick_us_both_shards": {k: round(v / TICKS * 1e6, 1) for k, v in zip(("phase1", "round1", "phase2", "round2", "phase3"), acc)},
                      "records_per_shard": {"round1_records": c1[0][0][1] - 32, "mask_payloads": c2[0][1][1], "explicit_payloads": c2[0][2][1]}}), flush=True)
    s.close()
