"""Cost of the sharded path on ONE GPU: the same 1M-member saturated workload as 1 handle vs G handles
(LocalFabric: the exchange is device-to-device copies), to separate exchange-kernel + host-side overhead
from interconnect time.  usage: shard_time.py [G ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swim_amd import Sim, workloads, _lib
from swim_amd.shard import LocalFabric, ShardedSim
abi = _lib.load()
N = int(os.environ.get("MEMBERS", 1 << 20)); WARM = int(os.environ.get("WARM", 150)); TICKS = int(os.environ.get("TICKS", 50))
for G in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    sc, crashes, _ = workloads.saturated(N, WARM + TICKS)
    s = Sim.create(abi, sc) if G == 1 else ShardedSim(abi, sc, LocalFabric(G), device="cuda:0")
    workloads.apply_crashes(s, crashes)
    s.step(WARM); torch.cuda.synchronize()
    t0 = time.time(); s.step(TICKS); torch.cuda.synchronize(); dt = time.time() - t0
    print(json.dumps({"shards_on_one_gpu": G, "members": N, "us_per_tick": round(dt / TICKS * 1e6, 1),
                      "Gmt_per_s": round(N * TICKS / dt / 1e9, 3), "digest": "%016x" % s.digest()}), flush=True)
    s.close()
