"""Cost of the sharded path on ONE GPU: the same saturated workload as 1 handle vs G handles of one population, both forms of the
exchange: `cluster` = swimsim_cluster_step (the exchange inside the library: the peers' buffers read in place,
ordered by events on the handles' streams, no host in the loop), `phases` = swimsim_shard_phase1/2/3 + LocalFabric's copies (what a
one-process-per-GPU embedder drives).  One GPU runs the handles' kernels one after the other (or overlapped where they fit): the
figure is what sharding COSTS in kernel work, not how it scales.  usage: shard_time.py [G ...]   env: MEMBERS WARM TICKS KERNELS=1 FORMS=cluster,phases"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swim_amd import Sim, workloads, _lib
from swim_amd.shard import LocalFabric, ShardedSim
abi = _lib.load()
N = int(os.environ.get("MEMBERS", 1 << 20)); WARM = int(os.environ.get("WARM", 150)); TICKS = int(os.environ.get("TICKS", 50))
base = None
for G, form in [(g, f) for g in ([int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]) for f in (tuple(os.environ.get("FORMS", "cluster,phases").split(",")) if g > 1 else ("one handle",))]:
    os.environ["SWIMSIM_CLUSTER_STEP"] = "0" if form == "phases" else "1"
    sc, crashes, _ = workloads.saturated(N, WARM + TICKS)
    s = Sim.create(abi, sc) if G == 1 else ShardedSim(abi, sc, LocalFabric(G), device="cuda:0")
    workloads.apply_crashes(s, crashes)
    s.step(WARM); torch.cuda.synchronize()
    if os.environ.get("KERNELS") == "1":
        s.kernelTimingEnable(True)
    t0 = time.time(); s.step(TICKS); torch.cuda.synchronize(); dt = time.time() - t0
    us = dt / TICKS * 1e6
    if G == 1:
        base = us
    out = {"shards_on_one_gpu": G, "exchange": form, "members": N, "us_per_tick": round(us, 1), "x_unsharded": round(us / base, 2) if base else None,
           "Gmt_per_s": round(N * TICKS / dt / 1e9, 3), "digest": "%016x" % s.digest()}
    if os.environ.get("KERNELS") == "1":
        kt = s.kernelTiming()
        out["shard0_probe_us"] = round(kt["probe_ms"] * 1e3 / max(1, kt["ticks"]), 1); out["shard0_merge_us"] = round(kt["merge_ms"] * 1e3 / max(1, kt["ticks"]), 1)
    print(json.dumps(out), flush=True)
    s.close()
