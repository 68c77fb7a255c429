#!/usr/bin/env python
"""bench.py -- member-ticks/sec of the SWIM tick on MI355X (BASELINE.json metric).

A "step" is one protocol period (tick) of the hot path over the whole simulated population.
Workload at N=1 = BASELINE config "1 048 576 members, k=3, on 1 MI355X" in the
dissemination-saturated regime: ~1 member crashes per tick, so
every Ping/Ack carries a full 8-rumour piggyback payload and each member accepts ~2 view
changes per tick.  State is resident in HBM before the timed region; faults are pre-scheduled.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     : dominant kernel (probe_kernel: it performs the 2P payload deliveries per member) algorithmic
                 bytes / HIP-event duration vs 8 TB/s, plus the per-kernel and whole-tick figures
  cpu_baseline : the CPU oracle (a "port": the Haskell reference cannot be built here, no GHC)
                 timed on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_MEMBERS = 1 << 20


def algorithmic_bytes(c0, c1, n_members, ticks, P, K):
    """A = 16 + P + 64 d + 16 r + 128 c + f k  bytes per member-tick, from the
    semantic event counters kept by the kernels (d payloads delivered, r view entries changed,
    c piggyback lines rewritten, f failed direct probes), split by the kernel that moves them."""
    mt = float(n_members) * ticks
    d = (c1["payloads"] - c0["payloads"]) / mt
    r = (c1["changes"] - c0["changes"]) / mt
    c = (c1["pb_writes"] - c0["pb_writes"]) / mt
    f = (c1["direct_failed"] - c0["direct_failed"]) / mt
    a = {"probe_kernel": P + f * K + 64.0 * d,         # liveness gathers + delivered piggyback payloads
         "merge_kernel": 16.0 + 16.0 * r + 128.0 * c}  # hot record, accepted rumours, own line read+write
    return a, {"d": d, "r": r, "c": c, "f": f}


def first_detection_latency(sim, crashes, lo_tick, hi_tick):
    fd = sim.firstDetection()
    lat = [fd[m] - t + 1 for (t, m) in crashes if lo_tick <= t < hi_tick and fd[m] is not None]
    return (sum(lat) / len(lat), len(lat)) if lat else (None, 0)


def cpu_baseline(budget_ticks=60, warm_ticks=150, n_members=65536):
    """The oracle (oracle/swim_oracle.c, single thread) on a 65 536-member slice of the workload at
    the same per-member rumour load (~1 crash per tick), warm-up excluded."""
    from swim_amd import Sim, workloads
    from tests import oracle_binding          # checker only: never the thing shipped
    total = warm_ticks + budget_ticks
    sc, crashes, _ = workloads.saturated(n_members, total)
    s = Sim.create(oracle_binding.load(), sc)
    workloads.apply_crashes(s, crashes)
    s.step(warm_ticks)
    t0 = time.perf_counter()
    s.step(budget_ticks)
    dt = time.perf_counter() - t0
    s.close()
    return {"value": n_members * budget_ticks / dt, "unit": "member-ticks/s", "cores": 1, "kind": "port",
            "sample": "%d-member slice, same per-member load (~1 crash/tick), ticks %d-%d, oracle/swim_oracle.c "
                      "single thread; reference Haskell not timed: no GHC in image" % (n_members, warm_ticks, total),
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=150)
    ap.add_argument("--members", type=int, default=N_MEMBERS, help="members per GPU")
    ap.add_argument("--regime", default="saturated", choices=["saturated", "quiescent"])
    ap.add_argument("--scheme", default="random", choices=["random", "robust"],
                    help="target scheme of the direct probes: random = the reference's kRandomMembers (the headline); "
                         "robust = round-robin rotation (src/Core.hs:232 FIXME), reported separately")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # SWIM_BENCH_SHARE_GPU=1 (test hook, not a reporting mode): all ranks share GPU 0 over gloo with
    # host-staged records, to exercise this file's multi-rank flow on a one-GPU box
    share_gpu = os.environ.get("SWIM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("cpu:gloo,cuda:nccl", device_id=torch.device("cuda", local_rank))

    from swim_amd import Sim, _lib, workloads
    total = args.warmup + args.steps
    n = args.members                          # per GPU: weak scaling, ONE cluster of world * n members
    nt = n * world
    # the same global failure rate at every size (~1 crash per tick): per-member rumour load, and so the
    # per-GPU work, stays what it is on one GPU
    if args.regime == "saturated":
        sc, crashes, _ = workloads.saturated(nt, total, seed=1)
    else:
        sc, crashes, _ = workloads.quiescent(nt, total, seed=1)
    sc.device = local_rank
    sc.targetScheme = 1 if args.scheme == "robust" else 0
    exchange = "none (one shard)"
    if world == 1:
        sim = Sim.create(_lib.load(), sc)
    else:
        from swim_amd.shard import DistFabric, ShardedSim
        fabric = DistFabric("cuda:%d" % local_rank, transport="host" if share_gpu else "auto")
        sim = ShardedSim(_lib.load(), sc, fabric, device="cuda:%d" % local_rank)
        exchange = "torch.distributed p2p, transport=%s%s" % (fabric.transport, (" [" + fabric.note + "]") if fabric.note else "")
    workloads.apply_crashes(sim, crashes)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sim.step(args.warmup)
    if world > 1:
        sim.phaseBreakdown(reset=True)
    c0 = sim.counters()
    sim.kernelTimingEnable(True)
    barrier()
    t0 = time.perf_counter()
    sim.step(args.steps)                      # blocking: returns after the stream has drained
    barrier()
    dt = time.perf_counter() - t0
    kt = sim.kernelTiming()
    c1 = sim.counters()
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if share_gpu else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    lat_all = first_detection_latency(sim, crashes, args.warmup, total - 2)   # collective on a sharded cluster
    if rank == 0:
        P = sim.resolved.probes_per_tick
        K = sim.resolved.indirect_k
        a_by, rates = algorithmic_bytes(c0, c1, nt, args.steps, P, K)
        nk = max(1, kt["ticks"])
        secs = {"probe_kernel": kt["probe_ms"] / 1e3 / nk, "merge_kernel": kt["merge_ms"] / 1e3 / nk}
        per_kernel = {k: {"algorithmic_bytes_per_member_tick": a_by[k], "avg_launch_us": secs[k] * 1e6,
                          "achieved_GBs": (a_by[k] * n / secs[k] / 1e9) if secs[k] > 0 else 0.0} for k in secs}
        dom = "probe_kernel"
        achieved = per_kernel[dom]["achieved_GBs"]
        a_tot, t_tot = sum(a_by.values()), sum(secs.values())
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("regime") == args.regime and tj.get("members") == n:
                traffic = tj.get(dom + "_hbm_bytes_per_launch")
        lat, nlat = lat_all
        out = {
            "metric": "member-ticks/sec at N=1M simulated members; mean first-detection latency (ticks)",
            "value": n * world * args.steps / dt,
            "unit": "member-ticks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "config3(%s): %d members/GPU, k=3, ~1 crash per tick (hashed schedule), "
                                   "suspicion %d ticks, retransmit %dx log2 N" % (
                                       args.regime, n, sim.resolved.suspicion_ticks, sim.resolved.retransmit_mult),
                       "members_per_gpu": n, "num_to_gossip": sim.resolved.num_to_gossip, "target_scheme": args.scheme,
                       "parallelism": "1 GPU" if world == 1 else "ONE cluster of %d members in %d shards (contiguous id ranges, one per GPU); piggyback payloads cross shards in two rounds per tick (%s)" % (nt, world, exchange)},
            "ticks_per_s": args.steps / dt,
            "mean_first_detection_latency_ticks": lat, "crashes_measured": nlat,
            "per_member_tick": rates,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_member_tick": a_by[dom],
                         "avg_launch_us": secs[dom] * 1e6,
                         "kernels": per_kernel,
                         "whole_tick": {"algorithmic_bytes_per_member_tick": a_tot,
                                        "kernel_us": t_tot * 1e6,
                                        "achieved_GBs": a_tot * n / t_tot / 1e9 if t_tot > 0 else 0.0,
                                        "frac": (a_tot * n / t_tot / 1e9 / HBM_PEAK_GBS) if t_tot > 0 else 0.0}},
        }
        if world > 1:
            out["shard_tick_breakdown_us_rank0"] = sim.phaseBreakdown()      # where a sharded tick goes (host view)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
