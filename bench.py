#!/usr/bin/env python
"""bench.py -- member-ticks/sec of the SWIM tick on MI355X (BASELINE.json metric).

A "step" is one protocol period (tick) of the hot path over the whole simulated population.
Workload at N=1 = BASELINE config "1 048 576 members, k=3, on 1 MI355X" in the
dissemination-saturated regime: ~1 member crashes per tick, so
every Ping/Ack carries a full 8-rumour piggyback payload and each member accepts ~2 view
changes per tick.  State is resident in HBM before the timed region; faults are pre-scheduled.

The regime does not depend on --warmup: before the W warm-up steps an untimed PRE-ROLL steps the
cluster until the kernels' own counters show the saturated load (d >= 5.9 payloads delivered per
member-tick over the last 10 ticks, and the suspicion timeout has passed so that Dead declarations
circulate as well: r ~ 2), and the line reports the regime actually measured
(`per_member_tick`).  `--regime quiescent` is the other regime of config 3 (one crash, empty payloads).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     : the DOMINANT kernel (the larger share of the tick), per launch: algorithmic bytes / HIP-event launch time vs
                 8 TB/s (`frac`), its PMC traffic and traffic_ratio; `whole_tick` and every kernel's own line (incl. the bytes
                 this layout moves, `impl_bytes_per_member_tick`) are separate objects
  cpu_baseline : the CPU oracle (a "port": the Haskell reference cannot be built here, no GHC) stepping the
                 SAME cluster on this box's host cores (member-range threads, all cores) over a bounded
                 window, plus its single-thread rate; the same replay checks the GPU's state digest and
                 counters against the oracle at the end of the timed region (`verified_vs_oracle`).
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
PREROLL_MAX = 400              # ticks; the saturated load is reached after ~60
PREROLL_CHUNK = 10
SATURATED_D = 5.9              # payloads delivered per member-tick (2P = 6 when every message carries one)


def rates(c0, c1, n_members, ticks):
    """d payloads delivered, r view entries changed, c piggyback lines rewritten, f failed direct probes:
    the semantic events per member-tick of SURVEY.md 8(d), from the kernels' counters."""
    mt = float(n_members) * ticks
    return {"d": (c1["payloads"] - c0["payloads"]) / mt, "r": (c1["changes"] - c0["changes"]) / mt,
            "c": (c1["pb_writes"] - c0["pb_writes"]) / mt, "f": (c1["direct_failed"] - c0["direct_failed"]) / mt}


def algorithmic_bytes(rt, P, K):
    """A = 16 + P + 64 d + 16 r + 128 c + f k  bytes per member-tick, split by the kernel
    that moves them."""
    return {"probe_kernel": P + rt["f"] * K + 64.0 * rt["d"],         # liveness gathers + delivered payloads
            "merge_kernel": 16.0 + 16.0 * rt["r"] + 128.0 * rt["c"]}  # hot record, accepted rumours, own line r+w


def implementation_bytes(rt, P, K):
    """A restated with the constants of THIS layout (last sentence) -- what the kernels
    move per member-tick when payloads travel as 8-byte masks instead of 64-byte lines.
    probe_kernel: own minfo word 4 + own queue mask 8 + per probe a 1-byte `mb` gather and, for a reached target, ONE 16-byte
    `pk` gather {queue mask, known-ring} (serves the Ack's payload and the push filter) + per pushed Ping payload an 8-byte
    atomicOr (read + write = 16; d/2 is an upper bound: pushes the target's ring already covers are skipped) + proxy bytes
    f k + its ackmask store 8 + probe_out 2.
    merge_kernel: the coalesced per-member streams it reads and rewrites whatever happens (probe_out 2, inmask 8 r + 8 clear,
    ackmask 8, hot 8, pk 16 r + 16 w, deadline cell 16 r + 16 w, minfo 4 r + 4 w + mb 1) + per accepted change a view cell
    8 r + 8 w + per rewritten queue the own line 64 r + 64 w."""
    return {"probe_kernel": 4.0 + 8.0 + P * (1.0 + 16.0) + 16.0 * rt["d"] / 2.0 + rt["f"] * K + 8.0 + 2.0,
            "merge_kernel": 2.0 + 16.0 + 8.0 + 8.0 + 32.0 + 32.0 + 9.0 + 16.0 * rt["r"] + 128.0 * rt["c"]}


def first_detection_latency(sim, crashes, lo_tick, hi_tick):
    fd = sim.firstDetection()
    lat = [fd[m] - t + 1 for (t, m) in crashes if lo_tick <= t < hi_tick and fd[m] is not None]
    return (sum(lat) / len(lat), len(lat)) if lat else (None, 0)


def oracle_replay(sc, crashes, t_window, t_end, gpu_digest, gpu_counters, budget_s=150.0, single_thread=True):
    """The CPU oracle (oracle/swim_oracle.c; checker and reported baseline, never the product) steps the
    same cluster with the same schedule: all host cores up to t_end, timed over [t_window, t_end); digest
    and counters must equal the GPU's; then a few more ticks on ONE thread for the single-core rate."""
    from swim_amd import Sim, workloads
    from tests import oracle_binding          # checker only: never the thing shipped
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # member-range threads stop paying off at ~32 on the 256-core GPU box (random access to the view columns;
    # measured 0.7 / 5.0 / 11.0 / 9.0 / 5.3 M member-ticks/s at 1 / 8 / 32 / 64 / 256 threads,
    # profiles/r02d_oracle_thread_scaling.txt): use the best count, report what the box has
    cores = min(avail, 32)
    n = sc.nMembers
    # rough cost model (1.0 M member-ticks/s per core, ~55 % parallel efficiency): skip what cannot finish
    est = n * t_end / ((0.1e6 if sc.viewCap else 1.0e6) * max(1.0, 0.55 * cores))     # (the set-based end of tick of bounded maps sorts: ~10x slower)
    if est > budget_s:
        return {"skipped": "oracle replay of %d ticks x %d members needs ~%.0f s on %d cores" % (t_end, n, est, cores)}, None
    s = Sim.create(oracle_binding.load(), sc)
    workloads.apply_crashes(s, crashes)
    if sc.viewCap:
        for (t, m) in crashes:
            s.scheduleFault(t + 8 + (m % 5), m, True)
    oracle_binding.set_threads(s, cores)
    s.step(t_window)
    t0 = time.perf_counter()
    s.step(t_end - t_window)
    dt_all = time.perf_counter() - t0
    ok = (s.digest() == gpu_digest) and (s.counters() == gpu_counters)
    k1, dt_one = 0, 1.0
    if single_thread:                          # (the second window of the default run skips it: the single-core rate is in the headline's object)
        oracle_binding.set_threads(s, 1)
        k1 = max(1, min(8, int(6.0e6 / n)))
        t0 = time.perf_counter()
        s.step(k1)
        dt_one = time.perf_counter() - t0
    s.close()
    base = {"value": n * (t_end - t_window) / dt_all, "unit": "member-ticks/s", "cores": cores, "kind": "port",
            "single_thread_value": (n * k1 / dt_one) if k1 else None,
            "sample": "oracle/swim_oracle.c on the SAME %d-member cluster and fault schedule: ticks %d-%d with %d "
                      "member-range threads (the count that scales best; the box has %d cores), then %d ticks on one "
                      "thread; reference Haskell not timed: no GHC in image" % (n, t_window, t_end, cores, avail, k1),
            "host_cores_available": avail}
    return base, ok


def main(argv=None, abi=None):
    """abi: tests hand in the host emulation of the kernels (tests/hostemu_binding) to drive this file's flows on a box without a
    GPU; the bench itself always loads libswimsim.so (no fallback: without a HIP device swimsim_create fails)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=150)
    ap.add_argument("--members", type=int, default=N_MEMBERS, help="members per GPU")
    ap.add_argument("--regime", default="saturated", choices=["saturated", "quiescent"])
    ap.add_argument("--scheme", default="random", choices=["random", "robust"],
                    help="target scheme of the direct probes: random = the reference's kRandomMembers (the headline); "
                         "robust = round-robin rotation (src/Core.hs:232 FIXME), reported separately")
    ap.add_argument("--loss-ppm", type=int, default=0, help="per-message loss (BASELINE config 5 uses 300000)")
    ap.add_argument("--num-to-gossip", type=int, default=3, help="P = k (the reference's default config has 10)")
    ap.add_argument("--gc", action="store_true", help="settling on (gc_ticks = auto): view rows are reclaimed")
    ap.add_argument("--crashes-per-tick", type=float, default=1.0,
                    help="saturated regime: crashes per tick cluster-wide (default 1; BASELINE.md section 3 row 3(s) as written -- 1 %% of "
                         "1 048 576 members over 1 100 ticks -- is 9.5, with --gc --max-subjects 8192 --steps 1000 --warmup 100)")
    ap.add_argument("--max-subjects", type=int, default=0, help="view rows (0 = sized by the workload)")
    ap.add_argument("--view-cap", type=int, default=0,
                    help="bounded member maps (view_cap = C; include/swimsim.h): BASELINE config 5's regime at its per-GPU size, e.g. "
                         "--members 2097152 --loss-ppm 300000 --view-cap 64 [--churn 10]; no crash schedule of its own, no pre-roll")
    ap.add_argument("--churn", type=int, default=0, help="with --view-cap: per mille of the members crash and rejoin per 100 ticks")
    ap.add_argument("--launch", default="auto", choices=["auto", "single", "dist"],
                    help="--gpus N > 1: 'single' = ONE process, N handles (swimsim_cluster_step: what the driver's plain `python bench.py --gpus N` "
                         "gets); 'dist' = one process per GPU under torch.distributed.run (re-executed that way if started without it); "
                         "'auto' = dist when WORLD_SIZE is set, single otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle replay (baseline + verification)")
    ap.add_argument("--no-as-written", action="store_true",
                    help="skip the second window of the default run: BASELINE.md row 3(s) as written (9.5 crashes per tick, settling, 8 192 rows)")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SWIM_BENCH_SHARE_GPU=1 (test hook, not a reporting mode): all shards share GPU 0, to exercise this file's multi-GPU flows on a
    # one-GPU box (one process per rank: gloo with host-staged records; one process: all handles on device 0)
    share_gpu = os.environ.get("SWIM_BENCH_SHARE_GPU") == "1"
    # `python bench.py --gpus N` as the driver starts it -- ONE process: N handles on N devices, the tick loop and the exchange
    # inside the library (swimsim_cluster_step: what a single Haskell host with eight GPUs calls).  Under torch.distributed.run
    # (WORLD_SIZE set): one process per GPU, swimsim_shard_step with torch.distributed as the embedder's exchange.
    if args.gpus > 1 and args.launch == "dist" and "WORLD_SIZE" not in os.environ and abi is None:
        # the other launch shape, without editing code or wrapping the command: the same arguments under torch.distributed.run
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        av = sys.argv[1:] if argv is None else list(argv)
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + av)
    if args.gpus > 1 and args.launch == "single" and "WORLD_SIZE" in os.environ:
        raise SystemExit("bench: --launch single under torch.distributed.run (start it as `python bench.py --gpus N`)")
    single = args.gpus > 1 and "WORLD_SIZE" not in os.environ
    if args.gpus > 1 and not single and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (or without it: one process steps all %d GPUs)" % (args.gpus, args.gpus))
    devices = None
    if single:
        ndev = torch.cuda.device_count()
        if not share_gpu and ndev < args.gpus and abi is None:
            raise SystemExit("bench: --gpus %d but %d device(s) visible (SWIM_BENCH_SHARE_GPU=1 puts every shard on device 0: a test hook)" % (args.gpus, ndev))
        devices = [0 if share_gpu else k for k in range(args.gpus)]
        world = args.gpus                     # shards of the cluster; this process is all of its ranks
    if share_gpu:
        local_rank = 0
    has_cuda = torch.cuda.is_available()
    if has_cuda:
        torch.cuda.set_device(local_rank)
    if single:
        world_procs = 1
    else:
        world_procs = world
    if world_procs > 1:
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("cpu:gloo,cuda:nccl", device_id=torch.device("cuda", local_rank))

    from swim_amd import Sim, _abi, _lib, workloads
    LIB = abi if abi is not None else _lib.load()
    n = args.members                          # per GPU: weak scaling, ONE cluster of world * n members
    nt = n * world
    saturated = args.regime == "saturated" and not args.view_cap
    horizon = (PREROLL_MAX if saturated else 0) + args.warmup + args.steps + 16
    # the same global failure rate at every size (~1 crash per tick from tick 0 on): per-member rumour load,
    # and so the per-GPU work, stays what it is on one GPU
    if saturated:
        sc, crashes, _ = workloads.saturated(nt, horizon, seed=1, t0=0, loss_ppm=args.loss_ppm, num_to_gossip=args.num_to_gossip,
                                              crashes_per_tick=args.crashes_per_tick)
    elif args.view_cap:
        from swim_amd import Config, SimConfig
        sc = SimConfig(cfg=Config(numToGossip=args.num_to_gossip), nMembers=nt, seed=1, lossPpm=args.loss_ppm, eventMask=0x10, viewCap=args.view_cap)
        crashes = workloads.hashed_crashes(nt, 9, max(1, args.churn * horizon // 100), 1000, 2, horizon) if args.churn else []
    else:
        sc, crashes, _ = workloads.quiescent(nt, horizon, seed=1)
        sc.cfg.numToGossip = args.num_to_gossip
        sc.lossPpm = args.loss_ppm
    sc.device = local_rank
    sc.targetScheme = 1 if args.scheme == "robust" else 0
    if args.gc:
        sc.gcTicks = _abi.GC_AUTO
    if args.max_subjects:
        sc.maxSubjects = args.max_subjects
    def run_one(sc, crashes, cpt, steps, warmup):
        """One cluster, one timed window: pre-roll (saturated regimes), warm-up, `steps` timed ticks, the oracle replay.  Returns rank
        0's line as a dict (None on the other ranks)."""
        exchange = "none (one shard)"
        if world == 1:
            sim = Sim.create(LIB, sc)
        elif single:
            from swim_amd.shard import LocalFabric, ShardedSim
            sim = ShardedSim(LIB, sc, LocalFabric(world), devices=devices if has_cuda else None)
            if not share_gpu and abi is None:
                # the in-library exchange reads the peers' buffers over xGMI: it needs peer access between the devices.  Where the node
                # cannot (or the first tick fails on the device), the same job is started again as one process per GPU under
                # torch.distributed.run -- the other launch shape of this file -- instead of failing the line.
                # The trial tick runs on a thread with a deadline: this path (peer reads over xGMI, events across devices) has never run
                # on two devices (one GPU per box here), and a tick that never returns must not hang the line either.
                from swim_amd.sim import SwimError
                import threading
                trial = {}

                def _trial():
                    try:
                        sim.step(1)
                        for d in sorted(set(devices)):
                            torch.cuda.synchronize(d)
                    except SwimError as e_:
                        trial["err"] = e_
                th = threading.Thread(target=_trial, daemon=True)
                th.start(); th.join(float(os.environ.get("SWIM_BENCH_TRIAL_S", "180")))
                if th.is_alive():
                    trial["err"] = "no answer from the first tick within the deadline"
                if "err" in trial:
                    e = trial["err"]
                    import socket
                    sys.stderr.write("bench: one-process cluster step failed (%s): restarting as one process per GPU\n" % e)
                    with socket.socket() as so:
                        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
                    av = sys.argv[1:] if argv is None else list(argv)
                    os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + av)
                sim.close()
                sim = ShardedSim(LIB, sc, LocalFabric(world), devices=devices)
            exchange = "inside the library (swimsim_cluster_step): the peers' buffers read in place over %s, ordered by events on the handles' streams" % (
                "device memory (all shards on ONE GPU: test hook)" if share_gpu else "xGMI peer access")
        else:
            from swim_amd.shard import DistFabric, ShardedSim
            fabric = DistFabric("cuda:%d" % local_rank, transport="host" if share_gpu else "auto")
            sim = ShardedSim(LIB, sc, fabric, device="cuda:%d" % local_rank)
            exchange = "swimsim_shard_step with torch.distributed as the exchange, transport=%s%s" % (fabric.transport, (" [" + fabric.note + "]") if fabric.note else "")
        workloads.apply_crashes(sim, crashes)
        if args.view_cap:
            for (t, m) in crashes:
                sim.scheduleFault(t + 8 + (m % 5), m, True)       # churn: back after 8-12 ticks

        def barrier():
            for d in ((sorted(set(devices)) if single else [local_rank]) if has_cuda else []):
                torch.cuda.synchronize(d)
            if world_procs > 1:
                dist.barrier()
                if has_cuda:
                    torch.cuda.synchronize()

        # ---- pre-roll (untimed): step until the cluster carries the saturated load, whatever --warmup is
        preroll = 0
        if saturated:
            cprev = sim.counters()
            while True:
                sim.step(PREROLL_CHUNK)
                preroll += PREROLL_CHUNK
                cnow = sim.counters()
                d = rates(cprev, cnow, nt, PREROLL_CHUNK)["d"]       # counters() is collective on a sharded cluster
                cprev = cnow
                # every Ping and Ack that arrives carries a payload: d -> P (1-l) + P (1-l)^2 (+ proxied hops)
                lv = 1.0 - args.loss_ppm / 1e6
                # ... and the first Dead declarations (suspicion timeout) are circulating: r -> ~2
                if d >= SATURATED_D / 6.0 * args.num_to_gossip * (lv + lv * lv) and preroll >= sim.resolved.suspicion_ticks + 20:
                    break
                if preroll >= PREROLL_MAX:
                    raise SystemExit("bench: the cluster did not reach the saturated regime in %d ticks (d = %.2f)" % (preroll, d))
        sim.step(warmup)
        if world > 1:
            sim.phaseBreakdown(reset=True)
        c0 = sim.counters()
        sim.kernelTimingEnable(True)
        barrier()
        t0 = time.perf_counter()
        sim.step(steps)                      # blocking: returns after the stream has drained
        barrier()
        dt = time.perf_counter() - t0
        kt = sim.kernelTiming()
        c1 = sim.counters()
        if world_procs > 1:
            tt = torch.tensor([dt], device="cpu" if share_gpu else "cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        t_window, t_end = preroll + warmup, preroll + warmup + steps
        lat_all = first_detection_latency(sim, crashes, t_window, t_end - 2)   # collective on a sharded cluster
        gpu_digest = sim.digest() if (world == 1 or single) else None
        if rank == 0:
            P = sim.resolved.probes_per_tick
            K = sim.resolved.indirect_k
            rt = rates(c0, c1, nt, steps)
            a_by = algorithmic_bytes(rt, P, K)
            nk = max(1, kt["ticks"])
            secs = {"probe_kernel": kt["probe_ms"] / 1e3 / nk, "merge_kernel": kt["merge_ms"] / 1e3 / nk}
            a_impl = implementation_bytes(rt, P, K)
            # HBM traffic per launch: PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs) of THIS workload and
            # regime, recorded by scripts/pmc_passes.sh -- a rocprof run cannot nest in here
            tj, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and saturated and not args.loss_ppm and args.num_to_gossip == 3 and args.scheme == "random" and cpt == 1.0:
                tj = json.load(open(tpath))
                # keyed by the sha of the kernel sources the counters were collected on: never quoted for other kernels
                if tj.get("regime") == args.regime and tj.get("members") == n and tj.get("kernels_sha") == _lib.kernel_sources_sha():
                    traffic_src = tj.get("source")
                else:
                    traffic_src = "profiles/traffic.json was measured on other kernel sources (sha %s, these: %s): not quoted" % (
                        tj.get("kernels_sha"), _lib.kernel_sources_sha())
                    tj = None

            def kernel_line(k):
                """One kernel, per launch: algorithmic bytes (SURVEY 8(d)'s A split by the kernel that moves them) x the members a
                launch processes / the HIP-event launch time; the same with the bytes the LAYOUT moves (A_impl);
                the PMC traffic per launch and its ratio to the algorithmic bytes (> 1: re-reads / bookkeeping the model does not price)."""
                t = secs[k]
                alg = a_by[k] * n
                tr = tj.get(k + "_hbm_bytes_per_launch") if tj else None
                return {"algorithmic_bytes_per_member_tick": a_by[k], "algorithmic_bytes_per_launch": alg,
                        "impl_bytes_per_member_tick": a_impl[k], "avg_launch_us": t * 1e6,
                        "achieved_GBs": (alg / t / 1e9) if t > 0 else 0.0,
                        "frac": (alg / t / 1e9 / HBM_PEAK_GBS) if t > 0 else 0.0,
                        "frac_impl": (a_impl[k] * n / t / 1e9 / HBM_PEAK_GBS) if t > 0 else 0.0,
                        "traffic": tr, "traffic_ratio": (tr / alg) if (tr and alg) else None,
                        "frac_traffic": (tr / t / 1e9 / HBM_PEAK_GBS) if (tr and t > 0) else None}
            per_kernel = {k: kernel_line(k) for k in secs}
            dom = max(secs, key=lambda k: secs[k])               # the dominant kernel: the larger share of the tick
            if args.view_cap:                                    # bounded member maps: the same two phases are swim_sparse.h's kernels
                per_kernel = {"sp_" + k: v for k, v in per_kernel.items()}
                for v in per_kernel.values():
                    v["impl_bytes_per_member_tick"] = None       # (A_impl above is the dense layout's; the map streams 2 x 12 C bytes per member-tick)
                    v["frac_impl"] = None
                secs = {"sp_" + k: v for k, v in secs.items()}
                a_by = {"sp_" + k: v for k, v in a_by.items()}
                dom = "sp_" + dom
            a_tot, t_tot = sum(a_by.values()), sum(secs.values())
            whole = a_tot * n / t_tot / 1e9 if t_tot > 0 else 0.0
            tr_tot = sum(per_kernel[k]["traffic"] for k in secs) if all(per_kernel[k]["traffic"] for k in secs) else None
            lat, nlat = lat_all
            out = {
                "metric": "member-ticks/sec at N=1M simulated members; mean first-detection latency (ticks)",
                "value": n * world * steps / dt,
                "unit": "member-ticks/s",
                "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": dt / steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u32", "data": "synthetic",
                "config": {"workload": "%s: %d members/GPU, k=%d, %s, loss %d ppm, suspicion %d ticks, retransmit %dx log2 N%s" % (
                               ("config5(bounded member maps, view_cap %d, churn %d per mille / 100 ticks)" % (args.view_cap, args.churn)) if args.view_cap else "config3(%s)" % args.regime,
                               n, args.num_to_gossip,
                               ("~%g crashes per tick" % cpt) + " from tick 0 (hashed schedule); untimed pre-roll of %d ticks until d >= %.1f and "
                               "the suspicion timeout has passed, then the warm-up" % (preroll, SATURATED_D) if saturated else ("message loss is the load: no pre-roll" if args.view_cap else "one crash at tick 2"),
                               args.loss_ppm, sim.resolved.suspicion_ticks, sim.resolved.retransmit_mult,
                               ", settling every %d quiet ticks" % sim.resolved.gc_ticks if sim.resolved.gc_ticks else ""),
                           "members_per_gpu": n, "num_to_gossip": sim.resolved.num_to_gossip, "target_scheme": args.scheme,
                           "preroll_ticks": preroll, "timed_ticks": [t_window, t_end],
                           "parallelism": "1 GPU" if world == 1 else "ONE cluster of %d members in %d shards (contiguous id ranges, one per GPU%s); piggyback payloads cross shards in two rounds per tick (%s)" % (
                               nt, world, ", ONE process" if single else ", one process per GPU", exchange)},
                "ticks_per_s": steps / dt,
                "mean_first_detection_latency_ticks": lat, "crashes_measured": nlat,
                "per_member_tick": rt,
                # every field of `roofline` describes the DOMINANT kernel, per launch (DESIGN.md section 5 has the formulas; the
                # rocprofv3 kernel trace of the same command is profiles/*_rocprof_timed_window.txt, the PMC passes *_pmc_summary.txt);
                # the whole tick is a separate object
                "roofline": {"bound": "hbm", "kernel": dom, "scope": "dominant kernel, per launch",
                             "achieved": per_kernel[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": per_kernel[dom]["frac"],
                             "traffic": per_kernel[dom]["traffic"], "traffic_ratio": per_kernel[dom]["traffic_ratio"],
                             "traffic_source": traffic_src,
                             "algorithmic_bytes_per_launch": per_kernel[dom]["algorithmic_bytes_per_launch"],
                             "avg_launch_us": per_kernel[dom]["avg_launch_us"],
                             "formula": "achieved = A_kernel x members / avg launch time (HIP events on the library's stream); "
                                        "A_probe = P + f k + 64 d, A_merge = 16 + 16 r + 128 c bytes per member-tick (SURVEY 8d) with d, r, c, f "
                                        "from the kernels' event counters over the timed ticks (per_member_tick); traffic = FETCH_SIZE + "
                                        "WRITE_SIZE per launch (PMC, separate passes); traffic_ratio = traffic / algorithmic bytes per launch",
                             "whole_tick": {"scope": "probe_kernel + merge_kernel", "algorithmic_bytes_per_member_tick": a_tot,
                                            "kernel_us": t_tot * 1e6, "achieved": whole, "frac": whole / HBM_PEAK_GBS,
                                            "frac_of_wall": (a_tot * n / (dt / steps) / 1e9 / HBM_PEAK_GBS),
                                            "traffic": tr_tot,
                                            "frac_traffic_of_wall": (tr_tot / (dt / steps) / 1e9 / HBM_PEAK_GBS) if tr_tot else None},
                             "kernels": per_kernel},
            }
            if not args.view_cap:
                # what the 8 TB/s yardstick hides ("Round 4: merge_kernel against the chip's RANDOM-access rate"): both
                # tick kernels are made of scattered 8-byte accesses, which the chip serves at fixed RATES (microbenchmarks under profiles/)
                chg = rt["r"] * n
                png = (c1["pings"] - c0["pings"]) / float(steps) / world
                out["roofline"]["scattered_access"] = {
                    "merge_kernel": {"view_cells_changed_per_launch": chg, "chip_rate_load_then_store_back_G_per_s": 19.8,
                                     "us_at_that_rate_if_no_two_cells_shared_a_sector": chg / 19.8e9 * 1e6},
                    "probe_kernel": {"ping_pushes_per_launch_upper_bound": png, "chip_rate_scattered_atomics_G_per_s": 26.7,
                                     "us_at_that_rate": png / 26.7e9 * 1e6},
                    "source": "profiles/r04o_microbench_random_access.txt, profiles/r01_microbench_gather_rate.txt"}
            if world > 1:
                out["shard_tick_breakdown_us_rank0"] = sim.phaseBreakdown()      # where a sharded tick goes (host view)
                # what crosses xGMI per GPU and tick (the last tick's counts, swimsim_shard_traffic): round 1 = the all-gather of replica
                # slices, dictionaries and lists (received from every peer), round 2 = the {dst, src} records; against 7 links x 153 GB/s
                tr = sim.traffic()[0]
                per_gpu = tr["round1_bytes_to_each_peer"] * (world - 1) + tr["round2_bytes_to_all_peers"]
                xgmi = 7 * 153.0
                out["exchange"] = {"bytes_per_gpu_per_tick": per_gpu, "detail_shard0": tr, "members_per_gpu": n,
                                   "xgmi_peak_GBs": xgmi, "achieved_GBs_if_on_xgmi": per_gpu / (dt / steps) / 1e9,
                                   "frac_of_xgmi": per_gpu / (dt / steps) / 1e9 / xgmi,
                                   "note": "all shards on ONE GPU (test hook): nothing crossed xGMI" if share_gpu else None}
            if not args.no_cpu_baseline and (world == 1 or single):
                base, ok = oracle_replay(sc, crashes, t_window, t_end, gpu_digest, c1, single_thread=(cpt == args.crashes_per_tick))
                out["cpu_baseline"] = base
                out["verified_vs_oracle"] = ok       # digest + counters at the end of the timed region; None = replay skipped
                if ok is False:
                    print(json.dumps(out))
                    raise SystemExit("bench: GPU state diverged from the oracle at tick %d" % t_end)
        sim.close()
        return out if rank == 0 else None

    out = run_one(sc, crashes, args.crashes_per_tick, args.steps, args.warmup)
    # BASELINE.md section 3 row 3(s) AS WRITTEN, in the same run: 1 % of the 1 048 576 members crash over the
    # 1 100 ticks = 9.5 crashes per tick, settling on, 8 192 view rows -- a second cluster, its own pre-roll, warm-up, timed window of
    # the same length, per-kernel roofline and oracle replay.  The headline above stays SURVEY 8(d)'s closed-form saturated regime
    # (d = 6, c = 1, r ~ 2: one crash per tick); this one carries 4x the accepted rumours per member-tick and is the slower number.
    if (world == 1 and saturated and args.crashes_per_tick == 1.0 and not args.loss_ppm and args.num_to_gossip == 3 and args.scheme == "random"
            and not args.gc and not args.max_subjects and n == N_MEMBERS and not args.no_as_written):
        sc2, crashes2, _ = workloads.saturated(nt, PREROLL_MAX + args.warmup + args.steps + 16, seed=1, t0=0, num_to_gossip=3, crashes_per_tick=9.5)
        sc2.device = local_rank
        sc2.gcTicks = _abi.GC_AUTO
        sc2.maxSubjects = min(8192, nt)
        o2 = run_one(sc2, crashes2, 9.5, args.steps, max(args.warmup, 100))      # (100 warm-up ticks)
        if rank == 0:
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "ticks_per_s", "config", "per_member_tick", "mean_first_detection_latency_ticks",
                    "crashes_measured", "roofline", "cpu_baseline", "verified_vs_oracle")
            out["config3s_as_written"] = {k: o2[k] for k in keep if k in o2}
    if rank == 0:
        print(json.dumps(out))
    if world_procs > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
