"""Worker for tests/test_shard_dist.py: one process = one shard (torch.distributed, gloo on CPU).
Every rank drives its shard through the host emulation of the product kernels; rank 0 also runs the
oracle and compares every observable.  usage: dist_worker.py <n_members> <p> <loss_ppm> <seed> <ticks> [mode]
(mode bit 0: settling on, suspicion 5 ticks, retransmit x1, and the member that went down comes back twice; bit 1: join pull;
bit 2: periodic state pull every 5 periods; bit 3: strict_reference_rules; bit 4: push_pull (with bit 2); bits 8..: view_cap)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n, p, loss, seed, ticks = (int(x) for x in sys.argv[1:6])
    mode = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    gc, pull, ppull = bool(mode & 1), bool(mode & 2), bool(mode & 4)       # 1: settling, 2: join-time pull (round 0), 4: periodic pull (round 0 in every tick)
    strict = bool(mode & 8)                         # 8: the literal suspectOrDeadNode' (every queue travels as a list)
    push = bool(mode & 16)                          # 16: the periodic pull is a push-pull
    cap = (mode >> 8)                               # bits 8..: bounded member maps with this view_cap (no other option)
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    from swim_amd import Config, Sim, SimConfig, _abi
    from swim_amd.shard import DistFabric, ShardedSim
    from tests import hostemu_binding, oracle_binding
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F,
                   suspicionTicks=6, maxSubjects=min(n, 1024))
    if gc:
        sc.suspicionTicks, sc.retransmitMult, sc.gcTicks = 5, 1, _abi.GC_AUTO
    if pull:
        sc.joinPull = 1
    if ppull:
        sc.pullTicks = 5
    if cap:
        sc.viewCap, sc.maxSubjects = cap, 0
    if strict:
        sc.strictReferenceRules = True
    if push:
        sc.pushPull = True
    if os.environ.get("SWIM_DIST_DEVICE", "cpu") == "cuda":
        # all ranks share GPU 0 (RCCL refuses two ranks on one device): the real HIP library, device
        # buffers wrapped zero-copy, records staged through host memory over gloo
        import torch
        from swim_amd import _lib
        torch.cuda.set_device(0)
        sh = ShardedSim(_lib.load(), sc, DistFabric("cuda:0", transport="host"), device="cuda:0")
    else:
        sh = ShardedSim(hostemu_binding.load(), sc, DistFabric("cpu"))
    ref = Sim.create(oracle_binding.load(), sc) if rank == 0 else None
    for s in (sh, ref):
        if s is None:
            continue
        s.crash(n // 2, 5)
        s.crash(3, 7)
        s.scheduleFault(30, n // 2, True)
        if pull:
            for m in range(10, 22):
                s.crash(m, 12)
                s.scheduleFault(26, m, True)                 # twelve joins in one tick, hosts on either shard
        if gc:
            s.crash(n - 2, 33)
            s.scheduleFault(70, n - 2, True)
            s.crash(n // 2, 75)
    done = 0
    while done < ticks:
        k = min(5, ticks - done)
        sh.step(k)
        done += k
        got = (sh.counters(), sh.digest(), sh.drainEventsRaw(), sh.members(0), sh.members(n - 1), sh.readMember(n // 2))
        if rank == 0:
            ref.step(k)
            want = (ref.counters(), ref.digest(), ref.drainEventsRaw(), ref.members(0), ref.members(n - 1), ref.readMember(n // 2))
            assert got == want, "sharded run (world %d) differs from the oracle after %d ticks" % (dist.get_world_size(), done)
    fd = sh.firstDetection()
    if rank == 0:
        assert fd == ref.firstDetection()
        if gc:
            assert got[0]["settled"] >= 3, got[0]
        print("DIST-OK world=%d digest=%016x" % (dist.get_world_size(), got[1]))
    sh.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
