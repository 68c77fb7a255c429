"""Randomised parity sweep: the product's kernels (host emulation) against the oracle over random sizes,
probe counts, loss rates, fault schedules (crashes and rejoins, with and without the join-time pull), both
target schemes, 1-8 shards (also with the robust scheme, settling and the join-time pull), tiny inbox capacities -- every observable, every 10 ticks.  Seeded, so a failure reproduces; a longer run of the
same generator (430 configurations) was clean when this was written."""
import os
import random

import pytest

from swim_amd import Config, Sim, SimConfig, _abi
from swim_amd.shard import LocalFabric, ShardedSim


@pytest.mark.parametrize("block", range(4))
def test_random_configurations(oracle_abi, block):
    from tests import hostemu_binding
    emu = hostemu_binding.load()
    rng = random.Random(1000 + block)
    for _ in range(10):
        n = rng.choice([2, 3, 5, 17, 64, 100, 129, 256, 300, 512, 777])
        p = rng.choice([1, 2, 3, 3, 3, 5, 10])
        loss = rng.choice([0, 0, 0, 10000, 100000, 300000])
        scheme = rng.choice([0, 0, 1])
        shards = 1
        if n >= 64 and rng.random() < 0.4:
            shards = rng.choice([g for g in (2, 3, 4, 8) if n % g == 0] or [1])
        seed = rng.randrange(1, 1 << 30)
        gc = _abi.GC_AUTO if rng.random() < 0.4 else 0
        sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F,
                       suspicionTicks=rng.choice([3, 6, 12]), retransmitMult=rng.choice([1, 1, 3]), maxSubjects=min(n, 1024),
                       targetScheme=scheme, inboxCap=rng.choice([0, 0, 1, 2]), gcTicks=gc,
                       joinPull=seed % 2, pullTicks=(0, 0, 2, 5, 17)[(seed >> 3) % 5])
        sc.pushPull = bool(sc.pullTicks) and (seed >> 7) % 2 == 1            # (on shards too since round 6)
        sc.strictReferenceRules = (seed >> 11) % 4 == 0                     # the literal rule with every option, on shards too (round 6)
        a = Sim.create(oracle_abi, sc)
        rm = shards > 1 and rng.random() < 0.5       # replicated queue masks instead of probe records (read at create)
        os.environ["SWIMSIM_CLUSTER_STEP"] = "0" if rm else "1"
        try:
            b = Sim.create(emu, sc) if shards == 1 else ShardedSim(emu, sc, LocalFabric(shards))
        finally:
            del os.environ["SWIMSIM_CLUSTER_STEP"]
        for _f in range(rng.randrange(0, max(1, n // 8) + 1)):
            m, t = rng.randrange(n), rng.randrange(1, 40)
            for s in (a, b):
                s.scheduleFault(t, m, False)
            if rng.random() < 0.5:
                t2 = t + rng.randrange(1, 30)
                for s in (a, b):
                    s.scheduleFault(t2, m, True)
        what = (n, p, loss, scheme, shards, gc, rm, seed)
        for _t in range(rng.choice([3, 6, 10])):
            a.step(10); b.step(10)
            assert a.counters() == b.counters(), ("counters", what)
            assert a.digest() == b.digest(), ("digest", what)
            assert a.drainEventsRaw() == b.drainEventsRaw(), ("events", what)
        assert a.firstDetection() == b.firstDetection(), ("first detection", what)
        a.close(); b.close()


def _random_bounded_case(rng, sizes, max_faults):
    n = rng.choice(sizes)
    p = rng.choice([1, 2, 3, 3, 3, 5, 7, 10])
    loss = rng.choice([0, 10000, 100000, 300000, 300000, 500000])
    cap = rng.choice([4, 5, 8, 16, 33, 64, 100, 128, 129, 200, 256])
    shards = 1
    if n >= 64 and rng.random() < 0.5:
        shards = rng.choice([g for g in (2, 3, 4, 8) if n % g == 0] or [1])
    seed = rng.randrange(1, 1 << 30)
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F, suspicionTicks=rng.choice([2, 3, 6, 12]),
                   retransmitMult=rng.choice([1, 1, 3]), inboxCap=rng.choice([0, 0, 0, 16]), viewCap=cap)
    faults = []
    for _f in range(rng.randrange(0, min(max_faults, max(1, n // 8)) + 1)):
        m, t = rng.randrange(n), rng.randrange(1, 30)
        faults.append((t, m, False))
        if rng.random() < 0.6:
            faults.append((t + rng.randrange(0, 20), m, True))       # (+0: down and up again in one tick)
    return sc, shards, faults, (n, p, loss, cap, shards, seed)


@pytest.mark.parametrize("block", range(3))
def test_random_bounded_configurations(oracle_abi, block):
    """The same for bounded member maps (view_cap, swim_sparse.h): capacities 4 ... 256 (one, two and four map entries per lane;
    the 256- / 512- / 1 024-slot tables), numToGossip up to 10, loss up to 50 %, crashes and rejoins, inboxes smaller than the
    fan-in, 1-8 shards -- every observable every 5 ticks."""
    from tests import hostemu_binding
    emu = hostemu_binding.load()
    rng = random.Random(7000 + block)
    for _ in range(9):
        sc, shards, faults, what = _random_bounded_case(rng, [8, 17, 64, 100, 128, 256, 300, 480], 12)
        a = Sim.create(oracle_abi, sc)
        b = Sim.create(emu, sc) if shards == 1 else ShardedSim(emu, sc, LocalFabric(shards))
        for (t, m, up) in faults:
            a.scheduleFault(t, m, up); b.scheduleFault(t, m, up)
        for _t in range(rng.choice([3, 5, 8])):
            a.step(5); b.step(5)
            assert a.counters() == b.counters(), ("counters", what)
            assert a.digest() == b.digest(), ("digest", what)
            assert a.drainEventsRaw() == b.drainEventsRaw(), ("events", what)
        assert a.firstDetection() == b.firstDetection(), ("first detection", what)
        a.close(); b.close()
