"""Bounded member maps (view_cap = C; include/swimsim.h "Bounded member maps"): every member keeps at most C
entries that differ from the default, the oldest evicted -- `Map String Member` (src/Types.hs:55) with a capacity, the
structure that lets BASELINE config 5 (30 % message loss x churn) run at millions of members per GPU.

CPU side of the parity ladder (no GPU in this container): the oracle's set-based end of tick against its own sequential one
where the capacity does not bind, the semantics of eviction, and the product kernels (swim_sparse.h: one wave per member, the
map as a hash table in LDS) on the host emulation against the oracle.  tests/test_hip_parity.py has the `-m gpu` twins at
65 536 and 262 144 members."""
import dataclasses
import random

import pytest

from swim_amd import Config, Sim, SimConfig, SwimError, _abi
from tests.helpers import compare_state


@pytest.fixture(scope="module")
def emu_abi():
    from tests import hostemu_binding
    return hostemu_binding.load()


def _faults(rng, n, horizon, crashes, rejoin=0.6):
    out = []
    for _ in range(crashes):
        m, t = rng.randrange(n), rng.randrange(1, horizon)
        out.append((t, m, False))
        if rng.random() < rejoin:
            out.append((t + rng.randrange(1, horizon), m, True))
    return out


@pytest.mark.parametrize("trial", range(5))
def test_a_capacity_that_does_not_bind_is_the_unbounded_tick(oracle_abi, trial):
    """The bounded tick is stated over sets (largest proposal per subject, then the capacity), the unbounded one applies one
    proposal after the other: where nothing is evicted the two must agree in every observable -- events with their causes,
    counters incl. the running event digest, views, queues, first-detection ticks.  Pins the restatement to the rules of
    src/Core.hs:142-218 as the sequential oracle has them."""
    rng = random.Random(100 + trial)
    n = rng.choice([40, 150, 256])
    sc = SimConfig(cfg=Config(numToGossip=rng.choice([2, 3, 4])), nMembers=n, seed=trial + 1, lossPpm=rng.choice([0, 20000, 100000]),
                   eventMask=0x1F, suspicionTicks=rng.choice([4, 7]), maxSubjects=n, retransmitMult=rng.choice([1, 2, 3]))
    a, b = Sim.create(oracle_abi, sc), Sim.create(oracle_abi, dataclasses.replace(sc, viewCap=256))
    for (t, m, u) in _faults(rng, n, 40, 6):
        a.scheduleFault(t, m, u); b.scheduleFault(t, m, u)
    for blk in range(12):
        a.step(5); b.step(5)
        compare_state(a, b, (0, 1, n - 1), (0, 1, n - 1), True, where="block %d:" % blk)
    assert a.firstDetection() == b.firstDetection() and b.counters()["evicted"] == 0
    a.close(); b.close()


def test_eviction_keeps_the_most_recent_entries_and_counts_what_leaves(oracle_abi):
    """Under 30 % loss every member hears of far more subjects than 8: the map never holds more than C entries, what it holds
    are the most recent changes (nothing older than what was evicted stays... by lastChange), entries that leave are counted,
    and an evicted subject is Alive again in that view -- it is probed again."""
    n, C = 600, 8
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=5, lossPpm=300000, eventMask=0x1F, viewCap=C)
    s = Sim.create(oracle_abi, sc)
    s.step(30)
    c = s.counters()
    assert c["evicted"] > 0
    full = 0
    for o in range(0, n, 7):
        v = s.members(o)
        assert len(v) <= C
        full += len(v) == C
        assert all(m.memberLastChange >= s.tick - 3 for m in v), v      # ~75 new subjects per tick through 8 entries: nothing grows old
    assert full > n // 7 // 2
    # the digest and the counters do not depend on how the members are split over threads
    from tests import oracle_binding
    s2 = Sim.create(oracle_abi, sc)
    oracle_binding.set_threads(s2, 3)
    s2.step(30)
    assert s2.digest() == s.digest() and s2.counters() == c
    s.close(); s2.close()


def test_what_cannot_be_combined_with_bounded_maps_is_refused(oracle_abi, emu_abi):
    base = SimConfig(cfg=Config(numToGossip=3), nMembers=64, viewCap=16)
    for abi in (oracle_abi, emu_abi):
        for bad in (dict(viewCap=3), dict(viewCap=257), dict(gcTicks=_abi.GC_AUTO), dict(joinPull=1), dict(pullTicks=5), dict(targetScheme=1)):
            with pytest.raises(SwimError):
                Sim.create(abi, dataclasses.replace(base, **bad))
        s = Sim.create(abi, base)
        with pytest.raises(SwimError):
            s.injectRumor(1, 2, 1, 0)
        with pytest.raises(SwimError):
            s.setView(1, 2, 1, 0)
        s.close()
    Sim.create(emu_abi, base, shard_index=0, n_shards=2).close()     # sharded clusters of bounded handles exist


CASES = [
    # members, view_cap, loss ppm, P = K, suspicion ticks, retransmit mult, crashes, ticks
    (64, 8, 0, 3, 5, 0, 3, 40),            # lossless: crashes detected, buried, rejoined; the capacity binds only mildly
    (200, 16, 100000, 3, 6, 0, 6, 30),     # 10 % loss
    (300, 64, 300000, 3, 8, 0, 5, 24),     # BASELINE config 5's loss rate; one map entry per lane
    (150, 4, 300000, 4, 5, 1, 5, 24),      # the smallest capacity: nearly everything is evicted at once
    (300, 200, 300000, 3, 8, 0, 4, 12),    # four map entries per lane, the 1 024-slot table, timers fire (nothing evicted before)
    (120, 130, 200000, 5, 4, 2, 6, 12),    # P = K = 5: the 16-wide probe arrays
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_cap%d_loss%d_k%d" % c[:4])
def test_kernels_on_the_emulation_match_the_oracle(oracle_abi, emu_abi, case):
    """swim_sparse.h (host emulation) vs the oracle, every tick: digest, every counter, the event stream record by record,
    views, queues, coverage, first-detection ticks -- with crashes and rejoins (churn) next to the loss."""
    n, cap, loss, P, S, rm, crashes, ticks = case
    rng = random.Random(n * 31 + cap)
    sc = SimConfig(cfg=Config(numToGossip=P), nMembers=n, seed=n + cap, lossPpm=loss, eventMask=0x1F, suspicionTicks=S, retransmitMult=rm, viewCap=cap)
    a, b = Sim.create(oracle_abi, sc), Sim.create(emu_abi, sc)
    for (t, m, u) in _faults(rng, n, ticks // 2, crashes):
        a.scheduleFault(t, m, u); b.scheduleFault(t, m, u)
    for _ in range(ticks):
        a.step(1); b.step(1)
        compare_state(a, b, (0, 1, n - 1), (0, 1, n - 1), True, where="tick %d:" % a.tick)
    assert a.firstDetection() == b.firstDetection()
    assert b.counters()["changes"] > 0
    a.close(); b.close()


@pytest.mark.parametrize("case", [(400, 16, 300000, 3, 5, 20), (400, 8, 300000, 6, 5, 12), (100, 4, 0, 3, 5, 20)], ids=lambda c: "n%d_cap%d_k%d" % (c[0], c[1], c[3]))
def test_a_tick_that_overflows_the_working_set_takes_the_rank_floor(oracle_abi, case):
    """The per-tick working set of a member (its map + every subject it hears of for the first time) is a 512- / 1 024-slot
    table in LDS.  A test build with 64 slots makes ordinary lossy ticks overflow it: subjects without an entry are then taken
    in only above a rank floor found by bisection (swim_sparse.h) -- exact, checked here against the oracle, which has no
    such table."""
    from tests import hostemu_binding
    emu = hostemu_binding.load_variant("spphys64", ["SWIM_SP_PHYS=64"])
    n, cap, loss, P, S, ticks = case
    sc = SimConfig(cfg=Config(numToGossip=P), nMembers=n, seed=7 + cap, lossPpm=loss, eventMask=0x1F, suspicionTicks=S, viewCap=cap)
    a, b = Sim.create(oracle_abi, sc), Sim.create(emu, sc)
    for s in (a, b):
        s.scheduleFault(2, 7, False); s.scheduleFault(9, 7, True)
    for _ in range(ticks):
        a.step(1); b.step(1)
        compare_state(a, b, (0, 7, n - 1), (0, 7), True, where="tick %d:" % a.tick)
    a.close(); b.close()


def test_inboxes_smaller_than_the_fan_in_go_through_the_overflow_list(oracle_abi, emu_abi):
    """inbox_cap = 16 against ~25 deliveries per member-tick: the exact overflow list (one source at a time) carries the rest."""
    n = 256
    sc = SimConfig(cfg=Config(numToGossip=4), nMembers=n, seed=3, lossPpm=250000, eventMask=0x1F, suspicionTicks=6, viewCap=32, inboxCap=16)
    a, b = Sim.create(oracle_abi, sc), Sim.create(emu_abi, sc)
    for _ in range(10):
        a.step(1); b.step(1)
        compare_state(a, b, (0, n - 1), (0,), True, where="tick %d:" % a.tick)
    a.close(); b.close()


def test_waves_that_step_several_members(oracle_abi, emu_abi, monkeypatch):
    """At millions of members a wave steps hundreds of members one after the other (grids of 16 384 workgroups); the small
    clusters above give every wave one.  Here 5 / 3 workgroups step 300 members: the per-wave tables are cleared and refilled
    between members, counters accumulate across them."""
    monkeypatch.setenv("SWIMSIM_SP_GRID", "5,3")           # measurement knob of the library, read at create
    n = 300
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=41, lossPpm=300000, eventMask=0x1F, suspicionTicks=6, viewCap=24)
    a, b = Sim.create(oracle_abi, sc), Sim.create(emu_abi, sc)
    for s in (a, b):
        s.scheduleFault(2, 9, False); s.scheduleFault(8, 9, True); s.scheduleFault(3, 200, False)
    for _ in range(16):
        a.step(1); b.step(1)
        compare_state(a, b, (0, 9, 200, n - 1), (0, 9), True, where="tick %d:" % a.tick)
    a.close(); b.close()


@pytest.mark.parametrize("n,shards,cap,p,loss,seed", [(128, 2, 16, 3, 0, 1), (256, 4, 8, 3, 300000, 2), (192, 3, 32, 2, 100000, 3), (512, 8, 64, 3, 300000, 4),
                                                      (240, 2, 200, 4, 250000, 5)])
def test_a_sharded_cluster_of_bounded_handles_matches_the_oracle(oracle_abi, emu_abi, monkeypatch, n, shards, cap, p, loss, seed):
    """BASELINE config 5 is 16 M members over 8 GPUs: the population of bounded handles split by contiguous id range (DESIGN.md
    section 7b).  Per tick ONE all-gather (everybody's start-of-tick queue line and byte) and ONE all-to-all-v of 8-byte
    delivery records; every observable of the cluster equals the unsharded oracle's, with crashes and rejoins on either side
    of the shard borders."""
    from swim_amd.shard import LocalFabric, ShardedSim
    from tests.test_shard_hostemu import lockstep
    # all shards in one process: by default the library steps the cluster itself (swimsim_cluster_step: the exchange as copies on
    # the handles' streams, counts in device memory); odd seeds take the phase calls + the fabric's exchange instead -- the path
    # a multi-process cluster takes (tests/test_shard_dist.py runs that one over gloo)
    monkeypatch.setenv("SWIMSIM_CLUSTER_STEP", "0" if seed % 2 else "1")
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F, suspicionTicks=6, viewCap=cap)
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(emu_abi, sc, LocalFabric(shards))
    per = n // shards
    for s in (a, b):
        s.crash(n // 2, 3); s.crash(per - 1, 4); s.crash(per, 4); s.crash(n - 1, 6)
        s.scheduleFault(12, n // 2, True); s.scheduleFault(15, per, True)
    lockstep(a, b, 30, 3, observers=(0, per - 1, per, n - 1), members=(0, per, n // 2, n - 1))
    assert b.counters()["changes"] > 0
    b.close(); a.close()


@pytest.mark.parametrize("n,cap,p,loss,shards", [(150, 16, 3, 300000, 1), (64, 64, 10, 200000, 1), (192, 64, 3, 300000, 4)])
def test_the_wave_per_member_probe_kernel_still_matches(oracle_abi, monkeypatch, n, cap, p, loss, shards):
    """sp_probe_kernel (one WAVE per member, the first form; SWIMSIM_SP_PROBE=wave, read when a handle is created) stays in the library
    for A/B measurements against sp_probe_lane_kernel (the default everywhere else in this file): it must stay exact."""
    from swim_amd.shard import LocalFabric, ShardedSim
    from tests import hostemu_binding
    emu = hostemu_binding.load()
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=17, lossPpm=loss, eventMask=0x1F, suspicionTicks=5, viewCap=cap)
    monkeypatch.setenv("SWIMSIM_SP_PROBE", "wave")
    a = Sim.create(oracle_abi, sc)
    b = Sim.create(emu, sc) if shards == 1 else ShardedSim(emu, sc, LocalFabric(shards))
    monkeypatch.delenv("SWIMSIM_SP_PROBE")
    for s in (a, b):
        s.scheduleFault(3, 7, False); s.scheduleFault(20, 7, True)
    for _ in range(8):
        a.step(5); b.step(5)
        assert a.counters() == b.counters() and a.digest() == b.digest()
        assert a.drainEventsRaw() == b.drainEventsRaw()
    a.close(); b.close()
