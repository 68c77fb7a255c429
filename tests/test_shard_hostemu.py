"""The sharded path (row e): the population split over several handles must be bit-identical to the
unsharded run and to the oracle.  Runs the product's kernel sources through the host emulation, with
all shards in one process (LocalFabric); tests/test_shard_dist.py repeats it with one process per shard
over torch.distributed (gloo)."""
import pytest

from swim_amd import Config, SimConfig, workloads
from swim_amd.shard import LocalFabric, ShardedSim
from swim_amd import Sim


@pytest.fixture(scope="module")
def emu_abi():
    from tests import hostemu_binding
    return hostemu_binding.load()


def lockstep(oracle, sharded, ticks, chunk, observers, members):
    done = 0
    while done < ticks:
        n = min(chunk, ticks - done)
        oracle.step(n); sharded.step(n)
        done += n
        assert oracle.counters() == sharded.counters(), "counters differ after %d ticks" % done
        assert oracle.digest() == sharded.digest(), "digest differs after %d ticks" % done
        assert oracle.drainEventsRaw() == sharded.drainEventsRaw(), "events differ after %d ticks" % done
        for o in observers:
            assert oracle.members(o) == sharded.members(o)
        for m in members:
            assert oracle.readMember(m) == sharded.readMember(m)
    assert oracle.firstDetection() == sharded.firstDetection()


@pytest.mark.parametrize("n,shards,p,loss,seed", [
    (128, 2, 3, 0, 1), (256, 4, 3, 0, 2), (192, 3, 2, 50000, 3), (512, 8, 3, 200000, 4), (64, 2, 10, 300000, 5),
])
def test_sharded_matches_oracle(oracle_abi, emu_abi, n, shards, p, loss, seed):
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F,
                   suspicionTicks=6, maxSubjects=min(n, 1024))
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(emu_abi, sc, LocalFabric(shards))
    for s in (a, b):
        s.crash(n // 2, 5)
        s.crash(3, 7)
        s.scheduleFault(40, n // 2, True)
    lockstep(a, b, 70, 1 if n <= 200 else 5, observers=(0, n - 1, n // 2), members=(0, n - 1, n // 2))
    b.close()


def test_sharded_saturated_queues(oracle_abi, emu_abi):
    """Many crashes: full queues, many rumours in flight, cross-shard payloads dominate."""
    n = 1024
    crashes = workloads.hashed_crashes(n, 5, 1, 8, 3, 33)
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=5, eventMask=0x1F, suspicionTicks=7,
                   maxSubjects=512)
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(emu_abi, sc, LocalFabric(4))
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
    lockstep(a, b, 50, 5, observers=(0, 1, n - 1), members=(0, 1, n - 1))
    b.close()


def test_sharded_with_tiny_mask_window(oracle_abi):
    """4-id mask window: most cross-shard payloads take the explicit 72-byte records and foreign lines,
    ticks after id bursts disable masks altogether."""
    from tests import hostemu_binding
    emu = hostemu_binding.load_variant("win4", ["SWIM_MASK_WIN=4", "SWIM_MASK_SLACK=2"])
    n = 384
    crashes = workloads.hashed_crashes(n, 11, 1, 8, 3, 33)
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=11, lossPpm=20000, eventMask=0x1F, suspicionTicks=6,
                   maxSubjects=384, inboxCap=2)
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(emu, sc, LocalFabric(3))
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        s.scheduleFault(40, crashes[0][1], True)
    lockstep(a, b, 60, 5, observers=(0, n - 1), members=(0, n - 1))
    b.close()


@pytest.mark.parametrize("shards,loss,ticks", [(2, 20000, 420), (4, 0, 200), (8, 50000, 140)])
def test_sharded_settling_with_churn(oracle_abi, emu_abi, shards, loss, ticks):
    """gc_ticks on a sharded cluster: a subject settles only when it is quiet on every shard, every shard commits
    the same base in the same tick (round 3), rows are reclaimed and reused per shard, members come back after
    their subject was removed -- every observable equals the (unsharded) oracle's after every block of ticks."""
    from swim_amd import _abi
    n = 320
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=12 + shards, lossPpm=loss, eventMask=0x1F, suspicionTicks=5,
                   retransmitMult=1, maxSubjects=60, gcTicks=_abi.GC_AUTO)
    crashes = [(3 + 4 * k, (7 * k + 11) % n) for k in range(70)]
    faults = [(t + 9 + (k % 5) * 14, m, True) for k, (t, m) in enumerate(crashes) if k % 3 == 0]
    faults += [(t + 2, (m + 1) % n, False) for (t, m) in crashes[::7]]            # neighbours that sleep through deadlines
    faults += [(t + 12, (m + 1) % n, True) for (t, m) in crashes[::7]]
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(emu_abi, sc, LocalFabric(shards))
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        for (t, m, up) in faults:
            s.scheduleFault(t, m, up)
    lockstep(a, b, ticks, 7, observers=(0, 12, n - 1), members=(0, 12, n - 1))
    c = b.counters()
    assert c["settled"] > (60 if ticks >= 420 else 10) and c["timers_fired"] > 0
    b.close()


@pytest.mark.parametrize("n,shards,p,loss,seed", [(256, 4, 3, 0, 1), (300, 3, 2, 100000, 2), (512, 8, 10, 30000, 3)])
def test_sharded_robust_target_scheme(oracle_abi, emu_abi, n, shards, p, loss, seed):
    """The robust (round-robin) target scheme on a sharded cluster: same targets, payloads pushed to the
    target's owner instead of pulled -- bit-identical to the unsharded oracle."""
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, targetScheme=1, eventMask=0x1F,
                   suspicionTicks=7, maxSubjects=min(n, 512))
    crashes = workloads.hashed_crashes(n, seed, 1, 16, 3, 33)
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(emu_abi, sc, LocalFabric(shards))
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        s.scheduleFault(45, crashes[0][1], True)
    lockstep(a, b, 70, 5, observers=(0, n - 1, crashes[0][1]), members=(0, n - 1, crashes[0][1]))
    b.close()


@pytest.mark.parametrize("shards,gc", [(2, False), (4, True)])
def test_sharded_join_pull_with_churn(oracle_abi, emu_abi, shards, gc):
    """join_pull on a sharded cluster: the join host of a member that comes up usually lives on another shard -- its
    owner sends what the host knows ahead of the tick's probes (round 0).  Several joins in one tick, hosts skipped
    because they change in the same tick, pulled Suspect entries, rows opened by a pull, with and without settling."""
    from swim_amd import _abi
    n = 300
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=21 + shards, lossPpm=30000, eventMask=0x1F, suspicionTicks=6,
                   retransmitMult=2, maxSubjects=200 if gc else 300, gcTicks=_abi.GC_AUTO if gc else 0, joinPull=1)
    crashes = [(2 + 3 * k, (11 * k + 5) % n) for k in range(60)]
    faults = [(t + 4 + (k % 6) * 5, m, True) for k, (t, m) in enumerate(crashes) if k % 2 == 0]
    faults += [(50, m, False) for m in range(100, 140)] + [(58, m, True) for m in range(100, 140)]   # 40 joins in one tick
    faults += [(58, m, False) for m in range(140, 170)]                                              # 30 busy non-hosts
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(emu_abi, sc, LocalFabric(shards))
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        for (t, m, up) in faults:
            s.scheduleFault(t, m, up)
    lockstep(a, b, 260, 1 if not gc else 4, observers=(0, 100, 139, n - 1), members=(0, 100, 139, n - 1))
    c = b.counters()
    assert c["timers_fired"] > 0 and (not gc or c["settled"] > 20)
    b.close()


@pytest.fixture
def phase_calls(monkeypatch):
    """SWIMSIM_CLUSTER_STEP=0: ShardedSim steps a one-process cluster through swimsim_shard_phase1/2/3 with LocalFabric's copies
    (the embedder's exchange) instead of swimsim_cluster_step (the exchange inside the library, peers' buffers read in place):
    the same tick through both forms of the exchange."""
    monkeypatch.setenv("SWIMSIM_CLUSTER_STEP", "0")


@pytest.mark.parametrize("n,shards,p,loss,seed", [(256, 4, 3, 0, 2), (512, 8, 3, 200000, 4), (64, 2, 10, 300000, 5)])
def test_phase_calls_match_oracle(oracle_abi, emu_abi, phase_calls, n, shards, p, loss, seed):
    test_sharded_matches_oracle(oracle_abi, emu_abi, n, shards, p, loss, seed)


def test_phase_calls_saturated_queues(oracle_abi, emu_abi, phase_calls):
    test_sharded_saturated_queues(oracle_abi, emu_abi)


def test_phase_calls_with_tiny_mask_window(oracle_abi, phase_calls):
    test_sharded_with_tiny_mask_window(oracle_abi)


def test_phase_calls_with_settling_and_join_pull(oracle_abi, emu_abi, phase_calls):
    test_sharded_settling_with_churn(oracle_abi, emu_abi, 2, 20000, 210)
    test_sharded_join_pull_with_churn(oracle_abi, emu_abi, 3, True)


def test_phase_calls_with_the_robust_scheme(oracle_abi, emu_abi, phase_calls):
    test_sharded_robust_target_scheme(oracle_abi, emu_abi, 256, 4, 3, 0, 1)
    test_sharded_robust_target_scheme(oracle_abi, emu_abi, 300, 3, 2, 100000, 2)


def test_shard_phases_are_refused_out_of_order(emu_abi):
    """The sharded entry points are a small state machine (phase0? -> phase1 -> phase2 -> phase3 [-> settle]); anything
    out of order is SWIMSIM_ERR_STATE with a message, never a silent step."""
    import ctypes as C
    from swim_amd import _abi
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=128, seed=1, gcTicks=_abi.GC_AUTO, suspicionTicks=5, retransmitMult=1)
    s = Sim.create(emu_abi, sc, shard_index=0, n_shards=2)
    a, h = emu_abi, s._h
    c3, c1 = (C.c_uint32 * 6)(), (C.c_uint32 * 2)()
    assert a.step(h, 1) == _abi.ERR_STATE                                   # a shard is not stepped alone
    assert a.shard_phase2(h, c1, c3) == _abi.ERR_STATE                      # before phase 1
    assert a.shard_settle_commit(h, c1) == _abi.ERR_STATE                   # no tick to end yet
    assert a.shard_phase1(h, c3) == _abi.OK
    assert a.shard_phase1(h, c3) == _abi.ERR_STATE
    assert b"order" in a.last_error(h)
    assert a.shard_phase2(h, c1, c3) == _abi.OK and a.shard_phase3(h, c1, c1) == _abi.OK
    assert a.shard_phase1(h, c3) == _abi.ERR_STATE                          # settling: the tick ends with round 3
    assert a.shard_settle_counts(h, c1) == _abi.OK and a.shard_settle_commit(h, c1) == _abi.OK
    assert a.shard_phase1(h, c3) == _abi.OK
    s.close()
    plain = Sim.create(emu_abi, SimConfig(cfg=Config(numToGossip=3), nMembers=64, seed=1))
    assert emu_abi.shard_phase1(plain._h, c3) == _abi.ERR_STATE             # not a sharded handle
    plain.close()


def test_cluster_step_checks_its_handles(emu_abi):
    """swimsim_cluster_step takes n DISTINCT shards of ONE resolved configuration, between ticks: a handle twice, a different
    seed or option are refused with a message -- never stepped into silent divergence; and
    swimsim_shard_traffic reports what a shard put on the wire."""
    import ctypes as C
    from swim_amd import _abi
    mk = lambda k, **kw: Sim.create(emu_abi, SimConfig(cfg=Config(numToGossip=3), nMembers=256, seed=kw.pop("seed", 5), suspicionTicks=6, **kw), shard_index=k, n_shards=2)
    a, b = mk(0), mk(1)
    arr = lambda *hs: (C.c_void_p * len(hs))(*[h._h for h in hs])
    assert emu_abi.cluster_step(arr(a, b), 2, 3) == _abi.OK and a.tick == b.tick == 3
    out = (C.c_uint64 * 4)()
    assert emu_abi.shard_traffic(a._h, out) == _abi.OK and out[0] >= 128 * 9 + 512 and out[1] % 8 == 0
    assert emu_abi.cluster_step(arr(a, a), 2, 1) == _abi.ERR_INVALID                       # the same handle twice
    c = mk(1, seed=6)
    assert emu_abi.cluster_step(arr(a, c), 2, 1) == _abi.ERR_INVALID and b"ONE cluster" in emu_abi.last_error(c._h)   # another seed (and another tick)
    for s in (a, b, c):
        s.close()


def _soak():
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak_hostemu.py")
    spec = importlib.util.spec_from_file_location("soak_hostemu", path)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def test_soak_case_703_24_a_message_from_outside_opens_its_row_on_every_shard():
    """The case a round-5 soak found (scripts/soak_hostemu.py 703, case 24): 130 members on 2 shards, 20 % loss, settling, join pulls,
    pull_ticks = 40, messages from outside.  A message from outside gives its subject a view row -- on EVERY shard:
    a member of another shard than the observer's that pulls the subject's own map in that very tick must take the subject's own
    incarnation over (src/Core.hs:110-117 feeds the rule; the row is what a state pull walks).  Before swimsim_note_outside_rumor the
    row existed on the observer's shard only and counters / settled counts parted from the oracle at tick 300."""
    ok, what = _soak().run_case(703, 24, log=lambda *a, **k: None)
    assert ok, what


def test_soak_case_832_8_an_id_exactly_one_window_short_of_a_turn():
    """Round 6's soak (scripts/soak_hostemu.py 832, case 8; 8-bit rumour ids, 1 500 members on 4 shards, robust scheme, 2 % loss, settling,
    pull_ticks = 3): a shard handed out 192 rumour ids within one tick; young_rid let the id H + 2^bits - KW_BITS travel with its id, whose
    low bits are those of H - KW_BITS, the oldest id of the receivers' wide window -- Suspect@0 and Dead@0 about one member under ONE id, and
    the receiver's test-and-set dropped the Dead.  An off-by-one since round 3 (`<=` for `<`), exposed when the ring directory changed
    which ids a tick hands out; the product's 16-bit ids reach it at 65 280 new rumours in one tick."""
    ok, what = _soak().run_case(832, 8, log=lambda *a, **k: None)
    assert ok, what


@pytest.mark.parametrize("shards,T,n", [(2, 3, 600), (4, 5, 1000), (3, 40, 300)])
def test_messages_from_outside_with_state_pulls_and_settling_on_shards(oracle_abi, emu_abi, shards, T, n):
    """swimsim_inject_rumor x pull_ticks x settling x shards, deliberately dense in the combination that diverged: many messages from
    outside per tick about members nobody else talks about (their row exists only because of the message), pullers on every shard in
    every tick (small T), hosts that ARE the named subjects."""
    from swim_amd import _abi
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=91 + T, lossPpm=100000, eventMask=0x1F, suspicionTicks=5,
                   maxSubjects=n, gcTicks=_abi.GC_AUTO, joinPull=1, pullTicks=T)
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(emu_abi, sc, LocalFabric(shards))
    for s in (a, b):
        for k in range(12):
            s.crash((53 * k + 7) % n, 4 + 3 * k)
            if k % 2 == 0:
                s.scheduleFault(4 + 3 * k + 11 + k, (53 * k + 7) % n, True)
    for blk in range(30):
        for s in (a, b):
            for j in range(10):
                x = blk * 10 + j
                s.injectRumor((977 * x + 5) % n, (1201 * x + 40) % n, x % 3, (x // 3) % 3)
        a.step(2); b.step(2)
        assert a.counters() == b.counters(), "counters differ at tick %d" % a.tick
        assert a.digest() == b.digest(), "digest differs at tick %d" % a.tick
        assert a.drainEventsRaw() == b.drainEventsRaw(), "events differ at tick %d" % a.tick
    assert a.firstDetection() == b.firstDetection()
    assert b.counters()["settled"] > 0
    a.close(); b.close()
