"""GPU parity: the HIP tick (through the C ABI of libswimsim.so) against the CPU oracle on the
same seeded inputs.  Integer path => bit-exact: state digest, event stream, counters, views,
piggyback buffers and first-detection ticks must be identical."""
import pytest

from swim_amd import Config, Sim, SimConfig
from swim_amd import workloads
from tests.helpers import compare_state, make_pair, run_lockstep

pytestmark = pytest.mark.gpu


def test_config1_every_tick(oracle_abi, hip_abi):
    """BASELINE config 1: 128 members, k=3, member 64 crashes at tick 10; compared every tick."""
    sc, crashes, ticks = workloads.config1()
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes)
    run_lockstep(a, b, ticks, 1, observers=(0, 1, 63, 64, 65, 127), members=(0, 5, 64, 100))
    fd = b.firstDetection()
    assert fd[64] is not None and fd[64] >= 10
    # every live member ends with m64 Dead and nothing else
    for o in (0, 17, 127):
        ms = b.members(o)
        assert [(m.memberName, int(m.memberAlive)) for m in ms] == [("m64", 2)]


@pytest.mark.parametrize("n,p,loss,seed", [
    (2, 1, 0, 1), (3, 3, 0, 2), (65, 3, 0, 3), (64, 1, 100000, 4), (200, 3, 300000, 5),
    (1000, 3, 50000, 6), (1000, 10, 200000, 7), (777, 5, 0, 8),
])
def test_small_populations_with_loss(oracle_abi, hip_abi, n, p, loss, seed):
    """Ragged sizes (not multiples of 64/256), heavy loss => refutations, false suspicions,
    indirect probes; P up to the reference default numToGossip=10."""
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F,
                   suspicionTicks=6, maxSubjects=min(n, 1024))
    crashes = [(5, n // 2)] if n > 2 else []
    faults = [(40, n // 2, True)] if n > 2 else []
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes, faults)
    run_lockstep(a, b, 80, 1 if n <= 200 else 8, observers=(0, n - 1, n // 2), members=(0, n - 1, n // 2))
    if loss:
        assert b.counters()["direct_failed"] > 0


def test_config2_full(oracle_abi, hip_abi):
    """BASELINE config 2: 65 536 members, k=3, 1 % hashed crashes, 400 ticks."""
    sc, crashes, ticks = workloads.config2()
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes)
    run_lockstep(a, b, ticks, 50, observers=(0, 4242, 65535), members=(1, 4242, 65535))
    c = b.counters()
    assert c["suspects"] >= len(crashes)
    assert c["false_suspects"] == 0


def test_kat_k_random_members_device(hip_abi, oracle_abi):
    """test/Spec.hs:108-139 on the device selection routine."""
    for abi in (hip_abi, oracle_abi):
        sc = SimConfig(cfg=Config(numToGossip=3), nMembers=4, seed=1)
        s = Sim.create(abi, sc)   # 0="alive" 1="suspect" 2="dead" 3="myself"
        s.setView(3, 1, 1, 0)
        s.setView(3, 2, 2, 0)
        assert s.kRandomMembers(3, 0, [0, 1, 2]) == []
        assert s.kRandomMembers(3, 3, []) == [0]
        assert s.kRandomMembers(3, 3, [0]) == []
        s.close()
    picks = []
    for abi in (hip_abi, oracle_abi):
        s = Sim.create(abi, SimConfig(cfg=Config(numToGossip=3), nMembers=201, seed=9))
        r = s.kRandomMembers(200, 50, [])
        assert len(r) == 50 and len(set(r)) == 50 and 200 not in r
        assert r != list(range(50))
        picks.append(r)
        s.close()
    assert picks[0] == picks[1]


def test_million_members_digest(oracle_abi, hip_abi):
    """BASELINE config 3 size (1 048 576 members): digest parity over a short seeded run."""
    sc, _, _ = workloads.config3(max_subjects=64)
    crashes = workloads.hashed_crashes(1 << 20, 1, 1, 100000, 2, 6)   # ~10 crashes, ticks 2..5
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes)
    run_lockstep(a, b, 12, 4, observers=(0, 1 << 19), members=(7,), check_events=True)


@pytest.mark.parametrize("name", ["config1_n128_k3", "lossy_n96_k3", "churn_n64_k2", "robust_n96_k3", "bounded_n96_cap8", "strict_n96_k3", "pushpull_n96_k3"])
def test_hip_matches_committed_golden_fixtures(hip_abi, name):
    """The HIP path against the committed golden vectors (tests/golden/*.json)."""
    import json, os
    from tests.test_oracle_semantics import GOLDEN, run_fixture
    fx = json.load(open(os.path.join(GOLDEN, name + ".json")))
    assert run_fixture(hip_abi, fx["spec"]) == fx["expect"]


def test_million_members_properties(hip_abi):
    """Full BASELINE size without the oracle: size-independent properties.  (a) determinism:
    two runs give the same digest; (b) completeness/accuracy at zero loss: every crashed member
    is Dead in sampled live views, no false suspicion; (c) detection latency >= 1 and its mean
    near 1/(1-e^-3); (d) conservation: changes == 2 * crashes * (live observers) once settled."""
    import math
    n = 1 << 20
    crashes = workloads.hashed_crashes(n, 3, 1, 20000, 2, 52)        # ~50 crashes, about one per tick
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=3, maxSubjects=128, suspicionTicks=20)
    digests = []
    for rep in range(2):
        s = Sim.create(hip_abi, sc)
        workloads.apply_crashes(s, crashes)
        s.step(130)
        digests.append(s.digest())
        if rep == 0:
            c = s.counters()
            fd = s.firstDetection()
            lat = [fd[m] - t + 1 for (t, m) in crashes]
            assert all(l >= 1 for l in lat)
            assert abs(sum(lat) / len(lat) - 1 / (1 - math.exp(-3))) < 0.25
            assert c["false_suspects"] == 0 and c["refutes"] == 0
            live = n - len(crashes)
            # every live observer saw Suspect then Dead for every crashed subject; crashed members
            # stop observing when they go down
            assert 2 * len(crashes) * live <= c["changes"] <= 2 * len(crashes) * n
            dead = {"m%d" % m for (_, m) in crashes}
            for o in (0, 12345, n - 1):
                if "m%d" % o in dead:
                    continue
                v = s.members(o)
                assert {m.memberName for m in v} == dead and all(int(m.memberAlive) == 2 for m in v)
        s.close()
    assert digests[0] == digests[1]


@pytest.mark.parametrize("n,shards,loss,seed", [(4096, 4, 0, 1), (3072, 3, 100000, 2), (65536, 8, 20000, 3)])
def test_sharded_cluster_on_one_gpu(oracle_abi, hip_abi, n, shards, loss, seed):
    """Row (e): the population split over several handles on this GPU (LocalFabric copies the exchange
    records between them) must be bit-identical to the oracle, every observable."""
    from swim_amd.shard import LocalFabric, ShardedSim
    # every gossip event is compared on the small clusters; on the large one only probe / refute / join
    # events (the per-handle event rings would overflow at different points, which is not protocol state)
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F if n <= 4096 else 0,
                   suspicionTicks=8, maxSubjects=min(n, 4096))
    crashes = workloads.hashed_crashes(n, seed, 1, 256, 3, 23)
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        s.scheduleFault(30, crashes[0][1], True)
    done = 0
    while done < 60:
        a.step(10); b.step(10); done += 10
        assert a.counters() == b.counters(), "counters differ after %d ticks" % done
        assert a.digest() == b.digest(), "digest differs after %d ticks" % done
        assert a.drainEventsRaw() == b.drainEventsRaw()
        for o in (0, n - 1, crashes[0][1]):
            assert a.members(o) == b.members(o)
            assert a.readMember(o) == b.readMember(o)
    assert a.firstDetection() == b.firstDetection()
    b.close()


def test_sharded_settling_on_one_gpu(oracle_abi, hip_abi):
    """gc_ticks on a sharded cluster (round 3: every shard's word about its rows; the same base committed by every
    shard in the same tick): rows are reclaimed and reused per shard, members come back after their subject
    was removed.  Several handles on this GPU; every observable against the (unsharded) oracle."""
    from swim_amd import _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    n, shards = 8192, 4
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=21, lossPpm=5000, eventMask=0x1F, suspicionTicks=5,
                   retransmitMult=1, maxSubjects=400, gcTicks=_abi.GC_AUTO)
    crashes = [(3 + 2 * k, (977 * k + 11) % n) for k in range(120)]
    faults = [(t + 9 + (k % 5) * 14, m, True) for k, (t, m) in enumerate(crashes) if k % 3 == 0]
    faults += [(t + 2, (m + 1) % n, False) for (t, m) in crashes[::7]]
    faults += [(t + 12, (m + 1) % n, True) for (t, m) in crashes[::7]]
    a = Sim.create(oracle_abi, sc)
    _oracle_threads(a)
    b = ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        for (t, m, up) in faults:
            s.scheduleFault(t, m, up)
    done = 0
    while done < 360:
        a.step(12); b.step(12); done += 12
        assert a.counters() == b.counters(), "counters differ after %d ticks" % done
        assert a.digest() == b.digest(), "digest differs after %d ticks" % done
        assert a.drainEventsRaw() == b.drainEventsRaw()
        for o in (0, n - 1, crashes[0][1]):
            assert a.members(o) == b.members(o)
    assert a.firstDetection() == b.firstDetection()
    assert b.counters()["settled"] > 100
    b.close()


@pytest.mark.parametrize("n,shards,p,loss,seed", [(4096, 4, 3, 0, 1), (3000, 3, 10, 50000, 2)])
def test_sharded_robust_scheme_on_one_gpu(oracle_abi, hip_abi, n, shards, p, loss, seed):
    """The robust target scheme on a sharded cluster (same targets, payloads pushed to the target's owner)."""
    from swim_amd.shard import LocalFabric, ShardedSim
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, targetScheme=1, eventMask=0x1F,
                   suspicionTicks=7, maxSubjects=min(n, 2048))
    crashes = workloads.hashed_crashes(n, seed, 1, 128, 3, 33)
    a = Sim.create(oracle_abi, sc)
    b = ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        s.scheduleFault(45, crashes[0][1], True)
    done = 0
    while done < 70:
        a.step(10); b.step(10); done += 10
        assert a.counters() == b.counters(), "counters differ after %d ticks" % done
        assert a.digest() == b.digest(), "digest differs after %d ticks" % done
        assert a.drainEventsRaw() == b.drainEventsRaw()
    assert a.firstDetection() == b.firstDetection()
    b.close()


def test_sharded_join_pull_on_one_gpu(oracle_abi, hip_abi):
    """join_pull on a sharded cluster (round 0: the owner of a join host sends what the host knows ahead of the tick's
    probes), with settling: 400 joins in one tick at 16 384 members in 4 shards, every observable against the oracle."""
    from swim_amd import _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    n, shards = 16384, 4
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=31, lossPpm=10000, eventMask=0, suspicionTicks=6,
                   retransmitMult=1, maxSubjects=2000, gcTicks=_abi.GC_AUTO, joinPull=1)
    crashes = [(2 + k // 8, (2731 * k + 5) % n) for k in range(400)]
    faults = [(70, m, True) for (_, m) in crashes] + [(40 + (k % 9), m, True) for k, (_, m) in enumerate(crashes[::5])]
    a = Sim.create(oracle_abi, sc)
    _oracle_threads(a)
    b = ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        for (t, m, up) in faults:
            s.scheduleFault(t, m, up)
    done = 0
    while done < 140:
        a.step(10); b.step(10); done += 10
        assert a.counters() == b.counters(), "counters differ after %d ticks" % done
        assert a.digest() == b.digest(), "digest differs after %d ticks" % done
        for o in (0, crashes[0][1], n - 1):
            assert a.members(o) == b.members(o)
    assert a.firstDetection() == b.firstDetection()
    b.close()


@pytest.mark.parametrize("n,loss,gc,T", [(65536, 20000, 0, 0), (262144, 0, 1, 0), (65536, 50000, 1, 16)])
def test_cluster_step_across_two_devices(oracle_abi, hip_abi, n, loss, gc, T):
    """swimsim_cluster_step with its handles on DIFFERENT devices: the kernels read the peers' buffers over xGMI (peer access),
    cross-device events order the handles' streams -- the path no one-GPU box can run.  Skipped unless two devices are visible; = the
    unsharded oracle."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (peer access between the handles' devices)")
    from swim_amd import _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    ndev = min(torch.cuda.device_count(), 8)
    while n % ndev:
        ndev -= 1
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=61, lossPpm=loss, eventMask=0, suspicionTicks=7, maxSubjects=4096,
                   gcTicks=_abi.GC_AUTO if gc else 0, joinPull=1 if T else 0, pullTicks=T)
    a, b = Sim.create(oracle_abi, sc), ShardedSim(hip_abi, sc, LocalFabric(ndev), devices=list(range(ndev)))
    _oracle_threads(a)
    crashes = [(3 + 2 * k, (37 * k + 11) % n) for k in range(30)]
    for s in (a, b):
        for t, m in crashes:
            s.crash(m, t)
        for t, m in crashes[::2]:
            s.scheduleFault(t + 9 + (m % 13), m, True)
    for _ in range(12):
        a.step(5); b.step(5)
        assert a.counters() == b.counters() and a.digest() == b.digest(), "tick %d" % a.tick
    assert a.firstDetection() == b.firstDetection()
    a.close(); b.close()


@pytest.mark.parametrize("n,shards,loss,seed", [(4096, 4, 0, 1), (65536, 8, 20000, 3)])
def test_sharded_cluster_by_phase_calls_on_one_gpu(oracle_abi, hip_abi, monkeypatch, n, shards, loss, seed):
    """SWIMSIM_CLUSTER_STEP=0: the same cluster stepped through swimsim_shard_phase1/2/3 with the embedder's copies (LocalFabric)
    instead of swimsim_cluster_step (the exchange inside the library) -- must give the same run."""
    monkeypatch.setenv("SWIMSIM_CLUSTER_STEP", "0")
    test_sharded_cluster_on_one_gpu(oracle_abi, hip_abi, n, shards, loss, seed)


def test_one_process_per_shard_on_one_gpu():
    """Two processes, one shard each, both on GPU 0, torch.distributed (gloo, records staged through
    host memory because RCCL refuses two ranks on one device): the DistFabric host code with the real
    HIP library."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", SWIM_DIST_DEVICE="cuda")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29655", os.path.join(root, "tests", "dist_worker.py"), "8192", "3", "50000", "4", "40"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DIST-OK world=2" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_million_members_sharded_digest_equals_unsharded(hip_abi):
    """BASELINE size, no oracle: the saturated 1 M-member run must give the same state digest, counters
    and first-detection ticks whether it is one handle or 2 / 8 shards (several handles on this GPU).
    Size-independent property of the exchange -- and a race detector for it."""
    from swim_amd.shard import LocalFabric, ShardedSim
    n, ticks = 1 << 20, 40
    sc, crashes, _ = workloads.saturated(n, ticks, crashes_per_tick=2.0, t0=2)
    results = []
    for shards in (1, 2, 8):
        s = Sim.create(hip_abi, sc) if shards == 1 else ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
        workloads.apply_crashes(s, crashes)
        s.step(ticks)
        c = s.counters()
        results.append((s.digest(), c, s.firstDetection()))
        s.close()
    assert results[0][1]["changes"] > 10 * n                      # the run really disseminates
    assert results[1] == results[0] and results[2] == results[0]


@pytest.mark.parametrize("n", [1 << 10, 1 << 15, 1 << 20])
def test_dissemination_is_logarithmic(hip_abi, n):
    """SURVEY 8c known-answer check: infection-style dissemination reaches everybody in O(log N) periods.
    One member crashes; from its first detection on, count the periods until every live member has
    changed its entry (Suspect) -- with 2P = 6 payloads per member-period the epidemic needs about
    log_7 N + a few periods; 2 log2 N is a generous bound that a broken piggyback path cannot meet."""
    import math
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=7, maxSubjects=16)
    s = Sim.create(hip_abi, sc)
    s.crash(n // 3, 2)
    s.step(2)
    detected_at = None
    for t in range(2, 2 + 4 * int(math.log2(n))):
        s.step(1)
        c = s.counters()
        if detected_at is None and c["suspects"] > 0:
            detected_at = t
        if c["changes"] >= n - 1:
            break
    else:
        raise AssertionError("rumour did not reach all %d members" % n)
    assert detected_at is not None
    assert t - detected_at <= 2 * math.log2(n), (t, detected_at)
    assert s.counters()["false_suspects"] == 0
    s.close()


@pytest.mark.parametrize("n,p,loss,seed", [(777, 3, 0, 1), (4096, 3, 50000, 2), (1000, 10, 150000, 3), (65536, 3, 5000, 4)])
def test_robust_target_scheme_parity(oracle_abi, hip_abi, n, p, loss, seed):
    """the robust (round-robin) target scheme (src/Core.hs:232), Ping payloads pulled by
    the target instead of pushed with atomics: bit-exact against the oracle."""
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, targetScheme=1, eventMask=0x1F if n <= 4096 else 0,
                   suspicionTicks=7, maxSubjects=min(n, 2048))
    crashes = workloads.hashed_crashes(n, seed, 1, 128, 3, 33)
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes, [(45, crashes[0][1], True)])
    run_lockstep(a, b, 70, 10, observers=(0, n - 1, crashes[0][1]), members=(0, n - 1, crashes[0][1]))
    fd = b.firstDetection()
    if loss == 0:
        assert all(fd[m] == t for (t, m) in crashes[1:])  # detected in the period of the crash (crashes[0] rejoined)


@pytest.mark.parametrize("n,p,loss,seed,robust", [(777, 3, 300000, 1, 0), (4096, 3, 150000, 2, 0), (1000, 10, 200000, 3, 0), (65536, 3, 50000, 4, 0),
                                                 (4096, 3, 150000, 5, 1), (2048, 3, 0, 6, 0)])
def test_strict_reference_rules_on_the_gpu(oracle_abi, hip_abi, n, p, loss, seed, robust):
    """the LITERAL suspectOrDeadNode' (/root/reference/src/Core.hs:151-152,182-184) under the canonical order
    (include/swimsim.h "Strict reference rules") on the device, bit-exact against the oracle's literal mode -- on runs where the literal
    rule and the merge do part (thousands of proposals decided differently), events record by record up to 4 096 members."""
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, targetScheme=robust, eventMask=0x1F if n <= 4096 else 0,
                   suspicionTicks=7, maxSubjects=n if n <= 4096 else 16384, strictReferenceRules=True)
    crashes = workloads.hashed_crashes(n, seed, 1, 128, 3, 33)
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes, [(45, crashes[0][1], True)])
    if n > 4096:
        _oracle_threads(a)
    run_lockstep(a, b, 70, 10, observers=(0, n - 1, crashes[0][1]), members=(0, n - 1, crashes[0][1]))
    if loss:
        assert oracle_abi.lib.swimoracle_d13_hits(a._h) > 0
    a.close(); b.close()


@pytest.mark.parametrize("n,shards,p,loss,seed", [(3000, 3, 3, 150000, 4), (4096, 4, 10, 200000, 5), (65536, 8, 3, 50000, 3)])
def test_strict_reference_rules_on_sharded_clusters_on_one_gpu(oracle_abi, hip_abi, n, shards, p, loss, seed):
    """The literal suspectOrDeadNode' (src/Core.hs:151-152,182-184) on 3-8 shards on MI355X (round 6): across shards every queue travels as
    a list, nothing is filtered, the owner applies a member's rumours in the canonical order.  = the UNSHARDED oracle's literal mode."""
    from swim_amd.shard import LocalFabric, ShardedSim
    events = n <= 4096
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F if events else 0, suspicionTicks=8,
                   maxSubjects=n if n <= 4096 else 16384, strictReferenceRules=True)
    a, b = Sim.create(oracle_abi, sc), ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    _oracle_threads(a)
    for s in (a, b):
        s.crash(7, 5); s.crash(n // 2, 9)
        s.scheduleFault(40, 7, True)
    for _ in range(12):
        a.step(5); b.step(5)
        assert a.counters() == b.counters() and a.digest() == b.digest(), "tick %d" % a.tick
        if events:
            assert a.drainEventsRaw() == b.drainEventsRaw()
    assert a.firstDetection() == b.firstDetection()
    assert oracle_abi.lib.swimoracle_d13_hits(a._h) > 0
    a.close(); b.close()


@pytest.mark.parametrize("n,shards,loss,T,push", [(3000, 1, 150000, 3, 1), (4096, 4, 100000, 5, 0), (65536, 1, 20000, 64, 0), (65536, 8, 20000, 0, 0)])
def test_strict_reference_rules_with_settling_and_state_pulls_on_the_gpu(oracle_abi, hip_abi, n, shards, loss, T, push):
    """The literal rule x settling x join pull x periodic (push-)pull x shards on MI355X (round 6): = the unsharded oracle's literal mode."""
    from swim_amd import _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    events = n <= 4096
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=77, lossPpm=loss, eventMask=0x1F if events else 0, suspicionTicks=6,
                   maxSubjects=n if n <= 4096 else 16384, strictReferenceRules=True, gcTicks=_abi.GC_AUTO, joinPull=1, pullTicks=T, pushPull=bool(push),
                   retransmitMult=1)
    a = Sim.create(oracle_abi, sc)
    b = Sim.create(hip_abi, sc) if shards == 1 else ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    _oracle_threads(a)
    for s in (a, b):
        for k in range(12):
            s.crash((37 * k + 11) % n, 3 + 2 * k)
            if k % 2 == 0:
                s.scheduleFault(3 + 2 * k + 9 + k, (37 * k + 11) % n, True)
    for _ in range(16):
        a.step(5); b.step(5)
        assert a.counters() == b.counters() and a.digest() == b.digest(), "tick %d" % a.tick
        if events:
            assert a.drainEventsRaw() == b.drainEventsRaw()
    assert a.firstDetection() == b.firstDetection()
    if n <= 4096:      # (the small, lossy cases are runs where the two rules part and rows settle; at 65 536 members and 2 % loss 80 ticks need not get there)
        assert oracle_abi.lib.swimoracle_d13_hits(a._h) > 0 and a.counters()["settled"] > 0
    assert b.counters()["changes"] > n
    a.close(); b.close()


@pytest.mark.parametrize("fold,n", [("0", 4096), ("1", 4096), ("1", 65536)])
def test_plain_ticks_with_and_without_begin_kernel_on_the_gpu(oracle_abi, hip_abi, monkeypatch, fold, n):
    """Ticks without scheduled changes run without begin_kernel (probe_kernel's workgroup 0 does its part on the side, merge_kernel's
    commits the window heads; SWIMSIM_FOLD_BEGIN=0 at create: never): both forms on MI355X against the oracle, plain ticks and ticks
    with crashes, rejoins, set_view and messages from outside interleaved -- here the probe's other workgroups really run next to
    workgroup 0."""
    monkeypatch.setenv("SWIMSIM_FOLD_BEGIN", fold)
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=77, eventMask=0x1F if n <= 4096 else 0, suspicionTicks=6, maxSubjects=2048)
    a, b = make_pair(oracle_abi, hip_abi, sc, [(3, 11), (4, 12), (9, 300), (30, 301)], [(20, 11, True), (41, 300, True)])
    monkeypatch.delenv("SWIMSIM_FOLD_BEGIN")
    if n > 4096:
        _oracle_threads(a)
    for k in range(30):
        if k == 12:
            for s in (a, b):
                s.injectRumor(5, 40, 1, 0); s.injectRumor(6, 41, 2, 0)
        if k == 16:
            for s in (a, b):
                s.setView(observer=8, subject=9, state=1, incarnation=0)
        a.step(2); b.step(2)
        assert a.counters() == b.counters() and a.digest() == b.digest(), "tick %d" % a.tick
        if n <= 4096:
            assert a.drainEventsRaw() == b.drainEventsRaw()
    assert a.firstDetection() == b.firstDetection()
    a.close(); b.close()


def _oracle_threads(sim):
    import os
    from tests import oracle_binding
    oracle_binding.set_threads(sim, min(32, os.cpu_count() or 1))   # scales to ~32 threads (profiles/r02d_oracle_thread_scaling.txt)


def test_saturated_million_members_window_vs_oracle(oracle_abi, hip_abi):
    """The BENCHMARKED regime against the oracle at full size: 1 048 576 members, ~1 crash per tick from
    tick 0, through the pre-roll into the saturated steady state (d = 6, r ~ 2, c = 1) and 40 ticks of it:
    state digest, every counter and the first-detection ticks must be identical (the oracle steps the
    cluster on all host cores; same schedule as bench.py)."""
    n = 1 << 20
    sc, crashes, _ = workloads.saturated(n, 140, seed=1, t0=0)
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes)
    _oracle_threads(a)
    for stop in (60, 80, 100, 120):
        a.step(stop - a.tick); b.step(stop - b.tick)
        ca, cb = a.counters(), b.counters()
        assert ca == cb, "counters differ at tick %d" % stop
        assert a.digest() == b.digest(), "digest differs at tick %d" % stop
    assert a.firstDetection() == b.firstDetection()
    mt = float(n) * 20
    # the window really is the saturated regime the bench times
    a.step(20); b.step(20)
    c2 = b.counters()
    assert a.counters() == c2 and a.digest() == b.digest()
    assert (c2["payloads"] - cb["payloads"]) / mt > 5.9 and (c2["changes"] - cb["changes"]) / mt > 1.2
    for o in (0, 777777):
        assert a.members(o) == b.members(o)


@pytest.mark.parametrize("block", range(3))
def test_random_configurations_on_the_gpu(oracle_abi, hip_abi, block):
    """Seeded randomised sweep (the generator of tests/test_random_sweep.py, sizes up to 65 536): sizes,
    probe counts, loss, crashes and rejoins (with and without the join-time pull), both target schemes, 1-8
    shards, tiny inboxes, settling."""
    import random
    from swim_amd import _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    rng = random.Random(4200 + block)
    for _ in range(8):
        n = rng.choice([2, 3, 17, 64, 129, 300, 777, 1024, 4096, 10000, 65536])
        p = rng.choice([1, 2, 3, 3, 3, 5, 10])
        loss = rng.choice([0, 0, 0, 10000, 100000, 300000]) if n <= 4096 else rng.choice([0, 0, 2000])
        scheme = rng.choice([0, 0, 1])
        shards = 1
        if scheme == 0 and n >= 64 and rng.random() < 0.4:
            shards = rng.choice([g for g in (2, 3, 4, 8) if n % g == 0] or [1])
        gc = _abi.GC_AUTO if shards == 1 and rng.random() < 0.5 else 0
        seed = rng.randrange(1, 1 << 30)
        sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=seed, lossPpm=loss, eventMask=0x1F if n <= 4096 else 0,
                       suspicionTicks=rng.choice([3, 6, 12]), retransmitMult=rng.choice([1, 3]), maxSubjects=min(n, 4096),
                       targetScheme=scheme, inboxCap=rng.choice([0, 0, 1, 2]) if n <= 4096 else 0, gcTicks=gc,
                       joinPull=1 if shards == 1 and seed % 2 else 0,
                       pullTicks=(0, 0, 2, 5, 17)[(seed >> 3) % 5])
        sc.pushPull = bool(sc.pullTicks) and shards == 1 and (seed >> 7) % 2 == 1
        a = Sim.create(oracle_abi, sc)
        _oracle_threads(a)
        b = Sim.create(hip_abi, sc) if shards == 1 else ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
        for _f in range(rng.randrange(0, min(256, max(1, n // 8)) + 1)):
            m, t = rng.randrange(n), rng.randrange(1, 40)
            for s in (a, b):
                s.scheduleFault(t, m, False)
            if rng.random() < 0.5:
                t2 = t + rng.randrange(1, 30)
                for s in (a, b):
                    s.scheduleFault(t2, m, True)
        what = (n, p, loss, scheme, shards, gc, seed)
        for _t in range(rng.choice([3, 6, 12])):
            a.step(10); b.step(10)
            assert a.counters() == b.counters(), ("counters", what)
            assert a.digest() == b.digest(), ("digest", what)
            assert a.drainEventsRaw() == b.drainEventsRaw(), ("events", what)
        assert a.firstDetection() == b.firstDetection(), ("first detection", what)
        a.close(); b.close()


def test_forced_fallback_paths_on_the_gpu(oracle_abi):
    """The gfx950 build with a 4-id mask window and a 1-slot inbox: nearly every delivery takes the explicit
    64-B-line records, the inbox overflow list and the "burst of new ids" fallback (the exact paths behind
    the mask transport), at a size where they race for real."""
    from swim_amd import _lib
    hip = _lib.load_variant("win4")
    n = 20000
    crashes = workloads.hashed_crashes(n, 9, 1, 40, 3, 33)          # ~500 crashes over 30 ticks
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=9, lossPpm=30000, eventMask=0, suspicionTicks=6,
                   maxSubjects=4096, inboxCap=1)
    faults = [(45, m, True) for (_, m) in crashes[:100]]
    a, b = make_pair(oracle_abi, hip, sc, crashes, faults)
    _oracle_threads(a)
    run_lockstep(a, b, 70, 10, observers=(0, 1, n - 1), members=(0, 1, n - 1), check_events=False)


def test_state_by_pointer_build_on_the_gpu(oracle_abi):
    """The gfx950 build whose tick kernels take the state through a pointer to a device copy (-DSWIM_STATE_BY_POINTER:
    no scalar spills, measurement knob of DESIGN.md 9): the same sources must give the same run."""
    from swim_amd import _lib
    hip = _lib.load_variant("sptr")
    n = 50000
    crashes = workloads.hashed_crashes(n, 6, 1, 200, 3, 43)
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=6, lossPpm=10000, eventMask=0, suspicionTicks=6, maxSubjects=4096)
    a, b = make_pair(oracle_abi, hip, sc, crashes, [(45, m, True) for (_, m) in crashes[:50]])
    _oracle_threads(a)
    run_lockstep(a, b, 70, 10, observers=(0, 1, n - 1), members=(0, 1, n - 1), check_events=False)


def test_rumour_id_counter_wraps_on_the_gpu(oracle_abi):
    """The gfx950 build with 10-bit rumour ids: the id counter wraps every 1 024 rumours, several times here
    (racing creators of one rumour take spare ids on the GPU: ~100 new ids per tick in this run).  After a tick
    with more than 3/4 of the id space in new ids, the lines of the next tick carry no ids at all (explicit
    records only): exact at any rate of new rumours; the 8-bit build of the CPU emulation tests
    (tests/test_hostemu_parity.py) lives in that mode most of the time."""
    from swim_amd import _lib
    hip = _lib.load_variant("rid10")
    n = 30000
    crashes = workloads.hashed_crashes(n, 4, 1, 120, 3, 103)        # ~250 crashes over 100 ticks: ~10 new ids per
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=4, lossPpm=1000, eventMask=0, suspicionTicks=6,   # tick (an 8-bit id
                   maxSubjects=8192)                                                                                  # space tolerates 64)
    faults = [(t + 25, m, True) for (t, m) in crashes[:80]]
    a, b = make_pair(oracle_abi, hip, sc, crashes, faults)
    _oracle_threads(a)
    run_lockstep(a, b, 150, 10, observers=(0, 1, n - 1), members=(0, 1, n - 1), check_events=False)


def test_config5_loss_at_16k_members(oracle_abi, hip_abi):
    """BASELINE config 5's message loss (30 %) at 16 384 members: about half of all direct probes escalate
    to the k indirect probes, nearly every member is falsely suspected at some point (one view row per
    member: the most a dense view holds), refutations everywhere.  Every observable, oracle-checked."""
    n = 16384
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=21, lossPpm=300000, eventMask=0, suspicionTicks=10,
                   maxSubjects=n)
    crashes = workloads.hashed_crashes(n, 21, 1, 512, 3, 13)
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes)
    _oracle_threads(a)
    run_lockstep(a, b, 40, 10, observers=(0, n - 1), members=(0, n - 1), check_events=False)
    c = b.counters()
    assert c["direct_failed"] > 0.3 * c["pings"] and c["ping_reqs"] > c["pings"] and c["refutes"] > 0


def test_default_capacities_survive_a_lossy_run(oracle_abi, hip_abi):
    """Library defaults (max_subjects = 0, no per-member timer capacity any more): 16 384 members at 5 % loss
    used to fail with 'timer_cap exceeded' at tick 7."""
    n = 16384
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=2, lossPpm=50000, eventMask=0)
    a, b = make_pair(oracle_abi, hip_abi, sc, [(5, 100)])
    _oracle_threads(a)
    run_lockstep(a, b, 120, 20, observers=(0,), members=(0,), check_events=False)


def test_settling_parity_and_bounded_rows(oracle_abi, hip_abi):
    """gc_ticks on the GPU.  (a) 50 000 members with churn (crash, come back, crash again) and a little loss,
    oracle-checked every 20 ticks through 3 settling horizons; (b) 262 144 members, one crash per tick for
    4 000 ticks with room for 400 subjects: the run needs ten times as many, rows are reclaimed and reused."""
    from swim_amd import _abi
    n = 50000
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=31, lossPpm=2000, eventMask=0, suspicionTicks=6,
                   retransmitMult=1, maxSubjects=1500, gcTicks=_abi.GC_AUTO)
    crashes = workloads.hashed_crashes(n, 31, 1, 100, 2, 162)        # ~500 crashes over 160 ticks
    faults = [(t + 20 + (m % 40), m, True) for (t, m) in crashes[::2]]
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes, faults)
    _oracle_threads(a)
    run_lockstep(a, b, 300, 20, observers=(0, n - 1, crashes[0][1]), members=(0, crashes[0][1]), check_events=False)
    assert b.counters()["settled"] > 400
    a.close(); b.close()
    n, ticks = 1 << 18, 4000
    sc, crashes, _ = workloads.saturated(n, ticks, seed=5, t0=0)
    sc.maxSubjects = 400
    sc.gcTicks = _abi.GC_AUTO
    s = Sim.create(hip_abi, sc)
    workloads.apply_crashes(s, crashes)
    s.step(ticks)                                                    # SWIMSIM_ERR_CAPACITY would raise
    c = s.counters()
    assert c["settled"] > 3500 and c["false_suspects"] == 0
    live = [o for o in (0, 5, n - 1) if o not in {m for (_, m) in crashes}]
    for o in live:
        assert len(s.members(o)) < 400                               # the removed ones are gone from the map
    s.close()


def test_join_pull_many_joins_at_64k(oracle_abi, hip_abi):
    """join_pull at a size where begin_kernel's block has real work: 2 000 members down, 1 200 of them
    back in three ticks (several hundred joins per tick, each merging a host's whole member map), settling on."""
    from swim_amd import _abi
    n = 65536
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=77, lossPpm=2000, suspicionTicks=8, maxSubjects=4096,
                   gcTicks=_abi.GC_AUTO, joinPull=1)
    a = Sim.create(oracle_abi, sc)
    _oracle_threads(a)
    b = Sim.create(hip_abi, sc)
    for k in range(2000):
        m = (k * 31 + 7) % n
        for s in (a, b):
            s.scheduleFault(2 + k % 5, m, False)
            if k < 1200:
                s.scheduleFault(30 + k % 3, m, True)
    for _t in range(8):
        a.step(10); b.step(10)
        assert a.counters() == b.counters()
        assert a.digest() == b.digest()
    for o in ((7 % n), (31 + 7) % n, 5):
        assert a.members(o) == b.members(o)
    a.close(); b.close()


def test_one_percent_loss_million_members_vs_oracle(oracle_abi, hip_abi):
    """The only lossy regime with a throughput claim, oracle-checked at full size ("the lossy regime is not
    oracle-checked at the benchmarked size"): 1 048 576 members, 1 % message loss, ~1 crash per tick, settling on --
    explicit records for every delivery after the id bursts, the wide known-ring, deadline overflow pools with their
    sub-pools (where both soak-found id-overflow bugs lived).  Digest, every counter and the first-detection ticks."""
    from swim_amd import _abi
    n = 1 << 20
    sc, crashes, _ = workloads.saturated(n, 90, seed=1, t0=0, loss_ppm=10000)
    sc.gcTicks = _abi.GC_AUTO
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes)
    _oracle_threads(a)
    for stop in (30, 60, 75):
        a.step(stop - a.tick); b.step(stop - b.tick)
        ca, cb = a.counters(), b.counters()
        assert ca == cb, "counters differ at tick %d" % stop
        assert a.digest() == b.digest(), "digest differs at tick %d" % stop
    assert a.firstDetection() == b.firstDetection()
    assert cb["false_suspects"] > 0 and cb["refutes"] > 0            # the loss really bites
    for o in (0, 777777):
        assert a.members(o) == b.members(o)


def test_wire_datagram_of_a_hip_member_equals_the_oracle_members(oracle_abi, hip_abi):
    """Rows f-2 / a18 under the driver's GPU run: what a simulated member puts on the wire in a period -- its control
    message and its piggyback queue as one compound Envelope (src/Types.hs:96-119, swim_wire.cpp) -- serialised from
    the HIP handle's state and from the oracle handle's must be the same bytes, and decode back to the same messages."""
    from swim_amd import wire
    from swim_amd.types import Ping, Ack, IndirectPing
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=4096, seed=11, lossPpm=20000, eventMask=0x1F, suspicionTicks=8)
    crashes = [(2, 100), (3, 2000), (5, 3000)]
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes)
    a.step(14); b.step(14)
    seen = 0
    for m in (0, 1, 99, 101, 1999, 2048, 4095):
        for ctl in (Ping(seqNo=14, node="m%d" % m), Ack(seqNo=14, payload=[]), IndirectPing(seqNo=14, target=5, port=4000, node="m7")):
            da, db = wire.datagram_of(a, m, ctl), wire.datagram_of(b, m, ctl)
            assert da == db
            err, msgs = wire.decode(db)
            assert err is None and msgs[0] == ctl
            seen += len(msgs) - 1
    assert seen > 0                                                  # queues were not empty: rumours rode along


def test_live_node_bridge_on_the_gpu(hip_abi):
    """Row f-4 under the driver's GPU run: the UDP endpoint answers Ping / IndirectPing for simulated members and hands the
    gossip that came in to the simulation (tests/test_bridge.py has the scenario; here the population is on the MI355X)."""
    from tests.test_bridge import bridge_scenario
    bridge_scenario(hip_abi)


def test_live_node_bridge_for_a_sharded_cluster_on_the_gpu(hip_abi):
    """swimbridge_open_cluster: the same scenario with the population as 4 shards on the MI355X behind one endpoint."""
    from tests.test_bridge import bridge_scenario
    bridge_scenario(hip_abi, shards=4, device="cuda:0")


def test_injected_rumours_match_the_oracle_on_the_gpu(oracle_abi, hip_abi):
    """swimsim_inject_rumor (messages from outside the simulation) at 20 000 members with loss: HIP vs oracle."""
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=20000, seed=8, lossPpm=10000, eventMask=0, suspicionTicks=8)
    a, b = make_pair(oracle_abi, hip_abi, sc, [(3, 17), (4, 1900)])
    a.step(2); b.step(2)
    for k in range(6):
        for s in (a, b):
            for j in range(40):
                s.injectRumor((977 * (k * 40 + j) + 5) % 20000, (1201 * (k * 40 + j) + 40) % 20000, 1 + (j & 1), j % 3)
        a.step(3); b.step(3)
        compare_state(a, b, (0, 5, 19999), (5, 40), False, where="block %d:" % k)
    assert b.counters()["refutes"] > 50


@pytest.mark.parametrize("loss", [100000, 300000])
def test_config5_metrics_on_the_gpu(oracle_abi, hip_abi, loss):
    """BASELINE config 5's two reported numbers -- the false-positive Dead count and the dissemination ticks-to-all of a
    crash (swimsim_coverage, polled every tick) -- and the whole coverage curve: MI355X = oracle, 4 096 members."""
    from tests.test_hostemu_parity import config5_metrics
    a = config5_metrics(oracle_abi, n=4096, loss=loss, ticks=90)
    b = config5_metrics(hip_abi, n=4096, loss=loss, ticks=90)
    assert a == b
    assert a[0] > 0 and a[2][3] == (0, 4095)


@pytest.mark.parametrize("T,gc,loss,n,push", [(2, 0, 0, 3000, 0), (7, 1, 50000, 3000, 0), (40, 1, 10000, 65536, 0),
                                              (2, 0, 0, 3000, 1), (3, 1, 50000, 3000, 1), (40, 1, 10000, 65536, 1)])
def test_periodic_state_pull_on_the_gpu(oracle_abi, hip_abi, T, gc, loss, n, push):
    """pull_ticks = T (the periodic state pull between up members; include/swimsim.h): join_pull_kernel's second kind of
    work item -- one block per puller of the tick --, with crashes, rejoins, join pulls, loss and settling: MI355X = oracle.
    push = 1: push_pull (push_kernel: every puller's host merges the puller's map, concurrent pullers of one host by atomics)."""
    from swim_amd import _abi
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=31 + T, lossPpm=loss, eventMask=0x1F if n <= 4096 else 0,
                   suspicionTicks=6, maxSubjects=min(n, 4096), gcTicks=_abi.GC_AUTO if gc else 0, joinPull=1, pullTicks=T, pushPull=bool(push))
    crashes = [(3 + 2 * k, (37 * k + 11) % n) for k in range(40)]
    faults = [(t + 9 + (m % 13), m, True) for (t, m) in crashes[::2]]
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes, faults)
    _oracle_threads(a)
    run_lockstep(a, b, 120, 6, observers=(0, 11, n - 1, n // 2), members=(0, 11, n - 1), check_events=n <= 4096)
    assert b.counters()["changes"] > 0


@pytest.mark.parametrize("push", [0, 1])
@pytest.mark.parametrize("n,loss,T,gc,join,shards", [(4096, 50000, 5, 1, 1, 4), (65536, 10000, 64, 0, 1, 8), (1500, 150000, 3, 0, 0, 2), (3000, 150000, 3, 0, 0, 2)])
def test_periodic_state_pull_on_sharded_clusters_on_one_gpu(oracle_abi, hip_abi, n, loss, T, gc, join, shards, push):
    """pull_ticks on a cluster of dense shards (several handles on this GPU, the exchange as device-to-device copies): pullers whose
    hosts live on other shards are served in exchange round 0 (pull_send_kernel / begin_kernel's record merge).  MI355X = the
    unsharded oracle."""
    from swim_amd import _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    events = n <= 4096 and not (n >= 3000 and loss >= 100000)      # (3 000 members at 15 % loss fill the event ring within 6 ticks: what is dropped then is implementation-defined)
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=41 + T, lossPpm=loss, eventMask=0x1F if events else 0, suspicionTicks=6,
                   maxSubjects=min(n, 4096), pullTicks=T, gcTicks=_abi.GC_AUTO if gc else 0, joinPull=join, pushPull=bool(push))   # push: push_pull on shards (round 6)
    a, b = Sim.create(oracle_abi, sc), ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    _oracle_threads(a)
    crashes = [(3 + 2 * k, (37 * k + 11) % n) for k in range(40)]
    for s in (a, b):
        for t, m in crashes:
            s.crash(m, t)
        for t, m in crashes[::2]:
            s.scheduleFault(t + 9 + (m % 13), m, True)
    for _ in range(15):
        a.step(6); b.step(6)
        assert a.counters() == b.counters() and a.digest() == b.digest(), "tick %d" % a.tick
        if events:
            assert a.drainEventsRaw() == b.drainEventsRaw()
    assert a.firstDetection() == b.firstDetection()
    a.close(); b.close()


@pytest.mark.parametrize("n,shards,T,loss", [(3000, 2, 3, 100000), (4096, 4, 5, 50000), (65536, 4, 64, 10000)])
def test_messages_from_outside_with_state_pulls_and_settling_on_sharded_clusters_on_one_gpu(oracle_abi, hip_abi, n, shards, T, loss):
    """swimsim_inject_rumor x pull_ticks x settling x 2-4 shards on MI355X (the soak case 703/24 is replayed on the
    emulation, tests/test_shard_hostemu.py): a message from outside opens its subject's view row on EVERY shard
    (swimsim_note_outside_rumor), so a puller of another shard that pulls the subject's own map in that tick sees it.  = the unsharded oracle."""
    from swim_amd import _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    events = n <= 4096
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=91 + T, lossPpm=loss, eventMask=0x1F if events else 0, suspicionTicks=5,
                   maxSubjects=min(n, 4096), gcTicks=_abi.GC_AUTO, joinPull=1, pullTicks=T)
    a, b = Sim.create(oracle_abi, sc), ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    _oracle_threads(a)
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
        assert a.counters() == b.counters() and a.digest() == b.digest(), "tick %d" % a.tick
        if events:
            assert a.drainEventsRaw() == b.drainEventsRaw()
    assert a.firstDetection() == b.firstDetection()
    assert b.counters()["settled"] > 0
    a.close(); b.close()


# ---- bounded member maps (view_cap; swim_sparse.h): BASELINE config 5 at the sizes a dense view cannot reach --------------------
def _config5_case(n, cap, churn_per_mille, ticks, seed=1):
    """30 % message loss x `churn_per_mille`/1000 of the members crash-and-rejoin per 100 ticks."""
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=seed, lossPpm=300000, eventMask=0x18, eventCap=1 << 22, viewCap=cap)
    churn = workloads.hashed_crashes(n, 9, max(1, churn_per_mille * ticks // 100), 1000, 2, ticks - 8) if churn_per_mille else []
    faults = [(t + 6 + (m % 7), m, True) for (t, m) in churn]
    tracked = [(3, n // 3), (5, n // 2 + 1)]            # two crashes that stay down: their dissemination is polled
    return sc, churn + tracked, faults


@pytest.mark.parametrize("n,cap,churn", [(65536, 64, 0), (65536, 64, 1), (65536, 256, 10), (262144, 64, 0), (262144, 128, 1), (262144, 64, 10)])
def test_config5_with_bounded_maps_vs_oracle(oracle_abi, hip_abi, n, cap, churn):
    """BASELINE config 5 -- 30 % loss x churn {0, 0.1, 1} % per 100 ticks -- at 65 536 and 262 144 members, where every member is
    somebody's subject all the time: bounded member maps (one wave per member, the map as a hash table in LDS) against the
    oracle's set-based end of tick.  Digest, every counter, JOIN / REFUTE events record by record, first-detection ticks,
    sampled views and queues, how far two real crashes have spread."""
    ticks = 36
    sc, crashes, faults = _config5_case(n, cap, churn, ticks)
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes, faults)
    _oracle_threads(a)
    run_lockstep(a, b, ticks, 12, observers=(n // 3, n // 2 + 1, 0, n - 1), members=(0, 1, n - 1), check_events=True)
    c = b.counters()
    assert c["evicted"] > 0 and c["direct_failed"] > 0.3 * c["pings"] and c["refutes"] > 0
    if churn:
        assert c["active_members"] < n * ticks


@pytest.mark.parametrize("n,cap,loss,p,S,rm", [(64, 8, 0, 3, 5, 0), (200, 16, 100000, 3, 6, 0), (300, 64, 300000, 3, 8, 0), (150, 4, 300000, 4, 5, 1),
                                                (300, 200, 300000, 3, 8, 0), (120, 130, 200000, 5, 4, 2), (5000, 32, 300000, 10, 6, 0)])
def test_bounded_maps_small_cases_every_tick(oracle_abi, hip_abi, n, cap, loss, p, S, rm):
    """The cases of tests/test_bounded_maps.py on the real thing (lanes run concurrently here: the LDS atomics race for real),
    every tick, every observable, the full event stream; plus the reference's default numToGossip = 10 at 30 % loss."""
    import random
    rng = random.Random(n * 31 + cap)
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=n + cap, lossPpm=loss, eventMask=0x1F, eventCap=1 << 22, suspicionTicks=S,
                   retransmitMult=rm, viewCap=cap)
    faults = []
    for _ in range(6):
        m, t = rng.randrange(n), rng.randrange(1, 12)
        faults.append((t, m, False))
        if rng.random() < 0.6:
            faults.append((t + rng.randrange(1, 12), m, True))
    a, b = make_pair(oracle_abi, hip_abi, sc, [], faults)
    _oracle_threads(a)
    run_lockstep(a, b, 24 if n <= 300 else 10, 1, observers=(0, 1, n - 1), members=(0, 1, n - 1), check_events=True)


def test_bounded_maps_rank_floor_on_the_gpu(oracle_abi):
    """The gfx950 build whose per-tick working set is 64 slots (-DSWIM_SP_PHYS=64): lossy ticks overflow it and take the
    rank-floor retries of swim_sparse.h -- exact, oracle-checked, at a size where the lanes' insertions race."""
    from swim_amd import _lib
    hip = _lib.load_variant("spphys")
    n = 30000
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=12, lossPpm=300000, eventMask=0x18, eventCap=1 << 22, suspicionTicks=6, viewCap=16)
    crashes = workloads.hashed_crashes(n, 5, 1, 100, 2, 12)
    a, b = make_pair(oracle_abi, hip, sc, crashes, [(t + 5, m, True) for (t, m) in crashes[:100]])
    _oracle_threads(a)
    run_lockstep(a, b, 24, 6, observers=(0, 1, n - 1), members=(0, 1, n - 1), check_events=True)


def test_bounded_maps_two_million_members_properties(hip_abi):
    """BASELINE config 5's shard size -- 2 097 152 members on one GPU at 30 % loss -- stepped (no oracle at this size within a
    test's time): size-independent properties.  No map exceeds its capacity, the probe statistics are the loss model's
    (P[direct probe fails] = 1 - 0.7^2 = 0.51, P[probe ends in Suspect] = 0.51 x (1 - 0.7^4)^3 = 0.223), two runs give the
    same digest, and a crash is detected in its first tick by somebody."""
    n, cap = 1 << 21, 64
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=3, lossPpm=300000, eventMask=0x10, viewCap=cap)
    digests = []
    for _ in range(2):
        s = Sim.create(hip_abi, sc)
        s.crash(n // 2, 4)
        s.step(10)
        c = s.counters()
        digests.append((s.digest(), tuple(sorted(c.items()))))
        assert abs(c["direct_failed"] / c["pings"] - 0.51) < 0.002
        assert abs(c["suspects"] / c["pings"] - 0.51 * (1 - 0.7 ** 4) ** 3) < 0.002
        assert c["evicted"] > 0
        for o in (0, 12345, n - 1):
            assert len(s.members(o)) <= cap
        fd = s.firstDetection()
        assert fd[n // 2] == 4
        s.close()
    assert digests[0] == digests[1]


def test_full_event_stream_at_65536_members(oracle_abi, hip_abi):
    """Above 4 096 members the parity tests run with the event ring off and cover the event stream through the running event
    digest only -- which is linear in the key and cannot see a wrong CAUSE or a dropped intermediate record.  Here: 65 536
    members, every cause recorded (an 8 M-record ring, drained every 4 ticks), ~3 M records compared one by one: probes'
    suspicions, the gossip that spreads them, suspicion timers, refutations under 1 % loss, joins."""
    n = 65536
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=17, lossPpm=10000, eventMask=0x1F, eventCap=1 << 23, suspicionTicks=12, maxSubjects=4096)
    crashes = workloads.hashed_crashes(n, 17, 1, 4096, 2, 26)         # ~16 crashes over 24 ticks
    faults = [(t + 20, m, True) for (t, m) in crashes[:6]]
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes, faults)
    _oracle_threads(a)
    total = 0
    for _ in range(14):
        a.step(4); b.step(4)
        ea, eb = a.drainEventsRaw(), b.drainEventsRaw()
        assert len(ea) == len(eb) and ea == eb, "event streams differ at tick %d (%d vs %d records)" % (a.tick, len(ea), len(eb))
        total += len(ea)
        assert a.digest() == b.digest() and a.counters() == b.counters()
    causes = {e[5] for e in ea}
    assert total > 1000000 and b.counters()["events_dropped"] == 0
    assert a.firstDetection() == b.firstDetection()


@pytest.mark.parametrize("n,ncrash,loss", [(1 << 20, 12, 0), (1 << 18, 24, 10000)])
def test_full_event_stream_at_a_million_members(oracle_abi, hip_abi, n, ncrash, loss):
    """Every record of the event stream at 1 048 576 (and 262 144, 1 % loss) members -- VERDICT r5 weak #3: above 4 096 members the
    stream was covered by one test and the linear digest.  A dozen crashes in one tick, every cause recorded: each member's suspicion
    by gossip, its own deadline or the Dead that overtakes it, the probers' own suspicions -- ~25 M records, drained every 2 ticks as
    numpy records and compared field by field (tick, observer, subject, incarnation, state, cause)."""
    import numpy as np
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=23, lossPpm=loss, eventMask=0x1F, eventCap=1 << 24, suspicionTicks=9, maxSubjects=4096)
    crashes = [(2, (977 * k + 5) % n) for k in range(ncrash)]
    a, b = make_pair(oracle_abi, hip_abi, sc, crashes, [(14, crashes[0][1], True)])
    _oracle_threads(a)
    total = 0
    for _ in range(15):
        a.step(2); b.step(2)
        ea, eb = a.drainEventsArray(), b.drainEventsArray()
        assert len(ea) == len(eb), "event streams differ at tick %d (%d vs %d records)" % (a.tick, len(ea), len(eb))
        assert np.array_equal(ea, eb), "event streams differ at tick %d: first at record %d" % (a.tick, int(np.argmax(ea != eb)))
        total += len(ea)
        assert a.digest() == b.digest() and a.counters() == b.counters()
    assert total > 15 * n and b.counters()["events_dropped"] == 0
    assert a.firstDetection() == b.firstDetection()


@pytest.mark.parametrize("n,shards,cap,churn", [(4096, 4, 16, 10), (65536, 8, 64, 10), (262144, 4, 64, 1)])
def test_sharded_cluster_of_bounded_handles_on_one_gpu(oracle_abi, hip_abi, n, shards, cap, churn):
    """BASELINE config 5 is a CLUSTER: 16 M members over 8 GPUs at 30 % loss.  Bounded handles sharded by id range (DESIGN.md 6:
    one all-gather of queue lines + member bytes and one all-to-all-v of 8-byte delivery records per tick), here as several
    handles on this GPU, against the unsharded oracle: digest, every counter, JOIN / REFUTE events, views and queues on
    either side of the shard borders, first-detection ticks."""
    from swim_amd.shard import LocalFabric, ShardedSim
    ticks = 24
    sc, crashes, faults = _config5_case(n, cap, churn, ticks)
    a = Sim.create(oracle_abi, sc)
    _oracle_threads(a)
    b = ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
    for s in (a, b):
        workloads.apply_crashes(s, crashes)
        for (t, m, up) in faults:
            s.scheduleFault(t, m, up)
    per, done = n // shards, 0
    while done < ticks:
        a.step(8); b.step(8); done += 8
        assert a.counters() == b.counters(), "counters differ after %d ticks" % done
        assert a.digest() == b.digest(), "digest differs after %d ticks" % done
        assert a.drainEventsRaw() == b.drainEventsRaw()
        for o in (0, per - 1, per, n - 1):
            assert a.members(o) == b.members(o)
            assert a.readMember(o) == b.readMember(o)
    assert a.firstDetection() == b.firstDetection()
    b.close()


@pytest.mark.parametrize("block", range(2))
def test_random_bounded_configurations_on_the_gpu(oracle_abi, hip_abi, block):
    """The randomised sweep of tests/test_random_sweep.py for bounded member maps on the real thing (sizes up to 20 000, 1-8
    shards on this GPU, capacities 4 ... 256, numToGossip up to 10, loss up to 50 %, churn)."""
    import random
    from swim_amd.shard import LocalFabric, ShardedSim
    from tests.test_random_sweep import _random_bounded_case
    rng = random.Random(9100 + block)
    for _ in range(8):
        sc, shards, faults, what = _random_bounded_case(rng, [8, 64, 300, 1024, 4096, 20000], 200)
        sc.eventCap = 1 << 22
        a = Sim.create(oracle_abi, sc)
        _oracle_threads(a)
        b = Sim.create(hip_abi, sc) if shards == 1 else ShardedSim(hip_abi, sc, LocalFabric(shards), device="cuda:0")
        for (t, m, up) in faults:
            a.scheduleFault(t, m, up); b.scheduleFault(t, m, up)
        for _t in range(rng.choice([3, 5])):
            a.step(5); b.step(5)
            assert a.counters() == b.counters(), ("counters", what)
            assert a.digest() == b.digest(), ("digest", what)
            assert a.drainEventsRaw() == b.drainEventsRaw(), ("events", what)
        assert a.firstDetection() == b.firstDetection(), ("first detection", what)
        a.close(); b.close()
