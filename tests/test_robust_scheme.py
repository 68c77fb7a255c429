"""The robust (round-robin) target scheme -- SURVEY 8(f) rank 1, the FIXME at src/Core.hs:232 -- as an option
behind the config (include/swimsim.h SWIMSIM_TARGETS_ROBUST).  Oracle properties, and parity of the product's
kernels (host emulation here; tests/test_hip_parity.py on the GPU)."""
import math

import pytest

from swim_amd import Config, Sim, SimConfig
from tests.helpers import make_pair, run_lockstep


@pytest.fixture(scope="module")
def emu_abi():
    from tests import hostemu_binding
    return hostemu_binding.load()


def robust(n, p=3, **kw):
    return SimConfig(cfg=Config(numToGossip=p), nMembers=n, targetScheme=1, **kw)


@pytest.mark.parametrize("n,p", [(65, 3), (128, 3), (10, 4), (2, 1), (97, 10)])
def test_round_visits_everybody_exactly_once(oracle_abi, n, p):
    """Time-bounded completeness: in one round of ceil((N-1)/P) periods every member probes every other
    member exactly once (zero loss, nobody down)."""
    s = Sim.create(oracle_abi, robust(n, p, seed=3))
    rounds = 2
    r = math.ceil((n - 1) / p)
    s.step(rounds * r)
    c = s.counters()
    assert c["pings"] == rounds * n * (n - 1)
    assert c["direct_failed"] == 0 and c["suspects"] == 0
    s.close()


def test_every_crash_is_detected_in_its_first_period(oracle_abi):
    """Every member is probed by numToGossip members every period: at zero loss a crash is detected in the
    period it happens (latency 1, against 1/(1-e^-P) for random targets)."""
    n = 512
    s = Sim.create(oracle_abi, robust(n, 3, seed=5, suspicionTicks=6))
    crashes = [(3 + 2 * k, 17 * k + 5) for k in range(20)]
    for (t, m) in crashes:
        s.crash(m, t)
    s.step(60)
    fd = s.firstDetection()
    assert all(fd[m] == t for (t, m) in crashes)
    assert s.counters()["false_suspects"] == 0
    for o in (0, 1, n - 1):
        assert {mm.memberName for mm in s.members(o)} == {"m%d" % m for (_, m) in crashes}
    s.close()


@pytest.mark.parametrize("n,p,loss,seed", [(2, 1, 0, 1), (65, 3, 0, 2), (128, 3, 100000, 3), (300, 5, 200000, 4), (777, 3, 0, 5)])
def test_robust_parity_hostemu(oracle_abi, emu_abi, n, p, loss, seed):
    sc = robust(n, p, seed=seed, lossPpm=loss, eventMask=0x1F, suspicionTicks=6, maxSubjects=min(n, 1024))
    crashes = [(5, n // 2)] if n > 2 else []
    faults = [(40, n // 2, True)] if n > 2 else []
    a, b = make_pair(oracle_abi, emu_abi, sc, crashes, faults)
    run_lockstep(a, b, 70, 1 if n <= 200 else 7, observers=(0, n - 1, n // 2), members=(0, n - 1, n // 2))


def test_robust_parity_hostemu_fallback_paths(oracle_abi):
    from tests import hostemu_binding
    from swim_amd import workloads
    emu = hostemu_binding.load_variant("win4", ["SWIM_MASK_WIN=4", "SWIM_MASK_SLACK=2"])
    n = 500
    crashes = workloads.hashed_crashes(n, 9, 1, 6, 3, 33)
    sc = robust(n, 3, seed=9, lossPpm=30000, eventMask=0x1F, suspicionTicks=6, maxSubjects=500, inboxCap=1)
    a, b = make_pair(oracle_abi, emu, sc, crashes, [(45, m, True) for (_, m) in crashes[:10]])
    run_lockstep(a, b, 60, 5, observers=(0, 1, n - 1), members=(0, 1, n - 1))


def test_robust_accepted_on_sharded_handles(emu_abi):
    """Round 1 refused the scheme on shards; now the targets are the same and the payloads are pushed
    (tests/test_shard_hostemu.py::test_sharded_robust_target_scheme checks the run against the oracle)."""
    err, sim = Sim.configure(emu_abi, robust(128), shard_index=0, n_shards=2)
    assert err is None and sim is not None
    sim.close()
