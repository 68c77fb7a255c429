"""Synthetic workloads = the configurations of BASELINE.json, as fault
schedules over a SimConfig.  Pure host-side integer arithmetic; used by bench.py and tests."""
from .types import Config, SimConfig

_M32 = 0xFFFFFFFF


def _mix32(x):
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def hashed_crashes(n_members, seed, percent_num, percent_den, t0, t1):
    """Members with hash(seed,id) % den < num crash at a hashed tick in [t0, t1).
    Returns [(tick, member)] sorted by (tick, member).  Vectorised with numpy."""
    import numpy as np
    ids = np.arange(n_members, dtype=np.uint64)

    def mix(x):
        x = x & _M32
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7FEB352D)) & _M32
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846CA68B)) & _M32
        x ^= x >> np.uint64(16)
        return x

    h1 = mix(ids ^ np.uint64(_mix32(seed * 2 + 1)))
    sel = (h1 % np.uint64(percent_den)) < np.uint64(percent_num)
    h2 = mix(h1 + np.uint64(0x9E3779B9))
    ticks = np.uint64(t0) + (h2 % np.uint64(max(1, t1 - t0)))
    members = ids[sel].astype(np.int64)
    ticks = ticks[sel].astype(np.int64)
    order = np.lexsort((members, ticks))
    return [(int(ticks[k]), int(members[k])) for k in order]


def config1(seed=1):
    """128 members, k=3, member 64 crashes at tick 10, 200 ticks (reference-sized case)."""
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=128, seed=seed, eventMask=0x1F)
    return sc, [(10, 64)], 200


def config2(seed=1):
    """65 536 members, k=3, 1 % hashed crashes in ticks [10,110), 400 ticks."""
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=65536, seed=seed, maxSubjects=1024)
    return sc, hashed_crashes(65536, seed, 1, 100, 10, 110), 400


def config3(seed=1, crash_per_mille=1, t0=10, t1=1010, max_subjects=2048):
    """1 048 576 members, k=3 on one MI355X: `crash_per_mille`/1000 of the members crash at
    hashed ticks in [t0,t1) (dissemination-loaded regime); crash_per_mille=0 -> one crash
    (quiescent regime)."""
    n = 1 << 20
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=seed, maxSubjects=max_subjects)
    if crash_per_mille == 0:
        return sc, [(t0, n // 2)], t1
    return sc, hashed_crashes(n, seed, crash_per_mille, 1000, t0, t1), t1


def apply_crashes(sim, crashes):
    for (tick, member) in crashes:
        sim.crash(member, tick)


def saturated(n_members, total_ticks, seed=1, crashes_per_tick=1.0, t0=10, loss_ppm=0, num_to_gossip=3):
    """Dissemination-saturated regime: about `crashes_per_tick` members crash per tick from t0 on,
    so every message carries a full piggyback payload (regime (s)).  With message loss the
    false suspicions add subjects of their own: the subject table is sized for them too."""
    span = max(1, total_ticks - t0)
    want = max(1, int(round(span * crashes_per_tick)))
    den = 1 << 20
    num = max(1, int(round(want * den / n_members)))
    loss = loss_ppm / 1e6
    p_false = (1.0 - (1.0 - loss) ** 2) * (1.0 - (1.0 - loss) ** 4) ** num_to_gossip    # per probe
    false_subjects = int(min(n_members, p_false * num_to_gossip * n_members * total_ticks))
    sc = SimConfig(cfg=Config(numToGossip=num_to_gossip), nMembers=n_members, seed=seed, lossPpm=loss_ppm,
                   maxSubjects=min(n_members, 60000, max(256, 2 * want + 64 + 2 * false_subjects)))
    return sc, hashed_crashes(n_members, seed, num, den, t0, t0 + span), total_ticks


def quiescent(n_members, total_ticks, seed=1):
    """One crash early on; afterwards only probes and empty payloads (regime (q))."""
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n_members, seed=seed, maxSubjects=64)
    return sc, [(2, n_members // 2)], total_ticks
