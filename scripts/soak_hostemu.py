"""Randomised long-run soak of the product's kernels in the host emulation against the oracle (CPU only): random
populations (130-3000) on 1-8 shards, both target schemes, fan-outs, loss rates up to 20 %, settling, join pull, periodic pulls (push-pull on one
handle), strict reference rules, messages from outside, plain ticks with and without begin_kernel, tiny inboxes, hundreds of crashes and
rejoins over 200-800 ticks, on the normal build and the knob-shrunk ones (8- and 10-bit rumour ids, 4-id mask
window).  Every 20 ticks: counters (the dropped-event count aside: implementation-defined once the ring overflows),
state digest, events (while nothing was dropped), first-detection ticks at the end.
usage: soak_hostemu.py <seed> <seconds>      -- prints one line per case; "DIVERGED" names the configuration."""
import sys, random, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Config, Sim, SimConfig, workloads, _abi
from swim_amd.shard import LocalFabric, ShardedSim
from tests import hostemu_binding, oracle_binding
_VARIANT_DEFS = {"": None, "rid10": ["SWIM_RID_BITS=10"], "rid8": ["SWIM_RID_BITS=8"], "win4": ["SWIM_MASK_WIN=4", "SWIM_MASK_SLACK=2"]}
_loaded = {}


def _variant(name):
    """The emulation build `name` (built on first use: a test that replays one case needs one build, not four)."""
    if name not in _loaded:
        _loaded[name] = hostemu_binding.load() if _VARIANT_DEFS[name] is None else hostemu_binding.load_variant(name, _VARIANT_DEFS[name])
    return _loaded[name]


def _oracle():
    if "orc" not in _loaded:
        _loaded["orc"] = oracle_binding.load()
    return _loaded["orc"]


def run_case(seed0, k, log=print):
    """Case k of the soak with seed seed0 (the whole case is a function of the two): returns (ok, what).  Every 20 ticks: counters,
    digest, events; first-detection ticks at the end.  tests/test_shard_hostemu.py replays the cases that once diverged."""
    rng = random.Random(seed0 * 100003 + k)
    rng2 = random.Random(seed0 * 100003 + k + 7777777)    # options added in later rounds draw from here
    vname = rng.choice(list(_VARIANT_DEFS))
    n = rng.choice([130, 300, 700, 1500, 3000])
    shards = rng.choice([1, 1, 2, 3, 4, 8])                      # sharded clusters take every option of the plain handle
    n -= n % shards
    scheme = 1 if rng.random() < 0.25 else 0
    p = rng.choice([1, 3, 3, 5, 10])
    loss = rng.choice([0, 20000, 50000, 100000, 200000])
    gc = rng.random() < 0.6
    jp = rng.random() < 0.5
    S = rng.choice([4, 7, 12])
    pt = rng.choice([0, 0, 2, 3, 9, 40])                          # periodic state pull (on shards: exchange round 0 in every tick)
    pp = bool(pt) and (rng.random() < 0.5 if shards == 1 else rng2.random() < 0.5)   # ... as a push-pull (on shards too since round 6: drawn from a generator of its own, so that the cases of earlier rounds replay as they were)
    strict = (rng.random() < 0.5 if shards == 1 else rng2.random() < 0.5) if (not gc and not jp and not pt) else rng2.random() < 0.25   # the literal suspectOrDeadNode' (round 6: on shards, with settling and state pulls too -- those draws come from rng2)
    fold = rng.choice(["0", "1"])                                # plain ticks with / without begin_kernel
    ticks = rng.choice([200, 400, 800])
    sc = SimConfig(cfg=Config(numToGossip=p), nMembers=n, seed=rng.randrange(1, 1 << 30), lossPpm=loss, eventMask=0x1F,
                   suspicionTicks=S, retransmitMult=rng.choice([1, 2, 3]), maxSubjects=n, gcTicks=_abi.GC_AUTO if gc else 0,
                   joinPull=1 if jp else 0, inboxCap=rng.choice([0, 0, 2]), targetScheme=scheme,
                   pullTicks=pt, pushPull=pp, strictReferenceRules=strict)
    os.environ["SWIMSIM_FOLD_BEGIN"] = fold
    a = Sim.create(_oracle(), sc)
    rm = shards > 1 and rng.random() < 0.5                       # replicated queue masks (read from the environment at create)
    os.environ["SWIMSIM_CLUSTER_STEP"] = "0" if rm else "1"
    rk = rng.choice(["", "0", "1"])                              # explicit records: the handle's own choice / phase in merge_kernel / records_kernel
    if rk: os.environ["SWIMSIM_RECORDS_KERNEL"] = rk
    else: os.environ.pop("SWIMSIM_RECORDS_KERNEL", None)
    inject = rng.random() < 0.4                                  # rumours from outside the simulation (swimsim_inject_rumor; on shards: to the observer's owner)
    b = Sim.create(_variant(vname), sc) if shards == 1 else ShardedSim(_variant(vname), sc, LocalFabric(shards))
    nf = rng.randrange(0, n // 4)
    for _ in range(nf):
        m, t = rng.randrange(n), rng.randrange(1, ticks)
        for s in (a, b): s.scheduleFault(t, m, False)
        if rng.random() < 0.7:
            t2 = t + rng.randrange(1, 150)
            for s in (a, b): s.scheduleFault(t2, m, True)
    what = (vname, n, p, loss, gc, jp, S, ticks, sc.seed, nf, shards, scheme, rm, "rk" + rk, inject, "pull%d" % pt, "push" if pp else "", "strict" if strict else "", "fold" + fold)
    ok = True
    try:
        for _ in range(ticks // 20):
            if inject:
                for _j in range(rng.randrange(0, 12)):
                    o_, s_, st_, inc_ = rng.randrange(n), rng.randrange(n), rng.randrange(3), rng.randrange(3)
                    a.injectRumor(o_, s_, st_, inc_); b.injectRumor(o_, s_, st_, inc_)
            a.step(20); b.step(20)
            ca, cb = a.counters(), b.counters()
            da, db = ca.pop("events_dropped"), cb.pop("events_dropped")
            assert ca == cb, "counters %s" % {k2: (ca[k2], cb[k2]) for k2 in ca if ca[k2] != cb[k2]}
            assert a.digest() == b.digest(), "digest"
            ea, eb = a.drainEventsRaw(), b.drainEventsRaw()
            if da == 0 and db == 0: assert ea == eb, "events"
        assert a.firstDetection() == b.firstDetection(), "fd"
    except AssertionError as e:
        ok = False; log("DIVERGED", e, what, "tick", a.tick, flush=True)
    except Exception as e:
        ok = False; log("ERROR", repr(e)[:200], what, flush=True)
    finally:
        for v in ("SWIMSIM_FOLD_BEGIN", "SWIMSIM_CLUSTER_STEP", "SWIMSIM_RECORDS_KERNEL"): os.environ.pop(v, None)
    log("ok" if ok else "FAIL", what, b.tableStats() if ok and shards == 1 else "", flush=True)
    a.close(); b.close()
    return ok, what


if __name__ == "__main__":
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 1200
    k = 0
    while time.time() < t_end:
        run_case(seed0, k); k += 1
