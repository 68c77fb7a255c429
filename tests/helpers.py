"""Shared helpers for the parity tests: run the same workload on two backends and compare
every observable the C ABI offers."""
from swim_amd import Sim
from swim_amd.workloads import apply_crashes


def make_pair(oracle_abi, hip_abi, sim_config, crashes=(), faults=()):
    a = Sim.create(oracle_abi, sim_config)
    b = Sim.create(hip_abi, sim_config)
    for s in (a, b):
        apply_crashes(s, crashes)
        for (tick, member, up) in faults:
            s.scheduleFault(tick, member, up)
    return a, b


def compare_state(a, b, observers=(), members=(), check_events=True, where=""):
    assert a.tick == b.tick, where
    ca, cb = a.counters(), b.counters()
    assert ca == cb, "%s counters differ: %r vs %r" % (where, ca, cb)
    assert a.digest() == b.digest(), "%s state digest differs at tick %d" % (where, a.tick)
    if check_events:
        ea, eb = a.drainEventsRaw(), b.drainEventsRaw()
        assert ea == eb, "%s events differ (%d vs %d)" % (where, len(ea), len(eb))
    for o in observers:                              # how far the rumours about o have got (swimsim_coverage)
        for (st, inc) in ((1, 0), (2, 0), (0, 1)):
            assert a.coverage(o, st, inc) == b.coverage(o, st, inc), "%s coverage of %d (%d@%d) differs" % (where, o, st, inc)
    for o in observers:
        assert a.members(o) == b.members(o), "%s view of %d differs" % (where, o)
    for m in members:
        assert a.readMember(m) == b.readMember(m), "%s member %d differs" % (where, m)


def run_lockstep(a, b, total_ticks, chunk, observers=(), members=(), check_events=True):
    done = 0
    while done < total_ticks:
        n = min(chunk, total_ticks - done)
        a.step(n)
        b.step(n)
        done += n
        compare_state(a, b, observers, members, check_events, where="after %d ticks:" % done)
    assert a.firstDetection() == b.firstDetection()
