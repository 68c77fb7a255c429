"""How the threaded CPU oracle scales on this box (cpu_baseline of bench.py): member-ticks/s of a saturated
1M-member cluster for several thread counts; prints the CPU affinity the process really has.
usage: python scripts/oracle_scaling.py [members]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads
from tests import oracle_binding as ob
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
sc, crashes, _ = workloads.saturated(n, 200, seed=1, t0=0)
s = Sim.create(ob.load(), sc); workloads.apply_crashes(s, crashes)
ob.set_threads(s, len(os.sched_getaffinity(0))); s.step(30)
for th in (1, 8, 32, 64, 128, 256):
    if th > 2 * (os.cpu_count() or 1): break
    ob.set_threads(s, th)
    k = 2 if th == 1 else 6
    t0 = time.perf_counter(); s.step(k); dt = time.perf_counter() - t0
    print("threads %3d: %.2f M member-ticks/s" % (th, n * k / dt / 1e6), flush=True)
