"""How often the explicit-record path runs (libswimsim built with -DSWIM_REC_STATS: scripts/mkvar.sh recstats:"-DSWIM_REC_STATS"): members
with records and new rumour ids per STEP ticks of the saturated 1M-member workload.  usage: rec_stats.py lib.so   env: LOSS, TICKS, STEP, GC"""
import ctypes as C, os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/swim_amd') else os.getcwd())
from swim_amd import Sim, workloads, _abi
abi = _abi.bind(C.CDLL(os.path.abspath(sys.argv[1])), "swimsim_")
N = 1 << 20
LOSS = int(os.environ.get('LOSS', 0)); T = int(os.environ.get('TICKS', 450)); STEP = int(os.environ.get('STEP', 10))
sc, crashes, _ = workloads.saturated(N, T, loss_ppm=LOSS)
if os.environ.get('GC'):
    sc.gcTicks = _abi.GC_AUTO
s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes)
prev = [0, 0, 0]
pids = 0
for k in range(T // STEP):
    s.step(STEP)
    buf = (C.c_uint64 * 10)(); abi.table_stats(s._h, buf, 10)
    cur = [buf[7], buf[8], buf[9]]
    print("ticks %3d-%3d: members with records %8d, waves %6d, new rumour ids %d, view rows live %d" % (k * STEP, k * STEP + STEP, cur[0] - prev[0], cur[1] - prev[1], buf[3] - pids, buf[1]))
    pids = buf[3]
    prev = cur
