import ctypes as C, os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/swim_amd') else os.getcwd())
from swim_amd import Sim, workloads, _abi
abi = _abi.bind(C.CDLL(os.path.abspath(sys.argv[1])), "swimsim_")
N = 1 << 20
sc, crashes, _ = workloads.saturated(N, 450)
s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes)
prev = [0, 0, 0]
for k in range(45):
    s.step(10)
    buf = (C.c_uint64 * 10)(); abi.table_stats(s._h, buf, 10)
    cur = [buf[7], buf[8], buf[9]]
    print("ticks %3d-%3d: members with records %8d, waves %6d, ticks with records %2d, rumour ids %d" % (k * 10, k * 10 + 10, cur[0] - prev[0], cur[1] - prev[1], cur[2] - prev[2], buf[3]))
    prev = cur
