"""Divergence profile of merge_kernel: the product's kernels in the host emulation with
-DSWIM_PATH_STATS count, for every marked site, how often it runs per lane and per wave (a wave executes a site
max-over-its-lanes times).  usage: python scripts/path_stats.py [members] [ticks measured]   env: LOSS (ppm), GC=1"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads                     # noqa: E402
from tests import hostemu_binding                        # noqa: E402

SITES = {0: "active member", 1: "own line read", 2: "group_put", 3: "group_put scan step", 4: "group_put shift step",
         5: "examine", 6: "accepted", 7: "full row scan", 8: "deadline entry", 9: "failed probe", 10: "fresh batch",
         11: "fresh rumour", 12: "record source", 13: "record line read", 14: "record entry examined", 15: "refute",
         16: "survivor kept", 17: "group size", 18: "line written", 20: "accepted: deadline", 21: "accepted: failed probe",
         22: "accepted: batch slot 0", 23: "accepted: batch slot 1", 24: "accepted: batch slot 2", 25: "accepted: batch slot 3",
         30: "accepted: record"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    abi = hostemu_binding.load_variant("pstat", ["SWIM_PATH_STATS"])
    lib = C.CDLL(os.path.join(hostemu_binding.EMU, "_build", "libswimsim_hostemu_pstat.so"))
    loss = int(os.environ.get("LOSS", 0))
    sc, crashes, _ = workloads.saturated(n, 400, seed=1, t0=0, loss_ppm=loss)
    if os.environ.get("GC"):
        from swim_amd import _abi
        sc.gcTicks = _abi.GC_AUTO
    s = Sim.create(abi, sc)
    for t, m in crashes:
        s.crash(m, t)
    pre = 0
    S = sc.suspicionTicks or 3 * max(1, (n - 1).bit_length())       # the library's default (include/swimsim.h)
    while True:                                            # bench.py's pre-roll: until the queues are full
        c0 = s.counters(); s.step(10); pre += 10; c1 = s.counters()
        d = (c1["payloads"] - c0["payloads"]) / (10.0 * n)
        if (d > 5.9 and pre >= S + 20) or pre >= 300:
            break
    print("pre-roll %d ticks, payloads per member-tick %.2f" % (pre, d))
    nthreads = (n + 255) // 256 * 256
    lib.hostemu_pstat_begin(C.c_uint32(nthreads))
    s.step(ticks)
    out = (C.c_uint64 * (3 * 48))()
    lib.hostemu_pstat_end(out)
    waves = nthreads // 64 * ticks
    print("%-28s %12s %12s %10s %8s" % ("site", "per lane", "per wave", "waves hit", "waste"))
    for k, name in sorted(SITES.items()):
        lanes, wsum, wn = out[3 * k], out[3 * k + 1], out[3 * k + 2]
        if not lanes:
            continue
        # per lane = mean count per member-tick; per wave = mean executions per wave-tick; waste = wave slots / useful
        print("%-28s %12.3f %12.3f %9.1f%% %8.1f" % (name, lanes / (n * ticks), wsum / waves, 100.0 * wn / (nthreads // 64),
                                                   64.0 * wsum / lanes))
    s.close()


if __name__ == "__main__":
    main()
