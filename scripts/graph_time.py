"""What a HIP graph does to the tick's kernel boundaries (VERDICT r3 item 4's "cheap first step").
The same cluster stepped K ticks per call as stream launches and -- SWIMSIM_GRAPH=1 -- as ONE captured graph (3 K kernel nodes)
whose launch-to-completion time the library reports; both interleaved, saturated and quiescent regimes.  Child processes:
the knob is read once per process.  usage: graph_time.py   env: MEMBERS, K (ticks per call), REPS"""
import ctypes as C, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(os.environ.get("MEMBERS", 1 << 20)); K = int(os.environ.get("K", 20)); REPS = int(os.environ.get("REPS", 10))


def child(regime):
    from swim_amd import Sim, workloads, _abi
    from swim_amd import _lib
    abi = _lib.load()
    mk = workloads.saturated if regime == "saturated" else workloads.quiescent
    sc, crashes, _ = mk(N, 150 + K * REPS + 10)
    s = Sim.create(abi, sc); workloads.apply_crashes(s, crashes)
    s.step(150)
    graph = os.environ.get("SWIMSIM_GRAPH") == "1"
    times = []
    for _ in range(REPS):
        t0 = time.perf_counter(); s.step(K); wall = (time.perf_counter() - t0) * 1e6
        if graph:
            buf = (C.c_uint64 * 10)(); abi.table_stats(s._h, buf, 10); times.append(buf[9] / K)
        else:
            times.append(wall / K)
    times.sort()
    print(json.dumps({"regime": regime, "mode": "one graph per %d ticks" % K if graph else "stream launches", "us_per_tick_median": round(times[len(times) // 2], 2),
                      "us_per_tick_min": round(times[0], 2), "digest": "%016x" % s.digest()}), flush=True)
    s.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1]); sys.exit(0)
    for regime in ("saturated", "quiescent"):
        for g in ("0", "1", "0", "1"):
            env = dict(os.environ, SWIMSIM_GRAPH=g)
            subprocess.call([sys.executable, os.path.abspath(__file__), regime], env=env)
