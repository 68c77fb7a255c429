"""The product's kernels in the host emulation under AddressSanitizer (CPU only): every "device" allocation is made at its exact
size (-DHOSTEMU_EXACT_ALLOC), so a load or store one element past a buffer is caught.  Randomised small clusters -- 1-4 shards through
both forms of the exchange, loss, settling, state pulls, crashes and rejoins, messages from outside -- are stepped and read through the
whole C ABI; `bench` runs bench.py's own flows (saturated, --gpus 2 in one process, quiescent, lossy, bounded maps) the same way.
Found nothing in round 5 (120 configurations + the five bench flows); kept as a tool: a GPU memory access fault on a box cannot be
told from a faulty box without it.
usage: asan_hostemu.py build | run <seed> <cases> | bench       (run / bench re-exec themselves with libasan preloaded)"""
import ctypes as C, glob, io, json, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tests", "hostemu", "_build", "libswimsim_hostemu_asan.so")
CSRC = os.path.join(ROOT, "swim_amd", "csrc")


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", os.path.join(ROOT, "tests", "hostemu"), "-Wno-unused-function",
                           "-Wno-attributes", "-Wl,-Bsymbolic", "-fsanitize=address", "-fno-omit-frame-pointer", "-DHOSTEMU_EXACT_ALLOC", "-o", LIB,
                           os.path.join(CSRC, "swimsim.hip"), os.path.join(CSRC, "swim_wire.cpp"), os.path.join(CSRC, "swim_bridge.cpp")])


def reexec():
    asan = sorted(glob.glob("/usr/lib/gcc/x86_64-linux-gnu/*/libasan.so"))[-1]
    if os.environ.get("LD_PRELOAD", "") != asan:
        env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1")
        os.execve(sys.executable, [sys.executable] + sys.argv, env)


def load():
    from swim_amd import _abi
    if not os.path.exists(LIB):
        build()
    return _abi.bind(C.CDLL(LIB), "swimsim_")


def run(seed, cases):
    from swim_amd import Sim, SimConfig, Config, _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    lib = load()
    rng = random.Random(seed)
    for case in range(cases):
        n = rng.choice([64, 130, 300, 700, 1024])
        shards = rng.choice([1, 1, 2, 4])
        n -= n % shards
        sc = SimConfig(cfg=Config(numToGossip=rng.choice([1, 3, 5])), nMembers=n, seed=rng.randrange(1, 1 << 30), lossPpm=rng.choice([0, 0, 20000, 200000]),
                       eventMask=0x1F, suspicionTicks=rng.choice([4, 7]), retransmitMult=rng.choice([1, 3]), maxSubjects=min(n, 1024),
                       gcTicks=_abi.GC_AUTO if rng.random() < 0.5 else 0, joinPull=int(rng.random() < 0.5), pullTicks=rng.choice([0, 0, 3, 9]), inboxCap=rng.choice([0, 2]))
        sc.pushPull = bool(sc.pullTicks) and rng.random() < 0.5                 # (round 6: on shards too)
        sc.strictReferenceRules = rng.random() < 0.25                         # (round 6: on shards, with settling and state pulls)
        os.environ["SWIMSIM_CLUSTER_STEP"] = rng.choice(["0", "1"])
        s = Sim.create(lib, sc) if shards == 1 else ShardedSim(lib, sc, LocalFabric(shards))
        for _ in range(n // 6):
            m, t = rng.randrange(n), rng.randrange(0, 60)
            s.scheduleFault(t, m, False)
            if rng.random() < 0.6:
                s.scheduleFault(t + rng.randrange(1, 30), m, True)
        for _blk in range(12):
            if rng.random() < 0.3:
                s.injectRumor(rng.randrange(n), rng.randrange(n), rng.randrange(3), rng.randrange(2))
            s.step(8)
            s.digest(); s.counters(); s.members(rng.randrange(n)); s.readMember(rng.randrange(n)); s.coverage(rng.randrange(n), 2, 0)
        s.drainEventsRaw(); s.firstDetection()
        print("case", case, "members", n, "shards", shards, "clean", flush=True)
        s.close()


def bench_flows():
    from contextlib import redirect_stdout
    import bench
    lib = load()
    bench.PREROLL_MAX = 60; bench.SATURATED_D = 0.0
    os.environ["SWIM_BENCH_SHARE_GPU"] = "1"
    for argv in (["--steps", "20", "--warmup", "5", "--members", "4096"], ["--gpus", "2", "--steps", "10", "--warmup", "3", "--members", "2048"],
                 ["--steps", "10", "--warmup", "3", "--members", "4096", "--regime", "quiescent", "--no-cpu-baseline"],
                 ["--steps", "10", "--warmup", "3", "--members", "4096", "--loss-ppm", "10000", "--gc"],
                 ["--steps", "8", "--warmup", "3", "--members", "4096", "--loss-ppm", "300000", "--view-cap", "64"]):
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main(argv, abi=lib)
        d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][0])
        print("bench.py", " ".join(argv), "clean; verified_vs_oracle:", d.get("verified_vs_oracle"), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "build"
    if what == "build":
        build()
    else:
        reexec()
        run(int(sys.argv[2]), int(sys.argv[3])) if what == "run" else bench_flows()
