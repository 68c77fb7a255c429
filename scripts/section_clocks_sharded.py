"""Where the waves of the SHARDED probe_kernel spend their time, next to an unsharded handle of the same size (MI355X,
-DSWIM_SECTION_CLOCKS build; see section_clocks.py): shard 0 of a G-handle cluster of MEMBERS members stepped by swimsim_cluster_step
against one unsharded handle of MEMBERS / G members in the same regime (~1 crash per tick cluster-wide / per handle).
usage: section_clocks_sharded.py [G]     env: WARM, TICKS, MEMBERS"""
import ctypes as C, json, os, sys
import torch
torch.zeros(1, device="cuda:0")          # torch creates its device context before the library does (tests/conftest.py says why)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads, _abi
from swim_amd.shard import LocalFabric, ShardedSim
from section_clocks import PROBE, MERGE

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.abspath(os.path.join(here, "..", "swim_amd", "csrc", "libswimsim_sect.so")))
abi = _abi.bind(lib, "swimsim_")
lib.swimsim_debug_sections.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
warm, ticks, n = int(os.environ.get("WARM", 150)), int(os.environ.get("TICKS", 60)), int(os.environ.get("MEMBERS", 1 << 20))


def sections(handle, kt):
    out = (C.c_uint64 * 64)()
    lib.swimsim_debug_sections(handle, out)
    res = {"probe_us": round(kt["probe_ms"] * 1e3 / max(1, kt["ticks"]), 1), "merge_us": round(kt["merge_ms"] * 1e3 / max(1, kt["ticks"]), 1)}
    for name, base, labels in (("probe_kernel", 32, PROBE), ("merge_kernel", 0, MERGE)):
        waves = out[base + 15]
        res[name] = {"waves": int(waves), "clocks_per_wave": {lab: round(out[base + k] / max(1, waves), 1) for k, lab in enumerate(labels) if out[base + k]}}
    return res


for what in ("sharded", "unsharded"):
    if what == "sharded":
        sc, crashes, _ = workloads.saturated(n, warm + ticks)
        s = ShardedSim(abi, sc, LocalFabric(G), device="cuda:0")
        h0 = s.shards[0].sim._h
    else:
        sc, crashes, _ = workloads.saturated(n // G, warm + ticks)
        s = Sim.create(abi, sc)
        h0 = s._h
    workloads.apply_crashes(s, crashes)
    s.step(warm)
    z = (C.c_uint64 * 64)(); lib.swimsim_debug_sections(h0, z)      # zero the table
    s.kernelTimingEnable(True)
    s.step(ticks)
    kt = s.kernelTiming()
    print(json.dumps({"what": "%s: shard 0 of %d x %d members" % (what, G, n // G) if what == "sharded" else "one unsharded handle of %d members" % (n // G), **sections(h0, kt)}), flush=True)
    s.close()
