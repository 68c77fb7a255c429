"""Where the waves of sp_probe_kernel / sp_merge_kernel (bounded member maps) spend their time: libswimsim_sect.so
(-DSWIM_SECTION_CLOCKS).  usage: bounded_sections.py <members> <cap>   env LOSS WARM TICKS K"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Config, Sim, SimConfig, _abi
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "swim_amd", "csrc", "libswimsim_sect.so"))
abi = _abi.bind(lib, "swimsim_")
lib.swimsim_debug_sections.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
n, cap = int(sys.argv[1]), int(sys.argv[2])
sc = SimConfig(cfg=Config(numToGossip=int(os.environ.get("K", 3))), nMembers=n, seed=1, lossPpm=int(os.environ.get("LOSS", 300000)), eventMask=0x10, viewCap=cap)
s = Sim.create(abi, sc)
s.step(int(os.environ.get("WARM", 12)))
out = (C.c_uint64 * 64)()
lib.swimsim_debug_sections(s._h, out)
ticks = int(os.environ.get("TICKS", 10))
s.kernelTimingEnable(True); s.step(ticks); kt = s.kernelTiming()
lib.swimsim_debug_sections(s._h, out)
MERGE = ["inputs (one round of loads)", "map -> hash table", "failed probes + rumours", "what changed, how many", "radix select", "who stays",
         "accounting", "rumours by subject (wave minima)", "queue line + state stores", "map write-back + clear"]
PROBE = ["own byte", "target selection (filter bits, map look-ups by the wave)", "outcomes, proxies, chains, inbox appends", "outputs"]   # sp_probe_lane_kernel (SWIMSIM_SP_PROBE=wave: own byte + map | target selection | outcomes, proxies, chains | inbox appends)
res = {"members": n, "view_cap": cap, "probe_us": kt["probe_ms"] * 1e3 / kt["ticks"], "merge_us": kt["merge_ms"] * 1e3 / kt["ticks"]}
for name, base, labels in (("sp_merge_kernel", 0, MERGE), ("sp_probe_kernel", 32, PROBE)):
    waves = out[base + 15]
    tot = sum(out[base + k] for k in range(len(labels)))
    res[name] = {"waves": waves, "clocks_per_wave": round(tot / max(1, waves)), "clocks_per_member_tick": round(tot / float(n * ticks)),
                 "sections": {lab: {"clocks_per_member_tick": round(out[base + k] / float(n * ticks)), "share": round(out[base + k] / max(1, tot), 3)} for k, lab in enumerate(labels)}}
print(json.dumps(res, indent=1))
