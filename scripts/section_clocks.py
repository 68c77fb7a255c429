"""Where the waves of probe_kernel / merge_kernel spend their time (MI355X): the -DSWIM_SECTION_CLOCKS build adds,
per wave, the shader clocks between marks in the kernels to a table (swim_kernels.h SECT).  Time waiting for a
load is charged to the section that first USES the value.  usage: section_clocks.py [libswimsim_sect.so]
env: WARM, TICKS, MEMBERS, SCHEME=robust, GC=1, LOSS (ppm), CPT (crashes per tick), MAXSUBJ"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swim_amd import Sim, workloads, _abi                      # noqa: E402

MERGE = ["inputs", "known-ring + own line + first deadline cells", "failed probes", "delivered rumours (masks)",
         "todo list (records' survivors) + refute", "queue rebuild", "state stores + counters", "wave sync", "line store", "counter flush",
         "deadlines", "records: counts, region, rings", "records: source word + line wait", "records: ring filter + survivor stores", "records: final stores (+ flag wait)"]
RECORDS = ["counts, rings -> LDS, prefix", "pair -> member, source words, lines", "ring filter (LDS atomics)", "chunk barrier + layout",
           "write-out", "serial pass + final stores"]
PROBE = ["target selection", "outcomes + pk gathers", "ping pushes", "acks", "indirect probes", "outputs + counters", "counter flush"]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "swim_amd", "csrc", "libswimsim_sect.so")
    warm, ticks, n = int(os.environ.get("WARM", 150)), int(os.environ.get("TICKS", 100)), int(os.environ.get("MEMBERS", 1 << 20))
    lib = C.CDLL(os.path.abspath(path))
    abi = _abi.bind(lib, "swimsim_")
    lib.swimsim_debug_sections.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    sc, crashes, _ = workloads.saturated(n, warm + ticks, loss_ppm=int(os.environ.get("LOSS", 0)), crashes_per_tick=float(os.environ.get("CPT", 1.0)))
    if os.environ.get("MAXSUBJ"):
        sc.maxSubjects = int(os.environ["MAXSUBJ"])
    sc.targetScheme = 1 if os.environ.get("SCHEME") == "robust" else 0
    if os.environ.get("GC"):
        sc.gcTicks = _abi.GC_AUTO
    churn = int(os.environ.get("CHURN", 0))        # per mille of the members crash and come back (50 ticks later) per 100 ticks
    if churn:
        from swim_amd import Config, SimConfig
        sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=9, maxSubjects=12000, gcTicks=_abi.GC_AUTO, eventMask=1)
        crashes = []
    s = Sim.create(abi, sc)
    workloads.apply_crashes(s, crashes)
    if churn:
        T = warm + ticks + 60
        for (t, m) in workloads.hashed_crashes(n, 9, churn * T // 100, 1000, 5, T - 60):
            s.crash(m, t); s.scheduleFault(t + 50, m, True)
    s.step(warm)
    out = (C.c_uint64 * 64)()
    lib.swimsim_debug_sections(s._h, out)   # zero the table
    s.kernelTimingEnable(True)
    s.step(ticks)
    kt = s.kernelTiming()
    lib.swimsim_debug_sections(s._h, out)
    res = {"members": n, "ticks": ticks, "probe_us": kt["probe_ms"] * 1e3 / kt["ticks"], "merge_us": kt["merge_ms"] * 1e3 / kt["ticks"]}
    for name, base, labels in (("merge_kernel", 0, MERGE), ("probe_kernel", 32, PROBE), ("records_kernel", 48, RECORDS)):
        waves = out[base + 15]
        tot = sum(out[base + k] for k in range(len(labels)))
        rows = {lab: {"clocks_per_wave": round(out[base + k] / max(1, waves), 1), "share": round(out[base + k] / max(1, tot), 4)}
                for k, lab in enumerate(labels)}
        res[name] = {"waves": waves, "clocks_per_wave": round(tot / max(1, waves), 1), "sections": rows}
    res["events_per_tick"] = {"full row scans (lanes)": out[16] / ticks, "deadline chain cells followed": out[17] / ticks,
                              "explicit record lines read": out[18] / ticks, "deadline pool exhausted (lanes)": out[19] / ticks,
                              "state rule calls (lanes)": out[20] / ticks, "accepted changes (lanes)": out[21] / ticks,
                              "deadline cells spilled (lanes)": out[22] / ticks
                              }
    print(json.dumps(res, indent=1))
    s.close()


if __name__ == "__main__":
    main()
