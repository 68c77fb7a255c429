#!/usr/bin/env python
"""profiles/traffic.json from a PMC summary (scripts/pmc_passes.sh -> summary.txt): HBM bytes per launch of the
two tick kernels = FETCH_SIZE + WRITE_SIZE (KiB, separate rocprofv3 --pmc passes, mean of the last 30 launches
of the saturated 1M-member regime).  bench.py copies the dominant kernel's figure into `roofline.traffic` of a
line measured on the SAME workload and regime, and names this file as its source.
usage: make_traffic_json.py <summary.txt> <tag>"""
import ast, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swim_amd import _abi, _lib
text = open(sys.argv[1]).read()
tag = sys.argv[2]
vals = {}
for line in text.splitlines():
    m = re.match(r"\s+(.*?)\s+(\{.*\})\s*$", line)
    if not m:
        continue
    name = "probe_kernel" if "probe_kernel" in m.group(1) else "merge_kernel" if "merge_kernel" in m.group(1) else None
    if name:
        for k, v in ast.literal_eval(m.group(2)).items():
            vals.setdefault(name, {})[k] = v
out = {"regime": "saturated", "members": 1 << 20, "kernels_rev": _abi.ABI_VERSION, "kernels_sha": _lib.kernel_sources_sha(),
       "source": "profiles/%s_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean of the last 30 "
                 "dispatches, scripts/pmc_passes.sh)" % tag,
       "note": "FETCH_SIZE / WRITE_SIZE in KiB. Calibrated on known-size patterns (profiles/r01_pmc_calibration.txt): a scattered "
               "gather of 4-64 B beyond L2 is tallied as one 64-B fetch, a scattered store or atomic as 32 B written; the gfx950 x2 "
               "correction of MI355X_MICROARCH.md applies to wide coalesced streaming reads only and is NOT applied (lower bound "
               "for that minor part)."}
for k in ("probe_kernel", "merge_kernel"):
    out[k + "_hbm_bytes_per_launch"] = int((vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024)
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(out)
