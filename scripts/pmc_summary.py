"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter per dispatch
over the last `tail` dispatches (steady state).  usage: pmc_summary.py <dir> [tail]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; tail = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per = defaultdict(lambda: defaultdict(list))
    for r in rows:
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.relpath(f, d))
    for k, cs in per.items():
        if "swim::" not in k or "fault" in k or "init" in k:
            continue
        print("  ", k.split("(")[0][-40:], {c: round(sum(v[-tail:]) / len(v[-tail:]), 1) for c, v in cs.items()})
