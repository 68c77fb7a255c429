#!/bin/bash
# Calibrate FETCH_SIZE / WRITE_SIZE on access patterns of known size (the guide: "calibrate on a known byte
# count in your own access pattern"): scripts/microbench/gather_rate does 6 accesses per thread, 1 Mi threads.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- $R/build/gather_rate > $OUT/f.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- $R/build/gather_rate > $OUT/w.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("f", "w"):
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(per.items()):
            print("%-62s %-11s launches=%3d  min=%12.1f  median=%12.1f KiB" % (k, c, len(v), min(v), sorted(v)[len(v) // 2]))
PY
