"""Kernel timeline of the tick from a rocprofv3 --kernel-trace CSV: per kernel the mean duration, and the mean gap between the
end of one kernel and the start of the next, over the last `tail` ticks.  usage: trace_gaps.py <kernel_trace.csv> [tail_ticks]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = [r for r in rows if "swim::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0]
    return n.split("::")[-1]
# keep the last `tail` merge ticks
idx = [k for k, r in enumerate(rows) if "merge_kernel" in r["Kernel_Name"]]
start = idx[-tail - 1] + 1 if len(idx) > tail else 0
rows = rows[start:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k, v in dur.items():
    print("%-28s n=%4d mean %.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
for k, v in gap.items():
    print("gap %-44s n=%4d mean %.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
t = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("span %.1f us over %d kernels" % (t, len(rows)))
