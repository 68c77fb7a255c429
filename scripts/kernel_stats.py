"""Per-kernel launch statistics from a rocprofv3 --kernel-trace --stats CSV directory (kernel names hold commas: a real CSV
parse).  usage: kernel_stats.py <dir> [top]"""
import csv, glob, os, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
if not f:
    sys.exit("no *kernel_stats.csv under " + d)
rows = list(csv.DictReader(open(f[0])))
print("%-48s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for r in rows[:top]:
    name = r["Name"].split("(")[0].replace("void ", "").replace("swim::", "")
    print("%-48s %8s %12.1f %10.2f %10.2f %10.2f %6s" % (name[:48], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                                      float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"][:5]))
