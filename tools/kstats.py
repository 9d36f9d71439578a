"""Top kernels of a rocprofv3 --kernel-trace --stats run:  python tools/kstats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(f)))
print("total kernel ms", sum(int(r["TotalDurationNs"]) for r in rows) / 1e6)
for r in rows[:n]:
    print("%5d %8.1f us %8.2f ms  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6, r["Name"][:110]))
