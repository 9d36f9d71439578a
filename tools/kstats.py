"""Print a rocprofv3 *_kernel_stats.csv as a readable table:  python tools/kstats.py <csv> [n_rows] [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms ({tot / 1e6 / steps:.2f} ms/step over {steps} steps)")
for r in rows[:n]:
    print(f"{int(r['TotalDurationNs']) / 1e6 / steps:9.2f} ms/step {int(r['Calls']) // steps:>6} calls avg {float(r['AverageNs']) / 1e3:9.1f} us  "
          f"{r['Name'][:120]}")
