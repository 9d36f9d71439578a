"""Post-processing of `rocprofv3 --kernel-trace` over tools/gemm_instep_probe.py: library GEMM kernels (Cijk_*) by section --
before the training steps (isolated / sustained probes), inside them, behind them -- with call counts and mean durations, so that
the SAME kernel name can be compared in and out of the step.   python tools/gemm_instep_analyze.py <rocprof dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
first_train = next(i for i, r in enumerate(rows) if "add_ln" in name(r))
last_train = max(i for i, r in enumerate(rows) if "adamw_kernel" in name(r))
sec = lambda i: "before" if i < first_train else ("in-step" if i <= last_train else "behind")
agg = collections.defaultdict(list)
for i, r in enumerate(rows):
    n = name(r)
    if n.startswith("Cijk") or n.startswith("Custom_Cijk") or "gemm_nt4" in n:
        agg[(sec(i), n[:150])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (s, n), ts in sorted(agg.items(), key=lambda kv: (kv[0][0], -sum(kv[1]))):
    if len(ts) >= 20:
        ts_s = sorted(ts)
        print(f"{s:8s} n={len(ts):5d} mean {sum(ts) / len(ts):8.1f} us  median {ts_s[len(ts) // 2]:8.1f}  min {ts_s[0]:8.1f}  {n}")
# gaps: how long after the previous kernel's end does each in-step 1024-class GEMM start, and what ran before it

# ---- the in-step calls of the kernel the r/k/v projections run (the name also serves other shapes): duration histogram per section
for section in ("before", "in-step", "behind"):
    ts = [t for (s_, n_), v in agg.items() if s_ == section and "SK3" in n_ for t in v]
    if ts:
        hist = collections.Counter(int(t // 10) * 10 for t in ts)
        print(f"SK3 {section}: {len(ts)} calls; 10-us buckets -> calls: {sorted(hist.items())}")
