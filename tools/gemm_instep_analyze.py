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

# ---- the in-step calls of the 1024-class forward kernel one by one: duration histogram, and for the slow ones what ran just before / concurrently
tgt = [i for i, r in enumerate(rows) if sec(i) == "in-step" and "SK3" in name(r)]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
import math
hist = collections.Counter(int(dur(rows[i]) // 10) * 10 for i in tgt)
print("SK3 in-step duration histogram (us bucket: calls):", sorted(hist.items()))
def short(n):
    return n.split("(")[0].replace("void rwkv7::", "").replace("(anonymous namespace)::", "")[:60]
slow = [i for i in tgt if dur(rows[i]) > 85]
ctx = collections.Counter()
for i in slow:
    s0, e0 = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    over = [short(name(rows[j])) for j in range(max(0, i - 6), min(len(rows), i + 7)) if j != i and int(rows[j]["Start_Timestamp"]) < e0 and int(rows[j]["End_Timestamp"]) > s0]
    ctx[(short(name(rows[i - 1])), tuple(sorted(set(over))))] += 1
print(f"{len(slow)} of {len(tgt)} in-step SK3 calls take > 85 us; (previous kernel, kernels overlapping in time) -> count:")
for k, v in ctx.most_common(12):
    print("   ", v, k)
fast = [i for i in tgt if dur(rows[i]) <= 85]
ctx = collections.Counter((short(name(rows[i - 1]))) for i in fast)
print("fast ones, previous kernel:", ctx.most_common(6))
if "Stream_Id" in rows[0] or "Queue_Id" in rows[0]:
    key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
    print("queues of slow:", collections.Counter(rows[i][key] for i in slow), "fast:", collections.Counter(rows[i][key] for i in fast))
