"""GPU busy fraction and per-class kernel time of the training step from a `rocprofv3 --kernel-trace` run of
`bench.py --no-decode --no-cpu-baseline`:  python tools/step_busy.py <rocprof dir> [steps=5]
A step = the span between two consecutive adamw launches; busy = union of kernel intervals (streams overlap) / span."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
ad = ad[-(nsteps + 1):]
lo, hi = rows[ad[0]][1], rows[ad[-1]][1]
win = [r for r in rows if r[0] >= lo and r[1] <= hi]
span = (hi - lo) / 1e6
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = []
pe = None
for s, e, _ in win:
    if pe is not None and s > pe:
        gaps.append(s - pe)
    pe = e if pe is None else max(pe, e)
print(f"{nsteps} steps: {span / nsteps:.2f} ms per step, GPU busy {busy / 1e6 / nsteps:.2f} ms per step ({100 * busy / (hi - lo):.1f} %), "
      f"{len(win) / nsteps:.0f} launches per step, idle gaps: {len(gaps) / nsteps:.0f} per step, mean {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us, "
      f"total {sum(gaps) / 1e6 / nsteps:.2f} ms per step; gaps > 20 us: {sum(1 for g in gaps if g > 20000) / nsteps:.1f} per step = {sum(g for g in gaps if g > 20000) / 1e6 / nsteps:.2f} ms")
def cls(n):
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"): return "library GEMM"
    if "gemm_nt4" in n or "wgrad_skinny" in n or "gemm_nt_" in n: return "own GEMM"
    if "wkv7c" in n: return "WKV7 chunked"
    if "rwkv7::" in n: return "own row-stream / optimizer"
    return "ATen / other"
agg = collections.defaultdict(float)
for s, e, n in win:
    agg[cls(n)] += (e - s) / 1e6 / nsteps
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"   {k:28s} {v:7.2f} ms per step")
big = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in win:
    k = n.split("(")[0].replace("void ", "").replace("rwkv7::", "").replace("(anonymous namespace)::", "")[:70]
    big[k][0] += 1; big[k][1] += (e - s) / 1e6 / nsteps
for k, (c, v) in sorted(big.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"   {c / nsteps:7.1f} x {v / (c / nsteps) * 1e3:8.1f} us = {v:6.2f} ms  {k}")

# where the large idle gaps sit: (kernel before, kernel after) -> count, total ms per step
short = lambda n: n.split("(")[0].replace("void ", "").replace("rwkv7::", "").replace("(anonymous namespace)::", "")[:48]
gp = collections.defaultdict(lambda: [0, 0.0])
pe, pn = None, None
for s_, e_, n_ in win:
    if pe is not None and s_ - pe > 20000:
        k = (short(pn), short(n_))
        gp[k][0] += 1; gp[k][1] += (s_ - pe) / 1e6 / nsteps
    if pe is None or e_ > pe:
        pe, pn = e_, n_
print("idle gaps > 20 us by (kernel before -> kernel after): count per step, ms per step")
for k, (c, v) in sorted(gp.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"   {c / nsteps:5.1f}  {v:6.3f} ms   {k[0]}  ->  {k[1]}")

# the three largest gaps of the last step, with their neighbourhood (full names, time since the step's first kernel)
last = [r for r in rows if r[0] >= rows[ad[-2]][1] and r[1] <= hi]
t0 = last[0][0]
cand = []
pe = None
for j, (s_, e_, n_) in enumerate(last):
    if pe is not None and s_ - pe > 100000:
        cand.append((s_ - pe, j))
    pe = e_ if pe is None else max(pe, e_)
for g_, j in sorted(cand, reverse=True)[:4]:
    print(f"gap {g_ / 1e3:.0f} us at +{(last[j][0] - t0) / 1e6:.2f} ms of the last step ({(hi - t0) / 1e6:.1f} ms long), launch index {j} of {len(last)}:")
    for q in range(max(0, j - 3), min(len(last), j + 3)):
        print(f"      {'>>' if q == j else '  '} +{(last[q][0] - t0) / 1e6:8.3f} ms  {(last[q][1] - last[q][0]) / 1e3:8.1f} us  {last[q][2][:150]}")
