#!/bin/bash
# PMC passes for the two late round-6 kernels (lora_down_fwd_kernel, wgrad_mid_kernel): separate rocprofv3 runs per counter group, no tracing.
# usage (on the GPU box): bash tools/pmc_lora.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_lora}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; for t in lora_down_ab wgrad_mid_ab; do rocprofv3 --pmc "$@" -d $REPO/$OUT/${name}_$t -o $name --output-format csv -- python $REPO/tools/$t.py > $REPO/$OUT/${name}_$t.log 2>&1 || echo "pass $name $t failed"; done; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
open(os.path.join(out, "names.txt"), "w").write("\n".join(sorted(agg)))
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k, cs in sorted(agg.items()):
        if not any(t in k for t in ("lora_down_fwd", "wgrad_mid", "MT160x256x64", "MT256x192x32", "combine_fwd")): continue
        fo.write(k + "\n")
        for c, vs in sorted(cs.items()):
            fo.write(f"   {c:28s} n={len(vs):3d} mean={sum(vs)/len(vs):.4g}\n")
        g = cs.get("GRBM_GUI_ACTIVE"); 
        if g and "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_ACTIVE_INST_VALU" in cs:
            ga = sum(g) / len(g) / 8 * 1024
            m = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"]); v = sum(cs["SQ_ACTIVE_INST_VALU"]) / len(cs["SQ_ACTIVE_INST_VALU"])
            fo.write(f"   -> mfma_util {m / ga:.3f}  valu_frac {4 * v / ga:.3f}\n")
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            fe = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]); wr = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            fo.write(f"   -> HBM bytes per launch (2 FETCH + WRITE) KiB: {(2 * fe + wr) * 1024 / 1e6:.1f} MB\n")
        if "SQ_LDS_BANK_CONFLICT" in cs and sum(cs.get("SQ_LDS_IDX_ACTIVE", [0])) > 0:
            fo.write(f"   -> LDS bank-conflict cycles / LDS active cycles {sum(cs['SQ_LDS_BANK_CONFLICT']) / sum(cs['SQ_LDS_IDX_ACTIVE']):.3f}\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
