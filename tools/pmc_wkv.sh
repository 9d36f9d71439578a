#!/bin/bash
# PMC passes for the WKV7 kernels (separate rocprofv3 runs per counter group, no tracing combined).
# usage (on the GPU box): bash tools/pmc_wkv.sh <outdir> [bench args]
set -e
OUT=${1:-gpurun_out/pmc}; shift || true
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" -d $REPO/$OUT/$name -o $name --output-format csv -- python $REPO/tools/bench_wkv.py --iters 2 $EXTRA > $REPO/$OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_WAVE32_LDS
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k, cs in sorted(agg.items()):
        if "wkv7" not in k: continue
        fo.write(k + "\n")
        for c, vs in sorted(cs.items()):
            fo.write(f"   {c:28s} n={len(vs):3d} mean={sum(vs)/len(vs):.4g}\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
