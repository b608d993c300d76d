#!/bin/bash
# dev: PMC counters of the PointNet++ training kernels (two passes: SQ activity, LDS). On the GPU box: bash tools/pn_pmc.sh [bf16] [v1]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pn_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/pn_train_probe.py 64 ${1:-0} ${2:-0} 2"
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/lds -o p -- $CMD > $OUT/lds.log 2>&1
cd $ROOT
python - <<EOF2
import csv, glob, collections
for sub in ("sq", "lds"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: print("no csv for", sub); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], )
        if key not in seen: seen.add(key); calls[k] += 1
    print("==", sub)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", kv[1].get("SQ_ACTIVE_INST_LDS", 0))):
        if "rows2" in k or "tn2" in k or "gemm" in k:
            print(k, "calls", calls[k], {n: f"{x/calls[k]:.3g}" for n, x in v.items()})
EOF2
rm -rf $OUT/sq $OUT/lds
