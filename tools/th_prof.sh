#!/bin/bash
# dev: per-kernel durations (+ optional PMC pass) of tools/text_head_probe.py. Usage on the GPU box: bash tools/th_prof.sh [pmc]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/th_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/text_head_probe.py ${TH_ARGS:-4096 16}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
if [ "$1" == "pmc" ]; then
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc_lds -o p -- $CMD > $OUT/pmc_lds.log 2>&1
fi
cd $ROOT
python - <<EOF2
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:16]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} {r["Percentage"]}%')
for sub in ("pmc_sq", "pmc_lds"):
    f = glob.glob("$OUT/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k] += 1
    for k, d in acc.items():
        if "th_" not in k: continue
        print(k, cnt[k], {c: round(v / cnt[k], 1) for c, v in d.items()})
EOF2
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_lds
