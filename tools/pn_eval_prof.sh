#!/bin/bash
# dev: per-kernel durations of the eval-mode PointNet++ backbone (tools/pn_eval_probe.py). On the GPU box: bash tools/pn_eval_prof.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pn_eval
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/pn_eval_probe.py > $OUT/probe.log 2>&1
cd $ROOT
python - <<EOF2
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:14]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} {r["Percentage"]}%')
EOF2
grep -v "^W2026\|^E2026\|amdgpu.ids" $OUT/probe.log | tail -n 6
rm -rf $OUT/trace
