#!/bin/bash
# dev: per-kernel launch counts / durations of the B=64 training step (tools/train_probe.py). On the GPU box: bash tools/train_prof.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/train_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/train_probe.py > $OUT/trace.log 2>&1
cd $ROOT
python - <<EOF2
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:40]:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:8.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} {r["Percentage"]}%')
EOF2
tail -n 3 $OUT/trace.log
rm -rf $OUT/trace
