#!/bin/bash
# dev: per-kernel launch counts / durations of the PointNet++ training-mode forward + backward (tools/pn_train_probe.py).
# On the GPU box: bash tools/pn_prof.sh [B] [bf16] [order]   (order = 1: also the last iteration's launches in stream order)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pn_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/pn_train_probe.py ${1:-64} ${2:-0} ${4:-0} > $OUT/trace.log 2>&1
cd $ROOT
python - <<EOF2
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:22]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:8.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} {r["Percentage"]}%')
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)
if f and "${3:-0}" == "1":
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    last = max(i for i, n in enumerate(names) if "pt_fps_kernel<4>" in n)  # the last iteration starts at its first FPS launch
    for r in rows[last:]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if d > 20: print(f'{d:9.1f} us  grid {r["Grid_Size_X"]:>10s}  {r["Kernel_Name"][:100]}')
EOF2
grep -v "^W2026\|^E2026\|amdgpu.ids" $OUT/trace.log | tail -n 4
rm -rf $OUT/trace
