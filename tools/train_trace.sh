#!/bin/bash
# dev: the launches of ONE training step in stream order with their durations (tools/train_probe.py). On the GPU box: bash tools/train_trace.sh [bf16]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/train_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/train_probe.py 12 64 1 ${1:-0} > $OUT/trace.log 2>&1
cd $ROOT
python - <<EOF2
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
zs = [i for i, n in enumerate(names) if "zero_kernel" in n]
a, b = zs[-2], zs[-1]  # one whole step: zero_grad .. the next zero_grad
tot = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f'{d:7.1f} us  grid {r["Grid_Size_X"]:>8s} x {r["Grid_Size_Y"]:>5s} x {r["Grid_Size_Z"]:>3s}  {r["Kernel_Name"][:90]}')
print("launches", b - a, "kernel time", round(tot, 1), "us; span", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3, "us")
EOF2
rm -rf $OUT/trace
