#!/bin/bash
# dev: dispatch gaps of the steady-state search loop (kernel-trace timestamps): scan end -> re-rank start, re-rank end -> next scan start
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out/gap_probe
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/listwrite_probe.py > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, statistics as st
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
g1, g2, d1, d2 = [], [], [], []
for a, b in zip(rows, rows[1:]):
    if "scanp_kernel" in a[2] and "rerank_kernel" in b[2]:
        g1.append(b[0] - a[1]); d1.append(a[1] - a[0])
    if "rerank_kernel" in a[2] and "scanp_kernel" in b[2]:
        g2.append(b[0] - a[1]); d2.append(a[1] - a[0])
tail = lambda v: v[len(v) // 2:]
print("launch pairs", len(g1), len(g2))
print("scan ns median", st.median(tail(d1)), "rerank ns median", st.median(tail(d2)))
print("gap scan->rerank ns median", st.median(tail(g1)), "gap rerank->scan ns median", st.median(tail(g2)))
PY
rm -rf $OUT/trace
