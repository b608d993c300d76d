#!/bin/bash
# dev: tools/pn_eval_probe.py under rocprofv3 for every experiment build text2loc_amd/libt2l_exp_*.so (timing only: their results are wrong).
# The builds it compared in round 5 were the one-wave-per-SIMD SetAbstraction kernel (and its first LDS-stream form) with one piece
# compiled out each (-DT2L_EXP_PN_NOSPLIT / NOX / NOPOOL / NOWAIT / NOBAR / NODMA): those hooks went with that kernel when the
# eight-wave form replaced it (DESIGN 3.6 keeps the numbers); the script itself works for any set of libt2l_exp_*.so.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in $ROOT/text2loc_amd/libt2l.so $ROOT/text2loc_amd/libt2l_exp_*.so; do
  OUT=$ROOT/gpurun_out/pn_exp; rm -rf $OUT; mkdir -p $OUT
  T2L_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/pn_eval_probe.py > $OUT/probe.log 2>&1
  echo "== $(basename $lib)"
  python - <<EOF2
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:8]:
        print(f'{r["Name"][:60]:60s} avg_us {float(r["AverageNs"])/1e3:9.1f}')
EOF2
  rm -rf $OUT/trace
done
