#!/bin/bash
# Dry run of the N > 1 bench line on ONE GPU: `bench.py --gpus N` for N in {2, 4, 8} with the data-path collectives staged
# through the host (T2L_DIST_BACKEND=gloo — RCCL refuses two ranks on one device), every rank an engine context on GPU 0.
# Proves the plumbing of the driver's 8-GPU run before it meets 8 GPUs: the self-launch through torch.distributed.run, the
# rendezvous on 127.0.0.1, ranks_seen == N, parity of the merged ids against the float64 oracle, and that the three N > 1
# side lines (weak_scaling_point, config5_coarse_plus_fine, alt_query_sharded) are present and error-free.
# Usage (on the GPU box): tools/scale_smoke.sh [out_dir]   — exits non-zero on the first failed assertion.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale_smoke}
mkdir -p "$OUT"
# (on a node with >= N GPUs: T2L_DIST_BACKEND=nccl tools/scale_smoke.sh runs the SAME command over RCCL — exactly what the driver launches)
export T2L_DIST_BACKEND=${T2L_DIST_BACKEND:-gloo} HSA_ENABLE_IPC_MODE_LEGACY=0
rc=0
for N in ${SCALE_SMOKE_NS:-2 4 8}; do
  timeout 900 python bench.py --gpus "$N" --steps 20 --warmup 5 --no-cpu-baseline --detail-out "$OUT/detail_g$N.json" > "$OUT/bench_g$N.json" 2> "$OUT/bench_g$N.err"
  st=$?
  if [ $st -ne 0 ]; then echo "scale_smoke: bench.py --gpus $N exited $st"; tail -20 "$OUT/bench_g$N.err"; rc=1; continue; fi
  python - "$OUT/bench_g$N.json" "$N" <<'EOF' || rc=1
import json, sys
path, n = sys.argv[1], int(sys.argv[2])
lines = open(path).read().splitlines()
line = lines[-1]  # THE line is the last thing on stdout, and the only one that parses as JSON
o = json.loads(line)
bad = []
if len(line) >= 4096: bad.append(f"headline is {len(line)} bytes (>= 4096)")
if sum(1 for l in lines if l.startswith("{")) != 1: bad.append("more than one JSON line on stdout")
for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline"):
    if o.get(key) is None: bad.append(f"headline lacks {key}")
if o["parity"]["ids_equal"] is not True: bad.append("headline parity.ids_equal is not true")
for key in ("weak_scaling_point", "config5_coarse_plus_fine", "alt_query_sharded"):
    if key not in o: bad.append(f"headline lacks {key}")
import os
o = json.load(open(os.path.join(os.path.dirname(path), os.path.basename(path).replace("bench_g", "detail_g"))))  # the full record: everything below reads it
if o.get("n_gpus") != n: bad.append(f"n_gpus {o.get('n_gpus')} != {n}")
if o.get("ranks_seen") != n: bad.append(f"ranks_seen {o.get('ranks_seen')} != {n}")
if o["parity"]["ids_equal_float64_oracle"] is not True: bad.append("merged ids differ from the float64 oracle")
for key in ("weak_scaling_point", "config5_coarse_plus_fine", "alt_query_sharded"):
    if key not in o: bad.append(f"{key} missing")
    elif "error" in o[key]: bad.append(f"{key}: {o[key]['error']}")
if not bad:
    if o["alt_query_sharded"].get("ids_equal_row_sharded") is not True: bad.append("query-sharded ids differ from row-sharded")
    if o["weak_scaling_point"].get("own_rows_in_merged_topk_are_in_local_topk") is not True: bad.append("weak point inconsistent")
    if o["config5_coarse_plus_fine"].get("offsets_finite") is not True: bad.append("config 5 offsets not finite")
print(f"scale_smoke N={n}: value {o['value']:.3e} {o['unit']}, ms_per_step {o['ms_per_step']:.4f}, "
      f"weak {o.get('weak_scaling_point', {}).get('ms_per_step')}, cfg5 {o.get('config5_coarse_plus_fine', {}).get('ms_per_step')}, "
      f"alt {o.get('alt_query_sharded', {}).get('ms_per_step')} -> {'OK' if not bad else 'FAIL: ' + '; '.join(bad)}")
sys.exit(1 if bad else 0)
EOF
done
exit $rc
