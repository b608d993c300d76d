#!/bin/bash
# dev: HBM bytes of one PointNet++ training step (forward + backward, tools/pn_train_probe.py, 2 iterations) from the FETCH_SIZE / WRITE_SIZE
# PMC counters (separate passes; KiB -> bytes with the gfx950 correction of tools/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE).
# On the GPU box: bash tools/pn_hbm.sh [bf16]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pn_hbm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/pn_train_probe.py 64 ${1:-0} 0 2"
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- $CMD > $OUT/f.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- $CMD > $OUT/w.log 2>&1
cd $ROOT
python - <<EOF2
import csv, glob
tot = {}
n_iter = None
for sub, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    s = 0.0; fps = 0
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "t2l::" not in k: continue
        if r["Counter_Name"] == name: s += float(r["Counter_Value"])
        if "pt_fps_kernel<4>" in k and r["Counter_Name"] == name: fps += 1
    tot[name] = s; n_iter = fps
steps = max(1, n_iter)
gb = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps / 1e9
print(f"mode ${1:-0}: {steps} steps profiled; per step FETCH_SIZE {tot['FETCH_SIZE']/steps/1e6:.2f} M KiB-units, WRITE_SIZE {tot['WRITE_SIZE']/steps/1e6:.2f} M; HBM bytes per step = {gb:.2f} GB (2 x FETCH + WRITE, KiB)")
EOF2
rm -rf $OUT/f $OUT/w
