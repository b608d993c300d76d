#!/bin/bash
# dev: PMC of encode_cells (both forms) — instruction mix and LDS behaviour
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out/enc_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/enc_only.py <<PY
import numpy as np, torch, sys
from text2loc_amd import synth
from text2loc_amd.engine import Engine
eng = Engine(0)
eng.load_weights({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(0).items()}, class_embed=True, color_embed=True)
cells = synth.make_cells(11259, seed=1)
pc = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}
for two in (1, 0):
    eng.set_option("encoder_two_cells", two)
    for _ in range(3): eng.encode_cells(pc)
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o p -- python /tmp/enc_only.py > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/b -o p -- python /tmp/enc_only.py > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("a", "b"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "encode_cells" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in cs.items()}, "(millions)")
PY
rm -rf $OUT/a $OUT/b
