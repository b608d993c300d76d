#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/profile.sh <tag>; python tools/pmc_summary.py <tag> gpurun_out/summary_<tag>; rm -rf gpurun_out/prof_<tag>
# (the raw traces exceed the 64 MiB that travel back; copy gpurun_out/summary_<tag>/* into profiles/)
# rocprofv3 of the HEADLINE command (bench.py without its side measurements, so per-kernel averages are those of the
# timed loop) — kernel-trace stats + PMC passes in separate runs, as the MI355X guide prescribes — and one
# kernel-trace + FETCH_SIZE/WRITE_SIZE pass of the side measurements (encoder, streaming scan, loss).
TAG=${1:-r03a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-pipelined"
# (the headline command keeps its clock-ramp steps — 1.6 k launches survive the PMC passes; the side measurements run --quick:
# rocprofv3 --pmc slows every launch ~100x and crashed / hung on the tens of thousands of launches of the full side loops)
SEC="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --quick"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_lds -o p -- $CMD > $OUT/pmc_lds.log 2>&1
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sec_trace -o t -- $SEC > $OUT/sec_trace.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/sec_fetch -o p -- $SEC > $OUT/sec_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/sec_write -o p -- $SEC > $OUT/sec_write.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $OUT/sec_sq -o p -- $SEC > $OUT/sec_sq.log 2>&1
find $OUT -name "*.csv" | wc -l
