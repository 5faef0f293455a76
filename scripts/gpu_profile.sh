#!/bin/bash
# rocprofv3 passes of the benchmark step (run on the GPU box through gpurun).  Outputs under gpurun_out/prof_*.
# Branches are serialised (--no-concurrency) so per-kernel durations are each kernel's own.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT
ARGS="--steps 10 --warmup 2 --no-graph --no-concurrency --cpu-utts 0"
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench --output-format csv -- python3 $R/bench.py $ARGS > $OUT/prof_trace.log 2>&1
PARGS="--steps 2 --warmup 1 --no-graph --no-concurrency --cpu-utts 0"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/prof_pmc_sq -o bench --output-format csv -- python3 $R/bench.py $PARGS > $OUT/prof_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o bench --output-format csv -- python3 $R/bench.py $PARGS > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o bench --output-format csv -- python3 $R/bench.py $PARGS > $OUT/prof_pmc_write.log 2>&1
ls $OUT/prof_trace $OUT/prof_pmc_sq | head -12
