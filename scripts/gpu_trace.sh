#!/bin/bash
# per-launch kernel trace of the eager, serialised C3 step (run on the GPU box through gpurun); scripts/trace_table.py condenses it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT; rm -rf $OUT/trace_eager
rocprofv3 --kernel-trace -d $OUT/trace_eager -o bench --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --no-extras --no-graph --no-concurrency > $OUT/trace_eager.log 2>&1
ls -la $OUT/trace_eager | head
