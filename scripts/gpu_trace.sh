#!/bin/bash
# kernel timeline of the graph-replayed benchmark step (run on the GPU box through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT; rm -rf $OUT/trace_graph
rocprofv3 --kernel-trace -d $OUT/trace_graph -o bench --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 > $OUT/trace_graph.log 2>&1
ls -la $OUT/trace_graph | head
