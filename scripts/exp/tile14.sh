#!/bin/bash
# 64 x 256 tile against 64 x 128 on the 64-output-channel convs of the towers
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3i; mkdir -p $O
TILES=12,14,11 python3 $R/scripts/gemm_bench.py 64,509440,64,9,15920 64,63680,64,9,1990 64,6400,512,1,200 64,128000,64,9,4000 2>&1 | grep "us " | sed 's/maxdiff.*//' | tee $O/tile14.log
