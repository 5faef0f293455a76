#!/bin/bash
# tile choice on the mid-size shapes whose grid is about one round of the chip
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3n; mkdir -p $O
TILES=,22,21,12,11 KSPLITS=,2 python3 $R/scripts/gemm_bench.py 512,3840,512,5,40 256,16000,256,9,500 512,2080,512,9,65 128,16000,128,9,500 256,4000,256,9,125 512,1280,512,3,40 512,3200,512,3,100 2>&1 | grep "us " | sed 's/maxdiff.*//' | tee $O/tiles.log
