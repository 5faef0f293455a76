#!/bin/bash
# C2 (one utterance): K-slice policy variants.  AS_GEMM_KSPLIT=1: never split; AS_GEMM_MINKT: minimum iterations per slice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3l; mkdir -p $O
run() { python3 $R/bench.py --config C2 --steps 200 --warmup 20 --no-extras --cpu-utts 0 --in-flight 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_classes']; print('$1', 'ms', round(d['ms_per_step'],3), 'gemm', round(k['conv_gemm']['ms_per_step'],3), k['conv_gemm']['launches_per_step'])"; }
run default
AS_GEMM_KSPLIT=1 run nosplit
AS_GEMM_KSPLIT=2 run split2
AS_GEMM_MINKT=12 run minkt12
AS_GEMM_MINKT=24 run minkt24
AS_GEMM_MINKT=48 run minkt48
