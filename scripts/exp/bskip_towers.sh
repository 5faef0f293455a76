#!/bin/bash
# upper bound of sharing one staged activation tile across a conv's taps, on the towers' few-channel 3x3 convs: the knock-out build
# stages the activations for the first tap only (results wrong)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3o; mkdir -p $O
SH="64,509440,64,9,15920 128,128000,64,9,4000 128,128000,128,9,4000 256,32000,256,9,1000 128,32000,128,9,1000 1024,6400,1024,3,200 512,6400,512,3,200"
for v in cur exp_bskip; do
  if [ $v = cur ]; then unset AS_LIB_PATH; else export AS_LIB_PATH=$R/artspeech_amd/lib/$v.so; fi
  echo "== $v"; python3 $R/scripts/gemm_bench.py $SH 2>&1 | grep "us " | sed 's/maxdiff.*//'
done | tee $O/bskip.log
