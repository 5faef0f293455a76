#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPES="1024,6400,1024,3,200 512,19200,512,3,200 1024,2560,512,9,40 128,128000,128,9,4000 512,6400,512,3,200"
for v in full b13; do
  if [ $v = full ]; then unset AS_LIB_PATH; else export AS_LIB_PATH=$R/artspeech_amd/lib/exp_$v.so; fi
  echo "== $v"
  python3 $R/scripts/gemm_bench.py $SHAPES 2>&1 | grep "us " | sed 's/maxdiff.*//'
done
