#!/bin/bash
# round 3, experiment 1: what the conv GEMM's k loop is waiting for -- operand traffic knock-outs, occupancy-2 builds of the 256x128 tile, 4 stages
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b; mkdir -p $O
SHAPES="1024,6400,1024,3,200 512,6400,512,3,200 1024,3840,512,9,40 512,19200,512,3,200 128,128000,128,9,4000 512,3840,512,1,40 1024,6400,1216,1,200"
for v in base occ2 bskip askip occ2bskip; do
  export AS_LIB_PATH=$R/artspeech_amd/lib/exp_$v.so
  echo "== $v"
  TILES=22,42 PIPES=13 python3 $R/scripts/gemm_bench.py $SHAPES 2>&1 | grep "us " | sed 's/maxdiff.*//'
done > $O/gemm.log 2>&1
export AS_LIB_PATH=$R/artspeech_amd/lib/exp_base.so
echo "== ns4" >> $O/gemm.log
TILES=22 PIPES=14 python3 $R/scripts/gemm_bench.py $SHAPES 2>&1 | grep "us " | sed 's/maxdiff.*//' >> $O/gemm.log
# in-step: the shipped tile policy against the 256x128 tile at two workgroups per CU
for v in base occ2; do
  export AS_LIB_PATH=$R/artspeech_amd/lib/exp_$v.so
  for use in 0 1; do
    if [ $use = 1 ]; then export AS_GEMM_USE42=1; else unset AS_GEMM_USE42; fi
    python3 $R/bench.py --steps 40 --warmup 10 --no-extras --cpu-utts 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v use42=$use', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), 'gemm_ms', round(d['kernel_classes']['conv_gemm']['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))"
  done
done > $O/instep.log 2>&1
cat $O/gemm.log $O/instep.log
