#!/bin/bash
# A/B of two builds on ONE box (boxes differ by +-3 %): artspeech_amd/lib/base_r3.so (a build of an earlier commit) against the current
# library, alternating, C3 step (two in flight / one) and optionally C2 / C5.   usage: ab_step.sh [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab; mkdir -p $O
N=${1:-3}; shift
for i in $(seq 1 $N); do
  for v in base cur; do
    if [ $v = base ]; then export AS_LIB_PATH=$R/artspeech_amd/lib/base_r3.so; else unset AS_LIB_PATH; fi
    python3 $R/bench.py --steps 60 --warmup 10 --no-extras --cpu-utts 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_classes']; print('$v', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), 'gemm', round(k['conv_gemm']['ms_per_step'],3), k['conv_gemm']['launches_per_step'], 'hbm', round(d['roofline_hbm']['ms_per_step'],3), 'launches', sum(c['launches_per_step'] for c in k.values()))"
  done
done 2>&1 | tee $O/ab.log
