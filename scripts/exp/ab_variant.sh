#!/bin/bash
# A/B of an experiment build (artspeech_amd/lib/$1.so) against the shipped library on one box.  usage: ab_variant.sh NAME [rounds] [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; N=${2:-3}; shift; shift
for i in $(seq 1 $N); do
  for v in cur $V; do
    if [ $v = cur ]; then unset AS_LIB_PATH; else export AS_LIB_PATH=$R/artspeech_amd/lib/$v.so; fi
    python3 $R/bench.py --steps 60 --warmup 10 --no-extras --cpu-utts 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_classes']; print('$v', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), 'gemm', round(k['conv_gemm']['ms_per_step'],3), 'adain', round(k['adain']['ms_per_step'],3), 'hbm', round(d['roofline_hbm']['ms_per_step'],3))"
  done
done
