#!/bin/bash
# the 256 x 128 tile (one workgroup per CU, a third less LDS traffic per product; -DAS_EXPERIMENTS builds) for the single launches with M % 256 == 0,
# inside the step: both arrangements of bench.py, the experiment build with and without AS_GEMM_USE42, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
export AS_LIB_PATH=$R/artspeech_amd/lib/exp_x42.so
for rep in 1 2 3; do
for v in off on; do
  if [ $v = on ]; then export AS_GEMM_USE42=1 AS_GEMM_T42=${T42:-1.2}; else unset AS_GEMM_USE42 AS_GEMM_T42; fi
  python3 $R/bench.py --steps 48 --warmup 8 --no-extras --cpu-utts 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['coalesced']
print('tile42 $v: coalesced 2x32', round(c['ms_per_step'],3), ' 4 x 32', round(d['ms_per_step_lanes_of_32'] or d['ms_per_step'],3), ' one chain', round(d['ms_per_step_one_chain_alone'],3), ' gemm TF/s by events', round(d['roofline']['achieved_by_events'],1), 'max abs', c['max_abs_vs_each_batch_alone'])"
done
done
