#!/bin/bash
# what a class of launches costs the STEP at the margin: a -DAS_EXPERIMENTS build of model.hip runs the launches whose call text matches
# AS_EXP_DUP twice (same arguments, same results) -- unlike the knock-out (knockout_step.sh) the data every kernel sees stays the same
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3q; mkdir -p $O
export AS_LIB_PATH=$R/artspeech_amd/lib/exp_skip.so
for dup in none adain avgpool,dwconv,stem_pool layernorm bilstm attention conv_gemm; do
  if [ $dup = none ]; then unset AS_EXP_DUP; else export AS_EXP_DUP=$dup; fi
  python3 $R/bench.py --no-extras --cpu-utts 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_classes']; print('dup=$dup', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), 'class ms', {c:round(v['ms_per_step'],2) for c,v in k.items()})"
done 2>&1 | tee $O/dup.log
