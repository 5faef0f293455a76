#!/bin/bash
# what the non-GEMM launches cost the STEP (not their own serial time): a -DAS_EXPERIMENTS build of model.hip drops the launches whose
# call text matches AS_EXP_SKIP (results are wrong, durations are forced, so the rest of the step is unchanged).
#   build:  hipcc ... -DAS_EXPERIMENTS -c artspeech_amd/csrc/model.hip -o /tmp/model_exp.o ; link with the shipped objects -> lib/exp_skip.so
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g; mkdir -p $O
export AS_LIB_PATH=$R/artspeech_amd/lib/exp_skip.so
for skip in none adain avgpool,dwconv,im2col layernorm adain,avgpool,dwconv,im2col,layernorm,crop,rows_to,mean_pool,linear_rows,project_cols,split_f16 bilstm attention; do
  if [ $skip = none ]; then unset AS_EXP_SKIP; else export AS_EXP_SKIP=$skip; fi
  python3 $R/bench.py --steps 60 --warmup 10 --no-extras --cpu-utts 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('skip=$skip', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), d.get('in_flight_note'))"
done 2>&1 | tee $O/knockout.log
