#!/bin/bash
# C5 (8 x 1024 tokens) as one merged serial chain, hipGraph replay: the duration predictor's recurrence on the plan's side stream beside the
# encoders' last layers (default) against everything on the one stream (AS_NO_SIDE_LSTM=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export AS_NO_SIDE_LSTM=1; else unset AS_NO_SIDE_LSTM; fi
  python3 $R/bench.py --config C5 --steps 20 --warmup 3 --no-extras --cpu-utts 0 --in-flight 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('no_side=$v', 'chain alone ms', round(d['ms_per_step_one_chain_alone'],3), ' side-streams one at a time', round(d['ms_per_step_one_in_flight'],3), ' 4 in flight', round(d['ms_per_step'],3), d.get('in_flight_note'))"
done
done
