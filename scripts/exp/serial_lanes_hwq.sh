#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for q in 6 8; do
  export GPU_MAX_HW_QUEUES=$q
  for n in 3 4 5 6 8; do
  python3 $R/bench.py --steps 90 --warmup 12 --no-extras --cpu-utts 0 --in-flight $n --no-concurrency 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('hwq=$q serial in_flight=$n', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), d.get('in_flight_note'))"
  done
done
