#!/bin/bash
# branches of a step on side streams (default) against one stream per batch (--no-concurrency), 1-6 batches in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for c in ${MODES:-"--no-concurrency"}; do
  [ "$c" = "-" ] && c=""
  for n in ${NS:-2 3 4 5 6}; do
  python3 $R/bench.py --steps 90 --warmup 12 --no-extras --cpu-utts 0 --in-flight $n $c 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$c] in_flight=$n', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), d.get('in_flight_note'))"
  done
done
done
