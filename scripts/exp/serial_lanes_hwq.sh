#!/bin/bash
# single-stream batches in flight (bench.py's default arrangement) against the number of hardware queues HIP may use
R=${GRAFT_REPO_ROOT:-/root/repo}
for q in ${QS:-default 5 6 8}; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for n in ${NS:-4 5 6 8}; do
  python3 $R/bench.py --no-extras --cpu-utts 0 --in-flight $n 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('hwq=$q chains=$n', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), d.get('in_flight_note'))"
  done
done
