#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two batches in flight use 2 x (1 + 3) streams
R=${GRAFT_REPO_ROOT:-/root/repo}
for q in ${QS:-default 1 2 3 4 5 6}; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for n in ${NS:-2}; do
  python3 $R/bench.py --steps 60 --warmup 10 --no-extras --cpu-utts 0 --in-flight $n 2>$R/gpurun_out/hwq_err.log | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('hwq=$q in_flight=$n', 'ms', round(d['ms_per_step'],3), 'one', round(d['ms_per_step_one_in_flight'],3), d.get('in_flight_note'))
except Exception as e:
    print('hwq=$q in_flight=$n failed', open('$R/gpurun_out/hwq_err.log').read()[-300:])"
  done
done
