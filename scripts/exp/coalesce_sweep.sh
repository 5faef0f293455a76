#!/bin/bash
# bench.py's K steps through the coalescing lanes: (batches in flight, batches per call) -> ms per step of 32 utterances
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for fc in "4 2" "6 2" "6 3" "8 4" "8 2" "4 4" "2 2"; do
  set -- $fc
  python3 $R/bench.py --steps 48 --warmup 8 --no-extras --cpu-utts 0 --in-flight $1 --coalesce $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['coalesced']
print('in flight $1, per call $2 x 32 (lanes', c['lanes'], '): coalesced', round(c['ms_per_step'],3), ' batches of 32 on $1 streams', round(d['ms_per_step_lanes_of_32'] or d['ms_per_step'],3), ' max abs', c['max_abs_vs_each_batch_alone'])"
done
done
