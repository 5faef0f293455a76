#!/bin/bash
# duration of each of phase A's four concurrent branches alone (AS_ONLY_BRANCH), graph-replayed whole step minus the rest
R=${GRAFT_REPO_ROOT:-/root/repo}
for k in -1 0 1 2 3; do
  if [ $k -ge 0 ]; then export AS_ONLY_BRANCH=$k; else unset AS_ONLY_BRANCH; fi
  python3 $R/bench.py --steps 30 --warmup 5 --cpu-utts 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('branch $k', 'ms_per_step', round(d['ms_per_step'],3), d['phase_ms_eager'])
"
done
