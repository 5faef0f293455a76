#!/bin/bash
# kernel timeline of ONE merged serial chain of a config replayed from its hipGraph: every kernel of the last replay in order with its
# duration and the gap in front of it.  usage: scripts/exp/c2_timeline.sh [C2|C3|C5] [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-C2}; TAG=${2:-c2tl}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
rocprofv3 --kernel-trace -d $O/raw -o t --output-format csv -- python3 $R/bench.py --config $CFG --steps 12 --warmup 3 --no-extras --cpu-utts 0 --in-flight 1 --no-concurrency > $O/run.log 2>&1
f=$(find $O/raw -name '*kernel_trace.csv' | head -1)
python3 - "$f" $O/timeline.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step: from the last ref_features_kernel on
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('ref_features_kernel')]
# eager profiling passes follow the replays: take the replay with the smallest span
best = None
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    span = int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])
    if best is None or span < best[0]:
        best = (span, seg)
span, seg = best
out = open(sys.argv[2], 'w')
t0 = int(seg[0]['Start_Timestamp'])
prev_end = t0
busy = 0
by = collections.Counter(); cnt = collections.Counter(); gaps = 0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][:48]
    gap = s - prev_end
    out.write(f"{(s - t0) / 1e3:9.1f} us  gap {gap / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d}  {name}\n")
    busy += e - s; by[name] += e - s; cnt[name] += 1; gaps += max(gap, 0)
    prev_end = max(prev_end, e)
out.write(f"\nspan {span / 1e3:.1f} us, {len(seg)} kernels, busy {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us\n")
for k, v in by.most_common():
    out.write(f"  {k:50s} n {cnt[k]:4d}  total {v / 1e3:8.1f} us  avg {v / cnt[k] / 1e3:7.2f}\n")
PY
rm -rf $O/raw
tail -45 $O/timeline.txt
grep -o '"ms_per_step": [0-9.]*' $O/run.log | head -2
