#!/bin/bash
# kernel timeline of the two-in-flight graph replay: who overlaps whom (analysed by scripts/exp/timeline.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3h; mkdir -p $O
for n in 1 2; do
rocprofv3 --kernel-trace -d $O/tl$n -o t --output-format csv -- python3 $R/bench.py --steps 12 --warmup 3 --no-extras --cpu-utts 0 --in-flight $n > $O/tl$n.log 2>&1
f=$(find $O/tl$n -name '*kernel_trace.csv' | head -1)
python3 - "$f" $O/tl$n.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
with open(sys.argv[2],'w') as f:
    for r in rows:
        f.write(f"{int(r['Start_Timestamp'])-t0},{int(r['End_Timestamp'])-t0},{r['Queue_Id']},{r.get('Stream_Id','')},{r['Kernel_Name'][:60].replace(',',';')},{r['Grid_Size_X']},{r['Workgroup_Size_X']}\n")
PY
rm -rf $O/tl$n
done
ls -la $O
