#!/bin/bash
# kernel timelines of arrangements "k x c" (32 k utterances per as_forward_test call, c chains in flight): who overlaps whom
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/tlw; mkdir -p $O
for arr in ${ARRS:-1x4 4x2}; do
NATIVE=${NATIVE:-} ARR=$arr REPS=6 rocprofv3 --kernel-trace -d $O/raw_$arr -o t --output-format csv -- python3 $R/scripts/exp/wide_batch.py > $O/$arr.log 2>&1
f=$(find $O/raw_$arr -name '*kernel_trace.csv' | head -1)
python3 - "$f" $O/tl_$arr.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
with open(sys.argv[2],'w') as f:
    for r in rows:
        f.write(f"{int(r['Start_Timestamp'])-t0},{int(r['End_Timestamp'])-t0},{r['Queue_Id']},{r.get('Stream_Id','')},{r['Kernel_Name'][:60].replace(',',';')},{r['Grid_Size_X']},{r['Workgroup_Size_X']}\n")
print("kernels", len(rows), "span ms", (int(rows[-1]['End_Timestamp'])-t0)/1e6)
PY
rm -rf $O/raw_$arr
# the last 40 % of the trace is the timed replays: analyse a window there
python3 - $O/tl_$arr.csv <<'PY'
import sys,subprocess
rows=[l.split(',') for l in open(sys.argv[1])]
end=int(rows[-1][1])/1e6
a=end-30.0 if end>60 else end*0.7
print(subprocess.run([sys.executable, __import__('os').environ.get('GRAFT_REPO_ROOT','/root/repo')+'/scripts/exp/timeline.py', sys.argv[1], str(a), str(a+20.0)],capture_output=True,text=True).stdout)
PY
tail -2 $O/$arr.log
done
