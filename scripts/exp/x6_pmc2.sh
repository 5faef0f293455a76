#!/bin/bash
# L2 hit/miss and TCP counters on ONE GEMM shape (tuning aid): bash scripts/exp/x6_pmc2.sh "512,6400,512,3,200" 22
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-512,6400,512,3,200}; export TILES=${2:-22}; export IMPLS=x6; export KSPLITS=1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/x6_pmc2; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum FETCH_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o g --output-format csv -- python3 $R/scripts/gemm_bench.py $SHAPE > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/g_counter_collection.csv")):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "conv_gemm_x6" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f.split("/")[-2], {k: round(v / n[k]) for k, v in agg.items()})
PY
tail -3 $OUT/p1.log | cut -c1-200
