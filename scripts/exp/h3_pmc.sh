#!/bin/bash
# PMC passes on ONE GEMM shape (tuning aid).  usage (on the GPU box): bash scripts/exp/h3_pmc.sh "512,6400,512,3,200" 22
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-1024,6400,1024,3,200}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/h3_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o g --output-format csv -- python3 $R/scripts/gemm_bench.py $SHAPE > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/g_counter_collection.csv")):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "conv_gemm_h3" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f.split("/")[-2], {k: round(v / n[k]) for k, v in agg.items()})
PY
tail -2 $OUT/p4.log
