#!/bin/bash
# effective clock (GRBM_GUI_ACTIVE / 8 / duration) and matrix-core busy fraction of the GEMM kernel, full build and knock-outs
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-1024,6400,1024,3,200}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/h3_clock; rm -rf $OUT; mkdir -p $OUT
for v in full nomfma nodma nofrag; do
  if [ $v = full ]; then unset AS_LIB_PATH; else export AS_LIB_PATH=$R/artspeech_amd/lib/exp_$v.so; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/$v -o g --output-format csv -- python3 $R/scripts/gemm_bench.py $SHAPE > $OUT/$v.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for v in ("full","nomfma","nodma","nofrag"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open("$OUT/%s/g_counter_collection.csv" % v)):
        if "conv_gemm_h3" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    dur = []
    for r in csv.DictReader(open("$OUT/%s/g_kernel_trace.csv" % v)):
        if "conv_gemm_h3" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    d = sum(dur) / len(dur)
    a = {k: x / n[k] for k, x in agg.items()}
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    print(v, "us %.1f" % (d / 1e3), "clock GHz %.2f" % (cyc / d), "mfma busy %.2f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)),
          "active %.2f wait_inst %.2f wait %.2f" % tuple(a[k] / a["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")))
PY
