#!/bin/bash
# Is the conv GEMM's FETCH_SIZE "one weight-image fetch per XCD L2 + the activations once"?  Three shapes that move the two terms apart.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch_model -o gb --output-format csv -- python3 $R/scripts/gemm_bench.py 1024,6400,1024,3,200 1024,12800,1024,3,200 512,6400,1024,3,200 1024,6400,256,3,200 > $OUT/fetch_model.log 2>&1
python3 - <<PY
import csv, collections
a = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/prof_fetch_model/gb_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("void conv_gemm_h3"):
        a[(r["Grid_Size"], r["Kernel_Name"][:40])].append(float(r["Counter_Value"]))
for k, v in a.items():
    print(k, len(v), "launches; FETCH_SIZE KiB median", sorted(v)[len(v)//2], "-> x2 corrected MB", 2 * sorted(v)[len(v)//2] * 1024 / 1e6)
PY
