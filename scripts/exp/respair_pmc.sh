cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  CASES="32,3,1 32,11,5 64,3,1 64,11,3" rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o b --output-format csv -- python3 $R/scripts/exp/respair_bench.py > /tmp/pmc_$c.log 2>&1
done
python3 - <<EOF
import csv, collections
for c in ["FETCH_SIZE","WRITE_SIZE"]:
    rows=list(csv.DictReader(open(f"/tmp/pmc_{c}/b_counter_collection.csv")))
    agg=collections.OrderedDict()
    for r in rows:
        if "respair" not in r["Kernel_Name"]: continue
        key=(r["Kernel_Name"][:60], r.get("Grid_Size"), r.get("LDS_Block_Size"))
        a=agg.setdefault(key,[0,0.0]); a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,(n,v) in agg.items(): print(c, k, n, "avg per launch:", v/n)
EOF
