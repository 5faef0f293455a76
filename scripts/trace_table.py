"""per-launch table of one step from a rocprofv3 kernel trace: python scripts/trace_table.py gpurun_out/trace_eager/bench_kernel_trace.csv [filter]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last ref_features_kernel on
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("ref_features_kernel")]
step = rows[starts[-1]:]
tot = 0.0
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if flt in n:
        print(f"{n[:60]:60s} grid {r['Grid_Size_X']:>9s}x{r['Grid_Size_Y']:>5s}x{r['Grid_Size_Z']:>4s} wg {r['Workgroup_Size_X']:>4s} {d:8.1f} us")
print("kernels", len(step), "sum us", round(tot, 1))
