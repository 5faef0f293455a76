"""Gaps between consecutive kernels of the LAST graph replay in a rocprofv3 kernel trace (tuning aid).
usage: trace_gaps.py kernel_trace.csv [n_kernels_per_replay]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if not n:                                   # find the period: the last kernel name's previous occurrence pattern
    names = [r["Kernel_Name"] for r in rows]
    for p in range(10, len(names) // 3):
        if names[-p:] == names[-2 * p:-p]:
            n = p
            break
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
busy = 0
prev_end = t0
print(f"{n} kernels per replay, wall {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end, e)
print(f"sum of kernel durations {busy / 1e3:.1f} us")
