"""One step's launches in order from the profiler's per-launch CSV (AS_PROF_CSV): class, tag, us (bracket cost removed), MB.
usage: launch_seq.py events.csv [steps=3] [bracket_us=4.3]"""
import csv, sys
rows = [r for r in csv.reader(open(sys.argv[1]))]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
br = float(sys.argv[3]) if len(sys.argv) > 3 else 4.3
n = len(rows) // steps
names = {'1': 'adain', '2': 'ln', '3': 'attn', '4': 'lstm', '6': 'other'}
tot = {}
for i in range(n):
    # median over the steps of the i-th launch
    ms = sorted(float(rows[s * n + i][2]) for s in range(steps))[steps // 2]
    r = rows[i]
    tag = r[1] if r[0] == '0' else names.get(r[0], r[0])
    us = ms * 1e3 - br
    tot[r[0]] = tot.get(r[0], 0.0) + us
    print(f"{i:4d} {r[0]} {tag:46s} {us:7.1f} us {float(r[4]) / 1e6:8.1f} MB")
print({names.get(k, 'gemm'): round(v) for k, v in tot.items()}, "sum", round(sum(tot.values())))
