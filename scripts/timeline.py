"""Timeline of ONE benchmark step from a rocprofv3 kernel trace: python scripts/timeline.py trace.csv [step_index_from_end]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps are delimited by the first kernel of the step: ref_features_kernel
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('ref_features_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = starts[-k-1], starts[-k]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print(f"step: {len(step)} kernels, wall {(t1-t0)/1e3:.1f} us, sum of kernel durations {sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)/1e3:.1f} us")
# busy timeline: fraction of wall time with >=1 kernel running, and concurrency histogram
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("concurrency histogram (us):", {k: round(v/1e3, 1) for k, v in sorted(hist.items())})
# phase view: 40 buckets of the wall time, top kernel by time in each
nb = 32; w = (t1 - t0) / nb
for i in range(nb):
    lo, hi = t0 + i*w, t0 + (i+1)*w
    acc = collections.Counter()
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        ov = min(e, hi) - max(s, lo)
        if ov > 0:
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:28]
            acc[f"{name}[{r['Grid_Size_X']}x{r['Grid_Size_Y']}]"] += ov
    tot = sum(acc.values())
    top = ", ".join(f"{n}:{v/w:.2f}" for n, v in acc.most_common(3))
    print(f"{(lo-t0)/1e3:8.0f}us  load {tot/w:4.2f}  {top}")
