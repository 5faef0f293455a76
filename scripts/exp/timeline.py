"""Overlap analysis of a kernel timeline written by scripts/exp/timeline.sh: python scripts/exp/timeline.py tl2.csv t0_ms t1_ms"""
import sys, collections
rows = [l.strip().split(',') for l in open(sys.argv[1])]
t0, t1 = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
K = [(int(r[0]), int(r[1]), r[2], r[4], int(r[5]) // max(int(r[6]), 1)) for r in rows]
K = [k for k in K if t0 <= k[0] < t1]
ev = []
for k in K:
    ev.append((k[0], 1, k)); ev.append((k[1], -1, k))
ev.sort(key=lambda e: (e[0], e[1]))
lvl = 0; last = ev[0][0]; hist = collections.Counter()
active = []
gem_alone = gem_with_gem = gem_with_other = other_alone = other_with_other = 0
def isg(k): return 'conv_gemm' in k[3] or 'conv_direct' in k[3]
for t, d, k in ev:
    dt = t - last
    if dt:
        hist[min(lvl, 4)] += dt
        ng = sum(1 for a in active if isg(a)); no = len(active) - ng
        if ng == 1 and no == 0: gem_alone += dt
        elif ng >= 2 and no == 0: gem_with_gem += dt
        elif ng >= 1 and no >= 1: gem_with_other += dt
        elif ng == 0 and no == 1: other_alone += dt
        elif ng == 0 and no >= 2: other_with_other += dt
    last = t
    if d == 1: active.append(k); lvl += 1
    else: active.remove(k); lvl -= 1
span = (t1 - t0)
print("span %.2f ms" % (span / 1e6), {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
print("gemm alone %.2f  gemm+gemm %.2f  gemm+other %.2f  other alone %.2f  other+other %.2f  idle %.2f" % tuple(
    x / 1e6 for x in (gem_alone, gem_with_gem, gem_with_other, other_alone, other_with_other, hist[0])))
tot = collections.Counter()
for k in K: tot['gemm' if isg(k) else 'other'] += k[1] - k[0]
print("sum of durations:", {k: round(v / 1e6, 2) for k, v in tot.items()})
