"""Compare per-shape conv-GEMM launch times (the library's HIP-event CSV, AS_PROF_CSV) between two bench runs.
usage: x6d_policy.py a.csv b.csv   (lines: class,tag,ms,flop,bytes; class 0 = conv GEMM)"""
import collections, csv, re, sys
def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.reader(open(path)):
        if r[0] != "0":
            continue
        key = re.sub(r" (x6d|x6|q|s)$", "", re.sub(r" tile\d+ S\d+", "", r[1]))
        agg[key][0] += 1
        agg[key][1] += float(r[2])
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
for k in sorted(a, key=lambda k: -a[k][1]):
    if k not in b:
        continue
    ma, mb = a[k][1] / a[k][0] * 1e3, b[k][1] / b[k][0] * 1e3
    ta += a[k][1]; tb += b[k][1]
    print(f"{k:34s} n={a[k][0]:4d}  {ma:8.1f} us -> {mb:8.1f} us  {100 * (mb / ma - 1):+6.1f} %   total {a[k][1]:.3f} -> {b[k][1]:.3f} ms")
print(f"all shapes: {ta:.3f} -> {tb:.3f} ms")
