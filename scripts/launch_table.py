"""Per-shape table from the profiler CSV (AS_PROF_CSV): python scripts/launch_table.py gpurun_out/launches.csv [steps]"""
import collections, csv, sys
path = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = collections.OrderedDict()
for r in csv.reader(open(path)):
    if r[0] != '0': continue
    d = g.setdefault(r[1], [0, 0.0, 0.0]); d[0] += 1; d[1] += float(r[2]); d[2] += float(r[3])
tot = sum(d[1] for d in g.values())
print(f"GEMM total {tot/steps:.3f} ms/step")
print(f"{'shape':42s} {'n/step':>6s} {'ms/step':>8s} {'%':>5s} {'TF/s':>6s}")
for k, d in sorted(g.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{k:42s} {d[0]/steps:6.1f} {d[1]/steps:8.3f} {100*d[1]/tot:5.1f} {d[2]/d[1]/1e9:6.1f}")
