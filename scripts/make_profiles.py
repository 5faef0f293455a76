"""Condense gpurun_out/prof_* (scripts/gpu_profile.sh) into the small summaries committed under profiles/."""
import collections, csv, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_id
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = "gpurun_out", "profiles"
os.makedirs(P, exist_ok=True)
shutil.copy(f"{G}/prof_trace/bench_kernel_stats.csv", f"{P}/{tag}_kernel_stats.csv")
shutil.copy(f"{G}/prof_graph/bench_kernel_stats.csv", f"{P}/{tag}_graph_kernel_stats.csv")
shutil.copy(f"{G}/prof_c5/bench_kernel_stats.csv", f"{P}/{tag}_c5_kernel_stats.csv")
if os.path.exists(f"{G}/prof_trace32/bench_kernel_stats.csv"):
    shutil.copy(f"{G}/prof_trace32/bench_kernel_stats.csv", f"{P}/{tag}_b32_kernel_stats.csv")
if os.path.exists(f"{G}/prof_c2/bench_kernel_stats.csv"):
    shutil.copy(f"{G}/prof_c2/bench_kernel_stats.csv", f"{P}/{tag}_c2_kernel_stats.csv")
for name in ("prof_trace", "prof_trace32", "prof_graph", "prof_c5", "prof_c2"):        # the JSON line each profiled command printed
    if not os.path.exists(f"{G}/{name}.log"):
        continue
    for ln in open(f"{G}/{name}.log"):
        if ln.startswith("{"):
            open(f"{P}/{tag}_{name[5:]}_bench_line.json", "w").write(ln)

if os.path.exists(f"{G}/gemm_launches_events.csv"):
    shutil.copy(f"{G}/gemm_launches_events.csv", f"{P}/{tag}_gemm_launches_events.csv")
for mode in ("weak", "c4"):                                  # the two-rank launches on the one-GPU box
    if os.path.exists(f"{G}/two_ranks_{mode}.log"):
        for ln in open(f"{G}/two_ranks_{mode}.log"):
            if ln.startswith("{"):
                open(f"{P}/{tag}_two_ranks_{mode}_bench_line.json", "w").write(ln)

def short(n):
    return n.split('(')[0].replace('void ', '')[:60]


def kernel_class(name):
    """the event profiler's classes (bench.py CLASSES) from a kernel name of the trace; None = not part of the step (profiler helpers, copies)"""
    n = short(name)
    if n.startswith(("conv_gemm_h3_kernel", "conv_direct_cin1", "splitk_reduce")):     # (incl. the reductions that write an AdaIN image: one conv call)
        return "conv_gemm"
    if n.startswith("adain"):
        return "adain"
    if n.startswith("channel_ln"):
        return "layernorm"
    if n.startswith(("relpos_attention", "xl_attention")):
        return "attention"
    if n.startswith(("bilstm", "lstm_step0")):
        return "lstm"
    if n.startswith("mas_"):
        return "mas"
    if "as_prof_" in n or n.startswith("__amd_rocclr") or n.startswith("make_meta"):
        return None
    return "other"


def trace_classes(stats_csv, bench_line_json, out_json, tag):
    """Sum of kernel-trace durations per class and PER STEP of the eager one-chain pass (prof_trace): what bench.py reports as the
    trace-based roofline fractions (`roofline.frac`, `roofline_hbm.frac`) for the build whose source id it carries."""
    line = json.load(open(bench_line_json))
    steps = line.get("forward_calls_in_process")
    if not steps:
        return None
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(stats_csv)):
        c = kernel_class(r["Name"])
        if c:
            agg[c][0] += float(r["TotalDurationNs"]) / 1e6
            agg[c][1] += int(r["Calls"])
    import re
    m = re.search(r"ONE call over (\d+) batches", line["config"]["workload"])
    bpc = int(m.group(1)) if m else 1                                   # batches of 32 per traced call (bench.py --call-batches)
    out = dict(source_id=source_id(), steps_in_trace=steps, batches_per_call=bpc, file=f"profiles/{tag}_kernel_stats.csv",
               command="rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-graph --no-concurrency --cpu-utts 0 --no-extras"
                       + (f" --call-batches {bpc}" if bpc > 1 else ""),
               classes={c: dict(ms_per_call=v[0] / steps, ms_per_step=v[0] / steps / bpc, launches_per_call=v[1] / steps) for c, v in agg.items()})
    json.dump(out, open(out_json, "w"), indent=1)
    return out

tc = trace_classes(f"{P}/{tag}_kernel_stats.csv", f"{P}/{tag}_trace_bench_line.json", f"{P}/latest_trace_classes.json", tag)
print("trace classes:", json.dumps(tc, indent=1) if tc else "no forward_calls_in_process in the bench line")
for extra in ("jdc", "ema", "vocoder", "surface"):                          # the test.py chain around the path (scripts/gpu_profile.sh)
    if os.path.exists(f"{G}/prof_{extra}/bench_kernel_stats.csv"):
        shutil.copy(f"{G}/prof_{extra}/bench_kernel_stats.csv", f"{P}/{tag}_{extra}_kernel_stats.csv")
if os.path.exists(f"{G}/prof_surface.log"):
    for ln in open(f"{G}/prof_surface.log"):
        if ln.startswith("{"):
            open(f"{P}/{tag}_surface_bench_line.json", "w").write(ln)


agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f"{G}/prof_pmc_sq/bench_counter_collection.csv")):
    k = short(r['Kernel_Name'])
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; cnt[k] += 1
sq = {}
for k in sorted(dur, key=lambda k: -dur[k])[:14]:
    a = agg[k]; gui = a['GRBM_GUI_ACTIVE']
    sq[k] = dict(launches=cnt[k], total_us=round(dur[k], 1), gui_cycles_per_ns=round(gui / 8 / (dur[k] * 1e3), 3) if dur[k] else None,
                 mfma_busy_frac=round(a['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8 * 256 * 4), 4) if gui else None,
                 wait_any=round(a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES'], 3), wait_inst_any=round(a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES'], 3),
                 active_inst=round(a['SQ_ACTIVE_INST_ANY'] / a['SQ_WAVE_CYCLES'], 3))
json.dump(sq, open(f"{P}/{tag}_pmc_sq_summary.json", "w"), indent=1)

def per_kernel(path, counter):
    a = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            k = short(r['Kernel_Name']); a[k] += float(r['Counter_Value']); n[k] += 1
    return a, n
f, nf = per_kernel(f"{G}/prof_pmc_fetch/bench_counter_collection.csv", 'FETCH_SIZE')
w, nw = per_kernel(f"{G}/prof_pmc_write/bench_counter_collection.csv", 'WRITE_SIZE')
tr = {}
gem_f = gem_w = gem_n = 0
for k in sorted(f, key=lambda k: -f[k])[:14]:
    fe = f[k] * 1024 / nf[k]; wr = w.get(k, 0) * 1024 / max(nw.get(k, 1), 1)
    tr[k] = dict(launches=nf[k], fetch_bytes_per_launch_raw=round(fe), fetch_bytes_per_launch_x2=round(2 * fe), write_bytes_per_launch=round(wr))
for k in f:
    if k.startswith("conv_gemm") or k.startswith("splitk_reduce") or k.startswith("split_f16x2"):
        gem_f += f[k] * 1024; gem_w += w.get(k, 0) * 1024
        if k.startswith("conv_gemm"): gem_n += nf[k]
json.dump(tr, open(f"{P}/{tag}_pmc_traffic.json", "w"), indent=1)
# FETCH_SIZE / WRITE_SIZE are KiB; gfx950 reports half of a wide coalesced read stream (MI355X_MICROARCH.md "HBM") -> x2
latest = dict(conv_gemm_hbm_bytes_per_launch=round((2 * gem_f + gem_w) / max(gem_n, 1)),
              conv_gemm_fetch_bytes_per_launch_x2=round(2 * gem_f / max(gem_n, 1)), conv_gemm_write_bytes_per_launch=round(gem_w / max(gem_n, 1)),
              launches=gem_n, source_id=source_id(), batches_per_call=(tc or {}).get("batches_per_call", 1),
              source=f"profiles/{tag}_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (separate runs), "
                     "conv GEMM + split-K reduce kernels, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced stream)")
json.dump(latest, open(f"{P}/latest_pmc_traffic.json", "w"), indent=1)
print(json.dumps(latest, indent=1))
rows = list(csv.DictReader(open(f"{P}/{tag}_kernel_stats.csv")))
for r in rows[:12]:
    print(f"{short(r['Name']):50s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
