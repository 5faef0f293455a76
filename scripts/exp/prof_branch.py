"""Per-launch HIP-event profile of ONE branch of the step (tuning aid): CASE=dur|style|arts|pred|dec python scripts/exp/prof_branch.py"""
import os, sys, ctypes, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["AS_PROF_CSV"] = "/tmp/prof_branch.csv"
import numpy as np, torch
import bench
from artspeech_amd import models, synth, ops, _lib
from artspeech_amd.weights import DEFAULT_STATS, load_distribution
dev = torch.device("cuda:0")
sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
net = model.ArtsSpeech
host, g = bench.make_inputs(dev)
models.CONCURRENT = False
out = net.forward_packed(g["tok"], g["tok_lay"], g["mel"], g["f0"], g["ema"], g["ref_lay"], forced=g["forced"], frames_hint=g["frames"])
torch.cuda.synchronize()
feat12, style = out["feat12"], out["style"]
FEAT = torch.cat([feat12, g["mel"][:, : feat12.shape[1]]], dim=0).contiguous()
TI = net.style_encoder.tower_inputs(FEAT, g["ref_lay"])
case = os.environ.get("CASE", "dur")
fn = {"dur": lambda: net.durationPredictor.forward_packed(g["tok"], g["tok_lay"], feat12[2:12], g["ref_lay"]),
      "arts": lambda: net.arts_encoder.forward_packed(g["tok"], g["tok_lay"]),
      "dec": lambda: net.decoder.forward_packed(ops.expand(out["t_en"], ops.durations(out["duration"].reshape(-1), g["forced"], g["tok_lay"], out["lay1"].N)[2], out["lay1"].N, 2, out["lay2"].new(512)), out["lay2"], style, out["F0"], out["N"], out["EMA"]),
      "towers": lambda: [net.style_encoder.tower(w, TI) for w in ("ema", "f0", "energy")],
      "mel": lambda: net.style_encoder.tower("mel", TI),
      }[case]
fn(); torch.cuda.synchronize()
L = _lib.lib()
if os.path.exists("/tmp/prof_branch.csv"): os.remove("/tmp/prof_branch.csv")
L.as_prof_enable(1)
fn(); torch.cuda.synchronize()
n = 7
ms = (ctypes.c_double * n)(); fl = (ctypes.c_double * n)(); by = (ctypes.c_double * n)(); cnt = (ctypes.c_int32 * n)()
L.as_prof_collect(ms, fl, by, cnt, n); L.as_prof_enable(0)
print(case, "per class ms:", dict(zip(bench.CLASSES, [round(v, 3) for v in ms])), "launches", list(cnt))
rows = list(csv.reader(open("/tmp/prof_branch.csv")))
agg = collections.OrderedDict()
for r in rows:
    k = (bench.CLASSES[int(r[0])], r[1])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r[2])
for (c, tag), (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{c:10s} {tag:40s} n={k:3d} {t*1e3:8.1f} us")
