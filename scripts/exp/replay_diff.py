"""debug aid: eager step vs hipGraph replays of the C3 step -- where and by how much do the mels differ (must be bitwise equal)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from artspeech_amd import models, synth
from artspeech_amd.weights import DEFAULT_STATS, load_distribution
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
net = model.ArtsSpeech
if os.environ.get("SERIAL"): net.rt.set_serial(True)
host, g = bench.make_inputs(dev)
r = bench.Runner(net, g)
keys = ["mel", "style", "duration", "F0", "N", "EMA", "t_en", "a_en"]
def snap():
    o = r.step() if not hasattr(r, "graph") else r.out
    return {k: o[k].clone() for k in keys if k in o and o[k] is not None}
first = snap(); torch.cuda.synchronize()
second = snap(); torch.cuda.synchronize()
print("eager vs eager:", {k: float((first[k] - second[k]).abs().max()) for k in first})
run = r.capture()
for i in range(3):
    run(); torch.cuda.synchronize()
    cur = {k: r.out[k] for k in first}
    print("replay", i, {k: float((first[k] - cur[k]).abs().max()) for k in first})
d = (first["mel"] - r.out["mel"]).abs()
if float(d.max()) > 0:
    cols = torch.nonzero(d.max(0).values > 0).flatten()
    print("differing mel columns:", cols.numel(), "of", d.shape[1], "first", cols[:10].tolist(), "last", cols[-5:].tolist())
bad = 0
for i in range(300):
    run()
    if i % 10 == 9:
        torch.cuda.synchronize()
        dd = float((first["mel"] - r.out["mel"]).abs().max())
        if dd > 0:
            bad += 1
            if bad <= 3:
                d = (first["mel"] - r.out["mel"]).abs()
                cols = torch.nonzero(d.max(0).values > 0).flatten()
                print("replay", i, "max diff", dd, "columns", cols.numel(), cols[:6].tolist(), "rows", torch.nonzero(d.max(1).values > 0).flatten()[:6].tolist())
print("mismatching checks:", bad, "of 30")
# two batches in flight: which lane fails, in which arrangement
class AuxRunner(bench.Runner):
    def step(self):
        g = self.g
        self.out = self.net.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                           frames_hint=g["frames"], out=self.out, aux=True)
        return self.out
mode = os.environ.get("MODE", "orig+replica")
ra = AuxRunner(net, g); ref = {k: v.clone() for k, v in ra.step().items() if torch.is_tensor(v)}
torch.cuda.synchronize()
if mode == "two replicas":
    ra = AuxRunner(net.replica(), g)
if mode == "serial lanes":
    net.rt.set_serial(True)
runa = ra.capture()
rep = net.replica()
if mode == "serial lanes":
    rep.rt.set_serial(True)
r2 = AuxRunner(rep, g)
if os.environ.get("LANE2_ONLY"):
    os.environ["AS_ONLY_BRANCH"] = os.environ["LANE2_ONLY"]               # lane 2's phase A: only this branch (its results are invalid)
run2 = r2.capture()
os.environ.pop("AS_ONLY_BRANCH", None)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = {"lane1": 0, "lane2": 0}
for i in range(100):
    if mode == "replica first":
        with torch.cuda.stream(s2): run2()
        with torch.cuda.stream(s1): runa()
    else:
        with torch.cuda.stream(s1): runa()
        with torch.cuda.stream(s2): run2()
    if i % 10 == 9:
        torch.cuda.synchronize()
        for name, rr in (("lane1", ra), ("lane2", r2)):
            diffs = {k: float((ref[k].float() - rr.out[k].float()).abs().max()) for k in ref}
            if any(v > 0 for v in diffs.values()):
                bad[name] += 1
print(mode, bad)
import ctypes
from artspeech_amd import _lib
L_ = _lib.lib()
if hasattr(L_, "as_debug_adain_counter"):
    buf = (ctypes.c_uint * 4)()
    L_.as_debug_adain_counter(buf)
    print("adain st checks:", buf[0], "mismatches:", buf[1])
