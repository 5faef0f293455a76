"""Wall time of the step's phases and of the individual concurrent branches, each captured in its own hipGraph
(tuning aid: shows the critical path).  Same workload as bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from artspeech_amd import models, synth, ops
from artspeech_amd.models import Fork, side_streams
from artspeech_amd.weights import DEFAULT_STATS, load_distribution

dev = torch.device("cuda:0")
sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
net = model.ArtsSpeech
host, g = bench.make_inputs(dev)
out = net.forward_packed(g["tok"], g["tok_lay"], g["mel"], g["f0"], g["ema"], g["ref_lay"], forced=g["forced"], frames_hint=g["frames"])
torch.cuda.synchronize()
stats24 = net._stats24
feat12 = out["feat12"]; style = out["style"]; t_en = out["t_en"]; a_en = out["a_en"]; duration = out["duration"]
lay1, lay2 = out["lay1"], out["lay2"]
dur_i, frame_off, tof = ops.durations(duration.reshape(-1), g["forced"], g["tok_lay"], lay1.N)
C = t_en.shape[0]
a_ex = ops.expand(a_en, tof, lay1.N, 1, lay1.new(C))
t_up = ops.expand(t_en, tof, lay1.N, 2, lay2.new(C))
f0, n, ema = out["F0"], out["N"], out["EMA"]

se = net.style_encoder
feat = torch.cat([feat12, g["mel"][:, : feat12.shape[1]]], dim=0).contiguous()
ti = se.tower_inputs(feat, g["ref_lay"])
torch.cuda.synchronize()


def phase1():
    with Fork(side_streams(dev, 4), uses=(feat12, feat, ti["c"], ti["mel_img"], ti["ema_img"])) as side:
        with side(0): a = net.arts_encoder.forward_packed(g["tok"], g["tok_lay"])
        with side(1): b = se.tower("mel", ti)
        with side(2): c = [se.tower(w, ti) for w in ("ema", "f0", "energy")]
        with side(3): d = net.durationPredictor.forward_packed(g["tok"], g["tok_lay"], feat12[2:12], g["ref_lay"])
        side.produced(a, b, d, *c)
    return a, b, c, d


tok2 = torch.cat([g["tok"], g["tok"]]).contiguous()
lay2x = ops.layout(list(g["tok_lay"].widths_host) * 2, dev)


def phase1_wide():                          # what a grouped (text + articulatory) double-width encoder would cost in phase 1
    with Fork(side_streams(dev, 4), uses=(feat12, feat, ti["c"], ti["mel_img"], ti["ema_img"])) as side:
        with side(0): a = net.arts_encoder.forward_packed(tok2, lay2x)
        with side(1): b = se.tower("mel", ti)
        with side(2): c = [se.tower(w, ti) for w in ("ema", "f0", "energy")]
        with side(3): d = net.durationPredictor.forward_packed(g["tok"], g["tok_lay"], feat12[2:12], g["ref_lay"])
        side.produced(a, b, d, *c)
    return a, b, c, d


def phase2():
    with Fork(side_streams(dev, 1, "text_encoder"), uses=(style,)) as side:
        with side(0):
            t = net.text_encoder.forward_packed(g["tok"], g["tok_lay"])
            gb = net.decoder.adain_params(style)
        r = net.artsPredictor.forward_packed(a_ex, lay1, style)
        side.produced(t, *[x for v in gb.values() for x in v])
    return t, r, gb


cases = {
    "phase 1 (4 concurrent branches)": phase1,
    "  arts encoder alone": lambda: net.arts_encoder.forward_packed(g["tok"], g["tok_lay"]),
    "  mel tower alone": lambda: se.tower("mel", ti),
    "  TV + F0 + energy towers alone": lambda: [se.tower(w, ti) for w in ("ema", "f0", "energy")],
    "    TV tower alone": lambda: se.tower("ema", ti),
    "    F0 tower alone": lambda: se.tower("f0", ti),
    "    energy tower alone": lambda: se.tower("energy", ti),
    "  duration predictor alone": lambda: net.durationPredictor.forward_packed(g["tok"], g["tok_lay"], feat12[2:12], g["ref_lay"]),
    "phase 2 (text encoder || predictor)": phase2,
    "  text encoder (+ decoder AdaIN fc) alone": lambda: (net.text_encoder.forward_packed(g["tok"], g["tok_lay"]), net.decoder.adain_params(style)),
    "  predictor alone (3 branches + LSTM)": lambda: net.artsPredictor.forward_packed(a_ex, lay1, style),
    "phase 1 with a double-width encoder": phase1_wide,
    "  double-width encoder alone": lambda: net.arts_encoder.forward_packed(tok2, lay2x),
    "phase 3 (decoder)": lambda: net.decoder.forward_packed(t_up, lay2, style, f0, n, ema),
    "whole step": lambda: net.forward_packed(g["tok"], g["tok_lay"], g["mel"], g["f0"], g["ema"], g["ref_lay"], forced=g["forced"], frames_hint=g["frames"]),
}
sel = os.environ.get('CASE')
for name, fn in cases.items():
    if sel is not None and str(list(cases).index(name)) != sel: continue
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            keep = fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_it = 20
    for _ in range(n_it): gr.replay()
    torch.cuda.synchronize()
    print(f"{name:42s} {(time.perf_counter()-t0)/n_it*1e3:8.3f} ms")
