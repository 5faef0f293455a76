"""Numerics experiment (CPU, test infrastructure only): emulate an fp16 two-way split GEMM (x = h + l, h, l fp16 RNE; products
hh + hl + lh, fp32 accumulation) inside the oracle and measure the distance to the reference-generated golden vectors.
Usage: python scripts/exp/f16x3_numerics.py [wscale_log2=auto] [xscale_log2=0] [flush]   (flush: fp16 subnormals -> 0, the
pessimistic model of a matrix core that would flush them)"""
import glob
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from artspeech_amd import synth                                                              # noqa: E402
from artspeech_amd.weights import DEFAULT_STATS, fold_state_dict, load_distribution        # noqa: E402
from oracle import acoustic                                                                  # noqa: E402

XS = float(2 ** int(sys.argv[2])) if len(sys.argv) > 2 else 1.0
WAUTO = len(sys.argv) < 2 or sys.argv[1] == "auto"
WS = 1.0 if WAUTO else float(2 ** int(sys.argv[1]))
FLUSH = "flush" in sys.argv
STATS = {"xmax": 0.0, "wmax": 0.0, "xmin_scale": 1e9}


def f16(x):
    y = x.half().float()
    if FLUSH:
        y = torch.where(y.abs() < 6.103515625e-05, torch.zeros_like(y), y)
    return y


def split(x):
    h = f16(x)
    return h, f16(x - h)


def wrap(fn, lin):
    def g(x, w, b=None, *a, **kw):
        groups = 1 if lin else kw.get("groups", a[2] if len(a) > 2 else 1)
        if groups != 1 or x.dtype != torch.float32:
            return fn(x, w, b, *a, **kw)
        ws = WS
        if WAUTO:                                   # per-tensor power of two: max |w| * ws in [2^13, 2^14)
            ws = float(2.0 ** (13 - np.floor(np.log2(float(w.abs().max()) + 1e-30))))
        STATS["xmax"] = max(STATS["xmax"], float(x.abs().max()))
        xh, xl = split(x * XS)
        wh, wl = split(w * ws)
        out = fn(xh, wl, None, *a, **kw) + fn(xl, wh, None, *a, **kw)
        out = out + fn(xh, wh, None, *a, **kw)
        out = out * (1.0 / (XS * ws))
        if b is not None:
            out = out + (b if lin else b.view(1, -1, *([1] * (out.dim() - 2))))
        return out
    return g


F_conv1d, F_conv2d, F_linear = F.conv1d, F.conv2d, F.linear


def run(patched):
    if patched:
        F.conv1d, F.conv2d, F.linear = wrap(F_conv1d, False), wrap(F_conv2d, False), wrap(F_linear, True)
    else:
        F.conv1d, F.conv2d, F.linear = F_conv1d, F_conv2d, F_linear
    gd = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
    for f in sorted(glob.glob(os.path.join(gd, "net_full_*.npz"))):
        g = np.load(f)
        W = fold_state_dict(synth.synth_state_dict(int(g["hidden_dim"]), int(g["dim_in"]), seed=int(g["weight_seed"])))
        mel, f0, ema = synth.synth_ref_features(int(g["t_ref"]), int(g["seed"]))
        f0_raw = (f0 * np.float32(DEFAULT_STATS["pitch"][3]) + np.float32(DEFAULT_STATS["pitch"][2])).astype(np.float32)
        ema_raw = (ema * np.asarray(DEFAULT_STATS["EMA"][3], np.float32)[:, None]
                   + np.asarray(DEFAULT_STATS["EMA"][2], np.float32)[:, None]).astype(np.float32)
        out = acoustic.forward_test(W, torch.from_numpy(g["tokens"]), torch.from_numpy(mel), torch.from_numpy(f0_raw),
                                    torch.from_numpy(ema_raw), load_distribution(DEFAULT_STATS))
        rep = {k: float(np.abs(out[k].numpy() - g["ref/" + k]).max()) for k in ("style", "duration", "F0", "N", "EMA", "mel")}
        print("f16x3  " if patched else "fp32   ", os.path.basename(f), "dur equal", np.array_equal(out["pred_dur"].numpy(), g["ref/pred_dur"]),
              {k: f"{v:.2e}" for k, v in rep.items()}, flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    run(False)
    run(True)
    print("max |activation| seen by a conv/linear:", STATS["xmax"])
