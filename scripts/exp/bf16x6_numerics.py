"""Numerics experiment (CPU, test infrastructure only): emulate the bf16x6 split-precision GEMM inside the oracle and
measure the distance to the reference-generated golden vectors.  x = h + m + l with h, m, l bf16 (RNE), six products
(hl, lh, mm, hm, mh, hh), fp32 accumulation.  Usage: python scripts/exp/bf16x6_numerics.py [terms=6] [trunc]"""
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

TERMS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
TRUNC = len(sys.argv) > 2 and sys.argv[2] == "trunc"


def to_bf16(x):
    if TRUNC:
        return (x.contiguous().view(torch.int32) & -65536).view(torch.float32)
    return x.bfloat16().float()


def split(x):
    h = to_bf16(x)
    r = x - h
    m = to_bf16(r)
    l = to_bf16(r - m)
    return h, m, l


PAIRS = [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0), (1, 2), (2, 1), (2, 2)]


def wrap(fn):
    def g(x, w, b=None, *a, **kw):
        groups = kw.get("groups", a[2] if len(a) > 2 else 1) if fn is not F_linear else 1
        if groups != 1 or x.dtype != torch.float32:
            return fn(x, w, b, *a, **kw)
        xs, ws = split(x), split(w)
        use = PAIRS[:TERMS] if TERMS != 3 else [(0, 1), (1, 0), (0, 0)]
        out = None
        for i, j in use:
            t = fn(xs[i], ws[j], None, *a, **kw)
            out = t if out is None else out + t
        if b is not None:
            out = out + (b.view(1, -1, *([1] * (out.dim() - 2))) if fn is not F_linear else b)
        return out
    return g


F_conv1d, F_conv2d, F_linear = F.conv1d, F.conv2d, F.linear


def run(patched):
    if patched:
        F.conv1d, F.conv2d, F.linear = wrap(F_conv1d), wrap(F_conv2d), wrap(F_linear)
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
        print("patched" if patched else "fp32   ", os.path.basename(f), "dur equal", np.array_equal(out["pred_dur"].numpy(), g["ref/pred_dur"]),
              {k: f"{v:.2e}" for k, v in rep.items()})


if __name__ == "__main__":
    torch.set_num_threads(8)
    run(False)
    run(True)
