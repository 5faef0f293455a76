"""Checkpoint loading: fold the reference's reparametrised tensors into plain weights.

The reference stores weight_norm convs as (weight_g, weight_v) and spectral_norm convs as
(weight_orig, weight_u, weight_v).  In eval mode the effective
weights are
    weight_norm      w = v * (g / ||v||_{dims != 0})           (torch._weight_norm, dim=0)
    spectral_norm    w = weight_orig / (u . (W2d v))           (no power iteration in eval)
This module computes those once at load time (models.py:685-701 does the equivalent implicitly on
every forward) and returns a flat  name -> fp32 tensor  dict with ``<prefix>.weight`` keys.
"""
import numpy as np
import torch

from .spec import EXTRACTOR_PREFIXES


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def fold_state_dict(sd):
    """reference-format state dict (tensors or numpy) -> plain fp32 CPU tensors."""
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    out = {}
    for k, v in sd.items():
        if k.startswith(EXTRACTOR_PREFIXES):
            continue
        v = _t(v).detach().to(torch.float32).cpu()
        if k.endswith(".weight_g"):
            p = k[: -len(".weight_g")]
            g, d = v, _t(sd[p + ".weight_v"]).detach().float().cpu()
            norm = d.reshape(d.shape[0], -1).norm(dim=1).reshape(g.shape)
            out[p + ".weight"] = (d * (g / norm)).contiguous()
        elif k.endswith(".weight_orig"):
            p = k[: -len(".weight_orig")]
            u = _t(sd[p + ".weight_u"]).detach().float().cpu()
            vv = _t(sd[p + ".weight_v"]).detach().float().cpu()
            sigma = torch.dot(u, torch.mv(v.reshape(v.shape[0], -1), vv))
            out[p + ".weight"] = (v / sigma).contiguous()
        elif k.endswith((".weight_v", ".weight_u")):
            continue
        else:
            out[k] = v.contiguous()
    return out


def load_distribution(stats, device="cpu"):
    """stats.json dict -> the ``distribution`` dict of utils.py:86-92 / test.py:75-79.

    Entries are [min?, max?, mean, std] (the shipped Data/stats.json); only [2], [3] are used."""
    dist = {}
    for key in ("EMA", "pitch", "energy"):
        _, _, mean_val, std_val = stats[key]
        dist[f"{key}_mean"] = torch.tensor(mean_val, dtype=torch.float32, device=device)
        dist[f"{key}_std"] = torch.tensor(std_val, dtype=torch.float32, device=device)
    return dist


# Data/stats.json of the reference (mean/std columns only are used; values are data, committed
# here so nothing reads /root/reference at run time).
DEFAULT_STATS = {
    "EMA": [None, None,
            [-0.025431977584958076, -0.010428862646222115, 0.004749640356749296, 0.022112606093287468,
             0.04373274743556976, 0.06785734742879868, 0.0906321108341217, 0.1101362556219101,
             0.12228765338659286, 0.12749071419239044],
            [0.8377999067306519, 0.8576095700263977, 0.872006356716156, 0.8813959360122681,
             0.88730788230896, 0.8949191570281982, 0.9049390554428101, 0.9150999784469604,
             0.9237231612205505, 0.9301097393035889]],
    "pitch": [None, None, 137.0945846179066, 78.25457848323684],
    "energy": [None, None, 4.601160882738853, 3.110472802617481],
}
