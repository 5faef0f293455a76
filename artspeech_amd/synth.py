"""Seeded synthetic checkpoint in the reference's state-dict format.

No pretrained acoustic checkpoint ships with the reference (README.md:48 is an
external link; SURVEY.md "Quick facts"), so parity is pinned with synthetic
weights that BOTH sides can regenerate from a seed: tests/golden/make_golden.py
loads them into the reference modules with ``load_state_dict`` and the GPU box
regenerates the identical tensors instead of shipping 600 MB.

The generator is a counter-based integer hash (splitmix64), so a tensor's
values depend only on (seed, tensor name, element index) -- not on numpy's
Generator streams, BLAS, or thread count.  Values are multiples of 2^-23 in
[-1, 1) scaled by a per-tensor constant, i.e. exactly representable fp32.
Scales follow PyTorch's default initialisers (U(+-1/sqrt(fan_in))) so the
activations stay O(1) through the ~40-layer path; the exceptions are noted
inline.
"""
import zlib
import numpy as np

from .spec import artsspeech_spec

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(name, n, seed):
    """n floats in [-1, 1): exact multiples of 2^-23, a pure function of (seed, name, index)."""
    with np.errstate(over="ignore"):
        base = np.uint64(zlib.crc32(name.encode("utf-8"))) * np.uint64(0x100000001B3) + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019)
        idx = np.arange(n, dtype=np.uint64) + _splitmix64(np.array([base], dtype=np.uint64))[0]
        bits = _splitmix64(idx) >> np.uint64(40)          # 24 random bits
    return (bits.astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)).astype(np.float32)


def hash_tensor(name, shape, seed, scale=1.0, shift=0.0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(name, n, seed)
    return (u * np.float32(scale) + np.float32(shift)).astype(np.float32).reshape(shape)


def _power_iterate(w2d, v0, iters=12):
    """u, v after `iters` power-iteration steps (fp64), as torch's spectral_norm would converge to."""
    w = w2d.astype(np.float64)
    v = v0.astype(np.float64)
    v /= max(np.linalg.norm(v), 1e-12)
    u = w @ v
    for _ in range(iters):
        u = w @ v
        u /= max(np.linalg.norm(u), 1e-12)
        v = w.T @ u
        v /= max(np.linalg.norm(v), 1e-12)
    u = w @ v
    u /= max(np.linalg.norm(u), 1e-12)
    return u.astype(np.float32), v.astype(np.float32)


def synth_state_dict(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80, seed=3407, spec=None):
    """name -> np.float32 array, in the reference's checkpoint layout."""
    spec = spec or artsspeech_spec(hidden_dim, dim_in, style_dim, n_mels)
    sd = {}
    pending_sn = {}
    for name, info in spec.items():
        shape, kind, fan = info["shape"], info["kind"], info["fan_in"]
        if kind in ("w", "wn_v", "sn_w", "lstm_w"):
            b = 1.0 / np.sqrt(fan)
            if name.endswith("norm1.fc.weight") or name.endswith("norm2.fc.weight"):
                b *= 0.5          # AdaIN gains: keep (1+gamma) away from 0
            sd[name] = hash_tensor(name, shape, seed, b)
            if kind == "sn_w":
                pending_sn[name[: -len(".weight_orig")]] = sd[name]
        elif kind in ("b", "lstm_b"):
            sd[name] = hash_tensor(name, shape, seed, 1.0 / np.sqrt(fan))
        elif kind == "wn_g":
            # torch initialises g = ||v|| ~= 0.577; use [0.7, 1.1] so unit-variance inputs stay O(1)
            sd[name] = hash_tensor(name, shape, seed, 0.2, 0.9)
        elif kind == "emb":
            t = hash_tensor(name, shape, seed, np.sqrt(3.0) * fan ** -0.5)
            t[0] = 0.0              # padding_idx=0 (RelTransformerEnc.py:354)
            sd[name] = t
        elif kind == "rel":
            sd[name] = hash_tensor(name, shape, seed, np.sqrt(3.0) * fan ** -0.5)
        elif kind == "ln_g":
            sd[name] = hash_tensor(name, shape, seed, 0.1, 1.0)
        elif kind == "ln_b":
            sd[name] = hash_tensor(name, shape, seed, 0.1)
        elif kind in ("sn_u", "sn_v"):
            sd[name] = None         # filled below from the power iteration
        else:
            raise ValueError(kind)
    for prefix, w in pending_sn.items():
        w2d = w.reshape(w.shape[0], -1)
        v0 = hash_tensor(prefix + ".weight_v", (w2d.shape[1],), seed)
        u, v = _power_iterate(w2d, v0)
        sd[prefix + ".weight_u"] = u
        sd[prefix + ".weight_v"] = v
    # durations: LinearNorm(512->1) over a tanh-bounded BiLSTM output; bias 2.5 puts
    # round(duration) in 1..4 so that N=40 tokens give ~100 half-rate frames.
    sd["durationPredictor.duration_proj.linear_layer.bias"] = np.full((1,), 2.5, np.float32)
    k = "durationPredictor.duration_proj.linear_layer.weight"
    sd[k] = (sd[k] * np.float32(3.0)).astype(np.float32)
    assert all(v is not None for v in sd.values())
    return sd


def synth_tokens(n, seed, pad_ends=True):
    """Token ids: uniform over 1..177 with id 0 at both ends (meldataset.py:112-113)."""
    u = hash_uniform(f"tokens/{n}", n, seed)
    ids = (np.floor((u.astype(np.float64) + 1.0) * 0.5 * 177.0).astype(np.int64) % 177) + 1
    if pad_ends and n >= 2:
        ids[0] = 0
        ids[-1] = 0
    return ids


def synth_ref_features(t_ref, seed):
    """Stand-ins for the reference utterance: mel [80,T] and the A14 extractor outputs
    f0 [1,T], energy-independent; values uniform with the variances SURVEY.md D2 names."""
    s3 = np.sqrt(3.0)
    mel = hash_tensor(f"ref/mel/{t_ref}", (80, t_ref), seed, 0.5 * s3)
    f0 = hash_tensor(f"ref/f0/{t_ref}", (1, t_ref), seed, s3)
    ema = hash_tensor(f"ref/ema/{t_ref}", (10, t_ref), seed, s3)
    return mel, f0, ema
