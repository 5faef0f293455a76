"""GPU: EMA_Predictor on the HIP path (artspeech_amd/ema.py, through the C ABI) against outputs of the reference's
EMA_Predictor (tests/golden/ema_*.npz) and, kernel by kernel, against plain torch: trajectories within 1e-5 abs (values are
O(0.2) with the synthetic weights; observed differences are printed)."""
import glob
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from artspeech_amd import ema as E
from artspeech_amd import ops

pytestmark = pytest.mark.gpu
TOL = 1e-5
_NET = {}


def net(cuda):
    if "n" not in _NET:
        _NET["n"] = E.EMA_Predictor(device=cuda).load_state_dict({"model": E.synth_ema_state_dict(seed=3407)})
    return _NET["n"]


def test_ema_matches_reference(cuda, golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "ema_T*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        out = net(cuda)(torch.from_numpy(g["f0"])[None], torch.from_numpy(g["n"])[None], torch.from_numpy(g["mel"])[None])
        assert out.shape == (1, 10, int(g["t"]))
        d = float(np.abs(out[0].cpu().numpy() - g["ema"]).max())
        print(os.path.basename(f), "ema max-abs", d)
        assert d <= TOL, (f, d)


def test_ema_ragged_batch_equals_single(cuda, golden_dir):
    gs = [np.load(f) for f in sorted(glob.glob(os.path.join(golden_dir, "ema_T*.npz")))]
    tmax = max(int(g["t"]) for g in gs)
    mel, f0, n = torch.zeros(len(gs), 80, tmax), torch.zeros(len(gs), 1, tmax), torch.zeros(len(gs), 1, tmax)
    for b, g in enumerate(gs):
        t = int(g["t"])
        mel[b, :, :t], f0[b, :, :t], n[b, :, :t] = torch.from_numpy(g["mel"]), torch.from_numpy(g["f0"]), torch.from_numpy(g["n"])
    out = net(cuda)(f0, n, mel, lengths=[int(g["t"]) for g in gs])
    assert out.shape == (len(gs), 10, tmax)
    for b, g in enumerate(gs):
        t = int(g["t"])
        assert float(np.abs(out[b, :, :t].cpu().numpy() - g["ema"]).max()) <= TOL
        assert t == tmax or float(out[b, :, t:].abs().max()) == 0.0


@pytest.mark.parametrize("lens", [[7], [150, 1, 66, 17], [64, 65, 129]])
def test_xl_attention_kernel(cuda, lens):
    """as_xl_attention_f32 against the reference's formulation written out in torch (pad-and-reshape relative shift)."""
    g = torch.Generator().manual_seed(sum(lens))
    C, heads, dh = 256, 4, 64
    lay = ops.layout(lens, cuda)
    qkv, pos = torch.randn(3 * C, lay.N, generator=g), torch.randn(C, lay.N, generator=g)
    u, v = torch.randn(heads, dh, generator=g) * 0.5, torch.randn(heads, dh, generator=g) * 0.5
    QKV, POS = lay.new(3 * C), lay.new(C)
    QKV[:, : lay.N].copy_(qkv)
    POS[:, : lay.N].copy_(pos)
    out = ops.xl_attention(QKV, C, heads, POS, u.to(cuda), v.to(cuda), 1.0 / math.sqrt(C), lay, lay.new(C))[:, : lay.N].cpu()
    o = 0
    for T in lens:
        q = qkv[:C, o:o + T].t().reshape(T, heads, dh)
        k = qkv[C:2 * C, o:o + T].t().reshape(T, heads, dh).permute(1, 0, 2)
        vv = qkv[2 * C:, o:o + T].t().reshape(T, heads, dh).permute(1, 0, 2)
        pe = pos[:, o:o + T].t().reshape(T, heads, dh)
        content = torch.matmul((q + u).transpose(0, 1), k.transpose(1, 2))
        ps = torch.matmul((q + v).transpose(0, 1), pe.permute(1, 2, 0))
        padded = torch.cat([ps.new_zeros(heads, T, 1), ps], -1).reshape(heads, T + 1, T)
        ps = padded[:, 1:].reshape(heads, T, T)
        attn = F.softmax((content + ps) / math.sqrt(C), -1)
        want = torch.matmul(attn, vv).transpose(0, 1).reshape(T, C).t()
        assert float((out[:, o:o + T] - want).abs().max()) <= 2e-5
        o += T


def test_glu_dwconv_bn_swish_and_lstm_step0(cuda):
    g = torch.Generator().manual_seed(9)
    C, k, lens = 8, 31, [40, 3, 1100]
    lay = ops.layout(lens, cuda)
    a = torch.randn(2 * C, lay.N, generator=g)
    w, sc, sh = torch.randn(C, k, generator=g) / 5, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    A = lay.new(2 * C)
    A[:, : lay.N].copy_(a)
    y = ops.glu_dwconv_bn_swish(A, C, w.to(cuda), sc.to(cuda), sh.to(cuda), lay, lay.new(C))[:, : lay.N].cpu()
    o = 0
    for T in lens:
        gl = a[:C, o:o + T] * torch.sigmoid(a[C:, o:o + T])
        z = F.conv1d(gl[None], w[:, None], padding=k // 2, groups=C)[0] * sc[:, None] + sh[:, None]
        assert float((y[:, o:o + T] - z * torch.sigmoid(z)).abs().max()) <= 1e-5
        o += T
    H, N = 16, 333
    gx = torch.randn(8 * H, N, generator=g)
    h = ops.lstm_step0(gx.to(cuda), H, N, torch.empty(2 * H, N, device=cuda)).cpu()
    for d in range(2):
        blk = gx[d * 4 * H:(d + 1) * 4 * H]
        c = torch.sigmoid(blk[:H]) * torch.tanh(blk[2 * H:3 * H])
        assert float((h[d * H:(d + 1) * H] - torch.sigmoid(blk[3 * H:]) * torch.tanh(c)).abs().max()) <= 1e-6
