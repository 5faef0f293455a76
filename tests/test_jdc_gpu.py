"""GPU: JDCNet on the HIP path (artspeech_amd/jdc.py, through the C ABI) against outputs of the reference's JDCNet
(tests/golden/jdc_*.npz) and against the oracle on a ragged batch: |classifier| within 1e-5 abs (values are O(0.3) with
the synthetic weights; observed differences are printed)."""
import glob
import os

import numpy as np
import pytest
import torch

from artspeech_amd import jdc as J
from artspeech_amd import ops

pytestmark = pytest.mark.gpu
TOL = 1e-5
_NET = {}


def net(cuda):
    if "n" not in _NET:
        _NET["n"] = J.JDCNet(num_class=1, seq_len=192, device=cuda).load_state_dict(J.synth_jdc_state_dict(1, seed=3407))
    return _NET["n"]


def test_jdc_matches_reference(cuda, golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "jdc_T*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        f0 = net(cuda)(torch.from_numpy(g["mel"])[None, None])                  # models.py:432: mel.unsqueeze(1)
        assert f0.shape == (1, 1, int(g["t"]))
        d = float(np.abs(f0[0].cpu().numpy() - g["f0"]).max())
        print(os.path.basename(f), "f0 max-abs", d)
        assert d <= TOL, (f, d)


def test_jdc_ragged_batch_and_padded_batch(cuda, golden_dir):
    """lengths=: every utterance equals its B = 1 result (the goldens); lengths=None: the reference's behaviour on a
    zero-padded batch, every item run at the full length (checked against the oracle on the padded item)."""
    import torch.nn.functional as F
    from oracle import jdc as ojdc
    gs = [np.load(f) for f in sorted(glob.glob(os.path.join(golden_dir, "jdc_T*.npz")))]
    tmax = max(int(g["t"]) for g in gs)
    mel = torch.zeros(len(gs), 80, tmax)
    for b, g in enumerate(gs):
        mel[b, :, : int(g["t"])] = torch.from_numpy(g["mel"])
    f0 = net(cuda)(mel.unsqueeze(1), lengths=[int(g["t"]) for g in gs])
    assert f0.shape == (len(gs), 1, tmax)
    for b, g in enumerate(gs):
        n = int(g["t"])
        assert float(np.abs(f0[b, :, :n].cpu().numpy() - g["f0"]).max()) <= TOL
        assert n == tmax or float(f0[b, :, n:].abs().max()) == 0.0
    dense = net(cuda)(mel.unsqueeze(1))
    W = {k: torch.from_numpy(np.asarray(v)) for k, v in J.synth_jdc_state_dict(1, seed=3407).items()}
    b = min(range(len(gs)), key=lambda i: int(gs[i]["t"]))                       # the most padded item
    want = ojdc.jdcnet(W, mel[b])
    assert float((dense[b].cpu() - want).abs().max()) <= TOL


def test_bn_lrelu_maxpool_rows(cuda):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    C, H, lens = 6, 10, [7, 1, 30]
    xs = [torch.randn(C, H, L, generator=g) for L in lens]
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    tok = ops.layout(lens, cuda)
    X = ops.layout(lens, cuda, H=H).new(C)
    X.copy_(torch.cat([x.reshape(C, -1) for x in xs], 1))
    for k in (2, 4):
        ref = [F.max_pool2d(F.leaky_relu(x * sc[:, None, None] + sh[:, None, None], 0.01)[None], (k, 1))[0] for x in xs]
        lo = ops.layout(lens, cuda, H=H // k)
        y = ops.bn_lrelu_maxpool_rows(X, tok, H, k, sc.to(cuda), sh.to(cuda), 0.01, lo.new(C))
        assert torch.allclose(y[:, : lo.N].cpu(), torch.cat([r.reshape(C, -1) for r in ref], 1), atol=1e-6)
        yc = ops.bn_lrelu_maxpool_rows(X, tok, H, k, sc.to(cuda), sh.to(cuda), 0.01, tok.new(C * (H // k)), to_channels=True)
        want = torch.cat([r.reshape(C * (H // k), -1) for r in ref], 1)         # row c * Hout + h
        assert torch.allclose(yc[:, : tok.N].cpu(), want, atol=1e-6)


def test_pipeline_with_attached_extractors(cuda, golden_dir):
    """models.py:431-433 on the test.py surface: with the HIP JDCNet and EMA_Predictor attached the model computes F0 and the
    EMA trajectories from the reference mel itself (phonemes + mel in, mel out, no `features`); the result equals feeding the
    extractors' outputs in by hand, and F0 alone can be left to the model too."""
    import json
    from artspeech_amd import ema as E
    from artspeech_amd import synth
    from artspeech_amd.pipeline import ArtSpeech
    from test_net_gpu import raw_features
    tts = ArtSpeech(config={"model_params": {"hidden_dim": 64, "dim_in": 8, "max_conv_dim": 64}},
                    checkpoint={"net": {"ArtsSpeech": synth.synth_state_dict(64, 8, seed=3407)}}, device=cuda)
    jd = tts.attach_pitch_extractor({"net": J.synth_jdc_state_dict(1, seed=3407)})
    with open(os.path.join(golden_dir, "text_golden.json"), encoding="utf-8") as f:
        cases = json.load(f)["cases"]
    ph = [cases[0]["text"][:30], cases[1]["text"][:18]]
    lens = (90, 70)
    mels, emas = [], []
    for i, t in enumerate(lens):
        mel, _, ema_raw = raw_features(t, 40 + i)
        mels.append(mel)
        emas.append(ema_raw)
    dense = torch.zeros(2, 80, 90)
    for b, m in enumerate(mels):
        dense[b, :, : m.shape[-1]] = torch.as_tensor(m)
    f0 = jd(dense.unsqueeze(1), lengths=lens)
    # F0 from the attached JDCNet, EMA given
    mel_a = tts.synthesis_mel(ph, mels, features=[(None, e) for e in emas])
    mel_b = tts.synthesis_mel(ph, mels, features=[(f0[b, :, : lens[b]].cpu(), emas[b]) for b in range(2)])
    assert mel_a.shape == mel_b.shape and bool(torch.isfinite(mel_a).all())
    assert float((mel_a - mel_b).abs().max()) <= 1e-6
    # both extractors attached: nothing but phonemes and the reference mel
    em = tts.attach_ema_extractor({"model": E.synth_ema_state_dict(seed=3407)})
    n_raw = torch.log(torch.exp(dense.to(cuda).unsqueeze(1) * 4 - 4).norm(dim=2))
    ema = em(f0, n_raw, dense, lengths=lens)
    mel_c = tts.synthesis_mel(ph, mels)
    mel_d = tts.synthesis_mel(ph, mels, features=[(f0[b, :, : lens[b]].cpu(), ema[b, :, : lens[b]].cpu()) for b in range(2)])
    assert bool(torch.isfinite(mel_c).all()) and float((mel_c - mel_d).abs().max()) <= 1e-6
    solo = tts.synthesis_mel(ph[1], mels[1])
    assert float((solo[0] - mel_c[1, :, : solo.shape[-1]]).abs().max()) <= 1e-5
