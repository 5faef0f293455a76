"""GPU: the HiFi-GAN generator on the HIP path (artspeech_amd/vocoder.py, through the C ABI) against outputs of the
reference's Generator (tests/golden/voc_*.npz): waveform within 1e-5 abs (samples are O(0.05) with the synthetic
weights; observed values are printed)."""
import glob
import os

import numpy as np
import pytest
import torch

from artspeech_amd import vocoder as V

pytestmark = pytest.mark.gpu
TOL = 1e-5
_GEN = {}


def gen(c0, cuda):
    if c0 not in _GEN:
        h = dict(V.DEFAULT_H, upsample_initial_channel=c0)
        _GEN[c0] = V.Generator(h, device=cuda).load_state_dict(V.synth_generator_state_dict(h, seed=3407))
    return _GEN[c0]


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_generator_matches_reference(cuda, golden_dir, tag):
    files = sorted(glob.glob(os.path.join(golden_dir, f"voc_{tag}_*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        wav = gen(int(g["c0"]), cuda)(torch.from_numpy(g["mel"])[None])
        assert wav.shape == (1, 1, 300 * int(g["t"]))
        d = float(np.abs(wav[0, 0].cpu().numpy() - g["wav"]).max())
        print(os.path.basename(f), "wav max-abs", d)
        assert d <= TOL, (f, d)


def test_ragged_batch_equals_single(cuda, golden_dir):
    gs = [np.load(f) for f in sorted(glob.glob(os.path.join(golden_dir, "voc_tiny_*.npz")))]
    net = gen(32, cuda)
    tmax = max(int(g["t"]) for g in gs)
    mel = torch.zeros(len(gs), 80, tmax)
    for b, g in enumerate(gs):
        mel[b, :, : int(g["t"])] = torch.from_numpy(g["mel"])
    wav = net(mel, lengths=[int(g["t"]) for g in gs])
    assert wav.shape == (len(gs), 1, 300 * tmax)
    for b, g in enumerate(gs):
        n = 300 * int(g["t"])
        assert float(np.abs(wav[b, 0, :n].cpu().numpy() - g["wav"]).max()) <= TOL
        assert float(wav[b, 0, n:].abs().max()) == 0.0 if n < 300 * tmax else True


def test_pipeline_mel_to_wav(cuda, golden_dir):
    """test.py's chain on the HIP path: phonemes + reference mel -> acoustic model -> generator -> samples."""
    import json
    from artspeech_amd import synth
    from artspeech_amd.pipeline import ArtSpeech
    from test_net_gpu import raw_features
    tts = ArtSpeech(config={"model_params": {"hidden_dim": 64, "dim_in": 8, "max_conv_dim": 64}},
                    checkpoint={"net": {"ArtsSpeech": synth.synth_state_dict(64, 8, seed=3407)}}, device=cuda)
    h = dict(V.DEFAULT_H, upsample_initial_channel=32)
    tts.attach_vocoder(h, V.synth_generator_state_dict(h, seed=3407))
    with open(os.path.join(golden_dir, "text_golden.json"), encoding="utf-8") as f:
        cases = json.load(f)["cases"]
    ph = [cases[0]["text"][:30], cases[1]["text"][:18]]
    mels, feats = [], []
    for i, t in enumerate((90, 70)):
        mel, f0_raw, ema_raw = raw_features(t, 40 + i)
        mels.append(mel)
        feats.append((f0_raw, ema_raw))
    wav = tts.synthesis_wav(ph, mels, features=feats)
    frames = tts._last_frames
    assert wav.shape == (2, 300 * max(frames)) and bool(torch.isfinite(wav).all())
    solo = tts.synthesis_wav(ph[1], mels[1], features=feats[1])
    assert solo.shape == (300 * frames[1],)
    assert float((solo - wav[1, : solo.shape[0]]).abs().max()) <= 1e-5


def test_long_utterance_matches_oracle(cuda):
    """230 mel frames = 69 000 samples: past 2^16 columns at the last stage (the column descriptors carry 22-bit positions), several
    column tiles of the fused residual steps per utterance, beside a short utterance in the same batch.  Against the CPU restatement
    of the reference's Generator (oracle/vocoder.py, pinned by tests/test_oracle_vocoder.py)."""
    from artspeech_amd.synth import hash_tensor
    from artspeech_amd.weights import fold_state_dict
    from oracle import vocoder as OV
    h = dict(V.DEFAULT_H)
    sd = V.synth_generator_state_dict(h, seed=3407)
    Wf = {k: v.float() for k, v in fold_state_dict(sd).items()}
    lens = [230, 9]
    mel = torch.zeros(2, 80, max(lens))
    for b, t in enumerate(lens):
        mel[b, :, :t] = torch.from_numpy(hash_tensor("voc/long%d" % b, (80, t), 77, 1.0))
    wav = gen(512, cuda)(mel, lengths=lens).cpu()
    for b, t in enumerate(lens):
        want = OV.generator(Wf, h, mel[b, :, :t])
        d = float((wav[b, 0, : 300 * t] - want).abs().max())
        print("frames", t, "wav max-abs", d)
        assert d <= TOL, (t, d)
