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
