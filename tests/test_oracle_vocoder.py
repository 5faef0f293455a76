"""CPU: the vocoder oracle (oracle/vocoder.py) and the synthetic-checkpoint generator against outputs of the
reference's own HiFi-GAN Generator (tests/golden/voc_*.npz, vocoder_inventory.json)."""
import glob
import json
import os

import numpy as np
import torch

from artspeech_amd import vocoder as V
from artspeech_amd.weights import fold_state_dict
from oracle import vocoder as ovoc


def test_generator_spec_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, "vocoder_inventory.json")) as f:
        inv = json.load(f)
    for tag, c0 in (("tiny", 32), ("full", 512)):
        assert {k: list(v) for k, v in V.generator_spec({"upsample_initial_channel": c0}).items()} == inv[tag]


def test_vocoder_oracle_matches_reference(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "voc_*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        h = dict(V.DEFAULT_H, upsample_initial_channel=int(g["c0"]))
        W = fold_state_dict(V.synth_generator_state_dict(h, seed=int(g["seed"])))
        wav = ovoc.generator(W, h, torch.from_numpy(g["mel"]))
        assert wav.shape == (300 * int(g["t"]),)
        d = float(np.abs(wav.numpy() - g["wav"]).max())
        assert d <= 2e-6, (f, d)
