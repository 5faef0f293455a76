"""CPU: the EMA_Predictor oracle (oracle/ema.py) and the synthetic-checkpoint generator against outputs of the reference's
own EMA_Predictor (tests/golden/ema_*.npz, ema_inventory.json; made by tests/golden/make_golden.py ema)."""
import glob
import json
import os

import numpy as np
import torch

from artspeech_amd import ema as E
from oracle import ema as oema

_W = {}


def weights(seed):
    if seed not in _W:
        _W[seed] = {k: torch.from_numpy(np.asarray(v)) for k, v in E.synth_ema_state_dict(seed=seed).items()}
    return _W[seed]


def test_ema_spec_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, "ema_inventory.json")) as f:
        inv = json.load(f)
    assert {k: list(v) for k, v in E.ema_spec().items()} == inv


def test_positional_table_is_the_oracles():
    assert torch.equal(E.positional_encoding(300), oema.positional_encoding(300))


def test_ema_oracle_matches_reference(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "ema_T*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        out = oema.ema_predictor(weights(int(g["seed"])), torch.from_numpy(g["f0"]), torch.from_numpy(g["n"]), torch.from_numpy(g["mel"]))
        assert out.shape == (10, int(g["t"]))
        d = float(np.abs(out.numpy() - g["ema"]).max())
        assert d <= 2e-6, (f, d)
