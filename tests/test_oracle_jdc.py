"""CPU: the JDCNet oracle (oracle/jdc.py) and the synthetic-checkpoint generator against outputs of the reference's own
JDCNet (tests/golden/jdc_*.npz, jdc_inventory.json; made by tests/golden/make_golden.py jdc)."""
import glob
import json
import os

import numpy as np
import torch

from artspeech_amd import jdc as J
from oracle import jdc as ojdc


def test_jdc_spec_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, "jdc_inventory.json")) as f:
        inv = json.load(f)
    assert {k: list(v) for k, v in J.jdc_spec(1).items()} == inv


def test_jdc_oracle_matches_reference(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "jdc_T*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        W = {k: torch.from_numpy(np.asarray(v)) for k, v in J.synth_jdc_state_dict(1, seed=int(g["seed"])).items()}
        f0 = ojdc.jdcnet(W, torch.from_numpy(g["mel"]))
        assert f0.shape == (1, int(g["t"]))
        d = float(np.abs(f0.numpy() - g["f0"]).max())
        assert d <= 2e-6, (f, d)
