"""CPU: the acoustic-path oracle (oracle/acoustic.py) and the checkpoint folding against outputs of
the reference's forward(step="test") (tests/golden/net_*.npz).  Tolerances: integer durations exact;
floating-point boundaries 2e-5 abs (observed <= 9e-6; the north-star bound for mels is 1e-4)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from artspeech_amd import synth
from artspeech_amd.spec import artsspeech_spec
from artspeech_amd.weights import DEFAULT_STATS, fold_state_dict, load_distribution
from oracle import acoustic

TOL = 2e-5


def test_param_inventory(golden_dir):
    inv = json.load(open(os.path.join(golden_dir, "param_inventory.json")))
    for tag, hd, di in (("full", 512, 64), ("tiny", 64, 8)):
        mine = {k: list(v["shape"]) for k, v in artsspeech_spec(hd, di).items()}
        assert mine == inv[tag]


def test_synth_is_deterministic():
    a = synth.hash_tensor("x.weight", (7, 5), 3407, 0.25)
    b = synth.hash_tensor("x.weight", (7, 5), 3407, 0.25)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert float(a.reshape(-1)[0]) == float(synth.hash_tensor("x.weight", (1,), 3407, 0.25)[0])
    assert np.abs(a).max() <= 0.25
    assert not np.array_equal(a, synth.hash_tensor("x.weight", (7, 5), 3408, 0.25))


def raw_features(t_ref, seed):
    mel, f0, ema = synth.synth_ref_features(t_ref, seed)
    f0_raw = (f0 * np.float32(DEFAULT_STATS["pitch"][3]) + np.float32(DEFAULT_STATS["pitch"][2])).astype(np.float32)
    ema_raw = (ema * np.asarray(DEFAULT_STATS["EMA"][3], np.float32)[:, None]
               + np.asarray(DEFAULT_STATS["EMA"][2], np.float32)[:, None]).astype(np.float32)
    return mel, f0_raw, ema_raw


_W = {}


def folded(hd, di, seed):
    key = (hd, di, seed)
    if key not in _W:
        _W[key] = fold_state_dict(synth.synth_state_dict(hd, di, seed=seed))
    return _W[key]


def _cases(golden_dir, tag):
    return sorted(glob.glob(os.path.join(golden_dir, f"net_{tag}_*.npz")))


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_oracle_matches_reference(golden_dir, tag):
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    files = _cases(golden_dir, tag)
    assert files
    for f in files:
        g = np.load(f)
        hd, di = int(g["hidden_dim"]), int(g["dim_in"])
        W = folded(hd, di, int(g["weight_seed"]))
        mel, f0_raw, ema_raw = raw_features(int(g["t_ref"]), int(g["seed"]))
        out = acoustic.forward_test(W, torch.from_numpy(g["tokens"]), torch.from_numpy(mel), torch.from_numpy(f0_raw),
                                    torch.from_numpy(ema_raw), load_distribution(DEFAULT_STATS))
        assert np.array_equal(out["pred_dur"].numpy(), g["ref/pred_dur"]), f
        for k in ("style", "duration", "F0", "N", "EMA", "mel", "t_en", "a_en", "f0_ext", "n_ext", "ema_ext"):
            if "ref/" + k in g.files:
                d = np.abs(out[k].numpy() - g["ref/" + k]).max()
                assert d <= TOL, (f, k, d)
        assert out["mel"].shape == (80, 2 * int(g["ref/pred_dur"].sum()))


def allin_extractor_weights(gain):
    """the extractors' seeded checkpoints as tests/golden/make_golden.py allin loads them into the reference (JDCNet's last Linear scaled)"""
    from artspeech_amd import ema as E
    from artspeech_amd import jdc as J
    jsd = {k: np.asarray(v) for k, v in J.synth_jdc_state_dict(1, seed=3407).items()}
    for k in ("classifier.weight", "classifier.bias"):
        jsd[k] = (jsd[k] * np.float32(gain)).astype(np.float32)
    esd = {k: np.asarray(v) for k, v in E.synth_ema_state_dict(seed=3407).items()}
    return jsd, esd


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_oracle_allin_matches_reference(golden_dir, tag):
    """forward(step="test") with the reference's real JDCNet and EMA_Predictor in the loop (models.py:426-449 feeding :356-371)"""
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    files = sorted(glob.glob(os.path.join(golden_dir, f"net_allin_{tag}_*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        W = folded(int(g["hidden_dim"]), int(g["dim_in"]), int(g["weight_seed"]))
        jsd, esd = allin_extractor_weights(float(g["jdc_classifier_gain"]))
        out = acoustic.forward_test_allin(W, {k: torch.from_numpy(v) for k, v in jsd.items()}, {k: torch.from_numpy(v) for k, v in esd.items()},
                                          torch.from_numpy(g["tokens"]), torch.from_numpy(g["mel_in"]), load_distribution(DEFAULT_STATS))
        assert np.array_equal(out["pred_dur"].numpy(), g["ref/pred_dur"]), f
        for k in ("f0_ext", "n_ext", "ema_ext", "style", "duration", "F0", "N", "EMA", "mel"):
            d = np.abs(out[k].numpy() - g["ref/" + k]).max()
            assert d <= TOL, (f, k, d)


def test_tokens_have_pad_ends():
    t = synth.synth_tokens(40, 1234)
    assert t[0] == 0 and t[-1] == 0 and t[1:-1].min() >= 1 and t.max() <= 177
