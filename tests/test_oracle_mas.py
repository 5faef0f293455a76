"""CPU: the MAS oracle (numpy + C restatements) against outputs of the reference itself
(tests/golden/mas_*.npz, made by tests/golden/make_golden.py from S_monotonic_align.maximum_path1/2)."""
import os

import numpy as np
import pytest

from oracle import mas as omas
from artspeech_amd import synth

CASES = ["ragged_4x7x15", "c3_32x40x100", "ties_8x12x30", "tx1_2x1x9", "square_3x16x16", "neglogp_4x20x50"]


def rows_of(path):
    has = path.sum(1) > 0
    idx = path.argmax(1).astype(np.int32)
    idx[~has] = -1
    return idx


@pytest.fixture(scope="module")
def small(golden_dir):
    return np.load(os.path.join(golden_dir, "mas_small.npz"))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tie_move", [True, False])
def test_oracle_matches_reference_small(small, name, tie_move):
    value, mask = small[name + "/value"], small[name + "/mask"]
    want = small[name + ("/rows_v1" if tie_move else "/rows_v2")]
    got_np = omas.maximum_path_np(value, mask, tie_move)
    got_c, dur = omas.maximum_path_c(value, mask, tie_move, want_dur=True)
    assert np.array_equal(rows_of(got_np), want)
    assert np.array_equal(got_c, got_np)                      # bit-identical dense paths
    assert np.array_equal(dur, got_c.sum(-1).astype(np.int32))
    # exactly one 1 per valid column
    y_lens = small[name + "/y_lens"]
    assert np.array_equal(got_c.sum((1, 2)).astype(np.int64), y_lens.astype(np.int64))
    # mask_from_lens restatement
    assert np.array_equal(omas.mask_from_lens(value.shape, small[name + "/x_lens"], y_lens), mask)


def test_tie_cases_differ(small):
    assert (small["ties_8x12x30/rows_v1"] != small["ties_8x12x30/rows_v2"]).any()


def test_oracle_matches_reference_large(golden_dir):
    g = np.load(os.path.join(golden_dir, "mas_large.npz"))
    B, Tx, Ty = (int(v) for v in g["shape"])
    u = synth.hash_tensor(f"mas/{B}x{Tx}x{Ty}", (B, Tx, Ty), int(g["seed"]))
    value = (u * u).astype(np.float32)
    assert value.astype(np.float64).sum() == float(g["value_checksum"])
    mask = omas.mask_from_lens(value.shape, g["x_lens"], g["y_lens"])
    for tie_move, key in ((True, "rows_v1"), (False, "rows_v2")):
        got = omas.maximum_path_c(value, mask, tie_move)
        assert np.array_equal(rows_of(got), g[key])
    assert (g["rows_v1"] != g["rows_v2"]).any()


def test_inputs_not_mutated(small):
    value, mask = small["ragged_4x7x15/value"].copy(), small["ragged_4x7x15/mask"].copy()
    v0, m0 = value.copy(), mask.copy()
    omas.maximum_path_c(value, mask, True)
    omas.maximum_path_np(value, mask, False)
    assert np.array_equal(value, v0) and np.array_equal(mask, m0)
