"""GPU: the HIP MAS kernel through the C ABI (as_mas_f32) against the oracle and the golden vectors.
Bit-exact: integer/0-1 results must be identical."""
import os

import numpy as np
import pytest
import torch

from artspeech_amd import mas, synth
from oracle import mas as omas

pytestmark = pytest.mark.gpu
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
@pytest.mark.parametrize("tie", ["move", "stay"])
def test_golden_small(cuda, small, name, tie):
    value = torch.from_numpy(small[name + "/value"]).to(cuda)
    mask = torch.from_numpy(small[name + "/mask"]).to(cuda)
    v0 = value.clone()
    path = mas.maximum_path(value, mask, tie=tie)
    assert torch.equal(value, v0), "input mutated"
    want_rows = small[name + ("/rows_v1" if tie == "move" else "/rows_v2")]
    assert np.array_equal(rows_of(path.cpu().numpy()), want_rows)
    want = omas.maximum_path_c(small[name + "/value"], small[name + "/mask"], tie == "move")
    assert np.array_equal(path.cpu().numpy(), want)
    out = mas.maximum_path_lens(value, torch.from_numpy(small[name + "/x_lens"]), torch.from_numpy(small[name + "/y_lens"]),
                                tie=tie, want=("dur", "rows"))
    assert np.array_equal(out["rows"].cpu().numpy(), want_rows)
    assert np.array_equal(out["dur"].cpu().numpy(), want.sum(-1).astype(np.int32))


@pytest.mark.parametrize("tie", ["move", "stay"])
def test_golden_large(cuda, golden_dir, tie):
    g = np.load(os.path.join(golden_dir, "mas_large.npz"))
    B, Tx, Ty = (int(v) for v in g["shape"])
    u = synth.hash_tensor(f"mas/{B}x{Tx}x{Ty}", (B, Tx, Ty), int(g["seed"]))
    value = torch.from_numpy((u * u).astype(np.float32)).to(cuda)
    out = mas.maximum_path_lens(value, torch.from_numpy(g["x_lens"]), torch.from_numpy(g["y_lens"]), tie=tie,
                                want=("path", "rows", "dur"))
    rows = out["rows"].cpu().numpy()
    assert np.array_equal(rows, g["rows_v1" if tie == "move" else "rows_v2"])
    path = out["path"]
    assert np.array_equal(path.sum((1, 2)).cpu().numpy().astype(np.int64), g["y_lens"].astype(np.int64))
    assert np.array_equal(rows_of(path.cpu().numpy()), rows)
    assert np.array_equal(out["dur"].cpu().numpy(), path.sum(-1).cpu().numpy().astype(np.int32))


@pytest.mark.parametrize("shape", [(3, 1, 1), (2, 64, 64), (2, 65, 130), (5, 130, 257), (2, 300, 301),
                                   (2, 700, 1403), (1, 1500, 1600), (3, 40, 99),
                                   # rows 16-byte aligned: the banded kernel (1, 2, 3, 6, 8 bands; batches that are not multiples of 8)
                                   (9, 130, 36), (3, 300, 260), (1, 700, 1000), (10, 129, 64), (2, 1024, 96), (17, 64, 4), (2, 5, 8)])
@pytest.mark.parametrize("tie", ["move", "stay"])
def test_random_shapes_vs_oracle(cuda, shape, tie):
    B, Tx, Ty = shape
    rng = np.random.default_rng(B * 1000 + Tx + Ty)
    value = rng.random((B, Tx, Ty), dtype=np.float32)
    value = (np.round(value * 8) / 8).astype(np.float32) if Tx % 2 == 0 else value    # force ties on some
    x_lens = np.array([max(1, Tx - 3 * i) for i in range(B)], np.int32)
    y_lens = np.array([max(int(x_lens[i]), Ty - 5 * i) for i in range(B)], np.int32)
    mask = omas.mask_from_lens(value.shape, x_lens, y_lens)
    want, wdur = omas.maximum_path_c(value, mask, tie == "move", want_dur=True)
    got = mas.maximum_path(torch.from_numpy(value).to(cuda), torch.from_numpy(mask).to(cuda), tie=tie)
    assert np.array_equal(got.cpu().numpy(), want)
    out = mas.maximum_path_lens(torch.from_numpy(value).to(cuda), torch.from_numpy(x_lens), torch.from_numpy(y_lens), tie=tie, want=("dur", "rows"))
    assert np.array_equal(out["dur"].cpu().numpy(), wdur)
    rows = out["rows"].cpu().numpy()
    for b in range(B):
        assert np.array_equal(rows[b, : y_lens[b]], want[b].argmax(0)[: y_lens[b]]) and (rows[b, y_lens[b]:] == -1).all()


def test_empty_and_zero_length(cuda):
    value = torch.rand(3, 9, 17, device=cuda)
    out = mas.maximum_path_lens(value, torch.tensor([9, 0, 4]), torch.tensor([17, 5, 0]), want=("path", "dur", "rows"))
    assert out["path"][1].sum() == 0 and out["path"][2].sum() == 0 and out["path"][0].sum() == 17
    assert (out["rows"][1] == -1).all()
    e = mas.maximum_path_lens(torch.zeros(0, 4, 4, device=cuda), torch.zeros(0), torch.zeros(0))
    assert e["path"].shape == (0, 4, 4)


def test_cpu_input_fails_loudly():
    from artspeech_amd._lib import HipLibraryError
    with pytest.raises(HipLibraryError):
        mas.maximum_path(torch.rand(1, 4, 8), torch.ones(1, 4, 8))


def test_forced_rows_per_lane(cuda, monkeypatch):
    """Every (rows-per-lane, waves) geometry of the one-workgroup kernel (the path of rows that are not 16-byte aligned; forced here)
    gives the same answer (multi-wave LDS mailbox path)."""
    monkeypatch.setenv("AS_MAS_IMPL", "one")
    rng = np.random.default_rng(5)
    value = rng.random((2, 200, 420), dtype=np.float32)
    mask = omas.mask_from_lens(value.shape, [200, 150], [420, 333])
    want = omas.maximum_path_c(value, mask, False)
    for r in ("1", "2", "4", "8", "16"):
        monkeypatch.setenv("AS_MAS_R", r)
        got = mas.maximum_path(torch.from_numpy(value).to(cuda), torch.from_numpy(mask).to(cuda))
        assert np.array_equal(got.cpu().numpy(), want), r


@pytest.mark.parametrize("tag", ["last", "dim1"])
def test_softmax_mas_fused(cuda, golden_dir, tag):
    """as_softmax_mas_f32 (train_second.py:181-185 from lengths): softmax within 1e-6 of the reference's, the path bit-exact
    with the oracle's MAS on the SAME probabilities (decisions compare fp32 sums, so the input must be identical), d_gt =
    path.sum(-1); on this fixture the path also equals the reference's own maximum_path1/2 output."""
    import os
    from artspeech_amd import mas
    from oracle import mas as omas
    g = np.load(os.path.join(golden_dir, "softmax_mas.npz"))
    feat, sl, ml, dim = torch.from_numpy(g[f"{tag}_feat"]), torch.from_numpy(g[f"{tag}_sl"]), torch.from_numpy(g[f"{tag}_ml"]), int(g[f"{tag}_dim"])
    for tie, key in (("stay", "path2"), ("move", "path1")):
        attn, path, dgt = mas.soft_maximum_path(feat.to(cuda), sl, ml, dim=dim, tie=tie)
        assert float((attn.cpu() - torch.from_numpy(g[f"{tag}_attn"])).abs().max()) <= 1e-6
        a = attn.cpu().numpy()
        want = omas.maximum_path_np(a, omas.mask_from_lens(a.shape, sl.numpy(), ml.numpy()), tie == "move")
        assert np.array_equal(path.cpu().numpy(), want.astype(np.float32))
        assert np.array_equal(dgt.cpu().numpy(), want.sum(-1).astype(np.int32))
        same = np.array_equal(path.cpu().numpy().astype(np.uint8), g[f"{tag}_{key}"])
        print(tag, tie, "path equals the reference's:", same)
        assert same
    assert np.array_equal(mas.soft_maximum_path(feat.to(cuda), sl, ml, dim=dim)[2].cpu().numpy(), g[f"{tag}_dgt2"])
