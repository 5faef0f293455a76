"""GPU: failures a kernel can only discover at run time are reported, never silent (include/artspeech_hip.h: as_device_status), and
the plan's layout cache stays bounded without ever invalidating a layout in use.

The reference's counterparts raise: nn.Embedding on an id >= n_token (RelTransformerEnc.py:11-16); cuDNN's LSTM (models.py:555-561)
and the loops of S_monotonic_align.py:5-95 have no cross-workgroup protocol that could time out."""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

from artspeech_amd import _lib, models, ops
from artspeech_amd.ops import Layout, taps_1d
from test_net_gpu import get_model, raw_features

pytestmark = pytest.mark.gpu
LSTM_TIMEOUT, MAS_TIMEOUT, BAD_TOKEN, F16_RANGE, BAD_LAYOUT = range(5)


@pytest.fixture(autouse=True)
def clean_status(cuda):
    L = _lib.lib()
    L.as_device_status(1)
    yield
    L.as_bilstm_cluster_test_hooks(-1, 0)
    L.as_set_range_probe(0)
    torch.cuda.synchronize()
    L.as_device_status(1)


def tiny_call(net, golden_dir):
    g = np.load(sorted(glob.glob(os.path.join(golden_dir, "net_tiny_*.npz")))[0])
    tokens = torch.from_numpy(g["tokens"])[None]
    mel, f0_raw, ema_raw = raw_features(int(g["t_ref"]), int(g["seed"]))
    batch = [tokens, torch.tensor([tokens.shape[1]]), torch.from_numpy(mel)[None], torch.tensor([mel.shape[1]])]
    feats = (torch.from_numpy(f0_raw)[None], torch.from_numpy(ema_raw)[None])
    return g, lambda: net(batch, None, None, step="test", features=feats)


def test_status_is_sticky_and_gates_the_module_entry_points(cuda, golden_dir):
    L = _lib.lib()
    net = get_model(64, 8, 3407, cuda)
    g, call = tiny_call(net, golden_dir)
    ref = call()
    assert L.as_device_status(0) == 0
    for kind in range(4):
        assert L.as_device_status_raise_for_test(kind, _lib.stream()) == 0
        torch.cuda.synchronize()
        assert L.as_device_status(0) == 1 << kind
        assert L.as_device_status(0) == 1 << kind                      # sticky
        with pytest.raises(_lib.HipLibraryError, match="kernel reported"):
            call()
        assert L.as_device_status(1) == 1 << kind                      # read and clear
        assert L.as_device_status(0) == 0
        assert torch.equal(call(), ref)
    assert L.as_device_status_raise_for_test(7, _lib.stream()) == -1


def test_bad_token_at_the_c_boundary(cuda):
    """the Python surface raises IndexError before any launch (models.py); a C host that hands over an id >= n_token gets the bit"""
    L = _lib.lib()
    C, V = 64, 178
    emb = torch.randn(V, C, device=cuda)
    for tok, bad in (([1, 5, 177], False), ([1, 178, 3], True), ([-1], True)):
        t = torch.tensor(tok, dtype=torch.int32, device=cuda)
        y = torch.empty(C, len(tok), device=cuda)
        assert L.as_embed_f32(t.data_ptr(), emb.data_ptr(), C, len(tok), V, 1.0, y.data_ptr(), len(tok), _lib.stream()) == 0
        torch.cuda.synchronize()
        assert L.as_device_status(1) == ((1 << BAD_TOKEN) if bad else 0), tok


def test_clustered_lstm_timeout_is_reported(cuda):
    """a cluster member that never shows up: its peers give up after the (shortened) spin bound, the launch ends, the bit is up;
    with the member back the same buffers give the right answer again"""
    L = _lib.lib()
    H, I, lens = 256, 64, [40, 12]
    g = torch.Generator().manual_seed(5)
    lstm = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True)
    W = models.Weights({"l." + k: v.detach() for k, v in lstm.state_dict().items()}, cuda)
    xchg = ops.bilstm_exchange_buffer(1, len(lens), cuda)
    lay = Layout(lens, cuda)
    xs = [torch.randn(I, n, generator=g) for n in lens]
    x = torch.cat(xs, 1).to(cuda)
    with torch.no_grad():
        want = torch.cat([lstm(v.t()[None])[0][0].t() for v in xs], 1)
    assert L.as_bilstm_cluster_test_hooks(3, 2000) == 0
    models.bilstm(W, "l", x, lay, xchg)
    torch.cuda.synchronize()
    assert L.as_device_status(0) == 1 << LSTM_TIMEOUT
    assert L.as_bilstm_cluster_test_hooks(-1, 0) == 0
    L.as_device_status(1)
    xchg.zero_()                                                        # (the dropped member's epoch words are stale: a fresh buffer, as after any failure)
    for _ in range(2):
        out = models.bilstm(W, "l", x, lay, xchg)
        assert float((out.cpu() - want).abs().max()) <= 2e-5
    assert L.as_device_status(0) == 0


@pytest.mark.parametrize("lens", [[1], [1, 1, 1, 1], [1, 7, 1]])
def test_clustered_lstm_one_step_sequences_replayed(cuda, lens):
    """Lmax == 1 clusters poll nothing during the recurrence: the epoch hand-over at the end must still keep a late member's words out
    of the next launch (many back-to-back launches on one exchange buffer, changing inputs)"""
    H, I = 256, 64
    g = torch.Generator().manual_seed(11)
    lstm = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True)
    W = models.Weights({"l." + k: v.detach() for k, v in lstm.state_dict().items()}, cuda)
    xchg = ops.bilstm_exchange_buffer(1, len(lens), cuda)
    lay = Layout(lens, cuda)
    outs, wants = [], []
    for rep in range(12):
        xs = [torch.randn(I, n, generator=g) for n in lens]
        with torch.no_grad():
            wants.append(torch.cat([lstm(v.t()[None])[0][0].t() for v in xs], 1))
        outs.append(models.bilstm(W, "l", torch.cat(xs, 1).to(cuda), lay, xchg).clone())
    for rep, (o, w) in enumerate(zip(outs, wants)):
        assert float((o.cpu() - w).abs().max()) <= 2e-5, rep
    assert _lib.lib().as_device_status(0) == 0


def test_range_probe_names_an_operand_beyond_fp16(cuda):
    L = _lib.lib()
    lens, cin, cout = [50, 13], 64, 128
    g = torch.Generator().manual_seed(1)
    w = torch.randn(cout, cin, 3, generator=g) / 14
    x = torch.randn(cin, sum(lens), generator=g)
    lay = Layout(lens, cuda)
    wt = ops.prep_weight(w, cuda)
    big = x.clone()
    big[7, 20] = 7.0e4                                                  # > 65504: h = inf, l = -inf
    y = ops.conv_gemm(wt, big.to(cuda), lay, lay.new(cout), taps_1d(3))
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(y).all())                            # loud in the values ...
    assert L.as_device_status(1) == 0                                   # ... but unreported while the probe is off
    L.as_set_range_probe(1)
    ops.conv_gemm(wt, x.to(cuda), lay, lay.new(cout), taps_1d(3))
    torch.cuda.synchronize()
    assert L.as_device_status(1) == 0
    ops.conv_gemm(wt, big.to(cuda), lay, lay.new(cout), taps_1d(3))
    torch.cuda.synchronize()
    assert L.as_device_status(1) == 1 << F16_RANGE


def test_post_max_w_smaller_than_an_utterance_is_reported(cuda):
    """as_conv_gemm_multi_post_f32: the reduction kernel that also writes the AdaIN image holds 64 NJ columns of an utterance, NJ chosen
    from the caller's post_max_w; an utterance wider than that raises AS_STATUS_BAD_LAYOUT instead of dropping its last columns."""
    L = _lib.lib()
    lens, cin, cout = [150], 512, 512
    g = torch.Generator().manual_seed(2)
    lay = Layout(lens, cuda)
    w = torch.randn(cout, cin, 3, generator=g) / 40
    xs = ops.split_act(torch.randn(cin, lay.N, generator=g).to(cuda), lay)
    gb = torch.randn(2 * cout, 1, generator=g).to(cuda)
    for claimed, bad in ((150, False), (64, True)):
        d = []
        ops.conv_gemm(ops.prep_weight(w, cuda), None, lay, lay.new(cout), taps_1d(3), xs=xs, K=cin, defer=d)
        ops.conv_gemm_multi_post(d, [(gb, 1, lay, ops.new_image(cout, lay.N, cuda), None, claimed)])
        torch.cuda.synchronize()
        assert L.as_device_status(1) == ((1 << BAD_LAYOUT) if bad else 0), claimed


def test_non_finite_mel_is_always_reported(cuda, golden_dir):
    """The debug probe is OFF: the path's last conv (`to_out`) still tests its accumulators, and inf / NaN anywhere upstream reaches it --
    an articulatory feature beyond fp16's range (the decoder's F0 / N / EMA convs and the towers read it), a NaN in the reference mel.
    The bit is up once the stream is synchronised, the next call is refused, and after the clear the same call is healthy again.
    (The reference's `int(pred_dur[i])` raises on a NaN duration, models.py:363-366: with predicted durations the duration kernel
    raises the same bit before the frame counts are read back.)"""
    L = _lib.lib()
    net = get_model(64, 8, 3407, cuda)
    g = np.load(sorted(glob.glob(os.path.join(golden_dir, "net_tiny_*.npz")))[0])
    n = len(g["tokens"])
    mel, f0_raw, ema_raw = raw_features(int(g["t_ref"]), int(g["seed"]))
    tok = torch.from_numpy(g["tokens"]).to(cuda, torch.int32)
    forced = torch.full((n,), 2, dtype=torch.int32, device=cuda)

    def call(mel_, ema_, **kw):
        return net.forward_packed(tok, [n], torch.from_numpy(mel_).to(cuda), torch.from_numpy(f0_raw).reshape(1, -1).to(cuda),
                                  torch.from_numpy(ema_).to(cuda), [int(g["t_ref"])], **kw)["mel"].clone()
    ref = call(mel, ema_raw, forced=forced, frames_hint=[2 * n])
    torch.cuda.synchronize()
    assert L.as_device_status(0) == 0 and bool(torch.isfinite(ref).all())
    bad_mel = mel.copy()
    bad_mel[3, 17] = np.nan
    bad_ema = ema_raw.copy()
    bad_ema[2, 5] = np.inf
    for m_, e_ in ((bad_mel, ema_raw), (mel, bad_ema)):
        out = call(m_, e_, forced=forced, frames_hint=[2 * n])
        torch.cuda.synchronize()
        assert not bool(torch.isfinite(out).all())
        assert L.as_device_status(0) == 1 << F16_RANGE
        with pytest.raises(_lib.HipLibraryError, match="kernel reported"):
            call(mel, ema_raw, forced=forced, frames_hint=[2 * n])
        assert L.as_device_status(1) == 1 << F16_RANGE
        assert torch.equal(call(mel, ema_raw, forced=forced, frames_hint=[2 * n]), ref)
    # predicted durations (the duration predictor is styled by the articulatory track, models.py:543-548): the call itself reads the
    # frame counts back, and refuses
    with pytest.raises(_lib.HipLibraryError, match="kernel reported"):
        call(mel, bad_ema)
    assert L.as_device_status(1) & (1 << F16_RANGE)


def test_layout_cache_stays_bounded_and_valid(cuda):
    """more geometries than the cap through ONE plan: the cache is dropped between calls (never inside one), results stay those of a
    fresh plan, and the flush count says it happened"""
    L = _lib.lib()
    net = get_model(64, 8, 3407, cuda).replica()                        # a plan of its own
    rt = net.rt
    assert L.as_plan_set_layout_cap(rt.plan, 8) == -1
    assert L.as_plan_set_layout_cap(rt.plan, 64) == 0
    enc = net.text_encoder
    g = torch.Generator().manual_seed(3)
    first = {}
    for rep in range(2):
        for n in range(5, 75):
            lens = [n, 1 + n % 7, 3]
            tok = torch.randint(1, 178, (sum(lens),), generator=torch.Generator().manual_seed(n), dtype=torch.int32).to(cuda)
            y = enc.forward_packed(tok, lens)
            if rep == 0:
                first[n] = y.clone()
            else:
                assert torch.equal(y, first[n]), n
    assert L.as_plan_layout_flushes(rt.plan) >= 2
    assert L.as_device_status(0) == 0


def test_output_reuse_follows_the_frame_count(cuda, golden_dir):
    """forward_packed(out=...) reuses a tensor only when its shape is what the call needs (predicted durations depend on the input
    values, not only on the lengths)"""
    net = get_model(64, 8, 3407, cuda)
    g = np.load(sorted(glob.glob(os.path.join(golden_dir, "net_tiny_*.npz")))[0])
    n = len(g["tokens"])
    mel, f0_raw, ema_raw = raw_features(int(g["t_ref"]), int(g["seed"]))
    tok = torch.from_numpy(g["tokens"]).to(cuda, torch.int32)
    args = (tok, [n], torch.from_numpy(mel).to(cuda), torch.from_numpy(f0_raw).reshape(1, -1).to(cuda), torch.from_numpy(ema_raw).to(cuda),
            [int(g["t_ref"])])
    short = torch.ones(n, dtype=torch.int32, device=cuda)
    long_ = torch.full((n,), 3, dtype=torch.int32, device=cuda)
    out = net.forward_packed(*args, forced=short, frames_hint=[n])
    m1 = out["mel"]
    assert m1.shape[1] == 2 * n
    out2 = net.forward_packed(*args, forced=long_, frames_hint=[3 * n], out=out)
    assert out2["mel"].shape[1] == 6 * n and out2["mel"].data_ptr() != m1.data_ptr()
    out3 = net.forward_packed(*args, forced=long_, frames_hint=[3 * n], out=out2)
    assert out3["mel"].data_ptr() == out2["mel"].data_ptr()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out3["mel"]).all())


def test_utterance_wider_than_the_descriptors(cuda):
    """Column descriptors hold 22-bit columns (AS_META_PACK): the Python layout refuses a wider utterance; the C entry point, whose
    widths are device data, raises AS_STATUS_BAD_LAYOUT from the kernel."""
    with pytest.raises(ValueError):
        Layout([ops.META_MAX_W + 1], cuda)
    L = _lib.lib()
    widths = torch.tensor([ops.META_MAX_W + 1, 7], dtype=torch.int32, device=cuda)
    off = torch.tensor([0, ops.META_MAX_W + 1, ops.META_MAX_W + 8], dtype=torch.int32, device=cuda)
    meta = torch.zeros(ops.META_MAX_W + 8, dtype=torch.int64, device=cuda)
    assert L.as_make_meta(ops._p(widths), ops._p(off), 2, 1, int(off[-1]), ops._p(meta), ops.stream()) == 0
    torch.cuda.synchronize()
    assert L.as_device_status(1) == 1 << BAD_LAYOUT
