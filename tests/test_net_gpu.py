"""GPU: the HIP acoustic path through the C ABI against (a) outputs of the reference itself
(tests/golden/net_*.npz) and (b) the CPU oracle.  Integer durations must be identical; mel frames within
1e-4 abs (the north-star bound; observed values are printed)."""
import glob
import os

import numpy as np
import pytest
import torch

from artspeech_amd import _lib, models, synth
from artspeech_amd.weights import DEFAULT_STATS, fold_state_dict, load_distribution

pytestmark = pytest.mark.gpu
MEL_TOL = 1e-4
AUX_TOL = 5e-5


def raw_features(t_ref, seed):
    mel, f0, ema = synth.synth_ref_features(t_ref, seed)
    f0_raw = (f0 * np.float32(DEFAULT_STATS["pitch"][3]) + np.float32(DEFAULT_STATS["pitch"][2])).astype(np.float32)
    ema_raw = (ema * np.asarray(DEFAULT_STATS["EMA"][3], np.float32)[:, None]
               + np.asarray(DEFAULT_STATS["EMA"][2], np.float32)[:, None]).astype(np.float32)
    return mel, f0_raw, ema_raw


_MODELS = {}


def get_model(hd, di, seed, device):
    key = (hd, di, seed)
    if key not in _MODELS:
        sd = synth.synth_state_dict(hd, di, seed=seed)
        m = models.build_model(models.Munch(hidden_dim=hd, dim_in=di, style_dim=256, n_mels=80, n_token=178, n_layer=3,
                                            max_conv_dim=hd, dropout=0.2), None, stage="second",
                               distribution=load_distribution(DEFAULT_STATS), device=device)
        models.load_checkpoint(m, None, {"net": {"ArtsSpeech": sd}})
        _MODELS[key] = m.ArtsSpeech
    return _MODELS[key]


def run_one(net, g):
    tokens = torch.from_numpy(g["tokens"])[None]
    mel, f0_raw, ema_raw = raw_features(int(g["t_ref"]), int(g["seed"]))
    batch = [tokens, torch.tensor([tokens.shape[1]]), torch.from_numpy(mel)[None], torch.tensor([mel.shape[1]]), None, None, None]
    out, aux = net(batch, None, None, step="test", features=(torch.from_numpy(f0_raw)[None], torch.from_numpy(ema_raw)[None]),
                   return_aux=True)
    return out, aux


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_forward_matches_reference(cuda, golden_dir, tag):
    files = sorted(glob.glob(os.path.join(golden_dir, f"net_{tag}_*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        net = get_model(int(g["hidden_dim"]), int(g["dim_in"]), int(g["weight_seed"]), cuda)
        out, aux = run_one(net, g)
        dur = aux["dur_i"][: len(g["tokens"])].cpu().numpy()
        assert np.array_equal(dur, g["ref/pred_dur"].astype(np.int32)), (f, "durations")
        rep = {}
        for k, ref_k in (("style", "style"), ("duration", "duration"), ("F0", "F0"), ("N", "N"), ("EMA", "EMA")):
            got = aux[k].cpu().numpy().reshape(g["ref/" + ref_k].shape)
            rep[k] = float(np.abs(got - g["ref/" + ref_k]).max())
            assert rep[k] <= AUX_TOL, (f, k, rep[k])
        assert out.shape == (1, 80, 2 * int(g["ref/pred_dur"].sum()))
        d = float(np.abs(out[0].cpu().numpy() - g["ref/mel"]).max())
        print(os.path.basename(f), "mel max-abs", d, rep)
        assert d <= MEL_TOL, (f, d)


def test_batched_equals_single(cuda, golden_dir):
    """A ragged batch gives, per utterance, the reference's B=1 result."""
    files = sorted(glob.glob(os.path.join(golden_dir, "net_tiny_*.npz")))
    gs = [np.load(f) for f in files]
    net = get_model(64, 8, int(gs[0]["weight_seed"]), cuda)
    B = len(gs)
    nmax = max(len(g["tokens"]) for g in gs)
    tmax = max(int(g["t_ref"]) for g in gs)
    texts = torch.zeros(B, nmax, dtype=torch.long)
    mels = torch.zeros(B, 80, tmax)
    f0s = torch.zeros(B, 1, tmax)
    emas = torch.zeros(B, 10, tmax)
    for b, g in enumerate(gs):
        n, t = len(g["tokens"]), int(g["t_ref"])
        mel, f0_raw, ema_raw = raw_features(t, int(g["seed"]))
        texts[b, :n] = torch.from_numpy(g["tokens"])
        mels[b, :, :t] = torch.from_numpy(mel)
        f0s[b, :, :t] = torch.from_numpy(f0_raw)
        emas[b, :, :t] = torch.from_numpy(ema_raw)
    tl = torch.tensor([len(g["tokens"]) for g in gs])
    ml = torch.tensor([int(g["t_ref"]) for g in gs])
    out = net([texts, tl, mels, ml], None, None, step="test", features=(f0s, emas))
    for b, g in enumerate(gs):
        M2 = 2 * int(g["ref/pred_dur"].sum())
        d = float(np.abs(out[b, :, :M2].cpu().numpy() - g["ref/mel"]).max())
        assert d <= MEL_TOL, (b, d)
        assert float(out[b, :, M2:].abs().max()) == 0.0 if out.shape[2] > M2 else True


def test_forced_durations(cuda, golden_dir):
    g = np.load(sorted(glob.glob(os.path.join(golden_dir, "net_tiny_*.npz")))[0])
    net = get_model(64, 8, int(g["weight_seed"]), cuda)
    tokens = torch.from_numpy(g["tokens"])[None]
    mel, f0_raw, ema_raw = raw_features(int(g["t_ref"]), int(g["seed"]))
    batch = [tokens, torch.tensor([tokens.shape[1]]), torch.from_numpy(mel)[None], torch.tensor([mel.shape[1]])]
    forced = [g["ref/pred_dur"].astype(np.int64)]
    out = net(batch, None, None, step="test", features=(torch.from_numpy(f0_raw)[None], torch.from_numpy(ema_raw)[None]),
              forced_durations=forced)
    assert float(np.abs(out[0].cpu().numpy() - g["ref/mel"]).max()) <= MEL_TOL


def test_submodule_surfaces(cuda, golden_dir):
    """The reference's sub-module call surface on padded [B,C,L] tensors."""
    from oracle import acoustic
    g = np.load(os.path.join(golden_dir, "net_tiny_N12_T70_s1.npz"))
    net = get_model(64, 8, int(g["weight_seed"]), cuda)
    W = fold_state_dict(synth.synth_state_dict(64, 8, seed=int(g["weight_seed"])))
    tokens = torch.from_numpy(g["tokens"])
    enc = net.text_encoder(tokens[None], torch.tensor([len(tokens)]))
    assert enc.shape == (1, len(tokens), 64)
    assert float(np.abs(enc[0].t().cpu().numpy() - g["ref/t_en"]).max()) <= AUX_TOL
    style = torch.from_numpy(g["ref/style"])[None]
    M = int(g["ref/pred_dur"].sum())
    a_ex = acoustic.expand(torch.from_numpy(g["ref/a_en"]), torch.from_numpy(g["ref/pred_dur"]))
    f0, n, ema = net.artsPredictor(a_ex[None], style)
    assert f0.shape == (1, 1, 2 * M) and ema.shape == (1, 10, 2 * M)
    assert float(np.abs(f0[0].cpu().numpy() - g["ref/F0"]).max()) <= AUX_TOL
    t_ex = acoustic.expand(torch.from_numpy(g["ref/t_en"]), torch.from_numpy(g["ref/pred_dur"]))
    mel = net.decoder(t_ex[None], style, torch.from_numpy(g["ref/F0"])[None], torch.from_numpy(g["ref/N"])[None],
                      torch.from_numpy(g["ref/EMA"])[None])
    assert float(np.abs(mel[0].cpu().numpy() - g["ref/mel"]).max()) <= MEL_TOL
    dur = net.durationPredictor(tokens[None], torch.from_numpy(g["ref/ema_ext"])[None], torch.tensor([len(tokens)]),
                                torch.tensor([int(g["t_ref"])]))
    assert float(np.abs(dur[0].cpu().numpy() - g["ref/duration"]).max()) <= AUX_TOL


def test_long_form_c5(cuda):
    """BASELINE config 5: 1024 phonemes -> 2048 mel frames, batch 8 (attention over 16 key tiles, 2048-step
    LSTMs, K1-sized lattices).  Forced all-ones durations as SURVEY.md section 8 prescribes; two of the eight
    utterances are checked against the oracle (CPU time), the rest through batch-independence."""
    from oracle import acoustic
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    hd, di, seed = 512, 64, 3407
    net = get_model(hd, di, seed, cuda)
    W = fold_state_dict(synth.synth_state_dict(hd, di, seed=seed))
    dist = load_distribution(DEFAULT_STATS)
    B, N, T = 8, 1024, 200
    toks = [synth.synth_tokens(N, 100 + b) for b in range(B)]
    feats = [raw_features(T, 100 + b) for b in range(B)]
    texts = torch.from_numpy(np.stack(toks))
    mels = torch.from_numpy(np.stack([f[0] for f in feats]))
    f0s = torch.from_numpy(np.stack([f[1] for f in feats]))
    emas = torch.from_numpy(np.stack([f[2] for f in feats]))
    forced = [np.ones(N, np.int64)] * B
    out = net([texts, torch.full((B,), N), mels, torch.full((B,), T)], None, None, step="test", features=(f0s, emas),
              forced_durations=forced)
    assert out.shape == (B, 80, 2 * N)
    for b in (0, 5):
        ref = acoustic.forward_test(W, torch.from_numpy(toks[b]), torch.from_numpy(feats[b][0]), torch.from_numpy(feats[b][1]),
                                    torch.from_numpy(feats[b][2]), dist, forced_dur=forced[b])
        d = float((out[b].cpu() - ref["mel"]).abs().max())
        print("C5 utterance", b, "mel max-abs", d)
        assert d <= MEL_TOL, (b, d)
    # batch independence: utterance 3 alone == utterance 3 inside the batch
    solo = net([texts[3:4], torch.tensor([N]), mels[3:4], torch.tensor([T])], None, None, step="test",
               features=(f0s[3:4], emas[3:4]), forced_durations=forced[3:4])
    # (not bitwise: the GEMM's tile / split-K choice depends on the batch's total column count)
    assert float((solo[0] - out[3]).abs().max()) <= 5e-5


def test_pipeline_surface(cuda, golden_dir):
    """test.py's ArtSpeech class, acoustic part: phoneme string + reference mel -> mel, batched."""
    import json
    from artspeech_amd.pipeline import ArtSpeech
    sd = synth.synth_state_dict(64, 8, seed=3407)
    tts = ArtSpeech(config={"model_params": {"hidden_dim": 64, "dim_in": 8, "max_conv_dim": 64}},
                    checkpoint={"net": {"ArtsSpeech": sd}}, device=cuda)
    with open(os.path.join(golden_dir, "text_golden.json"), encoding="utf-8") as f:
        g = json.load(f)
    ph = [g["cases"][0]["text"][:40], g["cases"][1]["text"][:25]]
    mels, feats = [], []
    for i, t in enumerate((90, 70)):
        mel, f0_raw, ema_raw = raw_features(t, 40 + i)
        mels.append(mel)
        feats.append((f0_raw, ema_raw))
    out = tts.synthesis_mel(ph, mels, features=feats)
    assert out.shape[0] == 2 and out.shape[1] == 80 and bool(torch.isfinite(out).all())
    solo = tts.synthesis_mel(ph[1], mels[1], features=feats[1])
    assert float((solo[0] - out[1, :, : solo.shape[2]]).abs().max()) <= 5e-5
    with pytest.raises(NotImplementedError):
        tts.synthesis("hello", "ref.wav", "out.wav")


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_allin_surface_matches_reference(cuda, golden_dir, tag):
    """The whole test.py:113 surface against the reference itself: pipeline.ArtSpeech with the HIP JDCNet and EMA_Predictor attached,
    phonemes + reference mel in, nothing else.  The fixture (tests/golden/net_allin_*.npz, make_golden.py allin) is the reference's
    ArtsSpeech.forward(step="test") with its REAL extractors in the loop (models.py:426-449 feeding :356-371): log_norm ->
    pitch_extractor(mel.unsqueeze(1)) -> ema_extractor(f0_ext, n_ext, mel) -> stats normalisation -> style towers, duration
    predictor, predictors, decoder.  Durations identical, features / style within 5e-5, mel within 1e-4."""
    from artspeech_amd.pipeline import ArtSpeech
    from artspeech_amd.text import symbols
    from test_oracle_golden import allin_extractor_weights
    files = sorted(glob.glob(os.path.join(golden_dir, f"net_allin_{tag}_*.npz")))
    assert files
    tts = None
    for f in files:
        g = np.load(f)
        hd, di = int(g["hidden_dim"]), int(g["dim_in"])
        if tts is None:
            tts = ArtSpeech(config={"model_params": {"hidden_dim": hd, "dim_in": di, "max_conv_dim": hd}},
                            checkpoint={"net": {"ArtsSpeech": synth.synth_state_dict(hd, di, seed=int(g["weight_seed"]))}}, device=cuda)
            jsd, esd = allin_extractor_weights(float(g["jdc_classifier_gain"]))
            tts.attach_pitch_extractor({"net": jsd})
            tts.attach_ema_extractor({"model": esd})
        tokens = [int(t) for t in g["tokens"]]
        # through the text front end when the ids survive the round trip (the table holds one character twice: test.py:19-38)
        text = "".join(symbols[t] for t in tokens)
        net = tts.model.ArtsSpeech
        if tts.textcleaner(text) == tokens:
            mel = tts.synthesis_mel(text, g["mel_in"])
            assert tts._last_frames == [2 * int(g["ref/pred_dur"].sum())]
            d = float((mel[0].cpu() - torch.from_numpy(g["ref/mel"])).abs().max())
            assert d <= MEL_TOL, (f, "pipeline mel", d)
        # and the model call itself, test.py:113, with the module-boundary tensors
        batch = [torch.tensor(tokens)[None], torch.tensor([len(tokens)]), torch.from_numpy(g["mel_in"])[None], torch.tensor([g["mel_in"].shape[-1]]),
                 None, None, None]
        out, aux = net(batch, None, None, step="test", return_aux=True)
        assert np.array_equal(aux["dur_i"][: len(tokens)].cpu().numpy(), g["ref/pred_dur"].astype(np.int32)), f
        f0_ext, n_ext, ema_ext, style = net.style_encoder(batch[2], batch[3])
        worst = {}
        for k, v in (("f0_ext", f0_ext[0]), ("n_ext", n_ext[0]), ("ema_ext", ema_ext[0]), ("style", style[0])):
            worst[k] = float((v.cpu() - torch.from_numpy(g["ref/" + k])).abs().max())
            assert worst[k] <= AUX_TOL, (f, k, worst[k])
        d = float((out[0].cpu() - torch.from_numpy(g["ref/mel"])).abs().max())
        print(f"{os.path.basename(f)}: mel max-abs {d:.2e} vs the reference with its extractors in the loop; {worst}")
        assert d <= MEL_TOL, (f, d)


def test_encoders_ragged_batch(cuda, golden_dir):
    """The text and articulatory encoders run inside the library as ONE double-width encoder (stacked weight sets, per-column
    parameter choice).  On a ragged batch whose column count needs the 128-column filler, each half must equal the reference's own
    encoder outputs for every utterance (tests/golden: t_en / a_en of the reference's forward), and the two must differ."""
    files = sorted(glob.glob(os.path.join(golden_dir, "net_tiny_*.npz")))
    gs = [np.load(f) for f in files]
    net = get_model(64, 8, int(gs[0]["weight_seed"]), cuda)
    lens = [len(g["tokens"]) for g in gs]
    assert sum(lens) % 128 != 0
    x = torch.zeros(len(gs), max(lens), dtype=torch.long)
    for b, g in enumerate(gs):
        x[b, : lens[b]] = torch.from_numpy(g["tokens"])
    t = net.text_encoder(x, torch.tensor(lens))
    a = net.arts_encoder(x, torch.tensor(lens))
    for b, g in enumerate(gs):
        assert float(np.abs(t[b, : lens[b]].t().cpu().numpy() - g["ref/t_en"]).max()) <= AUX_TOL
        assert float(np.abs(a[b, : lens[b]].t().cpu().numpy() - g["ref/a_en"]).max()) <= AUX_TOL
    assert float((a - t).abs().max()) > 1e-3


def test_c_abi_forward_on_full_goldens(cuda, golden_dir):
    """as_forward_test driven through ctypes alone (no Python orchestration): model from the blob, plan, workspaces, one call;
    durations identical to the reference's, mel within 1e-4 of the reference's own output (tests/golden/net_full_*.npz)."""
    import ctypes
    from artspeech_amd import _lib
    from artspeech_amd.blob import state_dict_to_blob
    L = _lib.lib()
    files = sorted(glob.glob(os.path.join(golden_dir, "net_full_*.npz")))
    g0 = np.load(files[0])
    sd = synth.synth_state_dict(int(g0["hidden_dim"]), int(g0["dim_in"]), seed=int(g0["weight_seed"]))
    blob = state_dict_to_blob(sd)
    cfg = _lib.ModelCfg()
    cfg.hidden_dim, cfg.dim_in, cfg.style_dim, cfg.n_mels, cfg.n_token = int(g0["hidden_dim"]), int(g0["dim_in"]), 256, 80, 178
    for i, v in enumerate(models.stats_floats(load_distribution(DEFAULT_STATS))):
        cfg.stats[i] = v
    model, plan = ctypes.c_void_p(), ctypes.c_void_p()
    torch.cuda.set_device(cuda)
    assert L.as_model_create(blob, len(blob), ctypes.byref(cfg), ctypes.byref(model)) == 0
    assert L.as_plan_create(model, ctypes.byref(plan)) == 0
    try:
        for f in files:
            g = np.load(f)
            tokens = g["tokens"].astype(np.int32)
            mel, f0_raw, ema_raw = raw_features(int(g["t_ref"]), int(g["seed"]))
            n, t = len(tokens), mel.shape[1]
            tl, rl = (ctypes.c_int32 * 1)(n), (ctypes.c_int32 * 1)(t)
            I32P = ctypes.POINTER(ctypes.c_int32)
            b = _lib.Batch(1, ctypes.cast(tl, I32P), ctypes.cast(rl, I32P), None)
            d_tok, d_mel = torch.from_numpy(tokens).to(cuda), torch.from_numpy(mel).to(cuda).contiguous()
            d_f0, d_ema = torch.from_numpy(f0_raw).to(cuda).contiguous(), torch.from_numpy(ema_raw).to(cuda).contiguous()
            cap = 2 * int(g["ref/pred_dur"].sum()) + 64
            out = torch.zeros(80, cap, device=cuda)
            dur_i = torch.zeros(n, dtype=torch.int32, device=cuda)
            io = _lib.ForwardIO()
            io.tokens, io.mel, io.ld_mel, io.f0_raw, io.ema_raw, io.ld_ema = d_tok.data_ptr(), d_mel.data_ptr(), t, d_f0.data_ptr(), d_ema.data_ptr(), t
            io.mel_out, io.ld_out, io.dur_i = out.data_ptr(), cap, dur_i.data_ptr()
            na = L.as_module_workspace_bytes(model, plan, _lib.AS_MOD_FORWARD_A, ctypes.byref(b))
            assert na > 0
            ws_a = torch.empty(na, dtype=torch.uint8, device=cuda)
            # frames unknown: workspace B sized for the capacity we allow
            fr = (ctypes.c_int32 * 1)(cap // 2)
            b_cap = _lib.Batch(1, ctypes.cast(tl, I32P), ctypes.cast(rl, I32P), ctypes.cast(fr, I32P))
            nb = L.as_module_workspace_bytes(model, plan, _lib.AS_MOD_FORWARD_B, ctypes.byref(b_cap))
            ws_b = torch.empty(nb, dtype=torch.uint8, device=cuda)
            frames = (ctypes.c_int32 * 1)(0)
            rc = L.as_forward_test(model, plan, ctypes.byref(b), ctypes.byref(io), ws_a.data_ptr(), na, ws_b.data_ptr(), nb, frames,
                                   torch.cuda.current_stream().cuda_stream)
            if rc == -2:
                # AS_ENOSPC: the header's contract -- a workspace sized for a CAPACITY need not cover the frame count that comes out
                # (split-K slabs exist only below 64 tiles); frames_host_out says what the batch needs: size B for it and call again
                assert frames[0] == int(g["ref/pred_dur"].sum())
                b_fit = _lib.Batch(1, ctypes.cast(tl, I32P), ctypes.cast(rl, I32P), ctypes.cast(frames, I32P))
                nb = L.as_module_workspace_bytes(model, plan, _lib.AS_MOD_FORWARD_B, ctypes.byref(b_fit))
                ws_b = torch.empty(nb, dtype=torch.uint8, device=cuda)
                rc = L.as_forward_test(model, plan, ctypes.byref(b), ctypes.byref(io), ws_a.data_ptr(), na, ws_b.data_ptr(), nb, frames,
                                       torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
            assert frames[0] == int(g["ref/pred_dur"].sum())
            assert np.array_equal(dur_i.cpu().numpy(), g["ref/pred_dur"].astype(np.int32))
            d = float(np.abs(out[:, : 2 * frames[0]].cpu().numpy() - g["ref/mel"]).max())
            print(os.path.basename(f), "C ABI mel max-abs", d)
            assert d <= MEL_TOL, (f, d)
            # the same utterance with NO read-back (as_forward_io.frame_cap; round 6): a capacity instead of a count, workspace B of
            # AS_MOD_FORWARD_B_CAP, the call only enqueues -- the reference's durations and mel again, and the frame offsets on the device
            out.zero_()
            dur_i.zero_()
            f_off = torch.zeros(2, dtype=torch.int32, device=cuda)
            io.frame_cap, io.frame_off = cap // 2, f_off.data_ptr()
            nb2 = L.as_module_workspace_bytes(model, plan, _lib.AS_MOD_FORWARD_B_CAP, ctypes.byref(b_cap))
            assert nb2 > 0
            ws_b2 = torch.empty(nb2, dtype=torch.uint8, device=cuda)
            rc = L.as_forward_test(model, plan, ctypes.byref(b), ctypes.byref(io), ws_a.data_ptr(), na, ws_b2.data_ptr(), nb2, None,
                                   torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
            assert L.as_device_status(0) == 0
            assert f_off.cpu().tolist() == [0, int(g["ref/pred_dur"].sum())]
            assert np.array_equal(dur_i.cpu().numpy(), g["ref/pred_dur"].astype(np.int32))
            d2 = float(np.abs(out[:, : 2 * frames[0]].cpu().numpy() - g["ref/mel"]).max())
            print(os.path.basename(f), "C ABI mel max-abs under a frame capacity", d2)
            assert d2 <= MEL_TOL, (f, d2)
    finally:
        L.as_plan_destroy(plan)
        L.as_model_destroy(model)


@pytest.mark.parametrize("arrangement", ["side_streams", "merged_chain"])
def test_c3_full_config_ragged_batch_vs_oracle(cuda, arrangement):
    """BASELINE config C3 at full size: 32 utterances of VARIED lengths through one as_forward_test call, every utterance against the
    oracle's batch-1 run of the same utterance (durations identical, mel within 1e-4).  Predicted durations (no forcing): the integer
    path is part of the check.  Both arrangements of the step: branches on side streams, and the one-stream chain whose independent
    branches share conv-GEMM launches (as_plan_set_serial + as_plan_set_merge: what as_lanes and bench.py's lanes run)."""
    from oracle import acoustic
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    hd, di, seed = 512, 64, 3407
    net = get_model(hd, di, seed, cuda)
    if arrangement == "merged_chain":
        net = net.replica()
        net.rt.set_serial(True)
    W = fold_state_dict(synth.synth_state_dict(hd, di, seed=seed))
    dist = load_distribution(DEFAULT_STATS)
    rng = np.random.default_rng(7)
    Bn = 32
    tl = [int(v) for v in rng.integers(24, 41, Bn)]
    ml = [int(v) for v in rng.integers(120, 201, Bn)]
    toks = [synth.synth_tokens(n, 500 + b) for b, n in enumerate(tl)]
    feats = [raw_features(t, 500 + b) for b, t in enumerate(ml)]
    texts = torch.zeros(Bn, max(tl), dtype=torch.long)
    mels, f0s, emas = torch.zeros(Bn, 80, max(ml)), torch.zeros(Bn, 1, max(ml)), torch.zeros(Bn, 10, max(ml))
    for b in range(Bn):
        texts[b, : tl[b]] = torch.from_numpy(toks[b])
        mels[b, :, : ml[b]], f0s[b, :, : ml[b]], emas[b, :, : ml[b]] = (torch.from_numpy(x) for x in feats[b])
    out, aux = net([texts, torch.tensor(tl), mels, torch.tensor(ml)], None, None, step="test", features=(f0s, emas), return_aux=True)
    dur = aux["dur_i"].cpu().numpy()
    worst, o = 0.0, 0
    for b in range(Bn):
        ref = acoustic.forward_test(W, torch.from_numpy(toks[b]), *(torch.from_numpy(x) for x in feats[b]), dist)
        assert np.array_equal(dur[o:o + tl[b]], ref["pred_dur"].numpy().astype(np.int32)), (b, "durations")
        o += tl[b]
        M2 = 2 * int(ref["pred_dur"].sum())
        assert aux["frames2"][b] == M2
        d = float((out[b, :, :M2].cpu() - ref["mel"]).abs().max())
        worst = max(worst, d)
        assert d <= MEL_TOL, (b, d)
        assert out.shape[2] == M2 or float(out[b, :, M2:].abs().max()) == 0.0
    print(f"C3 ragged batch of 32, full config, {arrangement}: worst mel max-abs vs oracle", worst)


@pytest.mark.parametrize("arrangement", ["side_streams", "merged_chain"])
def test_c3_predicted_durations_under_a_frame_capacity_vs_oracle(cuda, arrangement):
    """models.py:361-368 sizes the second half from the durations the first half PREDICTS.  as_forward_io.frame_cap: the same 32 ragged
    utterances as one as_forward_test call that never reads anything back -- the second half is laid out for a capacity (here 1.3 x what
    comes out, and once exactly what comes out), the utterances' extents are derived on the device.  Every utterance against the oracle's
    batch-1 run (durations identical, mel <= 1e-4), frame_off against the read-back path's, the eager call and its replay from a hipGraph
    bit for bit, the filler behind the last utterance is never mistaken for a non-finite mel, and a capacity that is too small raises
    AS_STATUS_CAPACITY without writing out of bounds."""
    from artspeech_amd import _lib
    from oracle import acoustic
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    hd, di, seed = 512, 64, 3407
    net = get_model(hd, di, seed, cuda).replica()
    if arrangement == "merged_chain":
        net.rt.set_serial(True)
    W = fold_state_dict(synth.synth_state_dict(hd, di, seed=seed))
    dist = load_distribution(DEFAULT_STATS)
    rng = np.random.default_rng(7)
    Bn = 32
    tl = [int(v) for v in rng.integers(24, 41, Bn)]
    ml = [int(v) for v in rng.integers(120, 201, Bn)]
    toks = [synth.synth_tokens(n, 500 + b) for b, n in enumerate(tl)]
    feats = [raw_features(t, 500 + b) for b, t in enumerate(ml)]
    with torch.cuda.device(cuda):
        tok = torch.from_numpy(np.concatenate(toks).astype(np.int32)).to(cuda)
        mel_p = torch.from_numpy(np.concatenate([f[0] for f in feats], 1)).to(cuda)
        f0_p = torch.from_numpy(np.concatenate([f[1] for f in feats], 1)).to(cuda)
        ema_p = torch.from_numpy(np.concatenate([f[2] for f in feats], 1)).to(cuda)
    base = net.forward_packed(tok, tl, mel_p, f0_p, ema_p, ml)                   # the read-back path
    off_ref = base["frame_off"].cpu().numpy()
    total = int(off_ref[-1])
    refs = [acoustic.forward_test(W, torch.from_numpy(toks[b]), *(torch.from_numpy(x) for x in feats[b]), dist) for b in range(Bn)]
    for cap in (int(1.3 * total) + 7, total):
        out = net.forward_packed(tok, tl, mel_p, f0_p, ema_p, ml, frame_cap=cap)
        torch.cuda.synchronize()
        assert _lib.lib().as_device_status(0) == 0, _lib.device_status()
        assert out["mel"].shape == (80, 2 * cap)
        off = out["frame_off"].cpu().numpy()
        assert np.array_equal(off, off_ref)
        assert np.array_equal(out["dur_i"].cpu().numpy(), base["dur_i"].cpu().numpy())
        worst, o = 0.0, 0
        for b in range(Bn):
            assert np.array_equal(out["dur_i"].cpu().numpy()[o:o + tl[b]], refs[b]["pred_dur"].numpy().astype(np.int32)), (b, "durations")
            o += tl[b]
            a, e = 2 * int(off[b]), 2 * int(off[b + 1])
            assert e - a == refs[b]["mel"].shape[1]
            d = float((out["mel"][:, a:e].cpu() - refs[b]["mel"]).abs().max())
            worst = max(worst, d)
            assert d <= MEL_TOL, (cap, b, d)
        assert float((out["mel"][:, : 2 * total] - base["mel"]).abs().max()) <= 3e-5
        print(f"C3 ragged 32, predicted durations, frame_cap {cap} (needed {total}), {arrangement}: worst mel max-abs vs oracle", worst)
    # the call as a hipGraph (nothing in it waits for the host): replays give the eager bits
    cap = int(1.3 * total) + 7
    eager = net.forward_packed(tok, tl, mel_p, f0_p, ema_p, ml, frame_cap=cap)
    want = eager["mel"][:, : 2 * total].clone()
    graph, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        net.forward_packed(tok, tl, mel_p, f0_p, ema_p, ml, frame_cap=cap, out=eager)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=st):
            net.forward_packed(tok, tl, mel_p, f0_p, ema_p, ml, frame_cap=cap, out=eager)
    for _ in range(2):
        eager["mel"].zero_()
        eager["frame_off"].zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(eager["mel"][:, : 2 * total], want)
        assert np.array_equal(eager["frame_off"].cpu().numpy(), off_ref)
    assert _lib.lib().as_device_status(0) == 0
    # not enough room: loud, and nothing past the buffer is touched
    small = total - 50
    try:
        cut = net.forward_packed(tok, tl, mel_p, f0_p, ema_p, ml, frame_cap=small)
        torch.cuda.synchronize()
        assert _lib.lib().as_device_status(0) & (1 << 5)                          # AS_STATUS_CAPACITY
        assert int(cut["frame_off"].cpu().numpy()[-1]) == total                   # (the durations themselves are what they are)
    finally:
        _lib.lib().as_device_status(1)
    out = net.forward_packed(tok, tl, mel_p, f0_p, ema_p, ml, frame_cap=cap)      # healthy again after the clear
    torch.cuda.synchronize()
    assert torch.equal(out["mel"][:, : 2 * total], want) and _lib.lib().as_device_status(0) == 0


@pytest.mark.parametrize("shape", ["one_utterance", "three_long", "merged_chain_pair"])
def test_frame_capacity_at_batch_one_and_long_form(cuda, shape):
    """as_forward_io.frame_cap where the launches are K-sliced (batch 1, BASELINE C2's shape: the slices store nothing for the filler and
    the reductions skip it; no fused reduction + AdaIN under a capacity) and where utterances are hundreds of tokens long (the duration
    predictor's recurrence on the side stream of a recording plan): frame offsets and durations equal to the read-back path's, mel within
    3e-5 of it, with room to spare of one frame, of 2 x and of 5 x what comes out; status clean."""
    import bench
    net = get_model(512, 64, 3407, cuda).replica()
    if shape == "one_utterance":
        host, g = bench.make_inputs(cuda, 1, 30, 75, 150, seed0=41)
    elif shape == "three_long":
        host, g = bench.make_inputs(cuda, 3, 260, 300, 120, vary=True, seed0=77)
    else:
        net.rt.set_serial(True)
        host, g = bench.make_inputs(cuda, 2, 220, 300, 100, vary=True, seed0=78)
    base = net.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"])
    off = base["frame_off"].cpu()
    total = int(off[-1])
    for cap in (total + 1, 2 * total, 5 * total):
        out = net.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], frame_cap=cap)
        torch.cuda.synchronize()
        assert _lib.lib().as_device_status(0) == 0, (cap, _lib.device_status())
        assert torch.equal(out["frame_off"].cpu(), off) and torch.equal(out["dur_i"], base["dur_i"])
        d = float((out["mel"][:, : 2 * total] - base["mel"]).abs().max())
        assert d <= 3e-5, (shape, cap, d)
        assert bool(torch.isfinite(out["mel"][:, : 2 * total]).all())


def test_f16_operand_mode_error_is_reported(cuda, golden_dir):
    """BASELINE config C2 names 16-bit operands ("error reported").  as_plan_set_operand_mode(plan, 1) runs every
    conv GEMM on the h parts only (plain fp16 operands, one matrix-core product, fp32 accumulate).  This mode is NOT held to the 1e-4
    bound: the test reports its distance to the reference's mel and only checks that it is the coarse mode (between the f16x3 error
    and 5e-2) and that switching back restores the fp32-accurate result."""
    g = np.load(os.path.join(golden_dir, "net_full_N30_T150_s7.npz"))
    net = get_model(int(g["hidden_dim"]), int(g["dim_in"]), int(g["weight_seed"]), cuda)
    forced = [g["ref/pred_dur"].astype(np.int64)]                      # same frames in both modes: the integer path is not under test here
    tokens = torch.from_numpy(g["tokens"])[None]
    mel, f0_raw, ema_raw = raw_features(int(g["t_ref"]), int(g["seed"]))
    batch = [tokens, torch.tensor([tokens.shape[1]]), torch.from_numpy(mel)[None], torch.tensor([mel.shape[1]])]
    kw = dict(step="test", features=(torch.from_numpy(f0_raw)[None], torch.from_numpy(ema_raw)[None]), forced_durations=forced)
    exact = float(np.abs(net(batch, None, None, **kw)[0].cpu().numpy() - g["ref/mel"]).max())
    net.rt.set_operand_mode(1)
    try:
        coarse = float(np.abs(net(batch, None, None, **kw)[0].cpu().numpy() - g["ref/mel"]).max())
    finally:
        net.rt.set_operand_mode(3)
    again = float(np.abs(net(batch, None, None, **kw)[0].cpu().numpy() - g["ref/mel"]).max())
    print(f"C2 mel max-abs vs the reference: f16x3 {exact:.2e}, fp16 operands {coarse:.2e}")
    assert exact <= MEL_TOL and again == exact
    assert exact < coarse <= 5e-2


def test_long_utterances_recurrence_beside_the_encoders(cuda, monkeypatch):
    """A recording plan (one chain per batch: what as_lanes runs) with utterances of >= 200 tokens puts the duration predictor's recurrence
    on the plan's side stream -- its blocks first, the text / articulatory encoders' last two layers beside it (models.py:356-360: they do
    not depend on it).  Eagerly, as a replayed hipGraph, and with PREDICTED durations (where the recurrence's result is read back by
    the call itself), against the chain without the fork."""
    import bench
    net = get_model(512, 64, 3407, cuda)
    host, g = bench.make_inputs(cuda, 3, 260, 300, 120, vary=True, seed0=77)
    assert max(g["tok_lens"]) >= 200

    def run(side, graph):
        if side:
            monkeypatch.delenv("AS_NO_SIDE_LSTM", raising=False)
        else:
            monkeypatch.setenv("AS_NO_SIDE_LSTM", "1")
        twin = net.replica()
        twin.rt.set_serial(True)
        r = bench.Runner(twin, g)
        r.step()
        if graph:
            fn = r.capture()
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        forced = r.out["mel"].clone()
        pred = twin.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], aux=True)
        torch.cuda.synchronize()
        return forced, pred["mel"].clone(), pred["duration"].clone()
    base = run(False, False)
    eager = run(True, False)
    replay = run(True, True)
    for a, b in zip(eager, replay):                                      # the fork survives capture: the replayed graph gives the eager bits
        assert a.shape == b.shape and torch.equal(a, b)
    # against the chain without the fork: the same arithmetic, but the duration predictor's convs now go out ahead of the encoders' instead
    # of sharing their launches -- another tile, another order of partial sums (the bound of chain_vs_side_streams in bench.py)
    assert float((base[0] - eager[0]).abs().max()) <= 3e-5
    assert float((base[2] - eager[2]).abs().max()) <= 3e-5
    if base[1].shape == eager[1].shape:                                  # (the same rounded durations)
        assert float((base[1] - eager[1]).abs().max()) <= 3e-5
    assert _lib.lib().as_device_status(0) == 0

