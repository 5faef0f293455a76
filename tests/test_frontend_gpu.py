"""GPU: the wav -> log-mel front end (artspeech_amd/frontend.py, through the C ABI) against the CPU restatement on torch.stft
(oracle/frontend.py; parity unpinned against the reference's torchaudio, see its header): normalised log-mel within 1e-4 abs."""
import numpy as np
import pytest
import torch

from artspeech_amd import frontend as FE

pytestmark = pytest.mark.gpu
TOL = 1e-4


def waves(seed, lens):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in lens:
        t = torch.arange(n) / 24000.0
        out.append(0.3 * torch.sin(2 * np.pi * (120 + 40 * len(out)) * t) * (1 + 0.5 * torch.sin(2 * np.pi * 3 * t)) + 0.02 * torch.randn(n, generator=g))
    return out


def test_logmel_matches_stft_restatement(cuda):
    from oracle import frontend as ofe
    fe = FE.LogMel(device=cuda)
    ws = waves(1, [24000, 5000, 1025, 36123])
    mel, lens = fe(ws)
    assert mel.shape == (4, 80, max(lens)) and lens == [1 + len(w) // 300 for w in ws]
    for b, w in enumerate(ws):
        want = ofe.preprocess(w)
        d = float((mel[b, :, : lens[b]].cpu() - want).abs().max())
        print("wave", len(w), "log-mel max-abs", d)
        assert d <= TOL, d
        assert lens[b] == mel.shape[2] or float(mel[b, :, lens[b]:].abs().max()) == 0.0
    solo = fe(ws[1])
    assert solo.shape == (1, 80, lens[1]) and float((solo[0] - mel[1, :, : lens[1]]).abs().max()) <= TOL   # (tile and K-slice choices follow the batch size: sums in another order)


def test_pipeline_from_wave_to_wave(cuda):
    """test.py's whole chain on the HIP path (minus espeak and the file I/O): reference wave -> log-mel -> JDCNet + EMA_Predictor ->
    acoustic model -> HiFi-GAN -> samples; a batch equals its items run alone."""
    from artspeech_amd import ema as E, jdc as J, synth, vocoder as V
    from artspeech_amd.pipeline import ArtSpeech
    tts = ArtSpeech(config={"model_params": {"hidden_dim": 64, "dim_in": 8, "max_conv_dim": 64}},
                    checkpoint={"net": {"ArtsSpeech": synth.synth_state_dict(64, 8, seed=3407)}}, device=cuda)
    tts.attach_pitch_extractor({"net": J.synth_jdc_state_dict(1, seed=3407)})
    tts.attach_ema_extractor({"model": E.synth_ema_state_dict(seed=3407)})
    h = dict(V.DEFAULT_H, upsample_initial_channel=32)
    tts.attach_vocoder(h, V.synth_generator_state_dict(h, seed=3407))
    ws = waves(3, [27000, 21000])                                       # 91 and 71 reference frames (>= 66: SURVEY.md A9)
    ph = ["ðə kənˈdɪʃən ɪz ðæt aɪ wɪl", "tə mˈeɪk lˈuːθɚ tˈɔːk"]
    wav = tts.synthesis_from_wave(ph, ws)
    frames = list(tts._last_frames)
    assert wav.shape == (2, 300 * max(frames)) and bool(torch.isfinite(wav).all())
    solo = tts.synthesis_from_wave(ph[1], ws[1])
    assert solo.shape == (300 * frames[1],)
    assert float((solo - wav[1, : solo.shape[0]]).abs().max()) <= 1e-4


def test_cli_synthetic_roundtrip(cuda, tmp_path):
    """python -m artspeech_amd.cli: wav file in, wav file out (synthetic tiny weights)."""
    from artspeech_amd import cli
    ref = tmp_path / "ref.wav"
    cli.write_wav(str(ref), waves(5, [26000])[0].numpy())
    out = tmp_path / "out.wav"
    assert cli.main(["--synthetic", "--tiny", "--phonemes", "ðə kənˈdɪʃən ɪz ðæt", "--ref-wav", str(ref), "--out", str(out)]) == 0
    x = cli.read_wav(str(out))
    assert x.size > 0 and x.size % 300 == 0 and np.isfinite(x).all()
