"""``test.py``-style command line: reference wave + phonemes in, synthesised wave out, every
stage on the HIP path (log-mel front end, JDCNet, EMA_Predictor, acoustic model, HiFi-GAN).

    python -m artspeech_amd.cli --config Configs/config.yaml --phonemes "ðə kənˈdɪʃən ..." --ref-wav ref.wav --out output.wav \\
        --jdc Utils/JDC/bst.t7 --ema Utils/EMA/200000.pth.tar --vocoder-config Vocoder/config.json --vocoder Vocoder/g_00935000

What test.py does and this does not: espeak phonemisation (pass the phoneme string the phonemizer prints, test.py:95), and
librosa's load / trim / resample (the wave must already be 24 kHz mono PCM; it is read with the standard library).
``--synthetic`` replaces every checkpoint by the seeded synthetic weights the tests use (the reference ships no weights).
"""
import argparse
import json
import sys
import wave

import numpy as np
import torch


def read_wav(path):
    with wave.open(path, "rb") as f:
        n, ch, sw, sr = f.getnframes(), f.getnchannels(), f.getsampwidth(), f.getframerate()
        raw = f.readframes(n)
    if sw not in (2, 4):
        raise ValueError(f"{path}: {8 * sw}-bit PCM is not supported (16- or 32-bit)")
    x = np.frombuffer(raw, dtype=np.int16 if sw == 2 else np.int32).astype(np.float32) / float(2 ** (8 * sw - 1))
    if ch > 1:
        x = x.reshape(-1, ch)[:, 0]                                   # test.py:101-102 keeps the first channel
    if sr != 24000:
        raise ValueError(f"{path}: {sr} Hz; resample to 24000 Hz first (test.py:105-106 uses librosa for that)")
    return x


def write_wav(path, x, sr=24000):
    pcm = np.clip(np.asarray(x, dtype=np.float32), -1.0, 1.0)
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((pcm * 32767.0).astype(np.int16).tobytes())


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", help="Configs/config.yaml of the reference (model_params, stats_path, pretrained_model)")
    ap.add_argument("--phonemes", required=True, help="the phoneme string espeak produces for the text (test.py:94-95)")
    ap.add_argument("--ref-wav", required=True, help="reference utterance, 24 kHz mono PCM wav")
    ap.add_argument("--out", default="output.wav")
    ap.add_argument("--jdc", help="Utils/JDC/bst.t7")
    ap.add_argument("--ema", help="Utils/EMA/200000.pth.tar")
    ap.add_argument("--vocoder-config", help="Vocoder/config.json")
    ap.add_argument("--vocoder", help="Vocoder/g_00935000")
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic weights instead of checkpoints (smoke / demo)")
    ap.add_argument("--tiny", action="store_true", help="with --synthetic: the small test configuration")
    a = ap.parse_args(argv)

    from . import ema as E, jdc as J, synth, vocoder as V
    from .pipeline import ArtSpeech
    torch.manual_seed(3407)                                            # test.py:129
    if a.synthetic:
        mp = {"hidden_dim": 64, "dim_in": 8, "max_conv_dim": 64} if a.tiny else {}
        hd, di = (64, 8) if a.tiny else (512, 64)
        tts = ArtSpeech(config={"model_params": mp}, checkpoint={"net": {"ArtsSpeech": synth.synth_state_dict(hd, di, seed=3407)}})
        tts.attach_pitch_extractor({"net": J.synth_jdc_state_dict(1, seed=3407)})
        tts.attach_ema_extractor({"model": E.synth_ema_state_dict(seed=3407)})
        h = dict(V.DEFAULT_H, upsample_initial_channel=32 if a.tiny else 512)
        tts.attach_vocoder(h, V.synth_generator_state_dict(h, seed=3407))
    else:
        if not (a.config and a.jdc and a.ema and a.vocoder):
            ap.error("--config, --jdc, --ema and --vocoder are required without --synthetic")
        tts = ArtSpeech(config=a.config)
        tts.attach_pitch_extractor(a.jdc)
        tts.attach_ema_extractor(a.ema)
        h = json.load(open(a.vocoder_config)) if a.vocoder_config else None
        tts.attach_vocoder(h, a.vocoder)
    audio = tts.synthesis_from_wave(a.phonemes, read_wav(a.ref_wav))
    write_wav(a.out, audio.cpu().numpy())
    print(f"{a.out}: {audio.numel() / 24000.0:.2f} s of audio from {tts._last_frames[0]} mel frames")
    return 0


if __name__ == "__main__":
    sys.exit(main())
