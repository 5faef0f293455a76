"""The reference's inference script (test.py:58-135) on the HIP path, from phonemes to samples.

``test.py`` does: espeak phonemizer -> TextCleaner -> reference wav -> log-mel -> ArtsSpeech(step="test") -> HiFi-GAN.
The phonemizer and the wav/mel front end are outside this path; this class covers
test.py:75-88 (distribution, build_model, load_checkpoint), test.py:96-113 (ids, tensor packing, the model call) and
test.py:115-125 (the generator, SURVEY.md section 8(f) N2: artspeech_amd/vocoder.py), so that a caller with phonemes and a
reference mel gets the mel -- and, with a vocoder attached, the waveform -- the reference would produce.
"""
import json

import torch

from . import models
from .text import TextCleaner
from .weights import DEFAULT_STATS, load_distribution


class ArtSpeech:
    def __init__(self, config=None, checkpoint=None, device=None, stats_path=None):
        """config: dict with the keys of Configs/config.yaml (model_params, stats_path, pretrained_model) or a path to
        such a YAML; checkpoint: a path / dict in the reference's format ({'net': {'ArtsSpeech': state_dict}})."""
        if isinstance(config, str):
            import yaml
            with open(config) as f:
                config = yaml.safe_load(f)
        config = config or {}
        mp = dict(hidden_dim=512, n_token=178, style_dim=256, n_layer=3, dim_in=64, max_conv_dim=512, n_mels=80, dropout=0.2)
        mp.update(config.get("model_params", {}))
        stats_path = stats_path or config.get("stats_path")
        if stats_path:                                                                  # test.py:75-79
            with open(stats_path) as f:
                stats = json.load(f)
        else:
            stats = DEFAULT_STATS
        dev = models._need_gpu(device if device is not None else config.get("device", "cuda"))
        self.model = models.build_model(models.Munch(mp), None, stage="second",
                                        distribution=load_distribution(stats, dev), device=dev)   # test.py:81
        ckpt = checkpoint if checkpoint is not None else config.get("pretrained_model")
        if ckpt:
            models.load_checkpoint(self.model, None, ckpt, load_only_params=True)       # test.py:85
        self.textcleaner = TextCleaner()
        self.device = dev
        self.generator = None                       # test.py:119-125: the HiFi-GAN generator (attach_vocoder)

    def attach_pitch_extractor(self, checkpoint=None):
        """models.py:377-379: ``JDCNet(num_class=1, seq_len=192)`` + ``torch.load("Utils/JDC/bst.t7")['net']``, on the HIP
        path (artspeech_amd/jdc.py).  With it attached, ``features=(None, ema_raw)`` lets the model extract F0 itself."""
        from .jdc import JDCNet
        net = JDCNet(num_class=1, seq_len=192, device=self.device)
        if checkpoint is not None:
            net.load_state_dict(checkpoint if isinstance(checkpoint, dict) else torch.load(checkpoint, map_location="cpu"))
        self.model.ArtsSpeech.style_encoder.pitch_extractor = net
        return net

    def attach_ema_extractor(self, checkpoint=None):
        """models.py:381-383: ``EMA_Predictor()`` + ``torch.load("Utils/EMA/200000.pth.tar")['model']``, on the HIP path
        (artspeech_amd/ema.py).  With both extractors attached ``synthesis_mel(phonemes, ref_mel)`` needs no ``features``."""
        from .ema import EMA_Predictor
        net = EMA_Predictor(device=self.device)
        if checkpoint is not None:
            net.load_state_dict(checkpoint if isinstance(checkpoint, dict) else torch.load(checkpoint, map_location="cpu"))
        self.model.ArtsSpeech.style_encoder.ema_extractor = net
        return net

    def attach_frontend(self):
        """test.py:40-47: the MelSpectrogram + log normalisation that turns the reference wave into ``mels`` (artspeech_amd/frontend.py)."""
        from .frontend import LogMel
        self.frontend = LogMel(device=self.device)
        return self.frontend

    @torch.no_grad()
    def synthesis_from_wave(self, phonemes, ref_wave, features=None, forced_durations=None):
        """test.py:94-116 from the phonemizer's output and the (already loaded, trimmed, 24 kHz) reference wave on: log-mel front
        end -> [JDCNet, EMA_Predictor] -> acoustic model -> generator.  Returns the samples (mel frames if no vocoder is attached).
        Loading / trimming / resampling the file (librosa, test.py:99-106) and espeak stay with the caller."""
        if getattr(self, "frontend", None) is None:
            self.attach_frontend()
        single = isinstance(phonemes, str)
        if single:
            mels = [self.frontend(ref_wave)[0]]
            phonemes = [phonemes]
            features = None if features is None else [features]
        else:
            mel, lens = self.frontend(list(ref_wave))
            mels = [mel[b, :, :n] for b, n in enumerate(lens)]
        fn = self.synthesis_wav if self.generator is not None else self.synthesis_mel
        out = fn(phonemes, mels, features=features, forced_durations=forced_durations)
        return out[0] if single and out.dim() > 1 and self.generator is not None else out

    def attach_vocoder(self, h=None, checkpoint=None):
        """test.py:119-125: build the generator from Vocoder/config.json-style `h` and load checkpoint['generator']."""
        from .vocoder import Generator
        self.generator = Generator(h, device=self.device)
        if checkpoint is not None:
            sd = checkpoint if isinstance(checkpoint, dict) else torch.load(checkpoint, map_location="cpu")
            self.generator.load_state_dict(sd)
        return self.generator

    @torch.no_grad()
    def synthesis_wav(self, phonemes, ref_mel, features=None, forced_durations=None):
        """test.py:113-116: mel from the acoustic model, then ``generator(mel).squeeze()`` -> [B, 300 * frames]
        (one utterance: 1-D), samples beyond an utterance's own length are zero.  The packed mel goes straight into the
        generator: no padding is ever synthesised."""
        if self.generator is None:
            raise RuntimeError("no vocoder attached: call attach_vocoder(h, checkpoint) first")
        single = isinstance(phonemes, str)
        mel = self.synthesis_mel(phonemes, ref_mel, features=features, forced_durations=forced_durations)
        lens = self._last_frames
        wav = self.generator(mel, lengths=lens)[:, 0]
        return wav[0] if single else wav

    @torch.no_grad()
    def synthesis_mel(self, phonemes, ref_mel, features=None, forced_durations=None, world=1, rank=0):
        """phonemes: the string the phonemizer returns (test.py:94-96) or a list of such strings; ref_mel: normalised
        log-mel [80,T] (test.py:43-47) or a list; features: (f0_raw, ema_raw) per utterance when no extractor modules
        are attached; (None, ema_raw) with a pitch extractor attached (attach_pitch_extractor).  Returns mel [B,80,2*max M] (what test.py:115 hands to the vocoder).
        world / rank: BASELINE config C4 -- the batch is one GLOBAL batch, this process (one per GPU, torch.distributed initialised by
        the caller) synthesises its length-sorted round-robin shard (artspeech_amd.shard: no data-path collective) and rank 0 gets the
        whole batch back in the caller's order (other ranks: None)."""
        if world > 1 and not isinstance(phonemes, str):
            from . import shard
            lens = [len(p) for p in phonemes]

            def step(idx):
                if not idx:
                    return []
                sub = self.synthesis_mel([phonemes[i] for i in idx], [ref_mel[i] for i in idx],
                                         features=None if features is None else [features[i] for i in idx],
                                         forced_durations=None if forced_durations is None else [forced_durations[i] for i in idx])
                return [sub[k, :, : self._last_frames[k]].cpu() for k in range(len(idx))]

            parts = shard.sharded_forward(step, lens, world, rank)
            if parts is None:
                return None
            self._last_frames = [p.shape[1] for p in parts]
            out = torch.zeros(len(parts), parts[0].shape[0], max(self._last_frames))
            for b, p in enumerate(parts):
                out[b, :, : p.shape[1]] = p
            return out
        if isinstance(phonemes, str):
            phonemes, ref_mel = [phonemes], [ref_mel]
            if features is not None:
                features = [features]
        ids = [torch.LongTensor(self.textcleaner(p)) for p in phonemes]               # test.py:96-97
        B = len(ids)
        nmax, tmax = max(len(i) for i in ids), max(m.shape[-1] for m in ref_mel)
        text = torch.zeros(B, nmax, dtype=torch.long)
        mels = torch.zeros(B, ref_mel[0].shape[0], tmax)
        for b in range(B):
            text[b, : len(ids[b])] = ids[b]
            mels[b, :, : ref_mel[b].shape[-1]] = torch.as_tensor(ref_mel[b])
        input_lengths = torch.LongTensor([len(i) for i in ids])                       # test.py:110
        mel_input_length = torch.LongTensor([m.shape[-1] for m in ref_mel])           # test.py:111
        feats = None
        if features is not None:
            f0 = None if any(f is None for f, _ in features) else torch.zeros(B, 1, tmax)     # None: the attached JDCNet
            ema = torch.zeros(B, 10, tmax)
            for b, (f, e) in enumerate(features):
                if f0 is not None:
                    f0[b, :, : f.shape[-1]] = torch.as_tensor(f).reshape(1, -1)
                ema[b, :, : e.shape[-1]] = torch.as_tensor(e)
            feats = (f0, ema)
        mel, aux = self.model.ArtsSpeech([text, input_lengths, mels, mel_input_length, None, None, None], None, None,
                                         step="test", features=feats, forced_durations=forced_durations, return_aux=True)   # test.py:113
        self._last_frames = list(aux["frames2"])
        return mel

    def synthesis(self, text, ref_wav, save_path):
        raise NotImplementedError("text -> phonemes (espeak) and reading / trimming the wav file (librosa) are outside this path "
                                  "(SURVEY.md section 2); use synthesis_from_wave(phonemes, wave) / synthesis_wav(phonemes, ref_mel)")
