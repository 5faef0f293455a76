#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The fixtures are data only: inputs (or the seeds that regenerate them) and the reference's outputs.

  python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json

What is pinned
  param_inventory.json   names + shapes of the reference's ArtsSpeech(stage="second").state_dict()
                         (hot-path keys) -- artspeech_amd/spec.py must reproduce it exactly
  mas_small.npz          inputs + maximum_path1 / maximum_path2 outputs (S_monotonic_align.py)
  mas_large.npz          [8,1024,2000] case: hash-generated input (seed only) + row-of-column paths
  net_tiny_*.npz         hidden_dim=64, dim_in=8: every module-boundary tensor of forward(step="test")
  net_full_*.npz         hidden_dim=512, dim_in=64 (the shipped config): outputs of forward(step="test")
  net_allin_*.npz        forward(step="test") with the reference's REAL JDCNet / EMA_Predictor in the loop (models.py:426-449 feeding
                         :356-371; no stub): tokens + mel in; f0_ext, n_ext, ema_ext, Style, pred_dur, mel out.  Tiny and shipped config.
Weights are artspeech_amd.synth.synth_state_dict(seed) loaded with load_state_dict -- the GPU box
regenerates the identical tensors from the seed.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import _refshim  # noqa: E402

Munch = _refshim.install()

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import yaml  # noqa: E402

import models as ref_models  # noqa: E402  (the reference)
import S_monotonic_align as ref_mas  # noqa: E402  (the reference)

from artspeech_amd import synth  # noqa: E402
from artspeech_amd.spec import EXTRACTOR_PREFIXES, artsspeech_spec  # noqa: E402
from artspeech_amd.weights import DEFAULT_STATS, fold_state_dict, load_distribution  # noqa: E402

torch.set_num_threads(8)
WEIGHT_SEED = 3407


# ------------------------------------------------------------------------------------------------
def make_param_inventory():
    cfg = yaml.safe_load(open(os.path.join(_refshim.REF, "Configs/config.yaml")))["model_params"]
    out = {}
    for tag, hd, di in (("full", 512, 64), ("tiny", 64, 8)):
        c = dict(cfg, hidden_dim=hd, dim_in=di)
        net = ref_models.ArtsSpeech(Munch(c), stage="second", distribution={})
        ref = {k: list(v.shape) for k, v in net.state_dict().items() if not k.startswith(EXTRACTOR_PREFIXES)}
        mine = {k: list(v["shape"]) for k, v in artsspeech_spec(hd, di).items()}
        assert ref == mine, "spec.py does not match the reference state_dict"
        out[tag] = ref
    stats = json.load(open(os.path.join(_refshim.REF, "Data/stats.json")))
    for k in ("EMA", "pitch", "energy"):
        assert stats[k][2] == DEFAULT_STATS[k][2] and stats[k][3] == DEFAULT_STATS[k][3]
    json.dump(out, open(os.path.join(HERE, "param_inventory.json"), "w"), indent=0, sort_keys=True)
    print("param inventory ok:", {k: len(v) for k, v in out.items()})


# ------------------------------------------------------------------------------------------------
def mas_large_input(B=8, Tx=1024, Ty=2000, seed=1234):
    """Hash-generated, transcendental-free lattice (bit-reproducible anywhere): u^2 of uniforms."""
    u = synth.hash_tensor(f"mas/{B}x{Tx}x{Ty}", (B, Tx, Ty), seed)
    return (u * u).astype(np.float32)


def row_of_col(path):
    """[B,Tx,Ty] 0/1 path -> int32 [B,Ty] row index per column (-1 where the column has no 1)."""
    has = path.sum(1) > 0
    idx = path.argmax(1).astype(np.int32)
    idx[~has] = -1
    return idx


def make_mas():
    rng = np.random.default_rng(20241003)
    cases = {}

    def add(name, value, x_lens, y_lens):
        v = torch.from_numpy(value)
        mask = ref_mas.mask_from_lens(v, torch.tensor(x_lens), torch.tensor(y_lens))
        p1 = ref_mas.maximum_path1(v.clone(), mask.clone()).numpy()
        p2 = ref_mas.maximum_path2(v.clone(), mask.clone()).numpy()
        cases[name + "/value"] = value
        cases[name + "/x_lens"] = np.asarray(x_lens, np.int32)
        cases[name + "/y_lens"] = np.asarray(y_lens, np.int32)
        cases[name + "/mask"] = mask.numpy()
        cases[name + "/rows_v1"] = row_of_col(p1)
        cases[name + "/rows_v2"] = row_of_col(p2)
        print(f"mas {name}: shape {value.shape} v1!=v2 cells {(p1 != p2).sum()}")

    def softmax_lattice(B, Tx, Ty):
        z = 3.0 * rng.standard_normal((B, Tx, Ty)).astype(np.float32)
        return torch.softmax(torch.from_numpy(z), dim=1).numpy()

    B, Tx, Ty = 4, 7, 15
    add("ragged_4x7x15", softmax_lattice(B, Tx, Ty), [7, 6, 5, 4], [15, 13, 11, 9])
    B, Tx, Ty = 32, 40, 100                      # C3-like, ragged
    add("c3_32x40x100", softmax_lattice(B, Tx, Ty), [max(1, Tx - i) for i in range(B)], [Ty - 2 * i for i in range(B)])
    # exact fp32 ties: quantised values -> v1 and v2 take different paths
    q = (rng.integers(0, 4, size=(8, 12, 30)).astype(np.float32)) * np.float32(0.25)
    add("ties_8x12x30", q, [12, 11, 10, 9, 8, 7, 6, 5], [30, 29, 28, 27, 26, 25, 24, 23])
    add("tx1_2x1x9", softmax_lattice(2, 1, 9), [1, 1], [9, 5])
    add("square_3x16x16", softmax_lattice(3, 16, 16), [16, 12, 8], [16, 12, 8])
    add("neglogp_4x20x50", -np.abs(rng.standard_normal((4, 20, 50)).astype(np.float32)) * 5.0, [20, 18, 3, 20], [50, 40, 50, 21])
    np.savez_compressed(os.path.join(HERE, "mas_small.npz"), **cases)

    B, Tx, Ty = 8, 1024, 2000
    value = mas_large_input(B, Tx, Ty)
    x_lens = [Tx - 16 * i for i in range(B)]
    y_lens = [Ty - 32 * i for i in range(B)]
    v = torch.from_numpy(value)
    mask = ref_mas.mask_from_lens(v, torch.tensor(x_lens), torch.tensor(y_lens))
    p1 = ref_mas.maximum_path1(v.clone(), mask.clone()).numpy()
    p2 = ref_mas.maximum_path2(v.clone(), mask.clone()).numpy()
    print(f"mas large: v1!=v2 cells {(p1 != p2).sum()}")
    np.savez_compressed(os.path.join(HERE, "mas_large.npz"), shape=np.array([B, Tx, Ty]), seed=np.array(1234),
                        x_lens=np.asarray(x_lens, np.int32), y_lens=np.asarray(y_lens, np.int32),
                        rows_v1=row_of_col(p1), rows_v2=row_of_col(p2),
                        value_checksum=np.array(value.astype(np.float64).sum()))


# ------------------------------------------------------------------------------------------------
class _Stub(nn.Module):
    """Stands in for a frozen extractor: returns the tensor it was given."""

    def __init__(self, out):
        super().__init__()
        self.out = out

    def forward(self, *a, **k):
        return self.out


def build_reference(hidden_dim, dim_in):
    cfg = yaml.safe_load(open(os.path.join(_refshim.REF, "Configs/config.yaml")))["model_params"]
    cfg = dict(cfg, hidden_dim=hidden_dim, dim_in=dim_in)
    dist = load_distribution(DEFAULT_STATS)
    net = ref_models.ArtsSpeech(Munch(cfg), stage="second", distribution=dist).eval()
    sd = synth.synth_state_dict(hidden_dim, dim_in, seed=WEIGHT_SEED)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(m.startswith(EXTRACTOR_PREFIXES) for m in missing), (missing[:5], unexpected[:5])
    return net, sd


def run_reference(net, tokens, mel, f0_raw, ema_raw):
    """forward(step='test') of the reference, models.py:356-371, grabbing the module-boundary tensors."""
    grabbed = {}

    def hook(name):
        def f(mod, inp, out):
            grabbed[name] = out
        return f

    hs = [getattr(net, n).register_forward_hook(hook(n)) for n in
          ("text_encoder", "arts_encoder", "style_encoder", "durationPredictor", "artsPredictor", "decoder")]
    net.style_encoder.pitch_extractor = _Stub(torch.from_numpy(f0_raw)[None])
    net.style_encoder.ema_extractor = _Stub(torch.from_numpy(ema_raw)[None])
    with torch.no_grad():
        text = torch.from_numpy(tokens)[None]
        out = net([text, torch.LongTensor([text.shape[-1]]), torch.from_numpy(mel)[None],
                   torch.LongTensor([mel.shape[-1]]), None, None, None], None, None, step="test")
    for h in hs:
        h.remove()
    f0_ext, n_ext, ema_ext, style = grabbed["style_encoder"]
    F0, N, EMA = grabbed["artsPredictor"]
    duration = grabbed["durationPredictor"][0]
    res = dict(t_en=grabbed["text_encoder"][0].t(), a_en=grabbed["arts_encoder"][0].t(),
               f0_ext=f0_ext[0], n_ext=n_ext[0], ema_ext=ema_ext[0], style=style[0], duration=duration,
               pred_dur=torch.round(duration).clamp(min=1).to(torch.int64),
               F0=F0[0], N=N[0], EMA=EMA[0], mel=out[0])
    return {k: v.detach().numpy().copy() for k, v in res.items()}


def raw_features(t_ref, seed):
    """un-normalised extractor outputs: normalised stand-ins mapped back through Data/stats.json."""
    mel, f0, ema = synth.synth_ref_features(t_ref, seed)
    f0_raw = (f0 * np.float32(DEFAULT_STATS["pitch"][3]) + np.float32(DEFAULT_STATS["pitch"][2])).astype(np.float32)
    ema_raw = (ema * np.asarray(DEFAULT_STATS["EMA"][3], np.float32)[:, None]
               + np.asarray(DEFAULT_STATS["EMA"][2], np.float32)[:, None]).astype(np.float32)
    return mel, f0_raw, ema_raw


def make_net():
    from oracle import acoustic
    for tag, hd, di, cases, keep in (
        ("tiny", 64, 8, [(12, 70, 1), (30, 150, 2), (5, 66, 3), (40, 200, 4)], None),
        ("full", 512, 64, [(40, 200, 1234), (30, 150, 7), (23, 97, 99)],
         ("style", "duration", "pred_dur", "F0", "N", "EMA", "mel", "t_en_sum", "a_en_sum")),
    ):
        net, sd = build_reference(hd, di)
        W = fold_state_dict(sd)
        dist = load_distribution(DEFAULT_STATS)
        for (n_tok, t_ref, seed) in cases:
            tokens = synth.synth_tokens(n_tok, seed)
            mel, f0_raw, ema_raw = raw_features(t_ref, seed)
            ref = run_reference(net, tokens, mel, f0_raw, ema_raw)
            ref["t_en_sum"] = np.array(ref["t_en"].astype(np.float64).sum())
            ref["a_en_sum"] = np.array(ref["a_en"].astype(np.float64).sum())
            frac = np.abs(ref["duration"] - np.floor(ref["duration"]) - 0.5)
            ora = acoustic.forward_test(W, torch.from_numpy(tokens), torch.from_numpy(mel),
                                        torch.from_numpy(f0_raw), torch.from_numpy(ema_raw), dist)
            diffs = {k: float(np.abs(ora[k].numpy().astype(np.float64) - ref[k]).max()) for k in
                     ("t_en", "a_en", "style", "duration", "F0", "N", "EMA", "mel") if ora[k].shape == ref[k].shape}
            print(f"net {tag} N={n_tok} T={t_ref} seed={seed}: M={int(ref['pred_dur'].sum())} "
                  f"dur margin {frac.min():.4f} |mel|max {np.abs(ref['mel']).max():.3f} |style|max {np.abs(ref['style']).max():.3f} "
                  f"oracle-vs-ref {diffs} dur_equal {bool((ora['pred_dur'].numpy() == ref['pred_dur']).all())}")
            out = dict(tokens=tokens, t_ref=np.array(t_ref), seed=np.array(seed), weight_seed=np.array(WEIGHT_SEED),
                       hidden_dim=np.array(hd), dim_in=np.array(di), dur_margin=np.array(frac.min()))
            for k, v in ref.items():
                if keep is None or k in keep:
                    out["ref/" + k] = v
            np.savez_compressed(os.path.join(HERE, f"net_{tag}_N{n_tok}_T{t_ref}_s{seed}.npz"), **out)


def make_allin():
    """ArtsSpeech.forward(step="test") as test.py:113 runs it: the reference's own JDCNet and EMA_Predictor (seeded synthetic weights:
    artspeech_amd.jdc.synth_jdc_state_dict / artspeech_amd.ema.synth_ema_state_dict, loaded strictly) inside StyleEncoder.forward --
    log_norm -> pitch_extractor(mel.unsqueeze(1)) -> ema_extractor(f0_ext, n_ext, mel) -> stats normalisation (models.py:426-449) --
    feeding the rest of the path (models.py:356-371).  B = 1, as the reference runs inference."""
    from artspeech_amd import ema as E
    from artspeech_amd import jdc as J
    # the seeded JDCNet's |classifier| output lives in 0.1 .. 0.4: against Data/stats.json's pitch scale (137 +- 78 Hz) that is a constant.
    # Its last Linear is scaled (fixture and test alike) so that F0 varies by a few tens of Hz per utterance and the glue between the
    # extractors -- which tensor goes where, raw or normalised -- shows in f0_ext, in what EMA_Predictor makes of it, and in the mel.
    F0_GAIN = 400.0
    jsd = {k: np.asarray(v) for k, v in J.synth_jdc_state_dict(1, seed=3407).items()}
    for k in ("classifier.weight", "classifier.bias"):
        jsd[k] = (jsd[k] * np.float32(F0_GAIN)).astype(np.float32)
    for tag, hd, di, cases in (("tiny", 64, 8, [(12, 70, 21), (30, 150, 22)]), ("full", 512, 64, [(30, 150, 23), (40, 200, 24)])):
        net, sd = build_reference(hd, di)
        se = net.style_encoder
        se.pitch_extractor.load_state_dict({k: torch.from_numpy(v) for k, v in jsd.items()})
        se.ema_extractor.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in E.synth_ema_state_dict(seed=3407).items()})
        for (n_tok, t_ref, seed) in cases:
            tokens = synth.synth_tokens(n_tok, seed)
            mel, _, _ = synth.synth_ref_features(t_ref, seed)
            grabbed = {}

            def hook(name):
                def f(mod, inp, out):
                    grabbed[name] = out
                return f

            hs = [getattr(net, n).register_forward_hook(hook(n)) for n in ("style_encoder", "durationPredictor", "artsPredictor")]
            with torch.no_grad():
                text = torch.from_numpy(tokens)[None]
                out = net([text, torch.LongTensor([text.shape[-1]]), torch.from_numpy(mel)[None], torch.LongTensor([mel.shape[-1]]),
                           None, None, None], None, None, step="test")
            for h in hs:
                h.remove()
            f0_ext, n_ext, ema_ext, style = grabbed["style_encoder"]
            duration = grabbed["durationPredictor"][0]
            F0, N, EMA = grabbed["artsPredictor"]
            frac = np.abs(duration.numpy() - np.floor(duration.numpy()) - 0.5)
            res = {"ref/f0_ext": f0_ext[0], "ref/n_ext": n_ext[0], "ref/ema_ext": ema_ext[0], "ref/style": style[0], "ref/duration": duration,
                   "ref/pred_dur": torch.round(duration).clamp(min=1).to(torch.int64), "ref/F0": F0[0], "ref/N": N[0], "ref/EMA": EMA[0],
                   "ref/mel": out[0]}
            res = {k: v.detach().numpy().copy() for k, v in res.items()}
            assert all(np.isfinite(v).all() for v in res.values())
            print(f"allin {tag} N={n_tok} T={t_ref} seed={seed}: M={int(res['ref/pred_dur'].sum())} dur margin {frac.min():.4f} "
                  f"f0_ext std {res['ref/f0_ext'].std():.3f} |ema_ext|max {np.abs(res['ref/ema_ext']).max():.3f} "
                  f"|style|max {np.abs(res['ref/style']).max():.3f} |mel|max {np.abs(res['ref/mel']).max():.3f}")
            np.savez_compressed(os.path.join(HERE, f"net_allin_{tag}_N{n_tok}_T{t_ref}_s{seed}.npz"), tokens=tokens, mel_in=mel,
                                t_ref=np.array(t_ref), seed=np.array(seed), weight_seed=np.array(WEIGHT_SEED), extractor_seed=np.array(3407),
                                jdc_classifier_gain=np.array(F0_GAIN),
                                hidden_dim=np.array(hd), dim_in=np.array(di), dur_margin=np.array(frac.min()), **res)


def make_text():
    """symbol table + token ids produced by the reference's own TextCleaner (meldataset.py:14-29; test.py:19-38 is a
    copy that prints and drops unknown characters) on real IPA lines of the shipped validation lists."""
    import meldataset as ref_md  # the reference
    tc = ref_md.TextCleaner()
    lines = []
    for lst in ("Data/val_list_LJspeech.txt", "Data/val_list_libritts.txt"):
        rows = open(os.path.join(_refshim.REF, lst), encoding="utf-8").read().splitlines()
        lines += [r.split("|")[1] for r in rows[:6]]
    gold = [{"text": t, "ids": tc(t)} for t in lines]
    json.dump({"symbols": ref_md.symbols, "cases": gold}, open(os.path.join(HERE, "text_golden.json"), "w", encoding="utf-8"),
              ensure_ascii=False)
    print("text golden:", len(ref_md.symbols), "symbols,", len(gold), "lines")


def make_vocoder():
    """HiFi-GAN Generator of the reference (Vocoder/vocoder.py:75-125) on seeded synthetic weights
    (artspeech_amd.vocoder.synth_generator_state_dict): a tiny configuration and the shipped one (Vocoder/config.json)."""
    from Vocoder.vocoder import Generator as RefGenerator                      # the reference
    from artspeech_amd import vocoder as V
    inv = {}
    for tag, c0, t_list in (("tiny", 32, (9, 16)), ("full", 512, (12,))):
        h = Munch(dict(V.DEFAULT_H, upsample_initial_channel=c0))
        ref = RefGenerator(h).eval()
        inv[tag] = {k: list(v.shape) for k, v in ref.state_dict().items()}
        assert inv[tag] == {k: list(v) for k, v in V.generator_spec(h).items()}, "generator_spec differs from the reference"
        sd = V.synth_generator_state_dict(h, seed=3407)
        ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        for t in t_list:
            mel = synth.hash_tensor(f"voc/mel/{tag}/{t}", (1, 80, t), 1234, 1.0)
            with torch.no_grad():
                wav = ref(torch.from_numpy(mel))
            assert wav.shape == (1, 1, 300 * t) and torch.isfinite(wav).all()
            print("vocoder golden", tag, t, "wav abs max %.4f mean |x| %.4f" % (float(wav.abs().max()), float(wav.abs().mean())))
            np.savez_compressed(os.path.join(HERE, f"voc_{tag}_T{t}.npz"), c0=c0, t=t, seed=3407, mel=mel[0], wav=wav[0, 0].numpy())
    json.dump(inv, open(os.path.join(HERE, "vocoder_inventory.json"), "w"), indent=0, sort_keys=True)


def make_jdc():
    """JDCNet of the reference (Utils/JDC/model.py, built as models.py:377 does) on seeded synthetic weights
    (artspeech_amd.jdc.synth_jdc_state_dict): one utterance per fixture, eval mode."""
    from Utils.JDC.model import JDCNet as RefJDC                               # the reference
    from artspeech_amd import jdc as J
    ref = RefJDC(num_class=1, seq_len=192).eval()
    inv = {k: list(v.shape) for k, v in ref.state_dict().items()}
    assert inv == {k: list(v) for k, v in J.jdc_spec(1).items()}, "jdc_spec differs from the reference"
    sd = J.synth_jdc_state_dict(1, seed=3407)
    ref.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    for t in (7, 66, 150):
        mel = synth.hash_tensor(f"jdc/mel/{t}", (1, 80, t), 1234, 1.0)
        with torch.no_grad():
            f0 = ref(torch.from_numpy(mel).unsqueeze(1))                       # models.py:432
        assert f0.shape == (1, 1, t) and torch.isfinite(f0).all()
        print("jdc golden T", t, "f0 max %.4f mean %.4f" % (float(f0.max()), float(f0.mean())))
        np.savez_compressed(os.path.join(HERE, f"jdc_T{t}.npz"), t=t, seed=3407, mel=mel[0], f0=f0[0].numpy())
    json.dump(inv, open(os.path.join(HERE, "jdc_inventory.json"), "w"), indent=0, sort_keys=True)


def make_ema():
    """EMA_Predictor of the reference (Utils/EMA/EMA_Predictor.py + its vendored conformer blocks) on seeded synthetic
    weights (artspeech_amd.ema.synth_ema_state_dict): one utterance per fixture (B = 1, as test.py runs it), eval mode."""
    from Utils.EMA.EMA_Predictor import EMA_Predictor as RefEMA                # the reference
    from artspeech_amd import ema as E
    ref = RefEMA().eval()
    inv = {k: list(v.shape) for k, v in ref.state_dict().items()}
    assert inv == {k: list(v) for k, v in E.ema_spec().items()}, "ema_spec differs from the reference"
    sd = E.synth_ema_state_dict(seed=3407)
    pe_ref = ref.state_dict()["decoder.0.sequential.1.module.positional_encoding.pe"]
    assert torch.equal(pe_ref, torch.from_numpy(sd["decoder.0.sequential.1.module.positional_encoding.pe"])), "PE table differs"
    ref.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    for t in (7, 66, 150):
        mel = synth.hash_tensor(f"ema/mel/{t}", (1, 80, t), 1234, 1.0)
        f0 = synth.hash_tensor(f"ema/f0/{t}", (1, 1, t), 1234, 1.0)
        n = synth.hash_tensor(f"ema/n/{t}", (1, 1, t), 1234, 1.0)
        with torch.no_grad():
            ema = ref(torch.from_numpy(f0), torch.from_numpy(n), torch.from_numpy(mel))   # models.py:433
        assert ema.shape == (1, 10, t) and torch.isfinite(ema).all()
        print("ema golden T", t, "abs max %.4f mean |x| %.4f" % (float(ema.abs().max()), float(ema.abs().mean())))
        np.savez_compressed(os.path.join(HERE, f"ema_T{t}.npz"), t=t, seed=3407, mel=mel[0], f0=f0[0], n=n[0], ema=ema[0].numpy())
    json.dump(inv, open(os.path.join(HERE, "ema_inventory.json"), "w"), indent=0, sort_keys=True)


def make_softmax_mas():
    """train_second.py:181-185 with the reference's own functions: softmax -> mask_from_lens -> maximum_path1/2 -> d_gt."""
    import torch.nn.functional as F
    import S_monotonic_align as ref
    out = {}
    for tag, (B, S, T, dim) in {"last": (5, 23, 61, -1), "dim1": (4, 17, 90, 1)}.items():
        feat = torch.from_numpy(synth.hash_tensor(f"smas/{tag}", (B, S, T), 1234, 3.0))
        sl = torch.tensor([S - (2 * i) % (S - 1) for i in range(B)])
        ml = torch.tensor([T - 3 * i for i in range(B)])
        attn = F.softmax(feat, dim=dim)
        mask = ref.mask_from_lens(attn, sl, ml)
        p2 = ref.maximum_path2(attn, mask)
        p1 = ref.maximum_path1(attn, mask)
        out.update({f"{tag}_feat": feat.numpy(), f"{tag}_sl": sl.numpy(), f"{tag}_ml": ml.numpy(), f"{tag}_attn": attn.numpy(),
                    f"{tag}_path2": p2.numpy().astype(np.uint8), f"{tag}_path1": p1.numpy().astype(np.uint8),
                    f"{tag}_dgt2": p2.sum(-1).numpy().astype(np.int32), f"{tag}_dim": np.int32(dim)})
        print("softmax+mas golden", tag, "d_gt sum", int(p2.sum()))
    np.savez_compressed(os.path.join(HERE, "softmax_mas.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["softmax_mas"]:
        make_softmax_mas()
        sys.exit(0)
    if sys.argv[1:] == ["ema"]:
        make_ema()
        sys.exit(0)
    if sys.argv[1:] == ["allin"]:
        make_allin()
        sys.exit(0)
    if sys.argv[1:] == ["vocoder"]:
        make_vocoder()
        sys.exit(0)
    if sys.argv[1:] == ["jdc"]:
        make_jdc()
        sys.exit(0)
    what = sys.argv[1:] or ["params", "mas", "net", "text"]
    if "text" in what:
        make_text()
    if "params" in what:
        make_param_inventory()
    if "mas" in what:
        make_mas()
    if "net" in what:
        make_net()
