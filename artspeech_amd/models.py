"""The reference's models.py call surface (build_model / load_checkpoint / ArtsSpeech.forward(step="test") and the sub-module
forwards) as a thin caller of the library's module-level C ABI (include/artspeech_hip.h: as_model_create,
as_forward_test, as_encoder_forward ...).  The launch sequences live in csrc/model.hip; this file packs the reference's padded
[B, C, L] tensors into packed frames, owns the output tensors and workspaces (PyTorch = device memory and streams), and unpacks.

Batched calls give, per utterance, exactly what the reference computes one utterance at a time (its own step="test" is batch-1
only, models.py:361-362): instance-norm statistics, conv zero padding and the reverse LSTM pass all see the utterance's own
frames only.

The two frozen feature extractors are pluggable: their OUTPUTS are inputs of this path.
Pass ``features=(f0_raw, ema_raw)`` or attach modules as ``style_encoder.pitch_extractor`` (artspeech_amd.jdc.JDCNet is
the HIP one) / ``style_encoder.ema_extractor``.
"""
import ctypes

import torch

from . import _lib
from ._lib import check
from .blob import state_dict_to_blob
from .hostutil import Munch, Weights, _Module, _need_gpu, bilstm, bilstm_many, pack, unpack  # noqa: F401  (re-exported)
from .ops import Layout, layout  # noqa: F401  (re-exported)
from .spec import N_VOCAB
from .weights import DEFAULT_STATS, load_distribution

_I32P = ctypes.POINTER(ctypes.c_int32)


def stats_floats(distribution):
    """distribution dict (utils.py:86-92 / test.py:75-79) -> the 24 floats as_model_cfg.stats holds."""
    d = distribution if distribution else load_distribution(DEFAULT_STATS)
    vals = [d["energy_mean"].reshape(1), d["energy_std"].reshape(1), d["pitch_mean"].reshape(1), d["pitch_std"].reshape(1),
            d["EMA_mean"].reshape(10), d["EMA_std"].reshape(10)]
    return torch.cat([v.detach().float().cpu() for v in vals]).tolist()


def _i32(values):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])


class Runtime:
    """One loaded model on one GPU: the library's as_model handle, an as_plan (geometry tables, side streams) and the
    workspaces.  A workspace is kept and reused while it is large enough; when a call needs more, a larger one REPLACES it and the
    old tensor stays alive in `_retired` (a hipGraph captured earlier still replays into it: capture after the largest geometry, or
    call `drop_retired()` once such graphs are gone)."""

    def __init__(self, state_dict, args, distribution, device):
        self.model, self.plan = ctypes.c_void_p(), ctypes.c_void_p()
        self.device = _need_gpu(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        L = _lib.lib()
        cfg = _lib.ModelCfg()
        cfg.hidden_dim, cfg.dim_in = int(args.get("hidden_dim", 512)), int(args.get("dim_in", 64))
        cfg.style_dim, cfg.n_mels, cfg.n_token = int(args.get("style_dim", 256)), int(args.get("n_mels", 80)), int(args.get("n_token", N_VOCAB))
        for i, v in enumerate(stats_floats(distribution)):
            cfg.stats[i] = v
        self.cfg = cfg
        blob = state_dict_to_blob(state_dict)
        with torch.cuda.device(self.device):
            check(L.as_model_create(blob, len(blob), ctypes.byref(cfg), ctypes.byref(self.model)), "as_model_create")
            check(L.as_plan_create(self.model, ctypes.byref(self.plan)), "as_plan_create")
        self._ws = {}
        self._retired = []
        self._parent = None

    def fork(self):
        """A second caller of the SAME weights: its own as_plan (geometry tables, side streams) and workspaces, so that its forwards can
        be in flight on another stream while this runtime's run (include/artspeech_hip.h: one model per GPU, one plan per stream)."""
        rt = Runtime.__new__(Runtime)
        rt.model, rt.plan, rt.device, rt.cfg = self.model, ctypes.c_void_p(), self.device, self.cfg
        rt._ws, rt._retired, rt._parent = {}, [], self                  # (keeps the owner of the model alive)
        with torch.cuda.device(self.device):
            check(_lib.lib().as_plan_create(self.model, ctypes.byref(rt.plan)), "as_plan_create")
        return rt

    def __del__(self):
        try:
            L = _lib.lib()
            if self.plan:
                L.as_plan_destroy(self.plan)
            if self.model and self._parent is None:
                L.as_model_destroy(self.model)
        except Exception:
            pass

    def set_serial(self, on):
        check(_lib.lib().as_plan_set_serial(self.plan, int(on)), "as_plan_set_serial")

    def set_merge(self, on):
        """serial plans: conv GEMMs of independent branches share launches (as_plan_set_merge; on by default)"""
        check(_lib.lib().as_plan_set_merge(self.plan, int(on)), "as_plan_set_merge")

    def set_operand_mode(self, n_prod):
        """3 = f16x3 (fp32-accurate, default); 1 = plain fp16 operands (BASELINE config C2's 16-bit mode)"""
        check(_lib.lib().as_plan_set_operand_mode(self.plan, int(n_prod)), "as_plan_set_operand_mode")

    def phase_ms(self, fn):
        """run fn() with phase marks on and return the four phase times of its (last) forward in ms"""
        L = _lib.lib()
        check(L.as_plan_set_timing(self.plan, 1), "as_plan_set_timing")
        try:
            fn()
            ms = (ctypes.c_float * 4)()
            check(L.as_plan_phase_ms(self.plan, ms, 4), "as_plan_phase_ms")
        finally:
            L.as_plan_set_timing(self.plan, 0)
        return list(ms)

    def batch(self, tok_lens=None, ref_lens=None, frames=None):
        """as_batch for host length lists (the ctypes arrays are kept alive on the returned struct)."""
        n = len(tok_lens if tok_lens is not None else ref_lens if ref_lens is not None else frames)
        b = _lib.Batch()
        b.B = n
        b._keep = []
        for name, vals in (("tok_lens", tok_lens), ("ref_lens", ref_lens), ("frames", frames)):
            if vals is not None:
                arr = _i32(vals)
                b._keep.append(arr)
                setattr(b, name, ctypes.cast(arr, _I32P))
        return b

    def workspace(self, slot, module, batch):
        """(tensor, bytes) of a kept workspace for `module` on this geometry; grown when a call needs more."""
        need = _lib.lib().as_module_workspace_bytes(self.model, self.plan, module, ctypes.byref(batch))
        if need == 0:
            raise _lib.HipLibraryError("as_module_workspace_bytes: invalid geometry (reference utterances need >= 66 frames, "
                                       "SURVEY.md A9; every length must be positive)")
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need:
            if ws is not None:
                self._retired.append(ws)                   # captured graphs may still point into it
            ws = self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws, ws.numel()

    def drop_retired(self):
        """free the workspaces that were outgrown (only when no captured hipGraph replays into them any more)"""
        self._retired.clear()

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream


def _dev(t, device, dtype=torch.float32):
    return t.to(device=device, dtype=dtype).contiguous()


def _p(t):
    return t.data_ptr() if t is not None else None


def _pack_tokens(x, lens, n_token):
    tok = torch.cat([torch.as_tensor(x[b])[: int(l)].reshape(-1) for b, l in enumerate(lens)]).to(torch.int64)
    if tok.numel() and (int(tok.min()) < 0 or int(tok.max()) >= n_token):
        # nn.Embedding raises on such ids (RelTransformerEnc.py:11-16); so does this path
        raise IndexError(f"token id out of range [0, {n_token}): {int(tok.min())} .. {int(tok.max())}")
    return tok


class _Sub(_Module):
    def __init__(self, rt):
        self.rt = rt


class RelTransformerEncoder(_Sub):
    """Utils/RelTransformerEnc.py:328-380.  forward(x int64 [B,N], x_lengths [B]) -> fp32 [B,N,C]."""

    def __init__(self, rt, which):
        super().__init__(rt)
        self.which = which                    # 0 text_encoder, 1 arts_encoder, 2 durationPredictor.text_encoder

    def forward_packed(self, tokens_i32, lens):
        rt, L = self.rt, _lib.lib()
        with torch.cuda.device(rt.device):
            b = rt.batch(tok_lens=lens)
            N = sum(lens)
            out = torch.empty((rt.cfg.hidden_dim, max(N, 1)), dtype=torch.float32, device=rt.device)
            ws, nb = rt.workspace("m", _lib.AS_MOD_ENCODER, b)
            check(L.as_encoder_forward(rt.model, rt.plan, self.which, ctypes.byref(b), _p(tokens_i32), _p(out), out.stride(0), _p(ws), nb,
                                       rt.stream()), "as_encoder_forward")
        return out

    def forward(self, x, x_lengths):
        lens = [int(v) for v in x_lengths]
        tok = _dev(_pack_tokens(x, lens, self.rt.cfg.n_token), self.rt.device, torch.int32)
        return unpack(self.forward_packed(tok, lens), layout(lens, self.rt.device)).transpose(1, 2)


class StyleEncoder(_Sub):
    """models.py:373-472.  forward(mel [B,80,T], mel_input_length, step, distribution, epoch) ->
    (f0_ext [B,1,T], n_ext [B,1,T], ema_ext [B,10,T], Style [B,512]).  The frozen extractors are pluggable
    torch modules; ``features=(f0_raw, ema_raw)`` bypasses them.  The normalisation statistics are the ones the
    model was built with (build_model's ``distribution``); the argument here is accepted for the call surface."""

    def __init__(self, rt):
        super().__init__(rt)
        self.pitch_extractor = None
        self.ema_extractor = None

    def _extract(self, mel, features, lengths=None):
        """(f0_raw, ema_raw) as models.py:431-433 computes them; an entry of `features` that is given is used as is.  The
        HIP extractors (artspeech_amd.jdc / .ema) get the utterance lengths, so every item equals its B = 1 result; a
        torch module in the slot is called exactly as the reference calls it."""
        f0, ema = features if features is not None else (None, None)
        if f0 is not None and ema is not None:
            return f0, ema
        if (f0 is None and self.pitch_extractor is None) or (ema is None and self.ema_extractor is None):
            raise RuntimeError("StyleEncoder: attach pitch_extractor / ema_extractor (artspeech_amd.jdc.JDCNet and "
                               "artspeech_amd.ema.EMA_Predictor, or the reference's torch modules) or pass features=(f0_raw, ema_raw)")
        kw = lambda ext: {"lengths": lengths} if (lengths is not None and hasattr(ext, "forward_packed")) else {}
        with torch.no_grad():
            if f0 is None:
                f0 = self.pitch_extractor(mel.unsqueeze(1), **kw(self.pitch_extractor))    # models.py:432
            if ema is None:
                m = mel.to(f0.device)
                n_raw = torch.log(torch.exp(m.unsqueeze(1) * 4 - 4).norm(dim=2))           # models.py:431, :655-660
                ema = self.ema_extractor(f0, n_raw, m, **kw(self.ema_extractor))           # models.py:433
        return f0, ema

    def forward(self, mel, mel_input_length, step="second", distribution=None, epoch=20, features=None):
        rt, L = self.rt, _lib.lib()
        dev = rt.device
        lens = [int(v) for v in mel_input_length]
        f0_raw, ema_raw = self._extract(mel, features, lens)
        with torch.cuda.device(dev):
            lay = layout(lens, dev)
            mel_p, f0_p, ema_p = pack(mel.to(dev), lens), pack(f0_raw.to(dev).reshape(len(lens), 1, -1), lens), pack(ema_raw.to(dev), lens)
            b = rt.batch(ref_lens=lens)
            feat12 = torch.empty((12, max(lay.N, 1)), dtype=torch.float32, device=dev)
            style = torch.empty((len(lens), 2 * rt.cfg.style_dim), dtype=torch.float32, device=dev)
            ws, nb = rt.workspace("m", _lib.AS_MOD_STYLE, b)
            check(L.as_style_forward(rt.model, rt.plan, ctypes.byref(b), _p(mel_p), mel_p.stride(0), _p(f0_p), _p(ema_p), ema_p.stride(0),
                                     _p(feat12), feat12.stride(0), _p(style), _p(ws), nb, rt.stream()), "as_style_forward")
        return unpack(feat12[1:2], lay), unpack(feat12[0:1], lay), unpack(feat12[2:12], lay), style


class DurationPredictor(_Sub):
    """models.py:519-571.  forward(texts [B,N], style=ema_ext [B,10,T], text_lengths, mel_input_length) -> [B,N]."""

    def forward(self, texts, style, text_lengths, mel_input_length):
        rt, L = self.rt, _lib.lib()
        dev = rt.device
        tl, ml = [int(v) for v in text_lengths], [int(v) for v in mel_input_length]
        with torch.cuda.device(dev):
            tok = _dev(_pack_tokens(texts, tl, rt.cfg.n_token), dev, torch.int32)
            ema_p = pack(style.to(dev), ml)
            b = rt.batch(tok_lens=tl, ref_lens=ml)
            dur = torch.empty((1, max(sum(tl), 1)), dtype=torch.float32, device=dev)
            ws, nb = rt.workspace("m", _lib.AS_MOD_DURATION, b)
            check(L.as_duration_forward(rt.model, rt.plan, ctypes.byref(b), _p(tok), _p(ema_p), ema_p.stride(0), _p(dur), _p(ws), nb,
                                        rt.stream()), "as_duration_forward")
        return unpack(dur, layout(tl, dev))[:, 0, :]


class ArtsPredictor(_Sub):
    """models.py:573-621.  forward(A_ens [B,512,M], style [B,512]) -> F0 [B,1,2M], N [B,1,2M], EMA [B,10,2M]."""

    def forward(self, A_ens, style, lengths=None):
        rt, L = self.rt, _lib.lib()
        dev = rt.device
        lens = [A_ens.shape[-1]] * A_ens.shape[0] if lengths is None else [int(v) for v in lengths]
        with torch.cuda.device(dev):
            a = pack(A_ens.to(dev), lens)
            st = _dev(style, dev)
            n2 = 2 * sum(lens)
            f0, n, ema = (torch.empty((c, max(n2, 1)), dtype=torch.float32, device=dev) for c in (1, 1, 10))
            b = rt.batch(frames=lens)
            ws, nb = rt.workspace("m", _lib.AS_MOD_ARTS, b)
            check(L.as_arts_forward(rt.model, rt.plan, ctypes.byref(b), _p(a), a.stride(0), _p(st), _p(f0), _p(n), _p(ema), f0.stride(0), _p(ws),
                                    nb, rt.stream()), "as_arts_forward")
        lay2 = layout([2 * l for l in lens], dev)
        return unpack(f0, lay2), unpack(n, lay2), unpack(ema, lay2)


class Decoder(_Sub):
    """models.py:474-517.  forward(asr [B,512,M], Style [B,512], F0 [B,1,2M], N [B,1,2M], EMA [B,10,2M]) -> [B,80,2M]."""

    def forward(self, asr, Style, F0, N, EMA, lengths=None):
        rt, L = self.rt, _lib.lib()
        dev = rt.device
        lens = [asr.shape[-1]] * asr.shape[0] if lengths is None else [int(v) for v in lengths]
        l2 = [2 * l for l in lens]
        with torch.cuda.device(dev):
            a = pack(asr.to(dev), lens)
            st = _dev(Style, dev)
            f0, n, ema = pack(F0.to(dev), l2), pack(N.to(dev), l2), pack(EMA.to(dev), l2)
            mel = torch.empty((rt.cfg.n_mels, max(sum(l2), 1)), dtype=torch.float32, device=dev)
            b = rt.batch(frames=lens)
            ws, nb = rt.workspace("m", _lib.AS_MOD_DECODER, b)
            check(L.as_decoder_forward(rt.model, rt.plan, ctypes.byref(b), _p(a), a.stride(0), _p(st), _p(f0), _p(n), _p(ema), f0.stride(0),
                                       _p(mel), mel.stride(0), _p(ws), nb, rt.stream()), "as_decoder_forward")
        return unpack(mel, layout(l2, dev))


class ArtsSpeech(_Module):
    """models.py:275-371, inference branch only: forward(batch, s2s_attn, s2s_attn_mono, step="test")."""

    def __init__(self, args, stage="second", distribution=None, device=None, state_dict=None):
        if stage == "first":
            raise NotImplementedError("only the inference path (stage='second' modules, step='test') is built")
        self.args = args
        self.device = _need_gpu(device)
        self.distribution = distribution if distribution else load_distribution(DEFAULT_STATS)
        self.rt = None
        self.style_encoder = StyleEncoder(None)      # (extractors may be attached before the weights are loaded)
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def replica(self):
        """Another ArtsSpeech on the same weights with its own plan and workspaces (Runtime.fork): run it on a second stream to keep two
        batches in flight."""
        twin = ArtsSpeech(self.args, "second", self.distribution, self.device)
        twin.style_encoder = self.style_encoder
        twin._bind(self.rt.fork())
        return twin

    def load_state_dict(self, sd, strict=False):
        return self._bind(Runtime(sd, dict(self.args), self.distribution, self.device))

    def _bind(self, rt):
        self.rt = rt
        self.device = rt.device
        self.text_encoder = RelTransformerEncoder(rt, 0)
        self.arts_encoder = RelTransformerEncoder(rt, 1)
        ext = self.style_encoder
        self.style_encoder = StyleEncoder(rt)
        if ext is not None:
            self.style_encoder.pitch_extractor, self.style_encoder.ema_extractor = ext.pitch_extractor, ext.ema_extractor
        self.durationPredictor = DurationPredictor(rt)
        self.durationPredictor.text_encoder = RelTransformerEncoder(rt, 2)
        self.artsPredictor = ArtsPredictor(rt)
        self.decoder = Decoder(rt)
        return self

    def forward(self, batch, s2s_attn=None, s2s_attn_mono=None, step="test", mode="train", epoch=0, features=None,
                forced_durations=None, return_aux=False):
        if step != "test":
            raise NotImplementedError("training branches (step='first'/'second') are out of scope (SURVEY.md section 2)")
        if self.rt is None:
            raise RuntimeError("no weights loaded: call load_checkpoint / load_state_dict first")
        texts, input_lengths, mels, mel_input_length = batch[0], batch[1], batch[2], batch[3]   # 7- or 4-tuple (B1)
        dev = self.device
        tl = [int(v) for v in input_lengths]
        ml = [int(v) for v in mel_input_length]
        B = len(tl)
        f0_raw, ema_raw = self.style_encoder._extract(mels, features, ml)
        with torch.cuda.device(dev):
            tok = _dev(_pack_tokens(texts, tl, self.rt.cfg.n_token), dev, torch.int32)
            mel_p = pack(mels.to(dev), ml)
            f0_p = pack(f0_raw.to(dev).reshape(B, 1, -1), ml)
            ema_p = pack(ema_raw.to(dev), ml)
            forced, frames = None, None
            if forced_durations is not None:
                fd = [torch.as_tensor(forced_durations[b])[: tl[b]].reshape(-1) for b in range(B)]
                forced = torch.cat(fd).to(device=dev, dtype=torch.int32)
                frames = [int(f.sum()) for f in fd]
            out = self.forward_packed(tok, tl, mel_p, f0_p, ema_p, ml, forced=forced, frames_hint=frames, aux=return_aux)
            mel = unpack(out["mel"], layout(out["frames2"], dev))
        if return_aux:
            return mel, out
        return mel

    def forward_packed(self, tok, tok_lens, mel_p, f0_p, ema_p, ref_lens, forced=None, frames_hint=None, aux=False, out=None, frame_cap=None):
        """The whole hot path on packed tensors (what bench.py times): one call of as_forward_test when the integer frame
        counts are known (forced durations), else as_forward_test_begin -> one device->host read of B + 1 integers ->
        as_forward_test_finish.  `out`: the dict of a previous call with the same geometry (its tensors are reused).
        frame_cap (predicted durations only): the half-rate frames to make room for, all utterances together -- ONE as_forward_test call
        with no read-back (capturable); the result's `frame_off` (device, [B + 1]) says where each utterance's frames lie in `mel`
        [n_mels][2 frame_cap], `frames` is None; more frames than room raises the AS_STATUS_CAPACITY bit (as_device_status)."""
        if frame_cap is not None:
            return self._forward_packed_cap(tok, tok_lens, mel_p, f0_p, ema_p, ref_lens, int(frame_cap), aux, out)
        rt, L = self.rt, _lib.lib()
        dev = rt.device
        tok_lens, ref_lens = [int(v) for v in tok_lens], [int(v) for v in ref_lens]
        B, Nt, Nr, C = len(tok_lens), sum(tok_lens), sum(ref_lens), rt.cfg.hidden_dim
        with torch.cuda.device(dev):
            io = _lib.ForwardIO()
            io.tokens, io.mel, io.ld_mel = _p(tok), _p(mel_p), mel_p.stride(0)
            io.f0_raw, io.ema_raw, io.ld_ema = _p(f0_p), _p(ema_p), ema_p.stride(0)
            io.forced_dur = _p(forced)
            res = out if out is not None else {}

            def new(key, shape, dtype=torch.float32):
                # a tensor of a previous call is reused only if it is exactly what this call needs: predicted durations depend on
                # the input VALUES, so the frame count (and with it mel / F0 / N / EMA) can change under an unchanged geometry
                t = res.get(key)
                if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev:
                    t = res[key] = torch.empty(shape, dtype=dtype, device=dev)
                return t

            io.dur_i, io.frame_off = _p(new("dur_i", (max(Nt, 1),), torch.int32)), _p(new("frame_off", (B + 1,), torch.int32))
            if aux:
                io.duration = _p(new("duration", (1, max(Nt, 1))))
                io.style = _p(new("style", (B, 2 * rt.cfg.style_dim)))
                io.feat12, io.ld_feat = _p(new("feat12", (12, max(Nr, 1)))), max(Nr, 1)
                io.t_en, io.a_en, io.ld_en = _p(new("t_en", (C, max(Nt, 1)))), _p(new("a_en", (C, max(Nt, 1)))), max(Nt, 1)
            ba = rt.batch(tok_lens=tok_lens, ref_lens=ref_lens, frames=frames_hint)
            ws_a, na = rt.workspace("a", _lib.AS_MOD_FORWARD_A, ba)
            s = rt.stream()
            frames = frames_hint
            if frames is None:
                check(L.as_forward_test_begin(rt.model, rt.plan, ctypes.byref(ba), ctypes.byref(io), _p(ws_a), na, s), "as_forward_test_begin")
                off = res["frame_off"].cpu().tolist()                            # the one device->host sync
                frames = [off[b + 1] - off[b] for b in range(B)]
                ba = rt.batch(tok_lens=tok_lens, ref_lens=ref_lens, frames=frames)
            frames = [int(f) for f in frames]
            n2 = 2 * sum(frames)
            io.mel_out, io.ld_out = _p(new("mel", (rt.cfg.n_mels, max(n2, 1)))), max(n2, 1)
            if aux:
                io.F0, io.N, io.EMA = _p(new("F0", (1, max(n2, 1)))), _p(new("N", (1, max(n2, 1)))), _p(new("EMA", (10, max(n2, 1))))
                io.ld_pred = max(n2, 1)
            ws_b, nb = rt.workspace("b", _lib.AS_MOD_FORWARD_B, ba)
            if frames_hint is None:
                check(L.as_forward_test_finish(rt.model, rt.plan, ctypes.byref(ba), ctypes.byref(io), _p(ws_a), na, _p(ws_b), nb, s),
                      "as_forward_test_finish")
            else:
                check(L.as_forward_test(rt.model, rt.plan, ctypes.byref(ba), ctypes.byref(io), _p(ws_a), na, _p(ws_b), nb, None, s),
                      "as_forward_test")
            res["frames"], res["frames2"] = frames, [2 * f for f in frames]
        return res


def _forward_packed_cap(self, tok, tok_lens, mel_p, f0_p, ema_p, ref_lens, frame_cap, aux, out):
    rt, L = self.rt, _lib.lib()
    dev = rt.device
    tok_lens, ref_lens = [int(v) for v in tok_lens], [int(v) for v in ref_lens]
    B, Nt, Nr, C = len(tok_lens), sum(tok_lens), sum(ref_lens), rt.cfg.hidden_dim
    with torch.cuda.device(dev):
        io = _lib.ForwardIO()
        io.tokens, io.mel, io.ld_mel = _p(tok), _p(mel_p), mel_p.stride(0)
        io.f0_raw, io.ema_raw, io.ld_ema = _p(f0_p), _p(ema_p), ema_p.stride(0)
        io.frame_cap = frame_cap
        res = out if out is not None else {}

        def new(key, shape, dtype=torch.float32):
            t = res.get(key)
            if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev:
                t = res[key] = torch.empty(shape, dtype=dtype, device=dev)
            return t
        n2 = 2 * frame_cap
        io.dur_i, io.frame_off = _p(new("dur_i", (max(Nt, 1),), torch.int32)), _p(new("frame_off", (B + 1,), torch.int32))
        io.mel_out, io.ld_out = _p(new("mel", (rt.cfg.n_mels, n2))), n2
        if aux:
            io.duration = _p(new("duration", (1, max(Nt, 1))))
            io.style = _p(new("style", (B, 2 * rt.cfg.style_dim)))
            io.feat12, io.ld_feat = _p(new("feat12", (12, max(Nr, 1)))), max(Nr, 1)
            io.t_en, io.a_en, io.ld_en = _p(new("t_en", (C, max(Nt, 1)))), _p(new("a_en", (C, max(Nt, 1)))), max(Nt, 1)
            io.F0, io.N, io.EMA = _p(new("F0", (1, n2))), _p(new("N", (1, n2))), _p(new("EMA", (10, n2)))
            io.ld_pred = n2
        ba = rt.batch(tok_lens=tok_lens, ref_lens=ref_lens)
        ws_a, na = rt.workspace("a", _lib.AS_MOD_FORWARD_A, ba)
        bc = rt.batch(tok_lens=tok_lens, ref_lens=ref_lens, frames=[frame_cap] + [0] * (B - 1))     # (only the sum counts)
        ws_b, nb = rt.workspace("b", _lib.AS_MOD_FORWARD_B_CAP, bc)
        check(L.as_forward_test(rt.model, rt.plan, ctypes.byref(ba), ctypes.byref(io), _p(ws_a), na, _p(ws_b), nb, None, rt.stream()),
              "as_forward_test")
        res["frames"], res["frames2"], res["frame_cap"] = None, None, frame_cap
    return res


ArtsSpeech._forward_packed_cap = _forward_packed_cap


class Lanes:
    """as_lanes (csrc/lanes.hip): n batches in flight on one model's weights, every batch one chain on a stream of its own -- the
    throughput arrangement as a piece of the library.  `submit` takes the packed tensors of `forward_packed`;
    keep them (and the returned dict) alive and unchanged until `wait`."""

    def __init__(self, net, n_lanes=4):
        self.rt = net.rt
        self.h = ctypes.c_void_p()
        check(_lib.lib().as_lanes_create(self.rt.model, n_lanes, ctypes.byref(self.h)), "as_lanes_create")
        self.n = n_lanes
        self._keep = [None] * n_lanes
        self.coalesce = 1

    def close(self):
        if self.h:
            _lib.lib().as_lanes_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_coalesce(self, k):
        """as_lanes_set_coalesce: submissions whose tensors are adjacent column ranges of one block (views: `block[:, a:b]`) and that
        bring their own `out["mel"]` view of one output block are launched k at a time as ONE call"""
        check(_lib.lib().as_lanes_set_coalesce(self.h, int(k)), "as_lanes_set_coalesce")
        self.coalesce = int(k)

    def flush(self):
        check(_lib.lib().as_lanes_flush(self.h), "as_lanes_flush")

    def merged_calls(self, lane):
        return int(_lib.lib().as_lanes_merged_calls(self.h, int(lane)))

    def submit(self, tok, tok_lens, mel_p, f0_p, ema_p, ref_lens, forced=None, frames=None, out=None, capacity=None, frame_cap=None):
        """-> (lane, dict with the output tensors).  frames (per-utterance half-rate frame counts) known: graph-replayed from the second
        submit of the same tensors on a lane; None: predicted durations, `capacity` = the mel frames the output buffer is made for (eager,
        one read-back per call) -- or frame_cap = the half-rate frames to make room for (as_forward_io.frame_cap: no read-back, replayed
        from hipGraphs and coalesced like a submission with known counts; `frame_off` of the result says where the utterances lie).
        2-D tensors may be column ranges of a wider block (row stride = the block's width)."""
        def _p(t):                                                  # (rows of a wider block: only the last axis has to be dense)
            if t is None:
                return None
            if not t.is_cuda or t.device != dev:
                raise _lib.HipLibraryError("expected a tensor on the model's GPU")
            if t.stride(-1) != 1:
                raise _lib.HipLibraryError("expected rows that are dense along the column axis")
            return t.data_ptr()
        rt, L = self.rt, _lib.lib()
        dev = rt.device
        tok_lens, ref_lens = [int(v) for v in tok_lens], [int(v) for v in ref_lens]
        B, Nt = len(tok_lens), sum(tok_lens)
        res = out if out is not None else {}
        with torch.cuda.device(dev):
            io = _lib.ForwardIO()
            io.tokens, io.mel, io.ld_mel = _p(tok), _p(mel_p), mel_p.stride(0)
            io.f0_raw, io.ema_raw, io.ld_ema = _p(f0_p), _p(ema_p), ema_p.stride(0)
            io.forced_dur = _p(forced)
            n2 = 2 * sum(int(f) for f in frames) if frames is not None else (2 * int(frame_cap) if frame_cap is not None else int(capacity))
            if frame_cap is not None:
                if frames is not None or forced is not None:
                    raise _lib.HipLibraryError("frame_cap goes with predicted durations (no frames, no forced durations)")
                io.frame_cap = int(frame_cap)
            if "mel" not in res:
                res["mel"] = torch.empty((rt.cfg.n_mels, max(n2, 1)), dtype=torch.float32, device=dev)
            # a submission that a lane may hold back for its group (frames known, coalescing on) has no per-submission home for the
            # optional outputs; every other one -- predicted durations above all, where frame_off is the only record of the split -- has
            can_merge = self.coalesce > 1 and (frames is not None or frame_cap is not None)
            if not can_merge and "dur_i" not in res:
                res["dur_i"] = torch.empty((max(Nt, 1),), dtype=torch.int32, device=dev)
            if (not can_merge or frame_cap is not None) and "frame_off" not in res:
                res["frame_off"] = torch.empty((B + 1,), dtype=torch.int32, device=dev)
            io.mel_out, io.ld_out = _p(res["mel"]), res["mel"].stride(0)
            if not can_merge:
                io.dur_i = _p(res["dur_i"])
            if not can_merge or frame_cap is not None:            # (under a capacity every submission of a merged call gets its own frame_off)
                io.frame_off = _p(res["frame_off"])
            ba = rt.batch(tok_lens=tok_lens, ref_lens=ref_lens, frames=frames)
            fr = (ctypes.c_int32 * B)()
            lane = ctypes.c_int32(-1)
            check(L.as_lanes_submit(self.h, ctypes.byref(ba), ctypes.byref(io), fr, ctypes.byref(lane)), "as_lanes_submit")
            res["frames"] = None if frame_cap is not None else ([int(v) for v in fr] if frames is None else [int(f) for f in frames])
            # (a lane's last group of submissions stays referenced: its launch may still be reading them)
            prev = self._keep[lane.value] or []
            self._keep[lane.value] = (prev + [(ba, io, tok, mel_p, f0_p, ema_p, forced, res)])[-2 * max(self.coalesce, 1):]
        return lane.value, res

    def set_debug(self, on=True):
        """as_lanes_set_debug: the device inputs of a held-back submission are checksummed at submit and at its group's launch"""
        check(_lib.lib().as_lanes_set_debug(self.h, int(bool(on))), "as_lanes_set_debug")

    def submit_host(self, tok, tok_lens, mel_p, f0_p, ema_p, ref_lens, forced, frames, out_mel, frame_cap=None, frame_off=None):
        """as_lanes_submit_host: HOST tensors in (pinned: `.pin_memory()`), the mel back into the host tensor `out_mel` [n_mels][>= 2 sum
        frames]; the lane owns the device side (its block, the copies, the group's launch).  -> lane.  Keep the tensors alive and unchanged
        until `wait(lane)`; `out_mel` is valid after it.  Predicted durations: frames=None, forced=None, frame_cap = the half-rate frames
        there is room for (out_mel [n_mels][>= 2 frame_cap]) and frame_off = a host int32 tensor [B + 1] that receives the offsets."""
        def _h(t):
            if t is None:
                return None
            if t.is_cuda or t.stride(-1) != 1:
                raise _lib.HipLibraryError("expected a host tensor whose rows are dense")
            return t.data_ptr()
        tok_lens, ref_lens = [int(v) for v in tok_lens], [int(v) for v in ref_lens]
        frames = [int(v) for v in frames] if frames is not None else None
        with torch.cuda.device(self.rt.device):
            io = _lib.HostIO()
            if frames is None:
                io.frame_cap, io.frame_off = int(frame_cap), _h(frame_off)
            io.tokens, io.mel, io.ld_mel = _h(tok), _h(mel_p), mel_p.stride(0)
            io.f0_raw, io.ema_raw, io.ld_ema = _h(f0_p), _h(ema_p), ema_p.stride(0)
            io.forced_dur = _h(forced)
            io.mel_out, io.ld_out = _h(out_mel), out_mel.stride(0)
            ba = self.rt.batch(tok_lens=tok_lens, ref_lens=ref_lens, frames=frames)
            lane = ctypes.c_int32(-1)
            check(_lib.lib().as_lanes_submit_host(self.h, ctypes.byref(ba), ctypes.byref(io), ctypes.byref(lane)), "as_lanes_submit_host")
            prev = self._keep[lane.value] or []
            self._keep[lane.value] = (prev + [(ba, io, tok, mel_p, f0_p, ema_p, forced, out_mel, frame_off)])[-2 * max(self.coalesce, 1):]
        return lane.value

    def wait(self, lane=-1):
        check(_lib.lib().as_lanes_wait(self.h, lane), "as_lanes_wait")

    def set_graph_cap(self, max_graphs):
        check(_lib.lib().as_lanes_set_graph_cap(self.h, int(max_graphs)), "as_lanes_set_graph_cap")

    def set_layout_cap(self, max_layouts):
        check(_lib.lib().as_lanes_set_layout_cap(self.h, int(max_layouts)), "as_lanes_set_layout_cap")

    def reserve(self, bytes_a, bytes_b):
        check(_lib.lib().as_lanes_reserve(self.h, int(bytes_a), int(bytes_b)), "as_lanes_reserve")

    def stats(self, lane):
        """dict(graphs, graph_drops, layout_flushes, graph_launches, eager_calls, captures) of one lane"""
        v = (ctypes.c_int64 * 6)()
        check(_lib.lib().as_lanes_stats(self.h, int(lane), v), "as_lanes_stats")
        return dict(zip(("graphs", "graph_drops", "layout_flushes", "graph_launches", "eager_calls", "captures"), [int(x) for x in v]))


def build_model(args, text_aligner=None, stage="second", distribution=None, device=None):
    """models.py:680-683.  Returns Munch(ArtsSpeech, discriminator, text_aligner); the discriminator is a
    training-only component and is None here."""
    if not isinstance(args, dict):
        args = Munch(vars(args))
    return Munch(ArtsSpeech=ArtsSpeech(Munch(args), stage, distribution=distribution, device=device),
                 discriminator=None, text_aligner=text_aligner)


def load_checkpoint(model, optimizer, path, load_only_params=True):
    """models.py:685-701: torch.load(path)['net'][key] per top-level key, non-strict, eval mode."""
    state = path if isinstance(path, dict) else torch.load(path, map_location="cpu")
    params = state["net"]
    for key in model:
        if key in params and model[key] is not None and hasattr(model[key], "load_state_dict"):
            model[key].load_state_dict(params[key], False)
    if not load_only_params:
        epoch, iters = state["epoch"], state["iters"]
    else:
        epoch, iters = 0, 0
    return model, optimizer, epoch, iters
