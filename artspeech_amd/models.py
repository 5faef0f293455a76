"""Host side of the acoustic-model inference path: the reference's models.py call surface
(build_model / load_checkpoint / ArtsSpeech.forward(step="test") and the sub-module forwards,
SURVEY.md 8(b) row B1) on top of the HIP kernels in csrc/.

Inside, every activation is a packed-frames tensor [C][N] (DESIGN.md "Data layout"): all utterances
of the batch concatenated along the contiguous axis, no padding.  Batched calls therefore give, per
utterance, exactly what the reference computes one utterance at a time (the reference's own
step="test" is batch-1 only, models.py:361-362) -- instance-norm statistics, conv zero padding and
the reverse LSTM pass all see the utterance's own frames only.

The two frozen feature extractors are pluggable: their OUTPUTS are inputs of this path.
Pass ``features=(f0_raw, ema_raw)`` or attach modules as ``style_encoder.pitch_extractor`` (artspeech_amd.jdc.JDCNet is
the HIP one) / ``style_encoder.ema_extractor``.
"""
import math

import torch

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, Layout, layout, taps_1d, taps_2d
from .spec import N_HEADS, WINDOW
from .weights import DEFAULT_STATS, fold_state_dict, load_distribution


class Munch(dict):
    """attribute-access dict (what the reference takes from the `munch` package)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _need_gpu(device):
    if not torch.cuda.is_available():
        raise ops._lib.HipLibraryError("the HIP path needs a GPU: torch.cuda.is_available() is False (no CPU fallback)")
    return torch.device(device if device is not None else "cuda")


# ------------------------------------------------------------------------------------------------
# Weight store: folded fp32 weights on the device in the layouts the kernels want
# ------------------------------------------------------------------------------------------------
class Weights:
    def __init__(self, folded, device):
        self.raw = folded                   # name -> CPU fp32 tensor (folded: plain `.weight` keys)
        self.device = device
        self._cache = {}

    def has(self, name):
        return name in self.raw

    def vec(self, name):
        """a tensor as-is (bias, gamma, table ...), on the device."""
        if name not in self._cache:
            self._cache[name] = self.raw[name].to(self.device).contiguous()
        return self._cache[name]

    def conv(self, name, name2=None):
        """conv weight [Cout,Cin,k] or [Cout,Cin,kh,kw] -> the GEMM's operand image (ops.GemmWeight).  name2: a second layer of
        the same shape stacked behind it (a grouped launch: the twin encoders)."""
        key = "T:" + name + ("|" + name2 if name2 else "")
        if key not in self._cache:
            self._cache[key] = ops.prep_weight(self.raw[name + ".weight"], self.device,
                                               stack=[self.raw[name2 + ".weight"]] if name2 else None)
        return self._cache[key]

    def bias(self, name, name2=None):
        if not self.has(name + ".bias"):
            return None
        if name2 is None:
            return self.vec(name + ".bias")
        return self.cached("B2:" + name + "|" + name2, lambda: torch.stack([self.raw[name + ".bias"], self.raw[name2 + ".bias"]], 0)
                           .contiguous().to(self.device))

    def qkv(self, p, p2=None):
        key = "QKV:" + p + ("|" + p2 if p2 else "")
        if key not in self._cache:
            ws = [torch.cat([self.raw[f"{q}.conv_{n}.weight"] for n in "qkv"], 0) for q in ([p, p2] if p2 else [p])]      # [3C, C, 1]
            bs = [torch.cat([self.raw[f"{q}.conv_{n}.bias"] for n in "qkv"], 0) for q in ([p, p2] if p2 else [p])]
            self._cache[key] = (ops.prep_weight(ws[0], self.device, stack=ws[1:]),
                                (torch.stack(bs, 0) if p2 else bs[0]).contiguous().to(self.device))
        return self._cache[key]

    def lstm(self, p):
        key = "LSTM:" + p
        if key not in self._cache:
            r = self.raw
            w_ih = torch.cat([r[p + ".weight_ih_l0"], r[p + ".weight_ih_l0_reverse"]], 0)          # [8H, I]
            b = torch.cat([r[p + ".bias_ih_l0"] + r[p + ".bias_hh_l0"],
                           r[p + ".bias_ih_l0_reverse"] + r[p + ".bias_hh_l0_reverse"]], 0)
            whh_t = torch.stack([r[p + ".weight_hh_l0"].t().contiguous(), r[p + ".weight_hh_l0_reverse"].t().contiguous()], 0)
            H = r[p + ".weight_hh_l0"].shape[1]
            self._cache[key] = (ops.prep_weight(w_ih[:, :, None], self.device), b.to(self.device), whh_t.to(self.device), H)
        return self._cache[key]

    def cached(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    def dw(self, name):
        """depthwise weight [C,1,...] -> [C][kh*3]."""
        key = "DW:" + name
        if key not in self._cache:
            w = self.raw[name + ".weight"]
            self._cache[key] = w.reshape(w.shape[0], -1).contiguous().to(self.device)
        return self._cache[key]


# ------------------------------------------------------------------------------------------------
# building blocks on packed frames
# ------------------------------------------------------------------------------------------------
def conv1d(W, name, X, lay, k, Y=None, name2=None, **kw):
    Wt = W.conv(name, name2)
    if Y is None:
        Y = lay.new(Wt.shape[2])
    return ops.conv_gemm(Wt, X, lay, Y, taps_1d(k), bias=W.bias(name, name2), **kw)


import os as _os

SHORTCUT_FORK = _os.environ.get("AS_SHORTCUT_FORK", "0") != "0"     # experiment (no gain measured): 1x1 shortcut on a side stream
LN_SPLIT = _os.environ.get("AS_LN_SPLIT", "1") != "0"               # encoder LayerNorms write the following conv's pre-split operand image
ENC_PAIR = _os.environ.get("AS_ENC_PAIR", "1") != "0"               # text + articulatory encoders as one double-width encoder
TOWER_BRANCHES = int(_os.environ.get("AS_TOWER_BRANCHES", "1"))     # streams for the TV / F0 / energy towers (1: back to back on one)
ADAIN_SPLIT = _os.environ.get("AS_ADAIN_SPLIT", "1") != "0"         # AdaIN writes the following conv's pre-split operand image


def adain_gb(W, p, style):
    """gamma/beta of one AdaIN1d: fc(style) -> [B][2C]   (models.py:237)."""
    return ops.linear_rows(style, W.vec(p + ".fc.weight"), W.vec(p + ".fc.bias"))


def adain_fc_batch(W, prefixes, style):
    """All AdaIN fc layers of several AdainResBlk1d blocks that take the SAME style vector, as one launch over
    the row-concatenated weights.  Returns {block prefix: (gamma_beta_norm1, gamma_beta_norm2)} (row-strided views)."""
    names = [f"{p}.{n}" for p in prefixes for n in ("norm1", "norm2")]
    key = "ADAINFC:" + "|".join(names)
    wcat, bcat, offs = W.cached(key, lambda: (
        torch.cat([W.raw[n + ".fc.weight"] for n in names], 0).contiguous().to(W.device),
        torch.cat([W.raw[n + ".fc.bias"] for n in names], 0).contiguous().to(W.device),
        [W.raw[n + ".fc.weight"].shape[0] for n in names]))
    gb = ops.linear_rows(style, wcat, bcat)
    out, o = {}, 0
    for i, p in enumerate(prefixes):
        n1, n2 = offs[2 * i], offs[2 * i + 1]
        out[p] = (gb[:, o:o + n1], gb[:, o + n1:o + n1 + n2])
        o += n1 + n2
    return out


def adain_resblk1d(W, p, X, lay, style, out=None, upsample=False, gb=None, fork_shortcut=False):
    """AdainResBlk1d.forward (models.py:189-202).  X [din][N] -> [dout][N or 2N].  Returns (Y, layout).
    gb: this block's precomputed (norm1, norm2) gamma/beta from adain_fc_batch.  The learned 1x1 shortcut
    (models.py:185-186) only depends on X, so it runs on a side stream next to norm1 -> conv1 -> norm2."""
    din = X.shape[0]
    gb1, gb2 = gb if gb is not None else (adain_gb(W, p + ".norm1", style), adain_gb(W, p + ".norm2", style))
    has_sc = W.has(p + ".conv1x1.weight")
    dout = W.conv(p + ".conv1").shape[2]
    lay2 = lay.scaled(2) if upsample else lay
    if out is None:
        out = lay2.new(dout)
    fork = None
    if has_sc and not upsample and fork_shortcut and SHORTCUT_FORK:
        cur = torch.cuda.current_stream()
        fork = Fork(side_streams(W.device, 1, f"shortcut{cur.stream_id}"), uses=(X, out))
        fork.__enter__()
        with fork(0):
            conv1d(W, p + ".conv1x1", X, lay2, 1, Y=out)
    if upsample:
        h = lay2.new(din)
        sc = lay2.new(din)
        ops.adain(X, gb1, lay, h, True, W.dw(p + ".pool"), W.vec(p + ".pool.bias"), sc)
    split = ADAIN_SPLIT                                # norm -> actv feeds only the conv: store it as that conv's operand image
    if not upsample:
        sc = X
    if upsample or not split:
        if not upsample:
            h = ops.adain(X, gb1, lay, lay.new(din), True)
        h = conv1d(W, p + ".conv1", h, lay2, 3)
    else:
        h = conv1d(W, p + ".conv1", None, lay2, 3, xs=ops.adain_split(X, gb1, lay), K=din)
    h2 = None if split else ops.adain(h, gb2, lay2, lay2.new(dout), True)
    if fork is not None:
        fork.__exit__(None, None, None)
        sc = out
    elif has_sc:
        sc = conv1d(W, p + ".conv1x1", sc, lay2, 1, Y=out)
    if split:
        conv1d(W, p + ".conv2", None, lay2, 3, Y=out, res=sc, div_sqrt2=True, xs=ops.adain_split(h, gb2, lay2), K=dout)
    else:
        conv1d(W, p + ".conv2", h2, lay2, 3, Y=out, res=sc, div_sqrt2=True)
    return out, lay2


def bilstm(W, p, X, lay):
    """nn.LSTM(bidirectional) on packed [I][N] -> [2H][N]: hoisted input GEMM + recurrence kernel."""
    return bilstm_many(W, [(p, X)], lay)[0]


def bilstm_many(W, items, lay):
    """Several independent BiLSTMs of the same size over the same layout (ArtsPredictor's three branches,
    models.py:606-618): one hoisted input GEMM each, ONE recurrence launch for all of them."""
    jobs, H = [], None
    for p, X in items:
        wih_t, b, whh_t, H = W.lstm(p)
        gx = torch.empty((max(lay.N, 1), 8 * H), dtype=torch.float32, device=X.device)
        ops.conv_gemm(wih_t, X, lay, gx, [(0, 0)], bias=b, transpose_out=True)
        jobs.append((gx, whh_t, lay.new(2 * H)))
    return ops.bilstm(jobs, lay, H)


def rel_encoder(W, p, tokens_i32, lay, n_layers, p2=None, n_split=0, b_split=0):
    """RelTransformerEncoder.forward (RelTransformerEnc.py:371-380) on packed tokens -> [C][N].
    p2: a SECOND encoder of the same shape (the text and articulatory encoders are twins on the same tokens) whose weights
    serve the columns >= n_split / utterances >= b_split of `lay`: both run as one double-width launch sequence."""
    emb = W.vec(p + ".emb.weight")
    C = emb.shape[1]
    pair = p2 is not None

    def g_conv(name):                       # second weight set of a conv (conv1d's name2= / group_cols=)
        return dict(name2=p2 + name, group_cols=n_split) if pair else {}

    def g_ln(name):
        return (W.vec(f"{p2}{name}.gamma"), W.vec(f"{p2}{name}.beta"), n_split) if pair else None

    x = ops.embed(tokens_i32, emb, math.sqrt(C), lay.new(C), group2=(W.vec(p2 + ".emb.weight"), n_split) if pair else None)
    split = LN_SPLIT                                # a LayerNorm here feeds only the next conv: store it as that conv's operand image

    def ln_conv(xin, ln, relu, conv, k, **kw):
        """conv(LayerNorm(xin)) -- the normalised activations exist only as the conv's pre-split operand when `split`."""
        g, b = W.vec(f"{p}{ln}.gamma"), W.vec(f"{p}{ln}.beta")
        if split:
            xs_ = ops.channel_layernorm_split(xin, lay, g, b, relu=relu, group2=g_ln(ln))
            return conv(None, xs=xs_, K=C, **kw)
        return conv(ops.channel_layernorm(xin, lay.N, g, b, lay.new(C), relu=relu, group2=g_ln(ln)), **kw)

    h = conv1d(W, f"{p}.pre.conv_layers.0", x, lay, 5, **g_conv(".pre.conv_layers.0"))                # ConvReluNorm :318-325
    for i in range(3):
        nxt = f".pre.conv_layers.{i + 1}" if i < 2 else ".pre.proj"
        kk, extra = (5, {}) if i < 2 else (1, {"res": x})
        h = ln_conv(h, f".pre.norm_layers.{i}", True,
                    lambda X_, _n=nxt, _k=kk, **kw: conv1d(W, p + _n, X_, lay, _k, **g_conv(_n), **kw), kk, **extra)
    x = h
    e = p + ".encoder"
    for i in range(n_layers):                                             # Encoder.forward :66-90
        a = f"{e}.attn_layers.{i}"
        a2 = f"{p2}.encoder.attn_layers.{i}" if pair else None
        wqkv, bqkv = W.qkv(a, a2)
        qkv = ln_conv(x, f".encoder.norm_layers_1.{i}", False,
                      lambda X_, **kw: ops.conv_gemm(wqkv, X_, lay, lay.new(3 * C), [(0, 0)], bias=bqkv,
                                                     group_cols=n_split if pair else 0, **kw), 1)
        att = ops.relpos_attention(qkv, C, N_HEADS, WINDOW, W.vec(a + ".emb_rel_k"), W.vec(a + ".emb_rel_v"), lay, lay.new(C),
                                   group2=(W.vec(a2 + ".emb_rel_k"), W.vec(a2 + ".emb_rel_v"), b_split) if pair else None)
        x = conv1d(W, a + ".conv_o", att, lay, 1, res=x, **g_conv(f".encoder.attn_layers.{i}.conv_o"))
        f = f".encoder.ffn_layers.{i}"
        y = ln_conv(x, f".encoder.norm_layers_2.{i}", False,
                    lambda X_, _f=f, **kw: conv1d(W, p + _f + ".conv_1", X_, lay, 9, act=ACT_RELU, **g_conv(_f + ".conv_1"), **kw), 9)
        x = conv1d(W, p + f + ".conv_2", y, lay, 1, res=x, **g_conv(f + ".conv_2"))
    return ops.channel_layernorm(x, lay.N, W.vec(e + ".last_ln.gamma"), W.vec(e + ".last_ln.beta"), lay.new(C),
                                 group2=g_ln(".encoder.last_ln"))


def rel_encoder_pair(W, p1, p2, tokens_i32, lay, n_layers):
    """Two encoders of the same shape on the same tokens (text_encoder / arts_encoder, models.py:358-359) as ONE double-width
    encoder: the tokens are laid out twice, [utterances | filler up to a multiple of 128 columns | utterances], and every kernel
    picks its parameter set by column (ConvGemmArgs.n_split, the *_groups_* entry points).  Half the launches, twice the columns
    per launch.  Returns (out1, out2), views [C][N] of the double-width result."""
    dev = tokens_i32.device
    lens = [int(v) for v in lay.widths_host]
    N = lay.N
    pad = (-N) % 128
    fill = [pad] if pad else []
    lay2 = layout(lens + fill + lens, dev)
    tok2 = torch.cat([tokens_i32[:N], tokens_i32.new_zeros(pad), tokens_i32[:N]])
    y = rel_encoder(W, p1, tok2, lay2, n_layers, p2=p2, n_split=N + pad, b_split=len(lens) + len(fill))
    return y[:, :N], y[:, N + pad: 2 * N + pad]


def resblk_down(W, p, X, lay, kind, one_d=False):
    """ResBlk (models.py:79-100) / ResBlk1d(downsample=True) (models.py:127-156).  Returns (Y, layout)."""
    h_too = kind == "half"
    lay2 = lay.halved(h_too)
    cin = X.shape[0]
    taps = taps_1d(3) if one_d else taps_2d(3, 3)
    r = ops.conv_gemm(W.conv(p + ".conv1"), X, lay, lay.new(cin), taps, bias=W.bias(p + ".conv1"), in_act=ACT_LRELU)
    dname = p + (".pool" if one_d else ".downsample_res.conv")
    r2 = ops.dwconv_down(r, lay, lay2.new(cin), lay2, W.dw(dname), W.vec(dname + ".bias"), 3 if h_too else 1, True)
    wt2 = W.conv(p + ".conv2")
    r3 = ops.conv_gemm(wt2, r2, lay2, lay2.new(wt2.shape[2]), taps, bias=W.bias(p + ".conv2"))
    if W.has(p + ".conv1x1.weight"):
        # shortcut = avgpool(conv1x1(x)) (models.py:79-84).  Both are linear and the 1x1 conv has no bias, so it is
        # evaluated as conv1x1(avgpool(x)): a quarter of the columns, and the residual merge (x + r)/sqrt(2) becomes
        # the GEMM's epilogue.  Same value up to fp32 summation order.
        xs = ops.avgpool_down(X, lay, lay2.new(cin), lay2, 2 if h_too else 1)
        out = ops.conv_gemm(W.conv(p + ".conv1x1"), xs, lay2, lay2.new(wt2.shape[2]), [(0, 0)], res=r3, div_sqrt2=True)
    else:
        out = ops.avgpool_down(X, lay, lay2.new(wt2.shape[2]), lay2, 2 if h_too else 1, res=r3)
    return out, lay2


def tower2d(W, p, X, lay, kinds, last_idx, last_stride, linear):
    """Mel_block / EMA_block / dur_block + their Linear (models.py:385-401,412-413,530-538) -> [B][S]."""
    wt = W.conv(p + ".0")
    x = ops.conv_gemm(wt, X, lay, lay.new(wt.shape[2]), taps_2d(3, 3), bias=W.bias(p + ".0"))
    for i, kind in enumerate(kinds):
        x, lay = resblk_down(W, f"{p}.{i + 1}", x, lay, kind)
    K = 5
    lout = lay.valid_conv(K, last_stride)
    if min(lout.widths_host) < 1 or lout.H < 1:
        raise ValueError(f"{p}: reference utterance too short for the {K}x{K} valid conv (SURVEY.md A9: T_ref >= 66)")
    C = x.shape[0]
    col = ops.im2col_valid(x, lay, lout.new(C * K * K), lout, K, last_stride, True)
    wraw = W.raw[f"{p}.{last_idx}.weight"]               # [Cout][C][5][5] -> one tap with K = C*25 (im2col row order)
    wl2 = W.cached("IM2COL:" + p, lambda: ops.prep_weight(wraw.reshape(wraw.shape[0], C * K * K, 1), W.device))
    y = ops.conv_gemm(wl2, col, lout, lout.new(wraw.shape[0]), [(0, 0)], bias=W.bias(f"{p}.{last_idx}"), act=ACT_LRELU)
    pooled = ops.mean_pool(y, lout, False)
    return ops.linear_rows(pooled, W.vec(linear + ".weight"), W.vec(linear + ".bias"))


def tower1d(W, p, X, lay, linear):
    """F0_block / energy_block + Linear (models.py:402-411,414-415) -> [B][S]."""
    wt = W.conv(p + ".0")
    x = ops.conv_gemm(wt, X, lay, lay.new(wt.shape[2]), taps_1d(3), bias=W.bias(p + ".0"))
    for i in (1, 2, 3, 4):
        x, lay = resblk_down(W, f"{p}.{i}", x, lay, "channelpreserve", one_d=True)
    pooled = ops.mean_pool(x, lay, True)
    return ops.linear_rows(pooled, W.vec(linear + ".weight"), W.vec(linear + ".bias"))


# ------------------------------------------------------------------------------------------------
# packing helpers (API boundary only)
# ------------------------------------------------------------------------------------------------
def pack(x, lens):
    """[B, C, Lmax] padded -> packed [C][sum lens]."""
    return torch.cat([x[b, :, : int(l)] for b, l in enumerate(lens)], dim=1).contiguous().float()


def unpack(X, lay, scale_cols=1):
    """packed [C][N] -> [B, C, Lmax] zero padded."""
    C = X.shape[0]
    out = torch.zeros((lay.B, C, lay.max_cols), dtype=X.dtype, device=X.device)
    for b in range(lay.B):
        o, n = lay.off_host[b], lay.off_host[b + 1] - lay.off_host[b]
        out[b, :, :n] = X[:, o:o + n]
    return out


class _Module:
    training = False

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


class RelTransformerEncoder(_Module):
    """Utils/RelTransformerEnc.py:328-380.  forward(x int64 [B,N], x_lengths [B]) -> fp32 [B,N,C]."""

    def __init__(self, W, prefix, n_layers):
        self.W, self.p, self.n_layers = W, prefix, n_layers

    def forward_packed(self, tokens_i32, lay):
        return rel_encoder(self.W, self.p, tokens_i32, lay, self.n_layers)

    def forward(self, x, x_lengths):
        dev = self.W.device
        lens = [int(v) for v in x_lengths]
        lay = layout(lens, dev)
        tok = torch.cat([x[b, :l] for b, l in enumerate(lens)]).to(device=dev, dtype=torch.int32)
        return unpack(self.forward_packed(tok, lay), lay).transpose(1, 2)


class StyleEncoder(_Module):
    """models.py:373-472.  forward(mel [B,80,T], mel_input_length, step, distribution, epoch) ->
    (f0_ext [B,1,T], n_ext [B,1,T], ema_ext [B,10,T], Style [B,512]).  The frozen extractors are pluggable
    torch modules; ``features=(f0_raw, ema_raw)`` bypasses them."""

    def __init__(self, W, prefix="style_encoder"):
        self.W, self.p = W, prefix
        self.pitch_extractor = None
        self.ema_extractor = None

    def tower_inputs(self, feat, lay_full):
        """feat [92][N] (rows 0 n, 1 f0, 2..11 ema, 12..91 mel; FULL reference lengths) -> the T-1 crop
        (models.py:459-471, start = randint(0,1) = 0) and the mel / TV images [1][sum H*L] of the 2-D towers."""
        dev = self.W.device
        lens = [w - 1 for w in lay_full.widths_host]
        l1 = layout(lens, dev)
        c = ops.crop(feat, lay_full, 0, l1.new(92), l1)
        lm, le = layout(lens, dev, H=80), layout(lens, dev, H=10)
        return dict(c=c, l1=l1, lm=lm, le=le, mel_img=ops.rows_to_images(c[12:92], l1, 0, 80, lm),
                    ema_img=ops.rows_to_images(c[2:12], l1, 0, 10, le))

    def tower(self, which, ti):
        """one of the four towers + its Linear (models.py:385-415) -> [B][S]"""
        W, p = self.W, self.p
        if which == "mel":
            return tower2d(W, p + ".Mel_block", ti["mel_img"], ti["lm"], ["half"] * 4, 6, 1, p + ".Mellinear")
        if which == "ema":
            return tower2d(W, p + ".EMA_block", ti["ema_img"], ti["le"], ["channelpreserve"] * 2 + ["half"], 5, 2, p + ".EMAlinear")
        if which == "f0":
            return tower1d(W, p + ".F0_block", ti["c"][1:2], ti["l1"], p + ".F0linear")
        return tower1d(W, p + ".energy_block", ti["c"][0:1], ti["l1"], p + ".Energylinear")

    def style_extractor_packed(self, feat, lay_full):
        """StyleEncoder.style_extractor (models.py:417-424) on packed features -> Style [B][512]."""
        ti = self.tower_inputs(feat, lay_full)
        return torch.cat([self.tower(w, ti) for w in ("mel", "ema", "f0", "energy")], dim=1).contiguous()

    def features_packed(self, mel_p, f0_raw_p, ema_raw_p, lay, stats24):
        return ops.ref_features(mel_p, f0_raw_p, ema_raw_p, lay.N, stats24, lay.new(12))

    def forward_packed(self, mel_p, f0_raw_p, ema_raw_p, lay, stats24):
        feat12 = self.features_packed(mel_p, f0_raw_p, ema_raw_p, lay, stats24)
        feat = torch.cat([feat12, mel_p[:, : feat12.shape[1]]], dim=0).contiguous()
        style = self.style_extractor_packed(feat, lay)
        return feat12, style

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
        dev = self.W.device
        lens = [int(v) for v in mel_input_length]
        lay = layout(lens, dev)
        f0_raw, ema_raw = self._extract(mel, features, lens)
        stats24 = stats_vector(distribution, dev)
        feat12, style = self.forward_packed(pack(mel.to(dev), lens), pack(f0_raw.to(dev), lens),
                                            pack(ema_raw.to(dev), lens), lay, stats24)
        return unpack(feat12[1:2], lay), unpack(feat12[0:1], lay), unpack(feat12[2:12], lay), style


class DurationPredictor(_Module):
    """models.py:519-571.  forward(texts [B,N], style=ema_ext [B,10,T], text_lengths, mel_input_length) -> [B,N]."""

    def __init__(self, W, prefix="durationPredictor"):
        self.W, self.p = W, prefix

    def style_tower(self, ema_p, ref_lay):
        """dur_block + dur_linear on the FULL-length TV track (models.py:543-546) -> [B][64]"""
        W, p = self.W, self.p
        limg = layout(ref_lay.widths_host, W.device, H=10)
        img = ops.rows_to_images(ema_p, ref_lay, 0, ema_p.shape[0], limg)
        return tower2d(W, p + ".dur_block", img, limg, ["channelpreserve"] * 2 + ["half"], 5, 2, p + ".dur_linear")

    def encoder(self, tokens_i32, tok_lay):
        return rel_encoder(self.W, self.p + ".text_encoder", tokens_i32, tok_lay, 2)

    def tail(self, d, ds, tok_lay):
        """3 x AdainResBlk1d -> BiLSTM -> duration_proj (models.py:549-566) -> [1][N_tok]"""
        W, p = self.W, self.p
        gbs = adain_fc_batch(W, [f"{p}.duration.{i}" for i in range(3)], ds)
        for i in range(3):
            d, _ = adain_resblk1d(W, f"{p}.duration.{i}", d, tok_lay, ds, gb=gbs[f"{p}.duration.{i}"])
        x = bilstm(W, p + ".LSTM", d, tok_lay)
        wt = W.cached("DP:" + p, lambda: ops.prep_weight(W.raw[p + ".duration_proj.linear_layer.weight"][:, :, None], W.device))
        return ops.conv_gemm(wt, x, tok_lay, tok_lay.new(1), [(0, 0)], bias=W.vec(p + ".duration_proj.linear_layer.bias"))

    def forward_packed(self, tokens_i32, tok_lay, ema_p, ref_lay):
        return self.tail(self.encoder(tokens_i32, tok_lay), self.style_tower(ema_p, ref_lay), tok_lay)

    def forward(self, texts, style, text_lengths, mel_input_length):
        dev = self.W.device
        tl = [int(v) for v in text_lengths]
        ml = [int(v) for v in mel_input_length]
        tok_lay, ref_lay = layout(tl, dev), layout(ml, dev)
        tok = torch.cat([texts[b, :l] for b, l in enumerate(tl)]).to(device=dev, dtype=torch.int32)
        d = self.forward_packed(tok, tok_lay, pack(style.to(dev), ml), ref_lay)
        return unpack(d, tok_lay)[:, 0, :]


class ArtsPredictor(_Module):
    """models.py:573-621.  forward(A_ens [B,512,M], style [B,512]) -> F0 [B,1,2M], N [B,1,2M], EMA [B,10,2M]."""

    def __init__(self, W, prefix="artsPredictor"):
        self.W, self.p = W, prefix

    def forward_packed(self, a_en, lay, style):
        W, p = self.W, self.p
        sl = {"EMA": style[:, 256:384].contiguous(), "F0": style[:, 384:448].contiguous(),
              "N": style[:, 448:512].contiguous()}
        gbs = adain_fc_batch(W, [p + ".shared"] + [f"{p}.{br}.0" for br in ("F0", "N", "EMA")], style)
        for br in ("F0", "N", "EMA"):
            gbs.update(adain_fc_batch(W, [f"{p}.{br}.1", f"{p}.{br}.2"], sl[br]))
        a, _ = adain_resblk1d(W, p + ".shared", a_en, lay, style, gb=gbs[p + ".shared"])
        outs, feats = {}, []
        lay2 = lay.scaled(2)
        with Fork(side_streams(W.device, 3), uses=(a,) + tuple(t for v in gbs.values() for t in v)) as side:   # F0 / N / EMA branches
            for i, br in enumerate(("F0", "N", "EMA")):
                with side(i):
                    x, _ = adain_resblk1d(W, f"{p}.{br}.0", a, lay, style, upsample=True, gb=gbs[f"{p}.{br}.0"])
                    x, _ = adain_resblk1d(W, f"{p}.{br}.1", x, lay2, sl[br], gb=gbs[f"{p}.{br}.1"])
                    x, _ = adain_resblk1d(W, f"{p}.{br}.2", x, lay2, sl[br], gb=gbs[f"{p}.{br}.2"])
                    feats.append((f"{p}.{br}_LSTM", x))
            side.produced(*[x for _, x in feats])
        hs = bilstm_many(W, feats, lay2)                  # the three recurrences share one launch
        for br, h in zip(("F0", "N", "EMA"), hs):
            outs[br] = conv1d(W, f"{p}.{br}_proj", h, lay2, 1)
        return outs["F0"], outs["N"], outs["EMA"], lay2

    def forward(self, A_ens, style, lengths=None):
        dev = self.W.device
        lens = [A_ens.shape[-1]] * A_ens.shape[0] if lengths is None else [int(v) for v in lengths]
        lay = layout(lens, dev)
        f0, n, ema, lay2 = self.forward_packed(pack(A_ens.to(dev), lens), lay, style.to(dev).contiguous().float())
        return unpack(f0, lay2), unpack(n, lay2), unpack(ema, lay2)


class Decoder(_Module):
    """models.py:474-517.  forward(asr [B,512,M], Style [B,512], F0 [B,1,2M], N [B,1,2M], EMA [B,10,2M]) -> [B,80,2M]."""

    def __init__(self, W, prefix="decoder"):
        self.W, self.p = W, prefix

    def adain_params(self, style):
        """gamma/beta of every AdaIN of the decoder (two batched launches over ~40 MB of fc weights): depends on the style
        vector only, so the caller can run it beside the predictors instead of at the head of the decoder."""
        W, p = self.W, self.p
        mel_style = style[:, :256].contiguous()
        gbs = adain_fc_batch(W, [p + ".encode"] + [f"{p}.decode.{i}" for i in (0, 1, 2)], style)
        gbs.update(adain_fc_batch(W, [f"{p}.decode.{i}" for i in (3, 4, 5)], mel_style))
        return gbs

    def forward_packed(self, asr_up, lay2, style, f0, n, ema, out=None, gbs=None):
        """asr_up: [C][2M-frames] (already nearest-x2, models.py:500)."""
        W, p = self.W, self.p
        C = asr_up.shape[0]
        mel_style = style[:, :256].contiguous()
        x0 = lay2.new(C + 128)
        x0[:C].copy_(asr_up)
        conv1d(W, p + ".F0_conv", f0, lay2, 1, Y=x0[C:C + 32])
        conv1d(W, p + ".N_conv", n, lay2, 1, Y=x0[C + 32:C + 64])
        conv1d(W, p + ".EMA_conv", ema, lay2, 1, Y=x0[C + 64:C + 128])
        bott = 2 * C
        cat_a, cat_b = lay2.new(bott + 64 + 128), lay2.new(bott + 64 + 128)
        if gbs is None:
            gbs = self.adain_params(style)
        adain_resblk1d(W, p + ".encode", x0, lay2, style, out=cat_a[:bott], gb=gbs[p + ".encode"], fork_shortcut=True)
        conv1d(W, p + ".asr_res.0", asr_up, lay2, 1, Y=cat_a[bott:bott + 64])
        cat_a[bott + 64:].copy_(x0[C:])
        cat_b[bott:].copy_(cat_a[bott:])
        adain_resblk1d(W, p + ".decode.0", cat_a, lay2, style, out=cat_b[:bott], gb=gbs[p + ".decode.0"], fork_shortcut=True)
        adain_resblk1d(W, p + ".decode.1", cat_b, lay2, style, out=cat_a[:bott], gb=gbs[p + ".decode.1"], fork_shortcut=True)
        x, _ = adain_resblk1d(W, p + ".decode.2", cat_a, lay2, style, gb=gbs[p + ".decode.2"], fork_shortcut=True)
        for i in (3, 4, 5):
            x, _ = adain_resblk1d(W, f"{p}.decode.{i}", x, lay2, mel_style, gb=gbs[f"{p}.decode.{i}"])
        return conv1d(W, p + ".to_out.0", x, lay2, 1, Y=out)

    def forward(self, asr, Style, F0, N, EMA, lengths=None):
        dev = self.W.device
        lens = [asr.shape[-1]] * asr.shape[0] if lengths is None else [int(v) for v in lengths]
        lay2 = layout([2 * l for l in lens], dev)
        asr_up = pack(asr.to(dev).repeat_interleave(2, dim=-1), lay2.widths_host)
        mel = self.forward_packed(asr_up, lay2, Style.to(dev).contiguous().float(), pack(F0.to(dev), lay2.widths_host),
                                  pack(N.to(dev), lay2.widths_host), pack(EMA.to(dev), lay2.widths_host))
        return unpack(mel, lay2)


_STREAMS = {}


def side_streams(device, n, tag="", priorities=None):
    """n pooled side streams; priorities: per-stream HIP priority (-1 = high) -- part of the pool key."""
    key = str(device) + "/" + tag + ("/" + ",".join(str(p) for p in priorities) if priorities else "")
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pr = priorities[len(pool)] if priorities else 0
        pool.append(torch.cuda.Stream(device=device, priority=pr))
    return pool[:n]


CONCURRENT = True        # False: every branch runs on the calling stream (bench.py's per-kernel timing pass)


class Fork:
    """Fork/join of independent branches over side HIP streams.  Every side stream first waits for the calling
    stream; on exit the calling stream waits for all of them.  Tensors that cross streams are registered with the
    caching allocator (`record_stream`) so their memory is not recycled while another stream still uses it."""

    def __init__(self, streams, uses=()):
        self.main = torch.cuda.current_stream()
        self.streams, self.uses = (streams if CONCURRENT else [self.main] * len(streams)), uses

    def __enter__(self):
        if not CONCURRENT:
            return self
        for s in self.streams:
            s.wait_stream(self.main)
            for t in self.uses:
                t.record_stream(s)
        return self

    def __call__(self, i):
        return torch.cuda.stream(self.streams[i])

    def produced(self, *tensors):
        if CONCURRENT:
            for t in tensors:
                t.record_stream(self.main)

    def __exit__(self, *exc):
        if CONCURRENT:
            for s in self.streams:
                self.main.wait_stream(s)
        return False


def stats_vector(distribution, device):
    """distribution dict (utils.py:86-92 / test.py:75-79) -> the 24 floats as_ref_features_f32 takes."""
    d = distribution if distribution else load_distribution(DEFAULT_STATS)
    vals = [d["energy_mean"].reshape(1), d["energy_std"].reshape(1), d["pitch_mean"].reshape(1), d["pitch_std"].reshape(1),
            d["EMA_mean"].reshape(10), d["EMA_std"].reshape(10)]
    return torch.cat([v.detach().float().cpu() for v in vals]).to(device)


class ArtsSpeech(_Module):
    """models.py:275-371, inference branch only: forward(batch, s2s_attn, s2s_attn_mono, step="test")."""

    def __init__(self, args, stage="second", distribution=None, device=None, state_dict=None):
        if stage == "first":
            raise NotImplementedError("only the inference path (stage='second' modules, step='test') is built")
        self.args = args
        self.device = _need_gpu(device)
        self.distribution = distribution if distribution else load_distribution(DEFAULT_STATS)
        self.W = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, sd, strict=False):
        self.W = W = Weights(fold_state_dict(sd), self.device)
        self.text_encoder = RelTransformerEncoder(W, "text_encoder", 4)
        self.arts_encoder = RelTransformerEncoder(W, "arts_encoder", 4)
        ext = getattr(self, "style_encoder", None)
        self.style_encoder = StyleEncoder(W)
        if ext is not None:
            self.style_encoder.pitch_extractor, self.style_encoder.ema_extractor = ext.pitch_extractor, ext.ema_extractor
        self.durationPredictor = DurationPredictor(W)
        self.artsPredictor = ArtsPredictor(W)
        self.decoder = Decoder(W)
        return self

    def forward(self, batch, s2s_attn=None, s2s_attn_mono=None, step="test", mode="train", epoch=0, features=None,
                forced_durations=None, return_aux=False):
        if step != "test":
            raise NotImplementedError("training branches (step='first'/'second') are out of scope (SURVEY.md section 2)")
        if self.W is None:
            raise RuntimeError("no weights loaded: call load_checkpoint / load_state_dict first")
        texts, input_lengths, mels, mel_input_length = batch[0], batch[1], batch[2], batch[3]   # 7- or 4-tuple (B1)
        dev = self.device
        tl = [int(v) for v in input_lengths]
        ml = [int(v) for v in mel_input_length]
        B = len(tl)
        tok_lay, ref_lay = layout(tl, dev), layout(ml, dev)
        tok = torch.cat([texts[b, :l] for b, l in enumerate(tl)]).to(device=dev, dtype=torch.int32)
        f0_raw, ema_raw = self.style_encoder._extract(mels, features, ml)
        mel_p, f0_p, ema_p = pack(mels.to(dev), ml), pack(f0_raw.to(dev), ml), pack(ema_raw.to(dev), ml)
        forced = None
        if forced_durations is not None:
            forced = torch.cat([torch.as_tensor(forced_durations[b])[: tl[b]].reshape(-1) for b in range(B)]).to(
                device=dev, dtype=torch.int32)
        out = self.forward_packed(tok, tok_lay, mel_p, f0_p, ema_p, ref_lay, forced=forced,
                                  frames_hint=None if forced is None else [int(torch.as_tensor(forced_durations[b])[: tl[b]].sum()) for b in range(B)])
        mel = unpack(out["mel"], out["lay2"])
        if return_aux:
            return mel, out
        return mel

    def forward_packed(self, tok, tok_lay, mel_p, f0_p, ema_p, ref_lay, forced=None, frames_hint=None):
        """The whole hot path on packed tensors (what bench.py times).  One host sync (reading the integer
        frame counts) unless `frames_hint` gives them (forced durations)."""
        dev = self.device
        if getattr(self, "_stats24", None) is None:
            self._stats24 = stats_vector(self.distribution, dev)
        stats24 = self._stats24
        feat12 = self.style_encoder.features_packed(mel_p, f0_p, ema_p, ref_lay, stats24)
        # The articulatory encoder, the style towers and the duration predictor are mutually independent
        # (models.py:358-360) and individually too small to fill 256 CUs: four concurrent branches on side HIP streams
        # (a fork/join that hipGraph capture records as parallel nodes) -- the mel tower, the longest chain, on its own.
        # Measured on MI355X (bench.py, C3): 3 branches (all towers in one) 9.35 ms, 4 branches 8.89 ms, 5 branches
        # (TV tower on its own too) 9.76 ms.  Stream-to-stream edges between side streams crash hipGraph instantiation
        # on ROCm 7.2, so every branch forks from / joins the caller.
        # Re-measured at the end of round 1 (scripts/phase_bench.py, each case its own hipGraph): the branches alone take
        # 1.02 + 1.81 + 0.98 + 1.67 = 5.5 ms, together 4.1 ms (one graph queue: 5.6; two: 4.35; four or eight: 4.1): most of
        # their kernels fill the 256 CUs on their own, so overlap only hides the small ones.  More queues with more branches
        # (GPU_MAX_HW_QUEUES=8, TV / F0 / energy towers on three streams: 12 ms per step), high-priority streams for the
        # small-kernel branches (10.4 ms) and the mel tower before / after / beside the others (+-0.1 ms) do not help.
        se = self.style_encoder
        feat = torch.cat([feat12, mel_p[:, : feat12.shape[1]]], dim=0).contiguous()
        ti = se.tower_inputs(feat, ref_lay)
        nb = TOWER_BRANCHES
        with Fork(side_streams(dev, 3 + nb), uses=(feat12, feat, ti["c"], ti["mel_img"], ti["ema_img"])) as side:
            with side(0):
                if ENC_PAIR:
                    a_en, t_en = rel_encoder_pair(self.W, "arts_encoder", "text_encoder", tok, tok_lay, 4)
                else:
                    a_en, t_en = self.arts_encoder.forward_packed(tok, tok_lay), None
            with side(1):
                s_mel = se.tower("mel", ti)
            with side(2):
                duration = self.durationPredictor.forward_packed(tok, tok_lay, feat12[2:12], ref_lay)
            s_rest = [None, None, None]
            for i, w in enumerate(("ema", "f0", "energy")):
                with side(3 + i % nb):
                    s_rest[i] = se.tower(w, ti)
            side.produced(a_en, s_mel, duration, *s_rest)
            if t_en is not None:
                side.produced(t_en)
        style = torch.cat([s_mel] + s_rest, dim=1).contiguous()
        if frames_hint is None:
            dur_i, frame_off, _ = ops.durations(duration.reshape(-1), forced, tok_lay, 0)
            off = frame_off.cpu().tolist()                                   # the one device->host sync
            frames = [off[b + 1] - off[b] for b in range(tok_lay.B)]
        else:
            frames = list(frames_hint)
        lay1 = layout(frames, dev)
        dur_i, frame_off, tof = ops.durations(duration.reshape(-1), forced, tok_lay, lay1.N)
        # The text encoder feeds only the decoder: it is deferred to run beside the articulatory predictors, whose
        # three branches and sequential LSTM recurrences leave most of the chip idle (critical path: scripts/phase_bench.py).
        with Fork(side_streams(dev, 1, "text_encoder"), uses=(style,)) as side:
            with side(0):
                if t_en is None:
                    t_en = self.text_encoder.forward_packed(tok, tok_lay)
                dec_gbs = self.decoder.adain_params(style)                   # style-only work of the decoder, off its critical path
            C = a_en.shape[0]
            a_ex = ops.expand(a_en, tof, lay1.N, 1, lay1.new(C))
            f0, n, ema, lay2 = self.artsPredictor.forward_packed(a_ex, lay1, style)
            side.produced(t_en, *[t for v in dec_gbs.values() for t in v])
        t_up = ops.expand(t_en, tof, lay1.N, 2, lay2.new(C))
        mel = self.decoder.forward_packed(t_up, lay2, style, f0, n, ema, gbs=dec_gbs)
        return dict(mel=mel, lay2=lay2, lay1=lay1, t_en=t_en, a_en=a_en, feat12=feat12, style=style,
                    duration=duration, dur_i=dur_i, F0=f0, N=n, EMA=ema)


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
