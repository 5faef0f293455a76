"""EMA_Predictor (articulatory-trajectory extractor) on the HIP path -- SURVEY.md section 8(f) row N1 (second half), on
the ``test.py`` surface right after JDCNet (models.py:433: ``ema_ext = self.ema_extractor(f0_ext, n_ext, mel)``).
Mirrors ``Utils/EMA/EMA_Predictor.py:18-82`` and the conformer blocks it instantiates (``conformer/encoder.py:33-110``,
d_model 256, 4 heads, FFN x4, conv kernel 31, half-step residuals): ``EMA_Predictor()``, ``load_state_dict``,
``forward(F0 [B,1,T], energy [B,1,T], mels [B,80,T]) -> [B,10,T]``.

  encoder1       Linear 82->256 (+BatchNorm folded in) + ReLU                                       conv GEMM
  3 x block      LayerNorm -> Linear 256->1024 (Swish epilogue) -> Linear 1024->256 (x0.5 folded in, residual)    x2
                 LayerNorm -> fused q/k/v GEMM, pos_proj(PE) GEMM -> as_xl_attention_f32 -> out_proj (residual)
                 LayerNorm -> pointwise 256->512 -> GLU + depthwise k31 + BatchNorm + Swish (one kernel) -> pointwise (residual)
                 LayerNorm
  decoder2       nn.LSTM without batch_first fed [1, T, 256]: ONE step per frame from the zero state = GEMM + gate kernel
  decoder3       Linear 512->128 (+BatchNorm) + ReLU, Linear 128->10
Per-utterance (B = 1) semantics for every item, as test.py runs it: the reference's LSTM would run ACROSS the items of a
larger batch (SURVEY.md N1 "batch-axis LSTM quirk"); here a batch is a stack of B = 1 results.  ``self.pool`` of the
reference is built but never called by forward.
"""
import math
import os

import numpy as np
import torch

from . import ops
from .ops import ACT_RELU, ACT_SWISH
from .synth import hash_tensor

D_MODEL, HEADS, FF, KSIZE, N_BLOCKS = 256, 4, 4, 31, 3


def ema_spec(max_len=10000):
    """name -> shape of the reference EMA_Predictor's state dict."""
    spec = {}

    def lin(name, o, i, bias=True):
        spec[name + ".weight"] = (o, i)
        if bias:
            spec[name + ".bias"] = (o,)

    def norm(name, c, bn=False):
        spec[name + ".weight"], spec[name + ".bias"] = (c,), (c,)
        if bn:
            spec[name + ".running_mean"], spec[name + ".running_var"], spec[name + ".num_batches_tracked"] = (c,), (c,), ()

    lin("encoder1.0", D_MODEL, 82)
    norm("encoder1.2", D_MODEL, bn=True)
    for i in range(N_BLOCKS):
        s = f"decoder.{i}.sequential"
        for j in (0, 3):
            norm(f"{s}.{j}.module.sequential.0", D_MODEL)
            lin(f"{s}.{j}.module.sequential.1.linear", D_MODEL * FF, D_MODEL)
            lin(f"{s}.{j}.module.sequential.4.linear", D_MODEL, D_MODEL * FF)
        a = f"{s}.1.module"
        spec[a + ".positional_encoding.pe"] = (1, max_len, D_MODEL)
        norm(a + ".layer_norm", D_MODEL)
        spec[a + ".attention.u_bias"] = spec[a + ".attention.v_bias"] = (HEADS, D_MODEL // HEADS)
        for n in ("query", "key", "value"):
            lin(f"{a}.attention.{n}_proj.linear", D_MODEL, D_MODEL)
        lin(a + ".attention.pos_proj.linear", D_MODEL, D_MODEL, bias=False)
        lin(a + ".attention.out_proj.linear", D_MODEL, D_MODEL)
        c = f"{s}.2.module.sequential"
        norm(c + ".0", D_MODEL)
        spec[c + ".2.conv.weight"], spec[c + ".2.conv.bias"] = (2 * D_MODEL, D_MODEL, 1), (2 * D_MODEL,)
        spec[c + ".4.conv.weight"] = (D_MODEL, 1, KSIZE)
        norm(c + ".5", D_MODEL, bn=True)
        spec[c + ".7.conv.weight"], spec[c + ".7.conv.bias"] = (D_MODEL, D_MODEL, 1), (D_MODEL,)
        norm(f"{s}.4", D_MODEL)
    spec["pool.bias"], spec["pool.weight_g"], spec["pool.weight_v"] = (D_MODEL,), (D_MODEL, 1, 1), (D_MODEL, 1, 3)
    for sfx in ("", "_reverse"):
        spec[f"decoder2.weight_ih_l0{sfx}"] = spec[f"decoder2.weight_hh_l0{sfx}"] = (4 * D_MODEL, D_MODEL)
        spec[f"decoder2.bias_ih_l0{sfx}"] = spec[f"decoder2.bias_hh_l0{sfx}"] = (4 * D_MODEL,)
    lin("decoder3.0", 128, 2 * D_MODEL)
    norm("decoder3.2", 128, bn=True)
    lin("decoder3.5", 10, 128)
    return spec


def positional_encoding(length, d_model=D_MODEL):
    """conformer/embedding.py:28-38, the same fp32 torch expressions (CPU) so that the table is bit-identical."""
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def synth_ema_state_dict(seed=3407, max_len=10000):
    """Seeded synthetic EMA_Predictor checkpoint (``Utils/EMA/200000.pth.tar`` is a 134-byte pointer in the tree): xavier-like
    uniform dense layers, near-identity norms with non-trivial BatchNorm statistics, small attention biases."""
    sd = {}
    for name, shape in ema_spec(max_len).items():
        tag = "ema." + name
        if name.endswith("num_batches_tracked"):
            sd[name] = np.asarray(1000, dtype=np.int64)
        elif name.endswith("positional_encoding.pe"):
            sd[name] = positional_encoding(max_len)[None].numpy()
        elif name.endswith("running_var"):
            sd[name] = (1.0 + 0.5 * hash_tensor(tag, shape, seed, 1.0)).astype(np.float32)
        elif name.endswith("running_mean"):
            sd[name] = hash_tensor(tag, shape, seed, 0.2)
        elif name.endswith("weight_g"):
            sd[name] = (1.0 + 0.2 * hash_tensor(tag, shape, seed, 1.0)).astype(np.float32)
        elif name.endswith(("u_bias", "v_bias")):
            sd[name] = hash_tensor(tag, shape, seed, 0.5)
        elif len(shape) == 1 and name.endswith(".weight"):                              # LayerNorm / BatchNorm gamma
            sd[name] = (1.0 + 0.2 * hash_tensor(tag, shape, seed, 1.0)).astype(np.float32)
        elif len(shape) == 1:
            sd[name] = hash_tensor(tag, shape, seed, 0.05)
        elif len(shape) == 3 and shape[1] == 1:                                         # depthwise kernels
            sd[name] = hash_tensor(tag, shape, seed, float(np.sqrt(3.0 / shape[2])))
        else:
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            sd[name] = hash_tensor(tag, shape, seed, float(np.sqrt(6.0 / (fan_in + fan_out))))
    return sd


def _bn_affine(w, p, eps=1e-5):
    scale = w[p + ".weight"] / torch.sqrt(w[p + ".running_var"] + eps)
    return scale, w[p + ".bias"] - w[p + ".running_mean"] * scale


class EMA_Predictor:
    def __init__(self, device=None):
        from .models import _need_gpu
        self.device = _need_gpu(device if device is not None else "cuda")
        self.W = None
        self._pos_cache = {}

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=False):
        sd = sd.get("model", sd) if isinstance(sd, dict) and "model" in sd else sd      # models.py:382: params['model']
        w = {k[len("module."):] if k.startswith("module.") else k: (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v)))
             for k, v in sd.items()}
        w = {k: v.detach().float().cpu() for k, v in w.items() if v.dtype.is_floating_point and not k.endswith(".pe")}
        dev, W = self.device, {}
        gemm = lambda m: ops.prep_weight(m.reshape(m.shape[0], m.shape[1], 1).contiguous(), dev)
        vec = lambda name: w[name].to(dev).contiguous()

        s, t = _bn_affine(w, "encoder1.2")
        W["enc1"] = (gemm(w["encoder1.0.weight"] * s[:, None]), (w["encoder1.0.bias"] * s + t).to(dev))
        blocks = []
        for i in range(N_BLOCKS):
            q = f"decoder.{i}.sequential"
            blk = {}
            for j, key in ((0, "ff1"), (3, "ff2")):
                f = f"{q}.{j}.module.sequential"
                blk[key] = dict(ln=(vec(f + ".0.weight"), vec(f + ".0.bias")),
                                w1=gemm(w[f + ".1.linear.weight"]), b1=vec(f + ".1.linear.bias"),
                                w2=gemm(w[f + ".4.linear.weight"] * 0.5), b2=(w[f + ".4.linear.bias"] * 0.5).to(dev))   # half-step residual
            a = f"{q}.1.module"
            at = a + ".attention"
            blk["att"] = dict(ln=(vec(a + ".layer_norm.weight"), vec(a + ".layer_norm.bias")),
                              wqkv=gemm(torch.cat([w[f"{at}.{n}_proj.linear.weight"] for n in ("query", "key", "value")], 0)),
                              bqkv=torch.cat([w[f"{at}.{n}_proj.linear.bias"] for n in ("query", "key", "value")], 0).to(dev),
                              wpos=gemm(w[at + ".pos_proj.linear.weight"]), u=vec(at + ".u_bias"), v=vec(at + ".v_bias"),
                              # matrix-core attention (ops.xl_attention_image): the query rows twice, u_bias / v_bias folded into the bias
                              wqkv4=gemm(torch.cat([w[f"{at}.{n}_proj.linear.weight"] for n in ("query", "query", "key", "value")], 0)),
                              bqkv4=torch.cat([w[f"{at}.query_proj.linear.bias"] + w[at + ".u_bias"].reshape(-1),
                                               w[f"{at}.query_proj.linear.bias"] + w[at + ".v_bias"].reshape(-1),
                                               w[f"{at}.key_proj.linear.bias"], w[f"{at}.value_proj.linear.bias"]], 0).to(dev),
                              wo=gemm(w[at + ".out_proj.linear.weight"]), bo=vec(at + ".out_proj.linear.bias"))
            c = f"{q}.2.module.sequential"
            s, t = _bn_affine(w, c + ".5")
            blk["conv"] = dict(ln=(vec(c + ".0.weight"), vec(c + ".0.bias")), w1=gemm(w[c + ".2.conv.weight"][:, :, 0]),
                               b1=vec(c + ".2.conv.bias"), dw=w[c + ".4.conv.weight"][:, 0].contiguous().to(dev), bn=(s.to(dev), t.to(dev)),
                               w2=gemm(w[c + ".7.conv.weight"][:, :, 0]), b2=vec(c + ".7.conv.bias"))
            blk["ln"] = (vec(f"{q}.4.weight"), vec(f"{q}.4.bias"))
            blocks.append(blk)
        W["blocks"] = blocks
        W["lstm"] = (gemm(torch.cat([w["decoder2.weight_ih_l0"], w["decoder2.weight_ih_l0_reverse"]], 0)),
                     torch.cat([w["decoder2.bias_ih_l0"] + w["decoder2.bias_hh_l0"],
                                w["decoder2.bias_ih_l0_reverse"] + w["decoder2.bias_hh_l0_reverse"]], 0).to(dev))
        s, t = _bn_affine(w, "decoder3.2")
        W["d3a"] = (gemm(w["decoder3.0.weight"] * s[:, None]), (w["decoder3.0.bias"] * s + t).to(dev))
        W["d3b"] = (gemm(w["decoder3.5.weight"]), vec("decoder3.5.bias"))
        self.W = W
        self._pos_cache.clear()
        return self

    def eval(self):
        return self

    def to(self, device):
        return self

    def modules(self):
        return iter(())

    # ------------------------------------------------------------------ forward
    def _pos_inputs(self, lay):
        """PE[frame index] for every packed column, [256][N]: the positional_encoding(seq_length) of attention.py:142-143."""
        key = tuple(int(v) for v in lay.widths_host)
        p = self._pos_cache.get(key)
        if p is None:
            if len(self._pos_cache) > 64:
                self._pos_cache.clear()
            pe = positional_encoding(max(key))
            idx = torch.cat([torch.arange(n) for n in key])
            p = lay.new(D_MODEL)
            p[:, : lay.N].copy_(pe[idx].t())
            ops._settle(self.device)
            self._pos_cache[key] = p
        return p

    def forward_packed(self, f0_p, n_p, mel_p, lay):
        """f0_p [1][N], n_p [1][N], mel_p [80][N] packed frames -> EMA [10][N]."""
        W, N = self.W, lay.N
        if W is None:
            raise RuntimeError("no weights loaded: call load_state_dict first")
        one = [(0, 0)]
        mm = lambda wt, x, **kw: ops.conv_gemm(wt, x, lay, lay.new(wt.shape[2]), one, **kw)
        ln = lambda g, x: ops.channel_layernorm(x, N, g[0], g[1], lay.new(x.shape[0]), eps=1e-5)
        x82 = lay.new(82)                                                             # EMA_Predictor.py:74: cat((F0, energy, mels), 1)
        x82[0:1, :N].copy_(f0_p[:, :N])
        x82[1:2, :N].copy_(n_p[:, :N])
        x82[2:, :N].copy_(mel_p[:, :N])
        x = mm(W["enc1"][0], x82, bias=W["enc1"][1], act=ACT_RELU)                    # :75
        pos_in = self._pos_inputs(lay)
        inv_scale = 1.0 / math.sqrt(D_MODEL)                                          # attention.py:57: sqrt(d_model), not d_head
        if os.environ.get("AS_XL_ATTENTION", "image") != "image":
            return self._forward_exact(x, pos_in, inv_scale, lay)
        # Activations travel between the GEMMs as operand images wherever their only reader is a GEMM: LayerNorm writes the image of the
        # GEMM that follows it, the FFN's first GEMM (Swish) that of the second, the attention that of its out-projection -- 5 split
        # passes per batch instead of 31.  pos = pos_proj(PE[frame index]) does not depend on the input: its image is made once per
        # batch geometry and block.
        img = lambda C: ops.new_image(C, N, x.device)
        lns = lambda g, v: ops.channel_layernorm_split(v, lay, g[0], g[1], eps=1e-5)
        gi = lambda wt, xs, K, Y=None, **kw: ops.conv_gemm(wt, None, lay, Y, one, xs=xs, K=K, **kw)

        def ffn(f, v):
            hh = gi(f["w1"], lns(f["ln"], v), D_MODEL, bias=f["b1"], act=ACT_SWISH, yh=img(4 * D_MODEL))
            return gi(f["w2"], hh, 4 * D_MODEL, lay.new(D_MODEL), bias=f["b2"], res=v)

        nb = len(W["blocks"])
        for bi, blk in enumerate(W["blocks"]):                                        # :77-78, encoder.py:74-110
            x = ffn(blk["ff1"], x)
            a = blk["att"]
            qh = img(4 * D_MODEL)
            qkv = gi(a["wqkv4"], lns(a["ln"], x), D_MODEL, lay.new(4 * D_MODEL), bias=a["bqkv4"], yh=qh)
            ctx = ops.xl_attention_image(qkv, qh, self._pos_image(lay, bi, a["wpos"], pos_in), D_MODEL, HEADS, inv_scale, lay, image=True)
            x = gi(a["wo"], ctx, D_MODEL, lay.new(D_MODEL), bias=a["bo"], res=x)
            c = blk["conv"]
            g = ops.glu_dwconv_bn_swish(gi(c["w1"], lns(c["ln"], x), D_MODEL, lay.new(2 * D_MODEL), bias=c["b1"]), D_MODEL, c["dw"],
                                        c["bn"][0], c["bn"][1], lay, lay.new(D_MODEL))
            x = mm(c["w2"], g, bias=c["b2"], res=x)
            x = ffn(blk["ff2"], x)
            if bi + 1 < nb:
                x = ln(blk["ln"], x)
            else:                                                                     # the last LayerNorm feeds only decoder2's GEMM
                gx = gi(W["lstm"][0], lns(blk["ln"], x), D_MODEL, lay.new(W["lstm"][0].shape[2]), bias=W["lstm"][1])
        h = ops.lstm_step0(gx, D_MODEL, N, lay.new(2 * D_MODEL))                      # :79
        d = ops.conv_gemm(W["d3a"][0], h, lay, None, one, bias=W["d3a"][1], act=ACT_RELU, yh=img(W["d3a"][0].shape[2]))   # :46-51
        return gi(W["d3b"][0], d, W["d3a"][0].shape[2], lay.new(W["d3b"][0].shape[2]), bias=W["d3b"][1])                  # :53, :80

    def _pos_image(self, lay, bi, wpos, pos_in):
        """the operand image of pos_proj(PE[frame index]) of block bi for this batch geometry (input-independent: made once)"""
        key = (tuple(int(v) for v in lay.widths_host), bi)
        ph = self._pos_cache.get(key)
        if ph is None:
            ph = ops.new_image(D_MODEL, lay.N, pos_in.device)
            ops.conv_gemm(wpos, pos_in, lay, None, [(0, 0)], yh=ph)
            ops._settle(self.device)
            self._pos_cache[key] = ph
        return ph

    def _forward_exact(self, x, pos_in, inv_scale, lay):
        """the blocks with the exact fp32 attention kernel (AS_XL_ATTENTION=exact): fp32 activations between all launches"""
        W, N = self.W, lay.N
        one = [(0, 0)]
        mm = lambda wt, v, **kw: ops.conv_gemm(wt, v, lay, lay.new(wt.shape[2]), one, **kw)
        ln = lambda g, v: ops.channel_layernorm(v, N, g[0], g[1], lay.new(v.shape[0]), eps=1e-5)
        for blk in W["blocks"]:
            f = blk["ff1"]
            x = mm(f["w2"], mm(f["w1"], ln(f["ln"], x), bias=f["b1"], act=ACT_SWISH), bias=f["b2"], res=x)
            a = blk["att"]
            qkv = mm(a["wqkv"], ln(a["ln"], x), bias=a["bqkv"])
            pos = mm(a["wpos"], pos_in)
            ctx = ops.xl_attention(qkv, D_MODEL, HEADS, pos, a["u"], a["v"], inv_scale, lay, lay.new(D_MODEL))
            x = mm(a["wo"], ctx, bias=a["bo"], res=x)
            c = blk["conv"]
            g = ops.glu_dwconv_bn_swish(mm(c["w1"], ln(c["ln"], x), bias=c["b1"]), D_MODEL, c["dw"], c["bn"][0], c["bn"][1], lay,
                                        lay.new(D_MODEL))
            x = mm(c["w2"], g, bias=c["b2"], res=x)
            f = blk["ff2"]
            x = mm(f["w2"], mm(f["w1"], ln(f["ln"], x), bias=f["b1"], act=ACT_SWISH), bias=f["b2"], res=x)
            x = ln(blk["ln"], x)
        h = ops.lstm_step0(mm(W["lstm"][0], x, bias=W["lstm"][1]), D_MODEL, N, lay.new(2 * D_MODEL))   # :79
        d = mm(W["d3a"][0], h, bias=W["d3a"][1], act=ACT_RELU)                        # :46-51
        return mm(W["d3b"][0], d, bias=W["d3b"][1])                                   # :53, :80

    @torch.no_grad()
    def forward(self, F0, energy, mels=None, lengths=None):
        """F0 [B,1,T], energy [B,1,T], mels [B,80,T] (models.py:433) -> EMA [B,10,T]."""
        from .models import pack, unpack
        B, _, T = mels.shape
        lens = [int(v) for v in lengths] if lengths is not None else [T] * B
        lay = ops.layout(lens, self.device)
        dev = self.device
        out = self.forward_packed(pack(F0.to(dev).float(), lens), pack(energy.to(dev).float(), lens), pack(mels.to(dev).float(), lens), lay)
        return unpack(out, lay)

    __call__ = forward
