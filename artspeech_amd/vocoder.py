"""HiFi-GAN generator on the HIP path -- SURVEY.md section 8(f) row N2, the step right after the acoustic model
(test.py:115: ``self.generator(mel_rec)``).  Mirrors ``Vocoder/vocoder.py:75-125``: ``Generator(h)``,
``load_state_dict`` (weight-norm folded), ``forward(mel [B, 80, T]) -> wav [B, 1, 300 T]``.

Everything dense runs through the conv GEMM of the acoustic path (``ops.conv_gemm``, bf16x6 arithmetic):
  conv_pre / conv_post   k = 7 convs (LeakyReLU(0.01) and tanh fused around conv_post)
  ResBlock1              18 dilated convs per stage, LeakyReLU(0.1) fused on the input operand, residual in the epilogue
  ConvTranspose1d(2u, u) ONE 3-tap conv with u x Cout output rows (phase, channel) + ``interleave_phases``:
                         out[u q + r] = sum_j w[:, :, (r+p)%u + u j] x[q + (r+p)//u - j]  (p = padding), i.e. taps
                         x[q-1], x[q], x[q+1] with a per-phase weight (zero where a phase does not use the tap)
Activations stay in the packed-frames layout, so a ragged batch costs nothing and every utterance equals its B = 1 result.
"""
import os

import numpy as np
import torch

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_TANH, layout, taps_1d
from .synth import hash_tensor
from .weights import fold_state_dict

LRELU_SLOPE = 0.1                      # vocoder.py:8

DEFAULT_H = dict(resblock="1", upsample_rates=[10, 5, 3, 2], upsample_kernel_sizes=[20, 10, 6, 4],
                 upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
                 resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=80)    # Vocoder/config.json


def generator_spec(h=None):
    """name -> shape of the reference Generator's state dict (weight-norm parametrisation, vocoder.py:75-99)."""
    h = {**DEFAULT_H, **(h or {})}
    c0, spec = h["upsample_initial_channel"], {}

    def wn(name, shape, g_len):
        spec[name + ".bias"] = (shape[1] if name.startswith("ups.") else shape[0],)
        spec[name + ".weight_g"] = (g_len, 1, 1)
        spec[name + ".weight_v"] = tuple(shape)

    wn("conv_pre", (c0, h["num_mels"], 7), c0)
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, cout = c0 // 2 ** i, c0 // 2 ** (i + 1)
        wn(f"ups.{i}", (cin, cout, k), cin)               # ConvTranspose1d weight [Cin, Cout, k]; weight_norm over dim 0
    n = 0
    for i in range(len(h["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for k, dil in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
            for j in range(len(dil)):
                wn(f"resblocks.{n}.convs1.{j}", (ch, ch, k), ch)
            for j in range(len(dil)):
                wn(f"resblocks.{n}.convs2.{j}", (ch, ch, k), ch)
            n += 1
    wn("conv_post", (1, c0 // 2 ** len(h["upsample_rates"]), 7), 1)
    return spec


def synth_generator_state_dict(h=None, seed=3407):
    """Seeded synthetic Generator checkpoint (no vocoder weights ship with the reference: README.md:52 is a link).
    Directions U(+-1/sqrt(fan_in)), gains chosen so that activations stay O(1) through the 4 x 18 residual convs."""
    sd = {}
    for name, shape in generator_spec(h).items():
        if name.endswith(".weight_v"):
            # conv [Cout, Cin, k]: fan-in Cin * k; ConvTranspose1d [Cin, Cout, 2u]: two taps reach an output sample
            fan = shape[0] * 2 if name.startswith("ups.") else shape[1] * shape[2]
            sd[name] = hash_tensor("voc." + name, shape, seed, 1.0 / np.sqrt(fan))
        elif name.endswith(".weight_g"):
            sd[name] = None                                 # filled below (needs ||v||)
        else:
            sd[name] = hash_tensor("voc." + name, shape, seed, 0.05)
    for name in list(sd):
        if name.endswith(".weight_g"):
            v = sd[name[:-len("weight_g")] + "weight_v"]
            norm = np.sqrt((v.reshape(v.shape[0], -1).astype(np.float64) ** 2).sum(1)).astype(np.float32).reshape(-1, 1, 1)
            gain = 0.6 if ".convs" in name else 1.0          # residual branches damped
            jitter = 1.0 + 0.1 * hash_tensor("voc." + name, (v.shape[0], 1, 1), seed, 1.0)
            sd[name] = (norm * np.float32(gain) * jitter).astype(np.float32)
    return sd


def _dil_taps(k, d):
    return [(0, d * (t - k // 2)) for t in range(k)]


class Generator:
    """``Generator(h)`` of Vocoder/vocoder.py:75-125 (ResBlock1 configuration)."""

    def __init__(self, h=None, device=None):
        self.h = {**DEFAULT_H, **(h if isinstance(h, dict) else (vars(h) if h is not None else {}))}
        if str(self.h["resblock"]) != "1":
            raise NotImplementedError("only the ResBlock1 generator of Vocoder/config.json is built")
        from .models import _need_gpu
        self.device = _need_gpu(device if device is not None else "cuda")
        self.W = None

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=False):
        sd = sd.get("generator", sd) if isinstance(sd, dict) else sd       # test.py:124: checkpoint['generator']
        w = fold_state_dict(sd)
        dev, h, W = self.device, self.h, {}
        W["pre"] = (ops.prep_weight(w["conv_pre.weight"], dev), w["conv_pre.bias"].to(dev))
        W["post"] = (ops.prep_weight(w["conv_post.weight"], dev), w["conv_post.bias"].to(dev))
        wp = w["conv_post.weight"]                                           # [1][C][7]: one output row -- plain FMAs (ops.conv_post)
        W["post32"] = wp[0].float().contiguous().to(dev) if wp.shape[0] == 1 and wp.shape[2] in (3, 5, 7) else None
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            if k != 2 * u:
                raise NotImplementedError("ConvTranspose1d with k != 2 * stride")
            wt = w[f"ups.{i}.weight"]                                      # [Cin, Cout, k]
            cin, cout = wt.shape[0], wt.shape[1]
            p = u // 2 + u % 2
            wc = torch.zeros(u * cout, cin, 3)                             # rows (phase r', channel m); taps dw = -1, 0, +1
            for r in range(u):
                rr, s = (r + p) % u, (r + p) // u
                for j in (0, 1):
                    dw = s - j
                    wc[r * cout:(r + 1) * cout, :, dw + 1] = wt[:, :, rr + u * j].t()
            W[f"ups{i}"] = (ops.prep_weight(wc, dev), w[f"ups.{i}.bias"].to(dev), u, cout)
            W[f"ups{i}b"] = w[f"ups.{i}.bias"].repeat(u).to(dev)               # per row (phase, channel): the interleaved store adds it
        n = 0
        for i in range(len(h["upsample_rates"])):
            for k, dil in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
                blk = []
                for j, d in enumerate(dil):
                    blk.append((ops.prep_weight(w[f"resblocks.{n}.convs1.{j}.weight"], dev), w[f"resblocks.{n}.convs1.{j}.bias"].to(dev),
                                _dil_taps(k, d),
                                ops.prep_weight(w[f"resblocks.{n}.convs2.{j}.weight"], dev), w[f"resblocks.{n}.convs2.{j}.bias"].to(dev),
                                _dil_taps(k, 1)))
                W[f"rb{n}"] = blk
                n += 1
        self.W = W
        return self

    def remove_weight_norm(self):                                          # folded at load
        return self

    def eval(self):
        return self

    def to(self, device):
        return self

    # ------------------------------------------------------------------ forward
    def forward_packed(self, mel_p, lay):
        """mel_p [80][sum T] packed -> (wav [1][300 sum T], layout of the samples)."""
        W, h = self.W, self.h
        if W is None:
            raise RuntimeError("no weights loaded: call load_state_dict first")
        wt, b = W["pre"]
        x = ops.conv_gemm(wt, mel_p, lay, lay.new(wt.shape[2]), taps_1d(7), bias=b)
        nk = len(h["resblock_kernel_sizes"])
        nst = len(h["upsample_rates"])
        xi = None                                   # LeakyReLU(x) as an operand image, when the producer of x wrote that instead of x
        for i in range(nst):
            wt, b, u, cout = W[f"ups{i}"]
            lay_up = lay.scaled(u)
            # ConvTranspose1d as one 3-tap conv with (phase, channel) rows: its epilogue stores the rows in time order (ileave) -- the
            # interleave_phases pass over the stage's input is gone -- when the channels are a multiple of 32
            il = cout % 32 == 0 and os.environ.get("AS_VOC_ILEAVE", "1") != "0"
            out = lay_up.new(cout) if il else lay.new(u * cout)
            kw = dict(bias=W[f"ups{i}b"], ileave=u) if il else {}
            if xi is not None:
                z = ops.conv_gemm(wt, None, lay, out, taps_1d(3), xs=xi, K=h["upsample_initial_channel"] // 2 ** i, **kw)
                xi = None
            else:
                z = ops.conv_gemm(wt, x, lay, out, taps_1d(3), in_act=ACT_LRELU, in_slope=LRELU_SLOPE, **kw)
            # 32 / 64 channels: the residual steps are fused launches (below); they address a tensor with 32-bit byte offsets: a batch
            # beyond 2 GiB per tensor takes the conv GEMM launches.  (They can read the conv's phase-major output in place -- the interleave
            # as an address computation of their six reads, AS_VOC_FOLD=1 --: measured 14.30 ms per batch against 14.08 with the interleave
            # kernel: the six strided reads cost more than the 0.28 ms the two passes take.)
            fused = cout in (32, 64) and nk == 3 and os.environ.get("AS_VOC_FUSED", "1") != "0" and 4 * u * cout * (lay.N + 1) < 2 ** 31
            if il:
                x = z
            else:
                x = (z, b, u) if fused and os.environ.get("AS_VOC_FOLD", "0") == "1" else ops.interleave_phases(z, b, cout, u, lay.N, lay_up.new(cout))
            lay = lay_up
            outs = []
            # LeakyReLU(x) as an operand image, once for the three residual stacks that start from x; inside a stack every conv hands its
            # LeakyReLU'd result to the next one as an image (ConvGemmArgs.Yh / yh_lrelu): no fp32 copy of conv1's output, no split passes
            # (the three stacks of a stage as ONE launch per step -- ops.conv_gemm_multi -- was measured: 18.2 ms per batch against 17.9 with
            # a launch per conv; these grids hold thousands of tiles each, there is no tail worth filling)
            if fused:
                # 32 / 64 channels: a residual step is ONE launch that keeps its column tile in LDS between the two convs (ops.respair):
                # x in, y out -- no operand images in HBM at all; the stage's mean rides in the last step of the third stack
                for j in range(nk):
                    y = x
                    blk = W[f"rb{i * nk + j}"]
                    for n, (w1, b1, t1, w2, b2, t2) in enumerate(blk):
                        last = j + 1 == nk and n + 1 == len(blk)
                        # (the stage's mean feeds only the next ConvTranspose1d: its last step writes LeakyReLU(mean) as that conv's image)
                        y = ops.respair(y, lay, w1, b1, w2, b2, len(t1), t1[1][1] - t1[0][1] if len(t1) > 1 else 1, LRELU_SLOPE,
                                        add=(outs[0], outs[1]) if last else None,
                                        image_slope=LRELU_SLOPE if last and i + 1 < nst else None)
                    outs.append(y)
                if i + 1 < nst:
                    x, xi = None, outs[2]
                else:
                    x = outs[2]
                continue
            xh = ops.split_act(x, lay, in_act=ACT_LRELU, in_slope=LRELU_SLOPE)
            for j in range(nk):
                y, yh = x, xh
                blk = W[f"rb{i * nk + j}"]
                for n, (w1, b1, t1, w2, b2, t2) in enumerate(blk):
                    xth = ops.new_image(cout, lay.N, x.device)
                    ops.conv_gemm(w1, None, lay, None, t1, bias=b1, xs=yh, K=cout, yh=xth, yh_lrelu=True, in_slope=LRELU_SLOPE)
                    last = n + 1 == len(blk)
                    yh = None if last else ops.new_image(cout, lay.N, x.device)
                    y = ops.conv_gemm(w2, None, lay, lay.new(cout), t2, bias=b2, res=y, xs=xth, K=cout, yh=yh, yh_lrelu=not last,
                                      in_slope=LRELU_SLOPE)
                outs.append(y)
            if nk != 3:
                raise NotImplementedError("three residual stacks per stage (Vocoder/config.json)")
            if i + 1 < nst and os.environ.get("AS_VOC_FUSED", "1") != "0":
                x, xi = None, ops.mean3_image(outs[0], outs[1], outs[2], lay.N, LRELU_SLOPE)
            else:
                x = ops.mean3(outs[0], outs[1], outs[2], lay.N, lay.new(cout))
        wt, b = W["post"]
        if W["post32"] is not None and os.environ.get("AS_VOC_FUSED", "1") != "0":
            wav = ops.conv_post(x, lay, W["post32"], b, 0.01)
        else:
            wav = ops.conv_gemm(wt, x, lay, lay.new(1), taps_1d(7), bias=b, in_act=ACT_LRELU, in_slope=0.01, act=ACT_TANH)
        return wav, lay

    @torch.no_grad()
    def forward(self, x, lengths=None):
        """x: mel [B, 80, T] (zero-padded beyond `lengths`) -> wav [B, 1, 300 * T], zero beyond each utterance."""
        from .models import pack, unpack
        B, _, T = x.shape
        lens = [int(v) for v in lengths] if lengths is not None else [T] * B
        lay = layout(lens, self.device)
        wav, lay_w = self.forward_packed(pack(x.to(self.device), lens), lay)
        return unpack(wav, lay_w)

    __call__ = forward
