"""JDCNet pitch extractor on the HIP path -- SURVEY.md section 8(f) row N1 (first half), the step right before the
acoustic model on the ``test.py`` surface (models.py:432: ``f0_ext = self.pitch_extractor(mel.unsqueeze(1))``).
Mirrors ``Utils/JDC/model.py:10-137`` as the reference builds it (models.py:377: ``JDCNet(num_class=1, seq_len=192)``):
``JDCNet()``, ``load_state_dict``, ``forward(mel [B, 1, 80, T]) -> f0 [B, num_class, T]``.

  conv_block      conv 3x3 1->64 (+BatchNorm folded into weight and bias, LeakyReLU(0.01) in the epilogue), conv 3x3 64->64
  res_block1..3   BatchNorm -> LeakyReLU -> MaxPool over the mel axis by 2 (one kernel), conv 3x3 (+BN, LeakyReLU), conv 3x3
                  with the 1x1 shortcut conv as its residual; 64 -> 128 -> 192 -> 256 channels, 80 -> 40 -> 20 -> 10 mel rows
  pool_block      BatchNorm -> LeakyReLU -> MaxPool by 4, written as the [256*2][frames] sequence the LSTM reads
  bilstm_classifier + classifier + abs          (the detector branch is built by the reference but never run by forward)
Every convolution is the conv GEMM of the acoustic path; images are [mel bins][frames] per utterance (the reference's
[frames][mel bins] transposed: kernels are transposed at load), so a ragged batch needs no padding.  The reference runs a
zero-padded batch as full-length items; ``lengths=None`` does the same (every item T frames long), ``lengths=`` gives
each utterance its own B = 1 result.
"""
import numpy as np
import torch

from . import ops
from .ops import ACT_ABS, ACT_LRELU, taps_2d
from .synth import hash_tensor

SLOPE = 0.01                                   # Utils/JDC/model.py:14
CHANNELS = (64, 128, 192, 256)                 # :27-29
BN_EPS = 1e-5


def jdc_spec(num_class=1):
    """name -> shape of the reference JDCNet's state dict (Utils/JDC/model.py:14-75)."""
    spec = {}

    def bn(name, c):
        spec[name + ".weight"], spec[name + ".bias"] = (c,), (c,)
        spec[name + ".running_mean"], spec[name + ".running_var"] = (c,), (c,)
        spec[name + ".num_batches_tracked"] = ()

    def lstm(name):
        for sfx in ("", "_reverse"):
            spec[f"{name}.weight_ih_l0{sfx}"], spec[f"{name}.weight_hh_l0{sfx}"] = (1024, 512), (1024, 256)
            spec[f"{name}.bias_ih_l0{sfx}"], spec[f"{name}.bias_hh_l0{sfx}"] = (1024,), (1024,)

    spec["conv_block.0.weight"] = (64, 1, 3, 3)
    bn("conv_block.1", 64)
    spec["conv_block.3.weight"] = (64, 64, 3, 3)
    for i in range(3):
        cin, cout, p = CHANNELS[i], CHANNELS[i + 1], f"res_block{i + 1}"
        bn(p + ".pre_conv.0", cin)
        spec[p + ".conv.0.weight"] = (cout, cin, 3, 3)
        bn(p + ".conv.1", cout)
        spec[p + ".conv.3.weight"] = (cout, cout, 3, 3)
        spec[p + ".conv1by1.weight"] = (cout, cin, 1, 1)
    bn("pool_block.0", 256)
    spec["detector_conv.0.weight"] = (256, 640, 1, 1)
    bn("detector_conv.1", 256)
    lstm("bilstm_classifier")
    lstm("bilstm_detector")
    spec["classifier.weight"], spec["classifier.bias"] = (num_class, 512), (num_class,)
    spec["detector.weight"], spec["detector.bias"] = (2, 512), (2,)
    return spec


def synth_jdc_state_dict(num_class=1, seed=3407):
    """Seeded synthetic JDCNet checkpoint (``Utils/JDC/bst.t7`` is not in the tree): convolutions U(+-sqrt(3/fan_in)),
    non-trivial BatchNorm statistics, LSTM / linear U(+-1/sqrt(H))."""
    sd = {}
    for name, shape in jdc_spec(num_class).items():
        tag = "jdc." + name
        if name.endswith("num_batches_tracked"):
            sd[name] = np.asarray(1000, dtype=np.int64)
        elif name.endswith("running_var"):
            sd[name] = (1.0 + 0.5 * hash_tensor(tag, shape, seed, 1.0)).astype(np.float32)            # 0.5 .. 1.5
        elif name.endswith("running_mean"):
            sd[name] = hash_tensor(tag, shape, seed, 0.2)
        elif ".weight" in name and len(shape) == 1:                                                     # BatchNorm gamma
            sd[name] = (1.0 + 0.2 * hash_tensor(tag, shape, seed, 1.0)).astype(np.float32)
        elif len(shape) == 4:
            sd[name] = hash_tensor(tag, shape, seed, float(np.sqrt(3.0 / (shape[1] * shape[2] * shape[3]))))
        elif len(shape) == 2:
            sd[name] = hash_tensor(tag, shape, seed, 1.0 / 16.0)
        else:
            sd[name] = hash_tensor(tag, shape, seed, 0.05)
    return sd


def _bn_affine(w, p):
    """BatchNorm (eval) as y = x * scale + shift."""
    scale = w[p + ".weight"] / torch.sqrt(w[p + ".running_var"] + BN_EPS)
    return scale, w[p + ".bias"] - w[p + ".running_mean"] * scale


class JDCNet:
    """``JDCNet(num_class=1, seq_len=192)`` of models.py:377 (seq_len is unused by forward)."""

    def __init__(self, num_class=1, seq_len=192, leaky_relu_slope=SLOPE, device=None):
        from .models import _need_gpu
        self.num_class, self.slope = num_class, float(leaky_relu_slope)
        self.device = _need_gpu(device if device is not None else "cuda")
        self.W = None
        self._xchg = {}                                 # exchange buffers of the clustered recurrence, per batch size

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=False):
        from .models import Weights
        sd = sd.get("net", sd) if isinstance(sd, dict) and "net" in sd else sd          # models.py:378: params['net']
        w = {k[len("module."):] if k.startswith("module.") else k: torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
             for k, v in sd.items()}
        w = {k: v.detach().float().cpu() for k, v in w.items() if v.dtype.is_floating_point}
        dev, W = self.device, {}

        def conv(name, bn=None):
            """[Cout, Cin, frames, mel] kernel -> ours [Cout, Cin, mel, frames]; a BatchNorm behind the conv folded in."""
            k = w[name + ".weight"].transpose(2, 3)
            if bn is None:
                return ops.prep_weight(k.contiguous(), dev), None
            scale, shift = _bn_affine(w, bn)
            return ops.prep_weight((k * scale[:, None, None, None]).contiguous(), dev), shift.to(dev)

        W["c0"] = conv("conv_block.0", "conv_block.1")
        W["c3"] = conv("conv_block.3")
        for i in range(3):
            p = f"res_block{i + 1}"
            s, t = _bn_affine(w, p + ".pre_conv.0")
            # conv.3 with the block's 1x1 shortcut (conv1by1, no bias: model.py:180-181,186-187) as extra channels of its reduction
            # (conv_gemm x2s / K2): one launch, the shortcut never exists as a tensor
            k3 = w[p + ".conv.3.weight"].transpose(2, 3).contiguous()
            ksc = w[p + ".conv1by1.weight"].reshape(k3.shape[0], -1).contiguous()
            W[p] = dict(pre=(s.to(dev), t.to(dev)), c0=conv(p + ".conv.0", p + ".conv.1"), c3=(ops.prep_weight(k3, dev, sc=[ksc]), None))
        s, t = _bn_affine(w, "pool_block.0")
        W["pool"] = (s.to(dev), t.to(dev))
        W["lstm"] = Weights({k: v for k, v in w.items() if k.startswith("bilstm_classifier.")}, dev)
        W["cls"] = (ops.prep_weight(w["classifier.weight"][:, :, None].contiguous(), dev), w["classifier.bias"].to(dev))
        self.W = W
        return self

    def eval(self):
        return self

    def to(self, device):
        return self

    def modules(self):
        return iter(())

    # ------------------------------------------------------------------ forward
    def forward_packed(self, mel_p, lay):
        """mel_p [80][sum T] packed frames of `lay`'s utterances -> f0 [num_class][sum T]."""
        from .models import bilstm
        W, dev, sl = self.W, self.device, self.slope
        if W is None:
            raise RuntimeError("no weights loaded: call load_state_dict first")
        lens = [int(v) for v in lay.widths_host]
        img = lambda H: ops.layout(lens, dev, H=H)
        t33, H = taps_2d(3, 3), mel_p.shape[0]
        l0 = img(H)
        x = ops.rows_to_images(mel_p, lay, 0, H, l0)                                                   # model.py:103 (transposed image)
        wt, b = W["c0"]
        # conv_block :19-22: the Cin = 1 conv (direct kernel) writes its BatchNorm-folded, LeakyReLU'd result only as the next conv's
        # operand image (64 x 512 000 values at 32 x 200 frames: no fp32 copy, no split pass)
        c0h = ops.new_image(wt.shape[2], l0.N, x.device)
        ops.conv_gemm(wt, x, l0, None, t33, bias=b, act=ACT_LRELU, act_slope=sl, yh=c0h)
        wt3, _ = W["c3"]
        x = ops.conv_gemm(wt3, None, l0, l0.new(wt3.shape[2]), t33, xs=c0h, K=wt.shape[2])             # :23
        for i in range(3):                                                                             # ResBlock.forward :184-190
            blk = W[f"res_block{i + 1}"]
            Ho = H // 2
            lo = img(Ho)
            p = ops.bn_lrelu_maxpool_rows(x, lay, H, 2, blk["pre"][0], blk["pre"][1], sl, lo.new(x.shape[0]))
            # p feeds conv.0 and the shortcut: split ONCE; conv.0 hands its (BatchNorm-folded, LeakyReLU'd) result to conv.3 as an operand
            # image (no fp32 copy, no split pass); conv.3 sums the 1x1 shortcut of p in the same launch
            ph = ops.split_act(p, lo)
            wt, b = blk["c0"]
            ah = ops.new_image(wt.shape[2], lo.N, p.device)
            ops.conv_gemm(wt, None, lo, None, t33, bias=b, act=ACT_LRELU, act_slope=sl, xs=ph, K=p.shape[0], yh=ah)
            wt3, _ = blk["c3"]
            x = ops.conv_gemm(wt3, None, lo, lo.new(wt3.shape[2]), t33, xs=ah, K=wt.shape[2], x2s=ph, K2=p.shape[0])
            H = Ho
        hout = H // 4
        f = ops.bn_lrelu_maxpool_rows(x, lay, H, 4, W["pool"][0], W["pool"][1], sl, lay.new(x.shape[0] * hout), to_channels=True)
        # H = 256: the recurrence of an utterance split over a cluster of four workgroups that keep W_hh in registers (as_bilstm_cluster_f32;
        # the single-workgroup kernel re-streams W_hh from L2 every step: 9.7 us per step, half of this module's time at 200 frames)
        xchg = self._xchg.get(lay.B)
        if xchg is None:
            xchg = self._xchg[lay.B] = ops.bilstm_exchange_buffer(1, lay.B, dev)
        h = bilstm(W["lstm"], "bilstm_classifier", f, lay, xchg=xchg)                                  # :127
        wt, b = W["cls"]
        return ops.conv_gemm(wt, h, lay, lay.new(wt.shape[2]), [(0, 0)], bias=b, act=ACT_ABS)         # :130, :137

    @torch.no_grad()
    def forward(self, x, lengths=None):
        """x: mel [B, 1, 80, T] (models.py:432) -> |classifier| [B, num_class, T] (model.py:137)."""
        from .models import pack, unpack
        if x.dim() == 3:
            x = x.unsqueeze(1)
        B, _, _, T = x.shape
        lens = [int(v) for v in lengths] if lengths is not None else [T] * B
        lay = ops.layout(lens, self.device)
        out = self.forward_packed(pack(x[:, 0].to(self.device).float(), lens), lay)
        return unpack(out, lay)

    __call__ = forward
