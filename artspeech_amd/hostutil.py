"""Host-side helpers shared by the Python mirrors: attribute dict, packing of padded [B, C, L] tensors into packed frames,
a small weight store and the BiLSTM wrapper for the modules that still order their launches in Python (the section 8(f) rows:
artspeech_amd.jdc / .ema / .vocoder / .frontend).  The acoustic model itself is ordered by the library (csrc/model.hip)."""
import torch

from . import ops


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


def bilstm(W, p, X, lay, xchg=None):
    """nn.LSTM(bidirectional) on packed [I][N] -> [2H][N]: hoisted input GEMM + recurrence kernel."""
    return bilstm_many(W, [(p, X)], lay, xchg)[0]


def bilstm_many(W, items, lay, xchg=None):
    """Several independent BiLSTMs of the same size over the same layout (ArtsPredictor's three branches,
    models.py:606-618): one hoisted input GEMM each, ONE recurrence launch for all of them."""
    jobs, H = [], None
    for p, X in items:
        wih_t, b, whh_t, H = W.lstm(p)
        gx = torch.empty((max(lay.N, 1), 8 * H), dtype=torch.float32, device=X.device)
        ops.conv_gemm(wih_t, X, lay, gx, [(0, 0)], bias=b, transpose_out=True)
        jobs.append((gx, whh_t, lay.new(2 * H)))
    return ops.bilstm(jobs, lay, H, xchg)


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


