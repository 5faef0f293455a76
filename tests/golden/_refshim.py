"""Import shims for running the UNMODIFIED reference (/root/reference) on CPU in the build container.

Only tests/golden/make_golden.py uses this, and only here: /root/reference does not exist on the
GPU box.  Recipe: SURVEY.md Appendix A.  Nothing from the reference is copied; we only stub the two
missing third-party imports, remap the hard-coded "cuda" device strings to "cpu", and hand
torch.load fresh state dicts for the two frozen extractors whose blobs are not in the tree.
"""
import os
import sys
import tempfile
import types

REF = "/root/reference"


def install():
    import torch
    import torch.nn as nn

    # --- stub modules -------------------------------------------------------------------------
    munch = types.ModuleType("munch")

    class Munch(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    munch.Munch = Munch
    sys.modules["munch"] = munch

    ta = types.ModuleType("torchaudio")
    taf = types.ModuleType("torchaudio.functional")
    tat = types.ModuleType("torchaudio.transforms")

    def create_dct(n_mfcc, n_mels, norm):
        import math
        n = torch.arange(float(n_mels))
        k = torch.arange(float(n_mfcc)).unsqueeze(1)
        dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
        if norm is None:
            dct *= 2.0
        else:
            dct[0] *= 1.0 / math.sqrt(2.0)
            dct *= math.sqrt(2.0 / float(n_mels))
        return dct.t()

    taf.create_dct = create_dct

    class MelSpectrogram(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tat.MelSpectrogram = MelSpectrogram
    ta.functional = taf
    ta.transforms = tat
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.functional"] = taf
    sys.modules["torchaudio.transforms"] = tat

    # --- "cuda" -> "cpu" ------------------------------------------------------------------------
    def _remap(args):
        return tuple("cpu" if isinstance(a, str) and a.startswith("cuda") else a for a in args)

    _mod_to = nn.Module.to
    _ten_to = torch.Tensor.to
    nn.Module.to = lambda self, *a, **k: _mod_to(self, *_remap(a), **k)
    torch.Tensor.to = lambda self, *a, **k: _ten_to(self, *_remap(a), **k)

    # --- missing extractor blobs ----------------------------------------------------------------
    _load = torch.load

    def load(path, *a, **k):
        p = str(path)
        if "JDC" in p:
            from Utils.JDC.model import JDCNet
            torch.manual_seed(11)
            return {"net": JDCNet(num_class=1, seq_len=192).state_dict()}
        if "EMA" in p:
            from Utils.EMA.EMA_Predictor import EMA_Predictor
            torch.manual_seed(12)
            return {"model": EMA_Predictor().state_dict()}
        return _load(path, *a, **k)

    torch.load = load

    sys.path.insert(0, REF)
    os.chdir(REF)
    return Munch
