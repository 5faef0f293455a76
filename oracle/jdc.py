"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's JDCNet pitch extractor,
``Utils/JDC/model.py:96-137`` (ResBlock ``:159-190``), eval mode, one utterance at a time -- plain torch functional
calls plus the LSTM restatement of oracle/acoustic.py.  Pinned by tests/test_oracle_jdc.py against outputs of the
reference itself (tests/golden/jdc_*.npz, made by tests/golden/make_golden.py jdc)."""
import torch
import torch.nn.functional as F

from .acoustic import bilstm

SLOPE = 0.01                                                                        # model.py:14


def _bn(W, p, x):                                                                   # nn.BatchNorm2d in eval mode
    return F.batch_norm(x, W[p + ".running_mean"], W[p + ".running_var"], W[p + ".weight"], W[p + ".bias"], False, 0.0, 1e-5)


def _resblock(W, p, x):                                                             # ResBlock.forward, model.py:184-190
    x = F.max_pool2d(F.leaky_relu(_bn(W, p + ".pre_conv.0", x), SLOPE), (1, 2))     # :166-170
    y = F.conv2d(x, W[p + ".conv.0.weight"], padding=1)                             # :173-179
    y = F.conv2d(F.leaky_relu(_bn(W, p + ".conv.1", y), SLOPE), W[p + ".conv.3.weight"], padding=1)
    return y + F.conv2d(x, W[p + ".conv1by1.weight"])                               # :186-187 (always in != out here)


def jdcnet(W, mel):
    """W: the reference's state dict (name -> tensor); mel [80, T] -> |classifier| [num_class, T] (model.py:96-137)."""
    T = mel.shape[1]
    x = mel.t()[None, None].float()                                                 # :103: [1, 1, T, 80]
    x = F.conv2d(x, W["conv_block.0.weight"], padding=1)                            # :19-24
    x = F.conv2d(F.leaky_relu(_bn(W, "conv_block.1", x), SLOPE), W["conv_block.3.weight"], padding=1)
    for i in (1, 2, 3):                                                             # :107-109
        x = _resblock(W, f"res_block{i}", x)
    x = F.max_pool2d(F.leaky_relu(_bn(W, "pool_block.0", x), SLOPE), (1, 4))        # :112-115 (dropout inactive)
    seq = x.permute(0, 2, 1, 3).reshape(T, -1)                                      # :126: [T, 256 * 2]
    h = bilstm(W, "bilstm_classifier", seq)                                         # :127
    out = h @ W["classifier.weight"].t() + W["classifier.bias"]                     # :129-131
    return out.abs().t()                                                            # :137
