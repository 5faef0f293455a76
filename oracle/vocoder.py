"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's HiFi-GAN generator,
``Vocoder/vocoder.py:75-125`` (ResBlock1, ``:11-48``), on folded weights -- plain torch functional calls, one
utterance at a time.  Pinned by tests/test_oracle_vocoder.py against outputs of the reference itself
(tests/golden/voc_*.npz, made by tests/golden/make_golden.py vocoder)."""
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1                                             # vocoder.py:8


def generator(W, h, mel):
    """W: folded state dict (name -> tensor, weight-norm already applied); mel [80, T] -> wav [300 T]."""
    x = F.conv1d(mel[None], W["conv_pre.weight"], W["conv_pre.bias"], padding=3)           # :81, :99
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)                                                   # :101
        x = F.conv_transpose1d(x, W[f"ups.{i}.weight"], W[f"ups.{i}.bias"], stride=u, padding=u // 2 + u % 2,
                               output_padding=u % 2)                                       # :84-89, :102
        xs = None
        for j in range(nk):                                                                # :103-109
            kk, dil = h["resblock_kernel_sizes"][j], h["resblock_dilation_sizes"][j]
            n = i * nk + j
            y = x
            for q, d in enumerate(dil):                                                    # ResBlock1.forward :37-43
                xt = F.leaky_relu(y, LRELU_SLOPE)
                xt = F.conv1d(xt, W[f"resblocks.{n}.convs1.{q}.weight"], W[f"resblocks.{n}.convs1.{q}.bias"],
                              padding=(kk * d - d) // 2, dilation=d)
                xt = F.leaky_relu(xt, LRELU_SLOPE)
                xt = F.conv1d(xt, W[f"resblocks.{n}.convs2.{q}.weight"], W[f"resblocks.{n}.convs2.{q}.bias"],
                              padding=(kk - 1) // 2)
                y = xt + y
            xs = y if xs is None else xs + y
        x = xs / nk                                                                        # :110
    x = F.leaky_relu(x)                                                                    # :111 (slope 0.01)
    x = F.conv1d(x, W["conv_post.weight"], W["conv_post.bias"], padding=3)                 # :112
    return torch.tanh(x)[0, 0]                                                             # :113
