"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's EMA_Predictor
(``Utils/EMA/EMA_Predictor.py:65-82``) and of the conformer blocks it is built from (``Utils/EMA/conformer/conformer``:
encoder.py:74-110, feed_forward.py:44-54, attention.py:77-151, convolution.py:137-149, modules.py:29-30), eval mode, ONE
utterance (the reference's decoder2 LSTM has no batch_first: fed [1, T, 256] it takes one step per frame from the zero
state -- SURVEY.md N1).  Plain torch calls on the reference's state dict.  Pinned by tests/test_oracle_ema.py against
outputs of the reference itself (tests/golden/ema_*.npz, made by tests/golden/make_golden.py ema)."""
import math

import torch
import torch.nn.functional as F


def positional_encoding(length, d_model=256):                                       # embedding.py:28-41
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _ln(W, p, x):
    return F.layer_norm(x, (x.shape[-1],), W[p + ".weight"], W[p + ".bias"], 1e-5)


def _bn(W, p, x):                                                                   # x [T, C]: BatchNorm1d over channels, eval
    return (x - W[p + ".running_mean"]) / torch.sqrt(W[p + ".running_var"] + 1e-5) * W[p + ".weight"] + W[p + ".bias"]


def _lin(W, p, x):
    return F.linear(x, W[p + ".weight"], W.get(p + ".bias"))


def _swish(x):
    return x * torch.sigmoid(x)


def _ffn(W, p, x):                                                                  # feed_forward.py:44-54
    return _lin(W, p + ".sequential.4.linear", _swish(_lin(W, p + ".sequential.1.linear", _ln(W, p + ".sequential.0", x))))


def _relative_shift(pos_score):                                                     # attention.py:111-119, literally
    h, t1, t2 = pos_score.shape
    padded = torch.cat([pos_score.new_zeros(h, t1, 1), pos_score], dim=-1)
    padded = padded.reshape(h, t2 + 1, t1)
    return padded[:, 1:].reshape(h, t1, t2)


def _mhsa(W, p, x, heads=4):                                                        # attention.py:140-151, :77-109
    T, D = x.shape
    dh = D // heads
    a = p + ".attention"
    y = _ln(W, p + ".layer_norm", x)
    pos = positional_encoding(T, D)
    q = _lin(W, a + ".query_proj.linear", y).view(T, heads, dh)
    k = _lin(W, a + ".key_proj.linear", y).view(T, heads, dh).permute(1, 0, 2)
    v = _lin(W, a + ".value_proj.linear", y).view(T, heads, dh).permute(1, 0, 2)
    pe = _lin(W, a + ".pos_proj.linear", pos).view(T, heads, dh)
    content = torch.matmul((q + W[a + ".u_bias"]).transpose(0, 1), k.transpose(1, 2))
    pos_score = _relative_shift(torch.matmul((q + W[a + ".v_bias"]).transpose(0, 1), pe.permute(1, 2, 0)))
    attn = F.softmax((content + pos_score) / math.sqrt(D), -1)
    ctx = torch.matmul(attn, v).transpose(0, 1).reshape(T, D)
    return _lin(W, a + ".out_proj.linear", ctx)


def _conv_module(W, p, x):                                                          # convolution.py:137-149
    s = p + ".sequential"
    y = _ln(W, s + ".0", x).t()[None]                                               # [1, C, T]
    y = F.conv1d(y, W[s + ".2.conv.weight"], W[s + ".2.conv.bias"])
    y = y[:, : y.shape[1] // 2] * torch.sigmoid(y[:, y.shape[1] // 2:])             # GLU(dim=1), activation.py:39-41
    k = W[s + ".4.conv.weight"].shape[-1]
    y = F.conv1d(y, W[s + ".4.conv.weight"], None, padding=(k - 1) // 2, groups=y.shape[1])
    y = _swish(_bn(W, s + ".5", y[0].t()))
    return F.conv1d(y.t()[None], W[s + ".7.conv.weight"], W[s + ".7.conv.bias"])[0].t()


def conformer_block(W, p, x):                                                       # encoder.py:74-110 (half-step residuals)
    s = p + ".sequential"
    x = _ffn(W, s + ".0.module", x) * 0.5 + x
    x = _mhsa(W, s + ".1.module", x) + x
    x = _conv_module(W, s + ".2.module", x) + x
    x = _ffn(W, s + ".3.module", x) * 0.5 + x
    return _ln(W, s + ".4", x)


def _lstm_step0(W, p, x, sfx):                                                      # one nn.LSTM step from h0 = c0 = 0
    g = F.linear(x, W[f"{p}.weight_ih_l0{sfx}"], W[f"{p}.bias_ih_l0{sfx}"]) + W[f"{p}.bias_hh_l0{sfx}"]
    H = g.shape[1] // 4
    c = torch.sigmoid(g[:, :H]) * torch.tanh(g[:, 2 * H:3 * H])
    return torch.sigmoid(g[:, 3 * H:]) * torch.tanh(c)


def ema_predictor(W, f0, n, mel, n_blocks=3):
    """W: the reference's state dict; f0 [1, T], n [1, T], mel [80, T] -> EMA [10, T] (EMA_Predictor.forward, B = 1)."""
    x = torch.cat((f0, n, mel), 0).t().float()                                      # :74-75: [T, 82]
    x = F.relu(_bn(W, "encoder1.2", _lin(W, "encoder1.0", x)))                      # :23-30
    for i in range(n_blocks):                                                       # :77-78
        x = conformer_block(W, f"decoder.{i}", x)
    x = torch.cat([_lstm_step0(W, "decoder2", x, ""), _lstm_step0(W, "decoder2", x, "_reverse")], 1)   # :79
    x = F.relu(_bn(W, "decoder3.2", _lin(W, "decoder3.0", x)))                      # :46-53
    return _lin(W, "decoder3.5", x).t()                                             # :80
