"""Parameter inventory of the acoustic model's checkpoint format.

Enumerates every tensor (name -> shape, kind) that the reference's
``ArtsSpeech(stage="second").state_dict()`` holds for the hot path
(reference: models.py:275-287 and the sub-module constructors it calls;
key layout listed in SURVEY.md section 8(a) row A13).  The two frozen feature
extractors (``style_encoder.pitch_extractor.*`` / ``style_encoder.ema_extractor.*``,
models.py:377-383) are NOT part of the path and are not listed.

The inventory is pinned against the reference by tests/golden/make_golden.py
(names and shapes must match the reference's state_dict exactly) and the
result of that comparison is stored in tests/golden/param_inventory.json.

``kind`` tells the synthetic-weight generator (synth.py) and the folding code
(weights.py) what a tensor is:
  w      plain conv/linear weight           b     bias
  wn_g / wn_v   weight_norm magnitude / direction (torch old-style, dim 0)
  sn_w / sn_u / sn_v   spectral_norm weight_orig / u / v
  emb    embedding table   rel  relative-position table   ln_g / ln_b  LayerNorm
  lstm_w / lstm_b
"""
from collections import OrderedDict

N_VOCAB = 178          # RelTransformerEnc.py:6-10 (len(symbols))
N_HEADS = 4            # RelTransformerEnc.py:333
WINDOW = 4             # RelTransformerEnc.py:335
FFN_KERNEL = 9         # RelTransformerEnc.py:332
PRENET_KERNEL = 5      # RelTransformerEnc.py:358
DUR_DIM_IN = 64        # models.py:529 (hard-coded in DurationPredictor)


class Spec(OrderedDict):
    def add(self, name, shape, kind, fan_in=None):
        assert name not in self, name
        self[name] = dict(shape=tuple(int(s) for s in shape), kind=kind, fan_in=fan_in)


def _conv(spec, p, cout, cin, k, bias=True):
    spec.add(p + ".weight", (cout, cin, k), "w", cin * k)
    if bias:
        spec.add(p + ".bias", (cout,), "b", cin * k)


def _linear(spec, p, cout, cin):
    spec.add(p + ".weight", (cout, cin), "w", cin)
    spec.add(p + ".bias", (cout,), "b", cin)


def _wn_conv1d(spec, p, cout, cin_per_group, k, bias=True, dim0=None):
    # weight_norm(nn.Conv1d / nn.ConvTranspose1d): weight_g [dim0,1,1], weight_v full shape
    d0 = cout if dim0 is None else dim0
    spec.add(p + ".weight_g", (d0, 1, 1), "wn_g", cin_per_group * k)
    spec.add(p + ".weight_v", (d0, cin_per_group, k), "wn_v", cin_per_group * k)
    if bias:
        spec.add(p + ".bias", (cout,), "b", cin_per_group * k)


def _sn_conv(spec, p, cout, cin_per_group, ks, bias=True):
    # spectral_norm(nn.ConvNd): weight_orig, weight_u [cout], weight_v [cin*prod(ks)]
    fan = cin_per_group
    for k in ks:
        fan *= k
    spec.add(p + ".weight_orig", (cout, cin_per_group) + tuple(ks), "sn_w", fan)
    spec.add(p + ".weight_u", (cout,), "sn_u", fan)
    spec.add(p + ".weight_v", (fan,), "sn_v", fan)
    if bias:
        spec.add(p + ".bias", (cout,), "b", fan)


def _layernorm(spec, p, c):
    spec.add(p + ".gamma", (c,), "ln_g")
    spec.add(p + ".beta", (c,), "ln_b")


def rel_encoder(spec, p, h, n_layers):
    """RelTransformerEnc.py:328-369."""
    spec.add(p + ".emb.weight", (N_VOCAB, h), "emb", h)
    for i in range(3):
        _conv(spec, f"{p}.pre.conv_layers.{i}", h, h, PRENET_KERNEL)
        _layernorm(spec, f"{p}.pre.norm_layers.{i}", h)
    _conv(spec, p + ".pre.proj", h, h, 1)
    dk = h // N_HEADS
    for i in range(n_layers):
        a = f"{p}.encoder.attn_layers.{i}"
        spec.add(a + ".emb_rel_k", (1, 2 * WINDOW + 1, dk), "rel", dk)
        spec.add(a + ".emb_rel_v", (1, 2 * WINDOW + 1, dk), "rel", dk)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            _conv(spec, f"{a}.{n}", h, h, 1)
        _layernorm(spec, f"{p}.encoder.norm_layers_1.{i}", h)
        _conv(spec, f"{p}.encoder.ffn_layers.{i}.conv_1", 2 * h, h, FFN_KERNEL)
        _conv(spec, f"{p}.encoder.ffn_layers.{i}.conv_2", h, 2 * h, 1)
        _layernorm(spec, f"{p}.encoder.norm_layers_2.{i}", h)
    _layernorm(spec, p + ".encoder.last_ln", h)


def adain_resblk1d(spec, p, din, dout, sdim, upsample=False):
    """models.py:158-202."""
    _wn_conv1d(spec, p + ".conv1", dout, din, 3)
    _wn_conv1d(spec, p + ".conv2", dout, dout, 3)
    _linear(spec, p + ".norm1.fc", 2 * din, sdim)
    _linear(spec, p + ".norm2.fc", 2 * dout, sdim)
    if din != dout:
        _wn_conv1d(spec, p + ".conv1x1", dout, din, 1, bias=False)
    if upsample:
        # depthwise ConvTranspose1d(k3,s2,p1,op1): weight [din, 1, 3]
        _wn_conv1d(spec, p + ".pool", din, 1, 3, dim0=din)


def resblk2d(spec, p, din, dout, downsample):
    """models.py:59-100 (+ LearnedDownSample models.py:19-36)."""
    _sn_conv(spec, p + ".conv1", din, din, (3, 3))
    _sn_conv(spec, p + ".conv2", dout, din, (3, 3))
    if din != dout:
        _sn_conv(spec, p + ".conv1x1", dout, din, (1, 1), bias=False)
    ks = {"half": (3, 3), "channelpreserve": (1, 3), "timepreserve": (3, 1)}.get(downsample)
    if ks is not None:
        _sn_conv(spec, p + ".downsample_res.conv", din, 1, ks)


def resblk1d(spec, p, din, dout):
    """models.py:102-156 with downsample=True."""
    _wn_conv1d(spec, p + ".conv1", din, din, 3)
    _wn_conv1d(spec, p + ".conv2", dout, din, 3)
    if din != dout:
        _wn_conv1d(spec, p + ".conv1x1", dout, din, 1, bias=False)
    _wn_conv1d(spec, p + ".pool", din, 1, 3)


def lstm(spec, p, inp, hid):
    for suf in ("", "_reverse"):
        spec.add(f"{p}.weight_ih_l0{suf}", (4 * hid, inp), "lstm_w", hid)
        spec.add(f"{p}.weight_hh_l0{suf}", (4 * hid, hid), "lstm_w", hid)
        spec.add(f"{p}.bias_ih_l0{suf}", (4 * hid,), "lstm_b", hid)
        spec.add(f"{p}.bias_hh_l0{suf}", (4 * hid,), "lstm_b", hid)


def style_towers(spec, p, d, style_dim):
    """models.py:385-415."""
    _sn_conv(spec, p + ".Mel_block.0", d, 1, (3, 3))
    resblk2d(spec, p + ".Mel_block.1", d, 2 * d, "half")
    resblk2d(spec, p + ".Mel_block.2", 2 * d, 4 * d, "half")
    resblk2d(spec, p + ".Mel_block.3", 4 * d, 8 * d, "half")
    resblk2d(spec, p + ".Mel_block.4", 8 * d, 8 * d, "half")
    _sn_conv(spec, p + ".Mel_block.6", 8 * d, 8 * d, (5, 5))
    _sn_conv(spec, p + ".EMA_block.0", d, 1, (3, 3))
    resblk2d(spec, p + ".EMA_block.1", d, 2 * d, "channelpreserve")
    resblk2d(spec, p + ".EMA_block.2", 2 * d, 4 * d, "channelpreserve")
    resblk2d(spec, p + ".EMA_block.3", 4 * d, 4 * d, "half")
    _sn_conv(spec, p + ".EMA_block.5", 4 * d, 4 * d, (5, 5))
    for blk in ("F0_block", "energy_block"):
        _sn_conv(spec, f"{p}.{blk}.0", d, 1, (3,))
        resblk1d(spec, f"{p}.{blk}.1", d, 2 * d)
        for i in (2, 3, 4):
            resblk1d(spec, f"{p}.{blk}.{i}", 2 * d, 2 * d)
    _linear(spec, p + ".Mellinear", style_dim, 8 * d)
    _linear(spec, p + ".EMAlinear", style_dim // 2, 4 * d)
    _linear(spec, p + ".F0linear", style_dim // 4, 2 * d)
    _linear(spec, p + ".Energylinear", style_dim // 4, 2 * d)


def decoder(spec, p, h, style_dim, n_mels):
    """models.py:474-495."""
    bott = 2 * h
    adain_resblk1d(spec, p + ".encode", h + 128, bott, style_dim * 2)
    _wn_conv1d(spec, p + ".F0_conv", 32, 1, 1)
    _wn_conv1d(spec, p + ".N_conv", 32, 1, 1)
    _wn_conv1d(spec, p + ".EMA_conv", 64, 10, 1)
    _wn_conv1d(spec, p + ".asr_res.0", 64, h, 1)
    adain_resblk1d(spec, p + ".decode.0", bott + 64 + 128, bott, style_dim * 2)
    adain_resblk1d(spec, p + ".decode.1", bott + 64 + 128, bott, style_dim * 2)
    adain_resblk1d(spec, p + ".decode.2", bott + 64 + 128, h, style_dim * 2)
    for i in (3, 4, 5):
        adain_resblk1d(spec, f"{p}.decode.{i}", h, h, style_dim)
    _wn_conv1d(spec, p + ".to_out.0", n_mels, h, 1)


def duration_predictor(spec, p, h, style_dim):
    """models.py:519-538."""
    rel_encoder(spec, p + ".text_encoder", h, 2)
    for i in range(3):
        adain_resblk1d(spec, f"{p}.duration.{i}", h, h, style_dim // 4)
    lstm(spec, p + ".LSTM", h, h // 2)
    _linear(spec, p + ".duration_proj.linear_layer", 1, h)
    d = DUR_DIM_IN
    _sn_conv(spec, p + ".dur_block.0", d, 1, (3, 3))
    resblk2d(spec, p + ".dur_block.1", d, 2 * d, "channelpreserve")
    resblk2d(spec, p + ".dur_block.2", 2 * d, 2 * d, "channelpreserve")
    resblk2d(spec, p + ".dur_block.3", 2 * d, 2 * d, "half")
    _sn_conv(spec, p + ".dur_block.5", 2 * d, 2 * d, (5, 5))
    _linear(spec, p + ".dur_linear", style_dim // 4, 2 * d)


def arts_predictor(spec, p, h, style_dim):
    """models.py:573-594."""
    adain_resblk1d(spec, p + ".shared", h, h, style_dim * 2)
    for br, s in (("F0", style_dim // 4), ("N", style_dim // 4), ("EMA", style_dim // 2)):
        adain_resblk1d(spec, f"{p}.{br}.0", h, h, style_dim * 2, upsample=True)
        adain_resblk1d(spec, f"{p}.{br}.1", h, h // 2, s)
        adain_resblk1d(spec, f"{p}.{br}.2", h // 2, h // 4, s)
    for br in ("F0", "N", "EMA"):
        lstm(spec, f"{p}.{br}_LSTM", h // 4, h // 4)
    _conv(spec, p + ".F0_proj", 1, h // 2, 1)
    _conv(spec, p + ".N_proj", 1, h // 2, 1)
    _conv(spec, p + ".EMA_proj", 10, h // 2, 1)


def artsspeech_spec(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80):
    """All hot-path tensors of ArtsSpeech(stage='second') (models.py:276-287)."""
    assert style_dim == 256, "style slices 256/384/448/512 are hard-coded (models.py:499,597-599)"
    spec = Spec()
    rel_encoder(spec, "arts_encoder", hidden_dim, 4)
    duration_predictor(spec, "durationPredictor", hidden_dim, style_dim)
    arts_predictor(spec, "artsPredictor", hidden_dim, style_dim)
    rel_encoder(spec, "text_encoder", hidden_dim, 4)
    style_towers(spec, "style_encoder", dim_in, style_dim)
    decoder(spec, "decoder", hidden_dim, style_dim, n_mels)
    return spec


EXTRACTOR_PREFIXES = ("style_encoder.pitch_extractor.", "style_encoder.ema_extractor.")
