"""GPU: individual kernels through the C ABI against plain fp32 torch (CPU) references of the same op."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from artspeech_amd import ops
from artspeech_amd.ops import Layout, taps_1d, taps_2d

pytestmark = pytest.mark.gpu


def packed(x_list):
    return torch.cat(x_list, dim=1).contiguous()


# tuning variants (two LDS stages, two k-blocks per iteration, the 256x128 tile) exist only in -DAS_EXPERIMENTS builds of the library
# (AS_BUILD_FLAGS=-DAS_EXPERIMENTS python -m artspeech_amd._build); AS_TEST_EXPERIMENTS=1 adds their ids to the matrix
EXPERIMENTS = bool(os.environ.get("AS_TEST_EXPERIMENTS"))


@pytest.fixture(params=["kt1ns3"] + (["kt1ns2", "kt2ns3"] if EXPERIMENTS else []))
def impl(request, monkeypatch):
    """pipeline shapes of the f16x3 conv GEMM that are compiled in: k-blocks per iteration x LDS stages"""
    monkeypatch.setenv("AS_H3_KT", request.param[2])
    monkeypatch.setenv("AS_H3_NS", request.param[5])
    return request.param


def image_parts(xs, K, N):
    """split image int16 [KBx][4][N+1][8] -> fp32 [2 parts][KBx*16][N+1]"""
    kbx, nx = ops.kbx(K), N + 1
    img = xs[: kbx * 4 * nx * 8].view(torch.float16).reshape(kbx, 2, 2, nx, 8).float()       # [kb][p][kh][n][8]
    return img.permute(1, 0, 2, 4, 3).reshape(2, kbx * 16, nx)


def split_ref(x):
    h = x.half().float()
    return h, (x - h).half().float()


@pytest.mark.parametrize("cin,cout,k,lens", [(64, 128, 3, [50, 13, 1, 200]), (10, 64, 1, [7, 9]), (1, 32, 1, [33]),
                                             (96, 80, 5, [40, 41]), (130, 257, 9, [17, 300, 64]), (512, 1024, 3, [128] * 8)])
@pytest.mark.parametrize("tile", ["", "11", "21", "22", "12", "14"] + (["42"] if EXPERIMENTS else []))
def test_conv1d_gemm(cuda, monkeypatch, impl, cin, cout, k, lens, tile):
    if tile:
        monkeypatch.setenv("AS_GEMM_TILE", tile)
    else:
        monkeypatch.delenv("AS_GEMM_TILE", raising=False)
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    xs = [torch.randn(cin, L, generator=g) for L in lens]
    res = [torch.randn(cout, L, generator=g) for L in lens]
    want = packed([(F.conv1d(x[None], w, b, padding=k // 2)[0] + r) / np.sqrt(2) for x, r in zip(xs, res)])
    lay = Layout(lens, cuda)
    wt = ops.prep_weight(w, cuda)
    y = ops.conv_gemm(wt, packed(xs).to(cuda), lay, lay.new(cout), taps_1d(k), bias=b.to(cuda), res=packed(res).to(cuda), div_sqrt2=True)
    err = float((y.cpu() - want).abs().max())
    assert err <= 2e-5, err


@pytest.mark.parametrize("M,K,K2,n_utt", [(1024, 48, 0, 48), (1024, 32, 80, 64), (512, 64, 0, 112), (1024, 48, 0, 33)])
def test_conv_gemm_mixed_tiles(cuda, monkeypatch, M, K, K2, n_utt):
    """AS_GEMM_MIX=1: a launch of between one and two rounds of the chip (513 .. 1024 tiles of 128 x 128 on 512 workgroup slots: the 1 024-row
    decoder convs at 64 utterances per call) runs its last sixth of columns on 128 x 64 tiles in the SAME launch (as_conv_gemm_h3_launch_mix):
    against float64 torch per utterance (walls between ragged utterances, bias, residual, folded shortcut, the result also as an operand
    image), and against the unmixed launch -- the big tiles' columns bit for bit, the small tiles' within rounding (another order of the
    partial sums).  The last case lies under the rule's threshold: mixed and unmixed are the same launch."""
    g = torch.Generator().manual_seed(M + K + n_utt)
    lens = [int(v) for v in torch.randint(150, 251, (n_utt,), generator=g)]
    lay = Layout(lens, cuda)
    w = torch.randn(M, K, 3, generator=g) / np.sqrt(3 * K + K2)
    w2 = torch.randn(M, K2, generator=g) / np.sqrt(3 * K + K2) if K2 else None
    b = torch.randn(M, generator=g)
    xs = [torch.randn(K, L, generator=g) for L in lens]
    x2 = [torch.randn(K2, L, generator=g) for L in lens] if K2 else None
    res = [torch.randn(M, L, generator=g) for L in lens]
    wt = ops.prep_weight(w, cuda, sc=w2) if K2 else ops.prep_weight(w, cuda)
    X = packed(xs).to(cuda)
    xh = ops.split_act(X, lay)
    x2h = ops.split_act(packed(x2).to(cuda), lay) if K2 else None
    R = packed(res).to(cuda)

    def run():
        yh = ops.new_image(M, lay.N, cuda)
        y = ops.conv_gemm(wt, None, lay, lay.new(M), taps_1d(3), bias=b.to(cuda), res=None if K2 else R, div_sqrt2=True, xs=xh, K=K, x2s=x2h, K2=K2, yh=yh)
        return y.clone(), yh.clone()
    monkeypatch.setenv("AS_GEMM_MIX", "1")                                      # (off by default: no gain inside the step, conv_gemm.hip)
    y_mix, yh_mix = run()
    monkeypatch.delenv("AS_GEMM_MIX")
    y_one, yh_one = run()
    o, worst = 0, 0.0
    for i, L in enumerate(lens):
        want = F.conv1d(xs[i][None].double(), w.double(), b.double(), padding=1)[0]
        want = want + (torch.einsum("mk,kl->ml", w2.double(), x2[i].double()) if K2 else res[i].double())
        want = (want / np.sqrt(2)).float()
        worst = max(worst, float((y_mix[:, o:o + L].cpu() - want).abs().max()))
        o += L
    assert worst <= 2e-5, worst
    assert torch.equal(yh_mix, ops.split_act(y_mix, lay))                       # the image is the split of what was stored
    tn = -(-lay.N // 128)
    tiles = (M // 128) * tn
    if 576 < tiles <= 1024:
        split = (tn - max(1, int(0.17 * tn + 0.5))) * 128
        assert torch.equal(y_mix[:, :split], y_one[:, :split])
        assert not torch.equal(y_mix[:, split:], y_one[:, split:])             # (another tile: another order of partial sums)
        assert float((y_mix - y_one).abs().max()) <= 1e-5
    else:
        assert torch.equal(y_mix, y_one) and torch.equal(yh_mix, yh_one)


@pytest.mark.parametrize("cin,cout,cin2,k,lens,groups", [(64, 128, 64, 3, [50, 13, 1, 200], 1), (128, 64, 40, 9, [300], 1), (16, 512, 1216, 3, [128] * 8, 1),
                                                          (512, 256, 512, 3, [40] * 9, 3), (96, 80, 200, 1, [40, 41], 1)])
@pytest.mark.parametrize("tile,ksplit", [("", ""), ("11", ""), ("21", "3"), ("22", ""), ("12", "2"), ("14", "")])
def test_conv_gemm_second_operand(cuda, monkeypatch, impl, cin, cout, cin2, k, lens, groups, tile, ksplit):
    """ConvGemmArgs.Xh2 / K2: a block's learned shortcut (models.py:79-84,185-186: conv1x1, no bias) as K2 more channels of the
    reduction of the block's last conv -- against conv(x) + conv1x1(x2) in float64, every tile, split-K, grouped weight sets."""
    if tile:
        monkeypatch.setenv("AS_GEMM_TILE", tile)
    if ksplit:
        monkeypatch.setenv("AS_GEMM_KSPLIT", ksplit)
    g = torch.Generator().manual_seed(cin + 3 * cout + cin2 + k)
    G = groups
    ws = [torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k) for _ in range(G)]
    w2 = [torch.randn(cout, cin2, generator=g) / np.sqrt(cin2) * 3.0 for _ in range(G)]       # (another magnitude: one common scale)
    b = torch.randn(G, cout, generator=g)
    per = len(lens) // G
    xs = [torch.randn(cin, L, generator=g) for L in lens]
    x2 = [torch.randn(cin2, L, generator=g) for L in lens]
    want = packed([(F.conv1d(x[None].double(), ws[i // per].double(), b[i // per].double(), padding=k // 2)[0] + w2[i // per].double() @ z.double()) / np.sqrt(2)
                   for i, (x, z) in enumerate(zip(xs, x2))])
    lay = Layout(lens, cuda)
    gc = sum(lens[:per]) if G > 1 else 0
    wt = ops.prep_weight(ws[0], cuda, stack=ws[1:], sc=w2)
    X, X2 = packed(xs).to(cuda), packed(x2).to(cuda)
    xh, x2h = ops.split_act(X, lay), ops.split_act(X2, lay)
    yh = ops.new_image(cout, lay.N, cuda)
    y = ops.conv_gemm(wt, None, lay, lay.new(cout), taps_1d(k), bias=b.to(cuda), div_sqrt2=True, xs=xh, K=cin, x2s=x2h, K2=cin2, group_cols=gc, yh=yh)
    err = float((y.double().cpu() - want).abs().max())
    assert err <= 3e-5, err
    assert torch.equal(yh, ops.split_act(y, lay))


@pytest.mark.parametrize("cin,cout,K,stride,H,widths", [(512, 512, 5, 1, 5, [13, 9, 13, 5]), (256, 128, 5, 2, 10, [25, 24, 6]), (16, 48, 3, 2, 7, [40]),
                                                         (32, 64, 5, 1, 5, [5, 5])])
@pytest.mark.parametrize("tile,ksplit", [("", ""), ("11", "4"), ("21", ""), ("22", "2"), ("12", "")])
def test_conv_gemm_valid_strided(cuda, monkeypatch, impl, cin, cout, K, stride, H, widths, tile, ksplit):
    """ConvGemmArgs.src_col / N_in: the K x K valid (strided) convs that close the 2-D towers (models.py:391,399,535) straight from the
    previous block's operand image -- the output columns are their own layout -- against F.conv2d in float64."""
    if tile:
        monkeypatch.setenv("AS_GEMM_TILE", tile)
    if ksplit:
        monkeypatch.setenv("AS_GEMM_KSPLIT", ksplit)
    g = torch.Generator().manual_seed(cin + cout + K + stride)
    w = torch.randn(cout, cin, K, K, generator=g) / np.sqrt(cin * K * K)
    b = torch.randn(cout, generator=g)
    xs = [torch.randn(cin, H, W, generator=g) for W in widths]
    want = packed([F.leaky_relu(F.conv2d(x[None].double(), w.double(), b.double(), stride=stride)[0], 0.2).reshape(cout, -1) for x in xs])
    lin = Layout(widths, cuda, H=H)
    lout = Layout([(W - K) // stride + 1 for W in widths], cuda, H=(H - K) // stride + 1)
    xh = ops.split_act(packed([x.reshape(cin, -1) for x in xs]).to(cuda), lin)
    col, meta = ops.strided_source(lin, lout, stride, cuda)
    taps = [(a, d) for a in range(K) for d in range(K)]
    y = ops.conv_gemm(ops.prep_weight(w, cuda), None, lout, lout.new(cout), taps, bias=b.to(cuda), act=ops.ACT_LRELU, xs=xh, K=cin,
                      src_col=col, src_meta=meta, N_in=lin.N)
    err = float((y.double().cpu() - want).abs().max())
    assert err <= 2e-5, err


@pytest.mark.parametrize("widths", [[23, 8, 40], [23, 23, 23], [12, 8, 40]])
def test_conv2d_gemm_and_transpose_out(cuda, impl, widths):
    g = torch.Generator().manual_seed(3)
    cin, cout, H = 16, 48, 10
    w = torch.randn(cout, cin, 3, 3, generator=g) / 12
    b = torch.randn(cout, generator=g)
    xs = [torch.randn(cin, H, W, generator=g) for W in widths]
    want = packed([F.conv2d(F.leaky_relu(x, 0.2)[None], w, b, padding=1)[0].reshape(cout, -1) for x in xs])
    lay = Layout(widths, cuda, H=H)
    wt = ops.prep_weight(w, cuda)
    X = packed([x.reshape(cin, -1) for x in xs]).to(cuda)
    y = ops.conv_gemm(wt, X, lay, lay.new(cout), taps_2d(3, 3), bias=b.to(cuda), in_act=ops.ACT_LRELU)
    assert float((y.cpu() - want).abs().max()) <= 2e-5
    yt = torch.empty(lay.N, cout, device=cuda)
    ops.conv_gemm(wt, X, lay, yt, taps_2d(3, 3), bias=b.to(cuda), in_act=ops.ACT_LRELU, transpose_out=True)
    assert torch.equal(yt.t().contiguous(), y)


@pytest.mark.parametrize("K,lens,lrelu", [(80, [50, 13, 1, 200], False), (512, [1000, 24], True), (7, [5], True), (33, [300], False)])
def test_split_activations(cuda, K, lens, lrelu):
    """as_split_f16x2_f32: h = fp16(x), l = fp16(x - h), laid out [kb][p*2+kh][column][8], zero rows past K and a zero column N;
    a conv fed with the image (xs=) equals the conv that lets the library split, bit for bit."""
    g = torch.Generator().manual_seed(K)
    lay = Layout(lens, cuda)
    X = (torch.randn(K, lay.N, generator=g) * torch.logspace(-3, 3, K)[:, None]).to(cuda)
    xs = ops.split_act(X, lay, ops.ACT_LRELU if lrelu else 0, 0.1)
    x = F.leaky_relu(X, 0.1) if lrelu else X
    parts = image_parts(xs, K, lay.N)
    assert torch.equal(parts[:, K:], torch.zeros_like(parts[:, K:])) and not parts[:, :, lay.N].any()
    parts = parts[:, :, : lay.N]
    h, l = split_ref(x)
    assert torch.equal(parts[0, :K], h) and torch.equal(parts[1, :K], l)
    assert float(((parts[0, :K] + parts[1, :K]) - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())
    w = ops.prep_weight(torch.randn(128, K, 3, generator=g) / np.sqrt(3 * K), cuda)
    y0 = ops.conv_gemm(w, X, lay, lay.new(128), taps_1d(3), in_act=ops.ACT_LRELU if lrelu else 0, in_slope=0.1)
    y1 = ops.conv_gemm(w, None, lay, lay.new(128), taps_1d(3), xs=xs, K=K)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("M,K,lens,tile,ksplit", [(128, 64, [50, 13, 1, 200], "22", "")] + ([(300, 64, [50, 13, 1, 200], "42", ""), (256, 48, [129], "42", "2")] if EXPERIMENTS else []) + [(80, 96, [40, 41], "21", ""), (257, 130, [17, 300, 64], "", ""),
                                                  (64, 64, [333], "12", ""), (64, 64, [333, 300], "14", ""), (40, 48, [700], "14", "2"), (200, 200, [33, 70], "", "3"), (512, 512, [40] * 32, "", "")])
@pytest.mark.parametrize("lrelu", [False, True])
def test_epilogue_writes_split_image(cuda, monkeypatch, M, K, lens, tile, ksplit, lrelu):
    """ConvGemmArgs.Yh: the GEMM's epilogue (and the split-K reduce kernel) writes its output as the split operand image of the
    next conv -- bit-identical to as_split_f16x2_f32 of the fp32 output, with or without the fp32 output itself."""
    if tile:
        monkeypatch.setenv("AS_GEMM_TILE", tile)
    if ksplit:
        monkeypatch.setenv("AS_GEMM_KSPLIT", ksplit)
    g = torch.Generator().manual_seed(M + K)
    lay = Layout(lens, cuda)
    w = ops.prep_weight(torch.randn(M, K, 3, generator=g) / np.sqrt(3 * K), cuda)
    b = torch.randn(M, generator=g).to(cuda)
    X = torch.randn(K, lay.N, generator=g).to(cuda)
    y = ops.conv_gemm(w, X, lay, lay.new(M), taps_1d(3), bias=b, act=ops.ACT_RELU)
    want = ops.split_act(y, lay, ops.ACT_LRELU if lrelu else 0, 0.2)
    yh = ops.new_image(M, lay.N, cuda)
    yh.fill_(0x3c00)
    y2 = ops.conv_gemm(w, X, lay, lay.new(M), taps_1d(3), bias=b, act=ops.ACT_RELU, yh=yh, yh_lrelu=lrelu)
    assert torch.equal(y2, y) and torch.equal(yh, want)
    yh2 = ops.new_image(M, lay.N, cuda)
    yh2.fill_(0x3c00)
    ops.conv_gemm(w, X, lay, None, taps_1d(3), bias=b, act=ops.ACT_RELU, yh=yh2, yh_lrelu=lrelu)
    assert torch.equal(yh2, want)


@pytest.mark.parametrize("B,M,K", [(1, 7, 64), (5, 256, 256), (32, 256, 512), (33, 19, 516), (32, 64, 1000), (3, 5, 1024), (32, 8, 2048),
                                   (4, 9, 130), (32, 130, 2052)])
def test_linear_rows(cuda, B, M, K):
    """as_linear_rows_f32 (nn.Linear on per-utterance vectors, models.py:237,412-415,538): every register-group count of the kernel,
    ragged tails, the unaligned fallback -- against float64."""
    g = torch.Generator().manual_seed(B * 1000 + K)
    x, w, b = torch.randn(B, K, generator=g), torch.randn(M, K, generator=g) / np.sqrt(K), torch.randn(M, generator=g)
    y = ops.linear_rows(x.to(cuda), w.to(cuda), b.to(cuda))
    want = x.double() @ w.double().T + b.double()
    assert float((y.double().cpu() - want).abs().max()) <= 2e-6 * (1 + float(want.abs().max()))


@pytest.mark.parametrize("M,M2,lens", [(16, 64, [9, 4]), (32, 128, [40]), (8, 64, [100, 3]), (32, 32, [300, 17])])
def test_small_channel_image_chain(cuda, M, M2, lens):
    """<= 32 output channels take the 32-row tile, which leaves the upper half of the image's 64-row block unwritten (whatever is
    there -- here fp16 NaNs): no consumer tile reads it, the chained result equals the one through the fp32 tensor."""
    g = torch.Generator().manual_seed(M * 7 + M2)
    lay = Layout(lens, cuda)
    w1 = ops.prep_weight(torch.randn(M, 48, 3, generator=g) / np.sqrt(3 * 48), cuda)
    w2 = ops.prep_weight(torch.randn(M2, M, 3, generator=g) / np.sqrt(3 * M), cuda)
    X = torch.randn(48, lay.N, generator=g).to(cuda)
    y = ops.conv_gemm(w1, X, lay, lay.new(M), taps_1d(3))
    yh = ops.new_image(M, lay.N, cuda)
    yh.fill_(0x7e00)
    ops.conv_gemm(w1, X, lay, None, taps_1d(3), yh=yh)
    z0 = ops.conv_gemm(w2, y, lay, lay.new(M2), taps_1d(3))
    z1 = ops.conv_gemm(w2, None, lay, lay.new(M2), taps_1d(3), xs=yh, K=M)
    assert bool(torch.isfinite(z1).all()) and torch.equal(z0, z1)


@pytest.mark.parametrize("G,cin,cout,k,cols", [(2, 64, 128, 3, 256), (3, 96, 80, 1, 128), (3, 512, 512, 3, 6400), (3, 64, 128, 3, 200), (2, 512, 1024, 9, 52)])
def test_grouped_launch(cuda, G, cin, cout, k, cols):
    """ConvGemmArgs.n_groups: G layers of the same shape side by side along the column axis equal G separate launches."""
    g = torch.Generator().manual_seed(G * cols)
    ws = [torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k) * (1 + 3 * i) for i in range(G)]
    bs = [torch.randn(cout, generator=g) for _ in range(G)]
    lens = [cols // 4] * 4 * G
    lay, lay1 = Layout(lens, cuda), Layout(lens[:4], cuda)
    X = torch.randn(cin, lay.N, generator=g).to(cuda)
    wt = ops.prep_weight(ws[0], cuda, stack=ws[1:])
    y = ops.conv_gemm(wt, X, lay, lay.new(cout), taps_1d(k), bias=torch.stack(bs).to(cuda), group_cols=cols)
    for i in range(G):
        yi = ops.conv_gemm(ops.prep_weight(ws[i], cuda), X[:, i * cols:(i + 1) * cols].contiguous(), lay1, lay1.new(cout), taps_1d(k),
                           bias=bs[i].to(cuda))
        assert float((y[:, i * cols:(i + 1) * cols] - yi).abs().max()) <= 1e-5 * (1 + 3 * i)    # (the tile, hence the summation order, may differ)


def test_adain_image_general_addressing(cuda):
    """as_adain_image_f32: gamma / beta read from a [rows][utterances] table through per-utterance offsets, three groups of
    utterances reading ONE shared input (src_off), the fused x2 up-sampler -- against as_adain_f32 + as_split_f16x2_f32 per group."""
    g = torch.Generator().manual_seed(31)
    C, lens, G = 48, [17, 64, 30, 1], 3
    B = len(lens)
    lay, lay3 = Layout(lens, cuda), Layout(lens * G, cuda)
    X = (torch.randn(C, lay.N, generator=g) * 2 + 1).to(cuda)
    gbT = (torch.randn(G * 2 * C, B, generator=g) * 0.3).to(cuda)                  # rows: group g -> gamma[C], beta[C]; columns: utterances
    pw = [torch.randn(C, 3, generator=g).to(cuda) for _ in range(G)]
    pb = [torch.randn(C, generator=g).to(cuda) for _ in range(G)]
    gb_off = torch.tensor([gi * 2 * C * B + b for gi in range(G) for b in range(B)], dtype=torch.int32, device=cuda)
    src_off = torch.tensor([lay.off_host[b] for _ in range(G) for b in range(B)], dtype=torch.int32, device=cuda)
    # plain: three groups in one launch on the tripled layout
    got = ops.adain_image(X, lay3, gbT, B, lay3.N, gb_off=gb_off, src_off=src_off)
    lay3x2 = lay3.scaled(2)
    up = torch.zeros(C, lay3x2.N, device=cuda)
    for gi in range(G):
        gb_rows = gbT[gi * 2 * C:(gi + 1) * 2 * C].t().contiguous()                # [B][2C] as as_adain_f32 wants it
        y = ops.adain(X, gb_rows, lay, lay.new(C), True)
        want = ops.split_act(y, lay)
        gotp, wantp = image_parts(got, C, lay3.N)[:, :C, gi * lay.N:(gi + 1) * lay.N], image_parts(want, C, lay.N)[:, :C, : lay.N]
        assert torch.equal(gotp, wantp)
    # up-sampling variant, one launch per group (the pool weights differ), all into one image
    xs = ops.new_image(C, lay3x2.N, cuda)
    a = ops._lib.AdainArgs()
    for gi in range(G):
        a.x, a.ldx, a.C, a.gb, a.ldgb, a.gb_sc = X.data_ptr(), lay.N, C, gbT.data_ptr(), 1, B
        a.gb_off, a.src_off = gb_off.data_ptr() + 4 * gi * B, src_off.data_ptr() + 4 * gi * B
        a.col_off, a.U, a.N, a.lrelu, a.yh = lay3.col_off.data_ptr() + 4 * gi * B, B, lay3x2.N, 1, xs.data_ptr()
        a.pool_w, a.pool_b, a.x_up, a.ld_up = pw[gi].data_ptr(), pb[gi].data_ptr(), up.data_ptr(), lay3x2.N
        ops.check(ops._lib.lib().as_adain_image_f32(ops.ctypes.byref(a), ops.stream()), "as_adain_image_f32")
    lay2 = lay.scaled(2)
    for gi in range(G):
        gb_rows = gbT[gi * 2 * C:(gi + 1) * 2 * C].t().contiguous()
        yu, xu = lay2.new(C), lay2.new(C)
        ops.adain(X, gb_rows, lay, yu, True, pw[gi], pb[gi], xu)
        want = image_parts(ops.split_act(yu, lay2), C, lay2.N)[:, :C, : lay2.N]
        assert torch.equal(image_parts(xs, C, lay3x2.N)[:, :C, gi * lay2.N:(gi + 1) * lay2.N], want)
        assert torch.equal(up[:, gi * lay2.N:(gi + 1) * lay2.N], xu)
    assert not image_parts(xs, C, lay3x2.N)[:, :, lay3x2.N].any()


def test_rows_image_and_project_cols(cuda):
    g = torch.Generator().manual_seed(32)
    x = torch.randn(5, 70, generator=g).to(cuda)
    parts = image_parts(ops.rows_image(x), 70, 5)
    h, l = split_ref(x.t().contiguous())
    assert torch.equal(parts[0, :70, :5], h) and torch.equal(parts[1, :70, :5], l) and not parts[:, 70:].any() and not parts[:, :, 5].any()
    for M in (1, 3, 10):
        X, w, b = torch.randn(256, 1000, generator=g).to(cuda), torch.randn(M, 256, generator=g).to(cuda), torch.randn(M, generator=g).to(cuda)
        y = ops.project_cols(X, 1000, w, b, torch.empty(M, 1000, device=cuda))
        assert float((y - (w @ X + b[:, None])).abs().max()) <= 2e-4


def test_f16_operand_mode(cuda):
    """n_prod = 1 (AS_GEMM_IMPL=h1): plain fp16 operands, one product -- the error of fp16 rounding, not of the split."""
    g = torch.Generator().manual_seed(11)
    cin, cout, lens = 256, 128, [100, 60]
    w = torch.randn(cout, cin, 3, generator=g) / np.sqrt(3 * cin)
    xs = [torch.randn(cin, L, generator=g) for L in lens]
    want = packed([F.conv1d(x[None], w, padding=1)[0] for x in xs])
    lay = Layout(lens, cuda)
    y = ops.conv_gemm(ops.prep_weight(w, cuda), packed(xs).to(cuda), lay, lay.new(cout), taps_1d(3), n_prod=1)
    err = float((y.cpu() - want).abs().max())
    assert 1e-5 < err <= 5e-3, err
    wh, xh = w.half().float(), [x.half().float() for x in xs]
    want_h = packed([F.conv1d(x[None], wh, padding=1)[0] for x in xh])
    assert float((y.cpu() - want_h).abs().max()) <= 2e-5          # (the weights are scaled by a power of two: same fp16 roundings)


def test_small_operands_are_not_flushed(cuda):
    """The l parts of small activations are fp16 subnormals: the matrix cores must multiply them, not flush them to zero
    (the f16x3 error model in the header of conv_gemm_h3.hip relies on it)."""
    g = torch.Generator().manual_seed(12)
    cin, cout, L = 64, 64, 128
    w = torch.randn(cout, cin, 1, generator=g)
    x = torch.randn(cin, L, generator=g) * 1e-6                    # h is an fp16 subnormal (< 6.1e-5), l below half its spacing
    want = F.conv1d(x.double()[None], w.double())[0]
    lay = Layout([L], cuda)
    y = ops.conv_gemm(ops.prep_weight(w, cuda), x.to(cuda), lay, lay.new(cout), [(0, 0)])
    err = float((y.cpu().double() - want).abs().max())
    # representation error of a subnormal h is <= 2^-25 per element: over 64 products of |w| ~ 0.8 that is ~2e-7 typical,
    # ~6e-7 at the worst of 8192 outputs (measured 5.5e-7); a flush would lose whole products: ~1e-6 * 0.8 * sqrt(64) = 6e-6
    assert err <= 1.5e-6, err


@pytest.mark.parametrize("C,lens", [(80, [50, 13, 1, 200]), (512, [100, 24]), (7, [5])])
def test_adain_split_equals_adain_then_split(cuda, C, lens):
    """as_adain_split_f32 (AdaIN + LeakyReLU stored only as the next conv's operand image) is bit-identical to as_adain_f32
    followed by as_split_f16x2_f32, and a conv fed with it (no fp32 activations at all) to the conv on the fp32 activations."""
    g = torch.Generator().manual_seed(C)
    lay = Layout(lens, cuda)
    X = (torch.randn(C, lay.N, generator=g) * 2 + 0.3).to(cuda)
    gb = torch.randn(len(lens), 2 * C, generator=g).to(cuda) * 0.3
    y = ops.adain(X, gb, lay, lay.new(C), True)
    want = ops.split_act(y, lay)
    got = ops.adain_split(X, gb, lay)
    assert torch.equal(got, want)
    w = ops.prep_weight(torch.randn(96, C, 3, generator=g) / np.sqrt(3 * C), cuda)
    y0 = ops.conv_gemm(w, y, lay, lay.new(96), taps_1d(3))
    y1 = ops.conv_gemm(w, None, lay, lay.new(96), taps_1d(3), xs=got, K=C)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("C,lens,relu", [(512, [40, 33, 7], False), (64, [50, 1, 200], True), (1024, [17], False)])
def test_layernorm_split_equals_layernorm_then_split(cuda, C, lens, relu):
    """as_channel_layernorm_split_f32: the parts sum to the LayerNorm output (to summation order and 2^-22), layout as the split
    image, second affine pair for the columns >= n_split."""
    g = torch.Generator().manual_seed(C + len(lens))
    lay = Layout(lens, cuda)
    X = (torch.randn(C, lay.N, generator=g) * 3 + 1).to(cuda)
    ga, be = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    ga2, be2 = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    n_split = (lay.N + 1) // 2                                    # two column groups (group g = column // n_split takes set g)
    grp = (ga2, be2, n_split)
    want = ops.channel_layernorm(X, lay.N, ga, be, lay.new(C), relu=relu, group2=grp)[:, : lay.N]
    xs = ops.channel_layernorm_split(X, lay, ga, be, relu=relu, group2=grp)
    parts = image_parts(xs, C, lay.N)
    assert not parts[:, :, lay.N].any() and not parts[:, C:].any()
    got = parts[0, :C, : lay.N] + parts[1, :C, : lay.N]
    assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


def test_mfma_layout_asymmetric(cuda, impl):
    """A = I with an asymmetric B: catches a transposed C fragment (guide section 3).  Integers up to 2^22 are exact in h + l."""
    n = 64
    w = torch.eye(n)[:, :, None]                                  # [cout, cin, 1]
    x = torch.arange(n * 96, dtype=torch.float32).reshape(n, 96)
    lay = Layout([96], cuda)
    y = ops.conv_gemm(ops.prep_weight(w, cuda), x.to(cuda), lay, lay.new(n), [(0, 0)])
    assert torch.equal(y.cpu(), x)


def test_adain_and_upsample(cuda):
    g = torch.Generator().manual_seed(5)
    C, lens = 48, [17, 64, 130, 1]
    xs = [torch.randn(C, L, generator=g) * 2 + 1 for L in lens]
    gb = torch.randn(len(lens), 2 * C, generator=g)
    lay = Layout(lens, cuda)
    y = ops.adain(packed(xs).to(cuda), gb.to(cuda), lay, lay.new(C), True)
    want = []
    for b, x in enumerate(xs):
        n = F.instance_norm(x[None], eps=1e-5)[0] if x.shape[1] > 1 else torch.zeros_like(x)
        want.append(F.leaky_relu((1 + gb[b, :C, None]) * n + gb[b, C:, None], 0.2))
    assert float((y.cpu() - packed(want)).abs().max()) <= 2e-5
    pw, pb = torch.randn(C, 1, 3, generator=g), torch.randn(C, generator=g)
    lay2 = lay.scaled(2)
    yu, xu = lay2.new(C), lay2.new(C)
    ops.adain(packed(xs).to(cuda), gb.to(cuda), lay, yu, True, pw.reshape(C, 3).contiguous().to(cuda), pb.to(cuda), xu)
    wantu = [F.conv_transpose1d(a[None], pw, pb, stride=2, padding=1, output_padding=1, groups=C)[0] for a in want]
    assert float((yu.cpu() - packed(wantu)).abs().max()) <= 2e-5
    assert torch.equal(xu.cpu(), packed([x.repeat_interleave(2, dim=1) for x in xs]))


def test_channel_layernorm_embed(cuda):
    g = torch.Generator().manual_seed(6)
    C, N = 96, 150
    x = torch.randn(C, N, generator=g)
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    y = ops.channel_layernorm(x.to(cuda), N, ga.to(cuda), be.to(cuda), torch.empty(C, N, device=cuda), relu=True)
    mean = x.mean(0, keepdim=True)
    var = ((x - mean) ** 2).mean(0, keepdim=True)
    want = torch.relu((x - mean) * torch.rsqrt(var + 1e-4) * ga[:, None] + be[:, None])
    assert float((y.cpu() - want).abs().max()) <= 2e-5
    emb = torch.randn(178, C, generator=g)
    tok = torch.randint(0, 178, (N,), generator=g)
    e = ops.embed(tok.to(cuda, torch.int32), emb.to(cuda), float(np.sqrt(C)), torch.empty(C, N, device=cuda))
    assert float((e.cpu() - (emb[tok] * np.sqrt(C)).t()).abs().max()) <= 1e-6


def test_durations_expand(cuda):
    d = torch.tensor([0.2, 0.5, 1.5, 2.5, 3.49, 2.51, 7.0, 0.99, 1.0, 4.5])
    lay = Layout([4, 6], cuda)
    dur_i, off, _ = ops.durations(d.to(cuda), None, lay, 0)
    want = torch.round(d).clamp(min=1).int()
    assert torch.equal(dur_i.cpu(), want)                        # [1,1,2,2,3,3,7,1,1,4]: half-to-even
    tot = off.cpu().tolist()
    assert tot == [0, int(want[:4].sum()), int(want.sum())]
    dur_i, off, tof = ops.durations(d.to(cuda), None, lay, tot[-1])
    x = torch.arange(30, dtype=torch.float32).reshape(3, 10)
    y = ops.expand(x.to(cuda), tof, tot[-1], 2, torch.empty(3, 2 * tot[-1], device=cuda))
    assert torch.equal(y.cpu(), x.repeat_interleave(want.long(), dim=1).repeat_interleave(2, dim=1))


@pytest.mark.parametrize("lens,C", [([40, 33, 7], 64), ([150], 512), ([1, 2, 70, 64, 65], 64), ([40, 33, 7, 64, 1, 63], 512), ([40] * 32, 512)])
def test_relpos_attention(cuda, lens, C):
    """as_relpos_attention_f32: the exact-fp32 kernel on fp32 q/k/v rows (any head width; the path's 128-channel heads run
    as_relpos_attention_image_f32, tested below against the same oracle)"""
    from oracle import acoustic
    g = torch.Generator().manual_seed(sum(lens) + C)
    W = {"a.emb_rel_k": torch.randn(1, 9, C // 4, generator=g) * 0.1, "a.emb_rel_v": torch.randn(1, 9, C // 4, generator=g) * 0.1}
    for n in "qkvo":
        W[f"a.conv_{n}.weight"] = torch.randn(C, C, 1, generator=g) / np.sqrt(C)
        W[f"a.conv_{n}.bias"] = torch.randn(C, generator=g) * 0.1
    xs = [torch.randn(C, L, generator=g) for L in lens]
    eye = torch.eye(C)[:, :, None]
    Wid = dict(W)
    Wid["a.conv_o.weight"], Wid["a.conv_o.bias"] = eye, torch.zeros(C)
    want = packed([acoustic.relpos_attention(Wid, "a", x) for x in xs])
    qkv = packed([torch.cat([acoustic.conv1d(x, W[f"a.conv_{n}.weight"], W[f"a.conv_{n}.bias"]) for n in "qkv"], 0) for x in xs])
    lay = Layout(lens, cuda)
    out = ops.relpos_attention(qkv.to(cuda), C, 4, 4, W["a.emb_rel_k"][0].contiguous().to(cuda),
                               W["a.emb_rel_v"][0].contiguous().to(cuda), lay, lay.new(C))
    assert float((out.cpu() - want).abs().max()) <= 2e-5


@pytest.mark.parametrize("lens", [[40, 33, 7, 64, 1, 63], [40] * 32, [1, 2, 70, 64, 65, 128, 129, 127], [700, 257, 3], [5]])
def test_relpos_attention_from_image(cuda, lens):
    """as_relpos_attention_image_f32: Q / K fragments straight from the q/k/v GEMM's operand image, V from its fp32 result; output as fp32
    and as the o-projection's operand image (h + l = the fp32 output to 22 bits; the image's zero column written)"""
    from oracle import acoustic
    C = 512
    g = torch.Generator().manual_seed(sum(lens))
    W = {"a.emb_rel_k": torch.randn(1, 9, C // 4, generator=g) * 0.1, "a.emb_rel_v": torch.randn(1, 9, C // 4, generator=g) * 0.1}
    for n in "qkvo":
        W[f"a.conv_{n}.weight"] = torch.randn(C, C, 1, generator=g) / np.sqrt(C)
        W[f"a.conv_{n}.bias"] = torch.randn(C, generator=g) * 0.1
    xs = [torch.randn(C, L, generator=g) for L in lens]
    Wid = dict(W)
    Wid["a.conv_o.weight"], Wid["a.conv_o.bias"] = torch.eye(C)[:, :, None], torch.zeros(C)
    want = packed([acoustic.relpos_attention(Wid, "a", x) for x in xs])
    lay = Layout(lens, cuda)
    wqkv = ops.prep_weight(torch.cat([W[f"a.conv_{n}.weight"] for n in "qkv"], 0), cuda)
    bqkv = torch.cat([W[f"a.conv_{n}.bias"] for n in "qkv"], 0).to(cuda)
    qkv, qkv_h = lay.new(3 * C), ops.new_image(3 * C, lay.N, cuda)
    ops.conv_gemm(wqkv, packed(xs).to(cuda), lay, qkv, taps_1d(1), bias=bqkv, yh=qkv_h)
    ek, ev = W["a.emb_rel_k"][0].contiguous().to(cuda), W["a.emb_rel_v"][0].contiguous().to(cuda)
    out, out_h = lay.new(C), ops.new_image(C, lay.N, cuda)
    out_h.fill_(0x3c00)                                            # (so that an unwritten zero column would be seen)
    ops.relpos_attention_image(qkv, qkv_h, C, 4, 4, ek, ev, lay, out=out, out_h=out_h)
    assert float((out.cpu() - want).abs().max()) <= 2e-5
    parts = image_parts(out_h.cpu(), C, lay.N)
    assert float(parts[:, :, lay.N].abs().max()) == 0.0
    got = (parts[0] + parts[1])[:C, :lay.N]
    assert float((got - out.cpu()).abs().max()) <= 2.0 ** -21 * float(out.abs().max())
    only_h = ops.new_image(C, lay.N, cuda)
    ops.relpos_attention_image(qkv, qkv_h, C, 4, 4, ek, ev, lay, out_h=only_h)
    assert torch.equal(only_h[: ops.kbx(C) * 4 * (lay.N + 1) * 8].cpu(), out_h[: ops.kbx(C) * 4 * (lay.N + 1) * 8].cpu())


@pytest.mark.parametrize("lens,mode", [([40, 12], None), ([40, 12, 0, 7, 1], "1"), ([40, 12, 0, 7, 1], "2"), ([33] * 32, None), ([300, 280], None),
                                       ([9] * 40, None)])
def test_bilstm_cluster(cuda, lens, mode, monkeypatch):
    """H = 256 split over clusters of four workgroups (as_bilstm_cluster_f32): same results as torch.nn.LSTM per utterance, repeated
    launches on one exchange buffer (the epochs), one or two utterances per cluster, the fallback when the grid does not fit (40 x 9)"""
    from artspeech_amd import models
    if mode:
        monkeypatch.setenv("AS_LSTM_CLUSTER", mode)
    H, I = 256, 64
    g = torch.Generator().manual_seed(len(lens))
    lstm = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True)
    raw = {"l." + k: v.detach() for k, v in lstm.state_dict().items()}
    W = models.Weights(raw, cuda)
    xchg = ops.bilstm_exchange_buffer(1, len(lens), cuda)
    lay = Layout(lens, cuda)
    for rep in range(3):
        xs = [torch.randn(I, L, generator=g) for L in lens]
        with torch.no_grad():
            want = packed([lstm(x.t()[None])[0][0].t() if x.shape[1] else torch.zeros(2 * H, 0) for x in xs])
        out = models.bilstm(W, "l", packed(xs).to(cuda), lay, xchg)
        assert float((out.cpu() - want).abs().max()) <= 2e-5, rep


@pytest.mark.parametrize("H,I,lens", [(16, 16, [5, 1, 9]), (128, 128, [40, 200, 3, 77, 50, 60, 70, 80, 11]), (256, 512, [40, 12])])
def test_bilstm(cuda, H, I, lens):
    from artspeech_amd import models
    g = torch.Generator().manual_seed(H + I)
    lstm = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True)
    xs = [torch.randn(I, L, generator=g) for L in lens]
    with torch.no_grad():
        want = packed([lstm(x.t()[None])[0][0].t() for x in xs])
    raw = {"l." + k: v.detach() for k, v in lstm.state_dict().items()}
    W = models.Weights(raw, cuda)
    lay = Layout(lens, cuda)
    out = models.bilstm(W, "l", packed(xs).to(cuda), lay)
    assert float((out.cpu() - want).abs().max()) <= 2e-5


def test_ref_features_crop_pool(cuda):
    g = torch.Generator().manual_seed(9)
    lens = [70, 66]
    N = sum(lens)
    mel, f0, ema = torch.randn(80, N, generator=g) * 0.5, torch.randn(1, N, generator=g) * 70 + 130, torch.randn(10, N, generator=g)
    stats = torch.cat([torch.tensor([4.6, 3.1, 137.0, 78.0]), torch.linspace(-0.1, 0.1, 10), torch.linspace(0.8, 0.9, 10)])
    lay = Layout(lens, cuda)
    feat = ops.ref_features(mel.to(cuda), f0.to(cuda), ema.to(cuda), N, stats.to(cuda), lay.new(12)).cpu()
    n = torch.log(torch.exp(mel * 4 - 4).norm(dim=0))
    assert float((feat[0] - (n - 4.6) / 3.1).abs().max()) <= 1e-5
    assert float((feat[1] - (f0[0] - 137.0) / 78.0).abs().max()) <= 1e-5
    assert float((feat[2:] - (ema - stats[4:14, None]) / stats[14:, None]).abs().max()) <= 1e-5
    l1 = Layout([l - 1 for l in lens], cuda)
    c = ops.crop(feat.to(cuda), lay, 0, l1.new(12), l1).cpu()
    assert torch.equal(c, torch.cat([feat[:, :69], feat[:, 70:70 + 65]], 1))
    limg = Layout([l - 1 for l in lens], cuda, H=10)
    img = ops.rows_to_images(c.to(cuda)[2:12], l1, 0, 10, limg).cpu()          # row-slice view in, H x W images out
    assert torch.equal(img, torch.cat([c[2:12, :69].reshape(1, -1), c[2:12, 69:].reshape(1, -1)], 1))
    p = ops.mean_pool(feat.to(cuda), lay, True).cpu()
    want = torch.stack([F.leaky_relu(feat[:, :70], 0.2).mean(1), F.leaky_relu(feat[:, 70:], 0.2).mean(1)])
    assert float((p - want).abs().max()) <= 1e-5


@pytest.mark.parametrize("kind", ["half", "channelpreserve"])
def test_down_sampling(cuda, kind):
    g = torch.Generator().manual_seed(10)
    C, H, widths = 12, 10, [25, 8, 13]
    xs = [torch.randn(C, H, W, generator=g) for W in widths]
    kh = 3 if kind == "half" else 1
    w, b = torch.randn(C, 1, kh, 3, generator=g), torch.randn(C, generator=g)
    lay = Layout(widths, cuda, H=H)
    lay2 = lay.halved(kind == "half")
    X = packed([x.reshape(C, -1) for x in xs]).to(cuda)
    y = ops.dwconv_down(X, lay, lay2.new(C), lay2, w.reshape(C, -1).contiguous().to(cuda), b.to(cuda), kh, True).cpu()
    stride, pad = ((2, 2), 1) if kind == "half" else ((1, 2), (0, 1))
    want = packed([F.leaky_relu(F.conv2d(x[None], w, b, stride=stride, padding=pad, groups=C)[0], 0.2).reshape(C, -1) for x in xs])
    assert float((y - want).abs().max()) <= 1e-5
    res = torch.randn(C, lay2.N, generator=g)
    z = ops.avgpool_down(X, lay, lay2.new(C), lay2, 2 if kind == "half" else 1, res=res.to(cuda)).cpu()
    pooled = []
    for x in xs:
        if x.shape[-1] % 2:
            x = torch.cat([x, x[..., -1:]], -1)
        pooled.append(F.avg_pool2d(x[None], (2, 2) if kind == "half" else (1, 2))[0].reshape(C, -1))
    assert float((z - (packed(pooled) + res) / np.sqrt(2)).abs().max()) <= 1e-5
    lo = lay.valid_conv(5, 2)
    col = ops.im2col_valid(X, lay, lo.new(C * 25), lo, 5, 2, False).cpu()
    want = packed([F.unfold(x[None], 5, stride=2)[0] for x in xs])
    assert torch.equal(col, want)


@pytest.mark.parametrize("kind,H,widths,C", [("half", 10, [25, 8, 13], 20), ("channelpreserve", 10, [25, 8, 13], 64), ("1d", 1, [199, 66, 7], 64),
                                            ("half", 80, [199, 120], 64)])
def test_stem_pool_image(cuda, kind, H, widths, C):
    """as_stem_pool_image_f32: the first tower block's shortcut input avgpool(stem(x)) from the one-channel input, against
    conv2d -> (replicate the last column of odd widths) -> avg_pool2d in float64 (the image's 22 bits)."""
    g = torch.Generator().manual_seed(5 + H)
    kh = 1 if kind == "1d" else 3
    w = torch.randn(C, 1, kh, 3, generator=g) / 3
    b = torch.randn(C, generator=g)
    xs = [torch.randn(1, H, W, generator=g) for W in widths]
    ph = 2 if kind == "half" else 1
    want = []
    for x in xs:
        y = F.conv2d(x[None].double(), w.double(), b.double(), padding=(kh // 2, 1))
        if y.shape[-1] % 2:
            y = torch.cat([y, y[..., -1:]], -1)
        want.append(F.avg_pool2d(y, (ph, 2))[0].reshape(C, -1))
    want = packed(want)
    lay = Layout(widths, cuda, H=H)
    lay2 = lay.halved(kind == "half")
    x1 = packed([x.reshape(1, -1) for x in xs])[0].contiguous().to(cuda)
    wt = ops.prep_weight(w.reshape(C, 1, kh * 3), cuda)
    img = ops.stem_pool_image(x1, lay, lay2, ph, wt, b.to(cuda), kh)
    parts = image_parts(img, C, lay2.N)
    got = (parts[0, :C, : lay2.N].double() + parts[1, :C, : lay2.N].double()).cpu()
    assert float((got - want).abs().max()) <= 3e-6 * float(want.abs().max())
    assert not parts[:, C:].any() and not parts[:, :, lay2.N].any()


@pytest.mark.parametrize("kind", ["half", "channelpreserve"])
def test_tower_producers_write_images(cuda, kind):
    """The style towers' non-GEMM producers (depthwise down-sampling conv, average pooling with / without the residual merge, im2col, the
    Cin = 1 first conv) written as the consumer conv's operand image: bit-identical to the fp32 kernel followed by as_split_f16x2_f32."""
    g = torch.Generator().manual_seed(13)
    C, H, widths = 20, 10, [25, 8, 13]
    xs = [torch.randn(C, H, W, generator=g) for W in widths]
    kh = 3 if kind == "half" else 1
    w, b = torch.randn(C, kh * 3, generator=g).to(cuda), torch.randn(C, generator=g).to(cuda)
    lay = Layout(widths, cuda, H=H)
    lay2 = lay.halved(kind == "half")
    X = packed([x.reshape(C, -1) for x in xs]).to(cuda)
    y = ops.dwconv_down(X, lay, lay2.new(C), lay2, w, b, kh, True)
    assert torch.equal(ops.dwconv_down_image(X, lay, lay2, w, b, kh, True), ops.split_act(y, lay2))
    ph = 2 if kind == "half" else 1
    z = ops.avgpool_down(X, lay, lay2.new(C), lay2, ph)
    assert torch.equal(ops.avgpool_down_image(X, lay, None, lay2, ph), ops.split_act(z, lay2))
    res = torch.randn(C, lay2.N, generator=g).to(cuda)
    z2 = ops.avgpool_down(X, lay, lay2.new(C), lay2, ph, res=res)
    y2 = lay2.new(C)
    img = ops.avgpool_down_image(X, lay, y2, lay2, ph, res=res, yh_lrelu=True)
    assert torch.equal(y2, z2) and torch.equal(img, ops.split_act(z2, lay2, ops.ACT_LRELU, 0.2))
    lo = lay.valid_conv(5, 2)
    col = ops.im2col_valid(X, lay, lo.new(C * 25), lo, 5, 2, True)
    assert torch.equal(ops.im2col_valid_image(X, lay, lo, 5, 2, True), ops.split_act(col, lo))
    # Cin = 1 direct kernel: fp32 output + LeakyReLU image
    w1, b1 = torch.randn(64, 1, 3, 3, generator=g) / 3, torch.randn(64, generator=g).to(cuda)
    wt = ops.prep_weight(w1, cuda)
    x1 = X[:1].contiguous()
    y1 = ops.conv_gemm(wt, x1, lay, lay.new(64), ops.taps_2d(3, 3), bias=b1)
    yh = ops.new_image(64, lay.N, cuda)
    y1b = ops.conv_gemm(wt, x1, lay, lay.new(64), ops.taps_2d(3, 3), bias=b1, yh=yh, yh_lrelu=True)
    assert torch.equal(y1, y1b) and torch.equal(yh, ops.split_act(y1, lay, ops.ACT_LRELU, 0.2))


@pytest.mark.parametrize("ksplit", ["2", "5", "16"])
def test_conv_gemm_split_k(cuda, monkeypatch, impl, ksplit):
    """split-K slabs + fixed-order reduce give the same result as the unsplit kernel (to rounding)."""
    monkeypatch.setenv("AS_GEMM_KSPLIT", ksplit)
    g = torch.Generator().manual_seed(int(ksplit))
    cin, cout, k, lens = 200, 96, 5, [33, 70]
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    xs = [torch.randn(cin, L, generator=g) for L in lens]
    res = [torch.randn(cout, L, generator=g) for L in lens]
    want = packed([F.leaky_relu((F.conv1d(x[None], w, b, padding=k // 2)[0] + r) / np.sqrt(2), 0.2) for x, r in zip(xs, res)])
    lay = Layout(lens, cuda)
    y = ops.conv_gemm(ops.prep_weight(w, cuda), packed(xs).to(cuda), lay, lay.new(cout), taps_1d(k), bias=b.to(cuda),
                      res=packed(res).to(cuda), div_sqrt2=True, act=ops.ACT_LRELU)
    assert float((y.cpu() - want).abs().max()) <= 2e-5


def test_kernels_side_by_side_on_two_streams(cuda):
    """Results must not depend on what else runs on the chip: the path's branches run on concurrent HIP streams, so a
    kernel shares CUs with other kernels (a timing-dependent fault in a kernel shows up only then).  Long-form shapes."""
    B, N, C = 8, 1024, 512
    lay = ops.layout([N] * B, cuda)
    g = torch.Generator().manual_seed(1)
    qkv = lay.new(3 * C)
    qkv.copy_(torch.randn(3 * C, lay.N, generator=g))
    ek, ev = (torch.randn(9, 128, generator=g) * 0.1).to(cuda), (torch.randn(9, 128, generator=g) * 0.1).to(cuda)
    w = ops.prep_weight(torch.randn(1024, 512, 9, generator=g) / 68, cuda)
    X = lay.new(512)
    X.copy_(torch.randn(512, lay.N, generator=g))
    att = lambda: ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, lay.new(C))
    gemm = lambda: ops.conv_gemm(w, X, lay, lay.new(1024), taps_1d(9))
    ref_a, ref_g = att().clone(), gemm().clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for first, second, r1, r2 in ((att, gemm, ref_a, ref_g), (gemm, gemm, ref_g, ref_g), (att, att, ref_a, ref_a)):
        for _ in range(3):
            with torch.cuda.stream(s1):
                a = first()
            with torch.cuda.stream(s2):
                b = second()
            torch.cuda.synchronize()
            assert torch.equal(a, r1) and torch.equal(b, r2)


def test_image_attention_side_by_side_and_equal_to_exact(cuda):
    """The path's attention kernel at 40-token utterances: within 5e-6 of the exact-fp32 kernel on the same q/k/v, and bit-stable when it
    shares the chip with a GEMM on another stream."""
    B, N, C = 32, 40, 512
    lay = ops.layout([N] * (B - 2) + [64, 1], cuda)
    g = torch.Generator().manual_seed(2)
    wq = ops.prep_weight(torch.randn(3 * C, C, 1, generator=g) / np.sqrt(C), cuda)
    Xq = lay.new(C)
    Xq.copy_(torch.randn(C, lay.N, generator=g))
    qkv, qkv_h = lay.new(3 * C), ops.new_image(3 * C, lay.N, cuda)
    ops.conv_gemm(wq, Xq, lay, qkv, taps_1d(1), yh=qkv_h)
    ek, ev = (torch.randn(9, 128, generator=g) * 0.1).to(cuda), (torch.randn(9, 128, generator=g) * 0.1).to(cuda)
    att = lambda: ops.relpos_attention_image(qkv, qkv_h, C, 4, 4, ek, ev, lay, out=lay.new(C))
    ref = att().clone()
    exact = ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, lay.new(C))
    assert float((ref - exact).abs().max()) <= 5e-6           # (22-bit operands; outputs are O(1))
    w = ops.prep_weight(torch.randn(1024, 512, 9, generator=g) / 68, cuda)
    X = lay.new(512)
    X.copy_(torch.randn(512, lay.N, generator=g))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(5):
        with torch.cuda.stream(s1):
            a = att()
        with torch.cuda.stream(s2):
            ops.conv_gemm(w, X, lay, lay.new(1024), taps_1d(9))
        with torch.cuda.stream(s1):
            a2 = att()
        torch.cuda.synchronize()
        assert torch.equal(a, ref) and torch.equal(a2, ref)


@pytest.mark.parametrize("two_d", [False, True])
def test_conv_cin1_direct_kernel(cuda, two_d):
    """Cin = 1 convs (first layer of each tower) take the direct kernel inside as_conv_gemm_f32: bias, LeakyReLU on the
    input and on the output, ragged batch, 1-D k=3 and 2-D 3x3."""
    g = torch.Generator().manual_seed(21)
    cout = 64
    if two_d:
        H, widths = 10, [23, 8, 40]
        w, b = torch.randn(cout, 1, 3, 3, generator=g) / 3, torch.randn(cout, generator=g)
        xs = [torch.randn(1, H, W, generator=g) for W in widths]
        want = packed([F.leaky_relu(F.conv2d(F.leaky_relu(x, 0.2)[None], w, b, padding=1)[0], 0.2).reshape(cout, -1) for x in xs])
        lay, taps, X = Layout(widths, cuda, H=H), taps_2d(3, 3), packed([x.reshape(1, -1) for x in xs])
    else:
        lens = [50, 13, 1, 200]
        w, b = torch.randn(cout, 1, 3, generator=g) / 2, torch.randn(cout, generator=g)
        xs = [torch.randn(1, L, generator=g) for L in lens]
        want = packed([F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.2)[None], w, b, padding=1)[0], 0.2) for x in xs])
        lay, taps, X = Layout(lens, cuda), taps_1d(3), packed(xs)
    y = ops.conv_gemm(ops.prep_weight(w, cuda), X.to(cuda), lay, lay.new(cout), taps, bias=b.to(cuda), in_act=ops.ACT_LRELU,
                      act=ops.ACT_LRELU)
    assert float((y.cpu() - want).abs().max()) <= 2e-6



def _multi_problems(cuda, g, specs):
    """specs: (cin, cout, k, lens, flavour) -- independent conv problems of the kinds the path merges.  Returns the deferred launch list,
    the outputs and float64 references."""
    deferred, outs, wants, checks = [], [], [], []
    for cin, cout, k, lens, flavour in specs:
        lay = Layout(lens, cuda)
        if flavour == "grouped":                                 # three weight sets side by side (the triple encoder / the predictor branches)
            G, per = 3, len(lens) // 3
            ws = [torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k) for _ in range(G)]
            b = torch.randn(G, cout, generator=g)
            xs = [torch.randn(cin, L, generator=g) for L in lens]
            want = packed([F.relu(F.conv1d(x[None].double(), ws[i // per].double(), b[i // per].double(), padding=k // 2)[0]) for i, x in enumerate(xs)])
            xh = ops.split_act(packed(xs).to(cuda), lay)
            y = lay.new(cout)
            ops.conv_gemm(ops.prep_weight(ws[0], cuda, stack=ws[1:]), None, lay, y, taps_1d(k), bias=b.to(cuda), act=ops.ACT_RELU, xs=xh, K=cin,
                          group_cols=sum(lens[:per]), defer=deferred)
        elif flavour == "shortcut":                              # learned shortcut folded in + the result also as an operand image
            cin2 = cin // 2 + 8
            w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
            w2 = torch.randn(cout, cin2, generator=g) / np.sqrt(cin2)
            b = torch.randn(cout, generator=g)
            xs = [torch.randn(cin, L, generator=g) for L in lens]
            x2 = [torch.randn(cin2, L, generator=g) for L in lens]
            want = packed([(F.conv1d(x[None].double(), w.double(), b.double(), padding=k // 2)[0] + w2.double() @ z.double()) / np.sqrt(2)
                           for x, z in zip(xs, x2)])
            xh, x2h = ops.split_act(packed(xs).to(cuda), lay), ops.split_act(packed(x2).to(cuda), lay)
            y, yh = lay.new(cout), ops.new_image(cout, lay.N, cuda)
            ops.conv_gemm(ops.prep_weight(w, cuda, sc=[w2]), None, lay, y, taps_1d(k), bias=b.to(cuda), div_sqrt2=True, xs=xh, K=cin, x2s=x2h,
                          K2=cin2, yh=yh, defer=deferred)
            checks.append((yh, y, lay))
        else:                                                    # plain conv + residual, LeakyReLU epilogue
            w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
            b = torch.randn(cout, generator=g)
            xs = [torch.randn(cin, L, generator=g) for L in lens]
            res = [torch.randn(cout, L, generator=g) for L in lens]
            want = packed([F.leaky_relu(F.conv1d(x[None].double(), w.double(), b.double(), padding=k // 2)[0] + r.double(), 0.2) for x, r in zip(xs, res)])
            xh = ops.split_act(packed(xs).to(cuda), lay)
            y = lay.new(cout)
            ops.conv_gemm(ops.prep_weight(w, cuda), None, lay, y, taps_1d(k), bias=b.to(cuda), res=packed(res).to(cuda), act=ops.ACT_LRELU, xs=xh,
                          K=cin, defer=deferred)
        outs.append(y)
        wants.append(want)
    return deferred, outs, wants, checks


MULTI_SETS = {
    "enc+dur": [(512, 1024, 9, [40] * 32, "plain"), (512, 512, 3, [40] * 12, "shortcut"), (256, 512, 1, [33, 40, 17] * 3, "grouped")],
    "towers": [(64, 128, 3, [199, 66, 150, 80], "plain"), (128, 256, 3, [100, 33, 75, 40], "shortcut"), (128, 128, 1, [7], "plain"),
               (256, 256, 5, [13, 9, 13, 5, 1, 2], "plain"), (96, 160, 9, [300], "plain"), (512, 130, 3, [64, 1, 129], "plain")],
    "short": [(64, 64, 3, [3000, 1000], "plain"), (64, 48, 9, [1500], "plain")],
    "long_k": [(512, 256, 9, [20, 9], "plain"), (1024, 128, 5, [11, 30], "plain")],
}


@pytest.mark.parametrize("name", sorted(MULTI_SETS))
@pytest.mark.parametrize("tile", ["", "22", "21", "12", "11"])
def test_conv_gemm_multi(cuda, monkeypatch, impl, name, tile):
    """as_conv_gemm_multi_f32: independent convolutions of the shapes the path's branches hold (encoder FFN beside the duration
    predictor's blocks, tower blocks of several widths, two closing long-K convs that need K slices even together) as ONE launch, each with
    its own epilogue / groups / second operand -- every problem against float64 torch, operand images against `h + l = y`, and bitwise
    against the same problems launched one by one with the same tile."""
    specs = MULTI_SETS[name]
    if tile:
        if name == "short" and tile in ("22", "21"):
            pytest.skip("128-row tiles are not offered for <= 64 output channels")
        monkeypatch.setenv("AS_GEMM_TILE", tile)
    else:
        monkeypatch.delenv("AS_GEMM_TILE", raising=False)
    g = torch.Generator().manual_seed(len(name) * 131 + len(specs))
    deferred, outs, wants, checks = _multi_problems(cuda, g, specs)
    chosen = []
    ops.conv_gemm_multi(deferred, tile_out=chosen)
    torch.cuda.synchronize()
    for (cin, cout, k, lens, fl), y, want in zip(specs, outs, wants):
        err = float((y.double().cpu() - want).abs().max())
        assert err <= 3e-5, (name, cin, cout, k, fl, err, chosen)
    for yh, y, lay in checks:
        assert torch.equal(yh, ops.split_act(y, lay))
    # one by one with the set's tile: the same bits (same k order inside a tile; K slices are summed in slice order either way)
    monkeypatch.setenv("AS_GEMM_TILE", str(chosen[0]))
    g = torch.Generator().manual_seed(len(name) * 131 + len(specs))
    deferred1, outs1, _, _ = _multi_problems(cuda, g, specs)
    L = ops._lib.lib()
    import ctypes
    alone_slices = []
    for a, _keep in deferred1:
        k, t, sl = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        ops.check(L.as_conv_gemm_plan(ctypes.byref(a), ctypes.byref(k), ctypes.byref(t), ctypes.byref(sl)), "as_conv_gemm_plan")
        assert t.value == chosen[0]
        alone_slices.append(sl.value)
        ops.check(L.as_conv_gemm_f32(ctypes.byref(a), ops.stream()), "as_conv_gemm_f32")
    torch.cuda.synchronize()
    for y, y1, sl in zip(outs, outs1, alone_slices):
        if sl == 1:                                             # (alone a tiny problem takes K slices: another order of partial sums)
            assert torch.equal(y, y1)
        else:
            assert float((y - y1).abs().max()) <= 2e-5


@pytest.mark.parametrize("case", ["single", "multi", "grouped", "wide", "unsliced"])
def test_conv_gemm_post_adain(cuda, monkeypatch, case):
    """as_conv_gemm_multi_post_f32 (models.py:189-202: conv -> AdaIN -> LeakyReLU -> conv): the AdaIN image of a conv's result from the
    same call.  At batch-1 sizes the launch is K-sliced and its reduction kernel writes the image (one dependent launch less) -- bitwise
    what the conv followed by as_adain_image_f32 gives, fp32 rows and the plain image included; utterances wider than 256 columns and
    unsliced launches take the two-launch route inside the call; with AS_NO_REDUCE_ADAIN every case does."""
    import ctypes
    g = torch.Generator().manual_seed(hash(case) % 1000)
    if case == "single":
        specs = [(1024, 1024, 3, [150], "plain")]
    elif case == "multi":
        specs = [(512, 512, 3, [30], "shortcut"), (256, 1024, 9, [90], "plain"), (512, 512, 3, [7, 30, 1], "plain")]
    elif case == "grouped":
        specs = [(256, 128, 3, [60, 41, 5] * 3, "grouped")]
    elif case == "wide":
        specs = [(512, 256, 3, [300, 20], "plain")]
    else:
        specs = [(64, 128, 3, [200] * 32, "plain")]

    def run(fused):
        if fused:
            monkeypatch.delenv("AS_NO_REDUCE_ADAIN", raising=False)
        else:
            monkeypatch.setenv("AS_NO_REDUCE_ADAIN", "1")
        gg = torch.Generator().manual_seed(1234)
        deferred, outs, wants, checks = _multi_problems(cuda, gg, specs)
        posts, imgs, gbs = [], [], []
        for (cin, cout, k, lens, fl), (a, keep) in zip(specs, deferred):
            lay = keep[10]
            gb = torch.randn(2 * cout, lay.B, generator=gg).to(cuda)            # [2C][U]: the fc GEMM's layout (gb_sc = U, utterance u at + u)
            img = ops.new_image(cout, lay.N, cuda)
            img.fill_(0x7e7e)
            posts.append((gb, lay.B, lay, img))
            imgs.append(img)
            gbs.append(gb)
        plans = []
        for a, _ in deferred:
            kk, tt, sl = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
            ops.check(ops._lib.lib().as_conv_gemm_plan(ctypes.byref(a), ctypes.byref(kk), ctypes.byref(tt), ctypes.byref(sl)), "as_conv_gemm_plan")
            plans.append(sl.value)
        ops.conv_gemm_multi_post(deferred, posts)
        torch.cuda.synchronize()
        return outs, wants, checks, imgs, gbs, [k[10] for _, k in deferred], plans
    outs, wants, checks, imgs, gbs, lays, plans = run(True)
    if case in ("single", "grouped", "wide"):
        assert plans[0] > 1, plans                                                # (the case is about K-sliced launches)
    if case == "unsliced":
        assert plans[0] == 1
    for (cin, cout, k, lens, fl), y, want in zip(specs, outs, wants):
        assert float((y.double().cpu() - want).abs().max()) <= 3e-5
    for yh, y, lay in checks:
        assert torch.equal(yh, ops.split_act(y, lay))
    # the image: AdaIN + LeakyReLU of the fp32 result by the stand-alone kernel, bit for bit
    for y, img, gb, lay in zip(outs, imgs, gbs, lays):
        ref = ops.adain_image(y, lay, gb, lay.B, lay.N)
        torch.cuda.synchronize()
        bad = (img[: ref.numel()] != ref).nonzero().flatten()
        where = [(int(i) // (8 * (lay.N + 1)) // 4, int(i) // (8 * (lay.N + 1)) % 4, int(i) // 8 % (lay.N + 1), int(i) % 8) for i in bad[:6]]
        assert bad.numel() == 0, (case, int(bad.numel()), "(k-block, plane, column, element):", where)
    outs0, _, checks0, imgs0, _, _, _ = run(False)
    for y, y0 in zip(outs, outs0):
        assert torch.equal(y, y0)
    for i, i0 in zip(imgs, imgs0):
        assert torch.equal(i, i0)
    for (yh, _, _), (yh0, _, _) in zip(checks, checks0):
        assert torch.equal(yh, yh0)


@pytest.mark.parametrize("case", ["single", "multi", "grouped", "wide_m", "unsliced"])
def test_conv_gemm_post_layernorm(cuda, monkeypatch, case):
    """as_conv_gemm_multi_post_f32 with a channel LayerNorm behind the conv (RelTransformerEnc.py:72-87, 318-325: conv -> residual add ->
    LayerNorm -> conv).  K-sliced launches of <= 512 channels store their partial sums time-major and the reduction kernel -- a wave per
    column -- writes the LayerNorm image: bitwise what the conv followed by as_channel_layernorm_split_f32 gives (fp32 rows included);
    1 024 channels and unsliced launches take the two-launch route inside the call; with AS_NO_REDUCE_LN every case does."""
    import ctypes
    if case == "single":
        specs = [(1024, 512, 1, [30, 30, 30], "plain")]
    elif case == "multi":
        specs = [(512, 512, 5, [90], "plain"), (1024, 512, 1, [7, 30, 1], "plain"), (256, 64, 3, [40], "shortcut_free")]
    elif case == "grouped":
        specs = [(512, 512, 5, [30, 11, 3] * 3, "grouped")]
    elif case == "wide_m":
        specs = [(512, 1024, 3, [90], "plain")]
    else:
        specs = [(64, 128, 3, [200] * 32, "plain")]
    specs = [(a, b, c, d, "plain" if e == "shortcut_free" else e) for a, b, c, d, e in specs]

    def run(fused):
        if fused:
            monkeypatch.delenv("AS_NO_REDUCE_LN", raising=False)
        else:
            monkeypatch.setenv("AS_NO_REDUCE_LN", "1")
        gg = torch.Generator().manual_seed(4321)
        deferred, outs, wants, checks = _multi_problems(cuda, gg, specs)
        lns, imgs, pars = [], [], []
        for (cin, cout, k, lens, fl), (a, keep) in zip(specs, deferred):
            lay = keep[10]
            if fl == "grouped":                                                  # three parameter sets, one per column group
                per = sum(lens[: len(lens) // 3])
                gam, bet = torch.randn(3, cout, generator=gg).to(cuda), torch.randn(3, cout, generator=gg).to(cuda)
                extra = (gam[1], bet[1], per)
            else:
                gam, bet = torch.randn(1, cout, generator=gg).to(cuda), torch.randn(1, cout, generator=gg).to(cuda)
                extra = ()
            img = ops.new_image(cout, lay.N, cuda)
            img.fill_(0x7e7e)
            lns.append((gam[0], bet[0], img, cin % 2 == 0) + extra)
            imgs.append(img)
            pars.append((gam, bet, extra))
        plans = []
        for a, _ in deferred:
            kk, tt, sl = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
            ops.check(ops._lib.lib().as_conv_gemm_plan(ctypes.byref(a), ctypes.byref(kk), ctypes.byref(tt), ctypes.byref(sl)), "as_conv_gemm_plan")
            plans.append(sl.value)
        ops.conv_gemm_multi_post(deferred, None, lns)
        torch.cuda.synchronize()
        return outs, wants, imgs, pars, [k[10] for _, k in deferred], plans, lns
    outs, wants, imgs, pars, lays, plans, lns = run(True)
    if case in ("single", "grouped", "wide_m"):
        assert plans[0] > 1, plans
    if case == "unsliced":
        assert plans[0] == 1
    for (cin, cout, k, lens, fl), y, want in zip(specs, outs, wants):
        assert float((y.double().cpu() - want).abs().max()) <= 3e-5
    for y, img, (gam, bet, extra), lay, ln in zip(outs, imgs, pars, lays, lns):
        ref = ops.channel_layernorm_split(y, lay, gam[0], bet[0], relu=ln[3], group2=(extra if extra else None))
        torch.cuda.synchronize()
        bad = (img[: ref.numel()] != ref).nonzero().flatten()
        where = [(int(i) // (8 * (lay.N + 1)) // 4, int(i) // (8 * (lay.N + 1)) % 4, int(i) // 8 % (lay.N + 1), int(i) % 8) for i in bad[:6]]
        assert bad.numel() == 0, (case, int(bad.numel()), "(k-block, plane, column, element):", where)
    outs0, _, imgs0, _, _, _, _ = run(False)
    for y, y0 in zip(outs, outs0):
        assert torch.equal(y, y0)
    for i, i0 in zip(imgs, imgs0):
        assert torch.equal(i, i0)


def test_conv_gemm_multi_rejects_what_it_cannot_merge(cuda):
    g = torch.Generator().manual_seed(1)
    lay = Layout([50, 20], cuda)
    w = torch.randn(64, 32, 3, generator=g)
    x = torch.randn(32, lay.N, generator=g).to(cuda)
    d = []
    ops.conv_gemm(ops.prep_weight(w, cuda), x, lay, lay.new(64), taps_1d(3), defer=d)                          # fp32 input: no operand image
    ops.conv_gemm(ops.prep_weight(w, cuda), None, lay, lay.new(64), taps_1d(3), xs=ops.split_act(x, lay), K=32, defer=d)
    with pytest.raises(ops._lib.HipLibraryError):
        ops.conv_gemm_multi(d)
    ops.conv_gemm_multi(d[1:])                                  # n == 1: the single launch
    with pytest.raises(ops._lib.HipLibraryError):
        ops.conv_gemm_multi(d[1:] * 7)                          # more than AS_MAX_MULTI


@pytest.mark.parametrize("K,M,lens", [(12, 128, [200, 33, 1]), (3, 64, [77]), (16, 128, [400] * 4)])
def test_pointwise_small(cuda, K, M, lens):
    """as_pointwise_small_f32: decoder.F0_conv / N_conv / EMA_conv as one 12 -> 128 1x1 conv written to two destinations, fp32 rows and
    operand-image rows (models.py:480-482,503-505) -- against torch in float64; the image parts are whole k-blocks inside larger images."""
    import ctypes
    g = torch.Generator().manual_seed(K * 31 + M)
    lay = Layout(lens, cuda)
    N = lay.N
    w = torch.randn(M, K, generator=g)
    b = torch.randn(M, generator=g)
    x = torch.randn(K, N, generator=g)
    want = (w.double() @ x.double() + b.double()[:, None]).float()
    xd = x.to(cuda).contiguous()
    C0 = 64                                                       # channels in front of the part inside the larger images
    y1 = torch.full((C0 + M, N), 7.0, device=cuda)
    y2 = torch.full((M + 32, N), 7.0, device=cuda)
    img1 = ops.new_image(C0 + M, N, cuda)
    img2 = ops.new_image(M, N, cuda)
    img1.zero_()
    blk_elems = 4 * (N + 1) * 8                                   # int16 elements of one k-block of an image over N columns
    part1 = img1[(C0 // 16) * blk_elems:]
    L = ops._lib.lib()
    wd, bd = w.to(cuda).contiguous(), b.to(cuda)
    ops.check(L.as_pointwise_small_f32(xd.data_ptr(), N, K, N, wd.data_ptr(), bd.data_ptr(), M, y1[C0:].data_ptr(), N, part1.data_ptr(),
                                       y2.data_ptr(), N, img2.data_ptr(), ops.stream()), "as_pointwise_small_f32")
    torch.cuda.synchronize()
    assert float((y1[C0:].cpu() - want).abs().max()) <= 1e-5 and float((y2[:M].cpu() - want).abs().max()) <= 1e-5
    assert bool((y1[:C0] == 7.0).all()) and bool((y2[M:] == 7.0).all())
    ref_img = ops.split_act(y2[:M].contiguous(), lay)
    assert torch.equal(img2[: ref_img.numel()], ref_img)
    parts = image_parts(img1, C0 + M, N)
    assert float(parts[:, :C0].abs().max()) == 0.0                                # the part in front is untouched
    got = (parts[0] + parts[1])[C0:C0 + M, :N]
    assert float((got.cpu() - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-6
    assert float(parts[:, C0:, N].abs().max()) == 0.0                             # the zero column


def test_down_multi_equals_single_launches(cuda):
    """as_down_multi_f32: the towers' down-sampling steps that are ready together as ONE launch (a 3-row depthwise conv of the mel
    tower, a channel-preserving one of the TV tower, an average pool with the residual merge, a 1-D average pool, the stem + pool of a
    first block) -- every output bitwise equal to the single entry points', which test_dwconv_down* / test_avgpool* / test_stem_pool*
    hold against torch."""
    import ctypes
    from artspeech_amd._lib import DownArgs
    g = torch.Generator().manual_seed(17)
    L = ops._lib.lib()
    probs = []

    def lay_pair(widths, H, half):
        lin = Layout(widths, cuda, H=H)
        lout = Layout([(w + 1) // 2 for w in widths], cuda, H=H // 2 if half else H)
        return lin, lout

    def base(kind, X, lin, lout, C):
        a = DownArgs()
        a.kind, a.x, a.ldx = kind, X.data_ptr(), X.stride(0) if X.dim() > 1 else 0
        a.in_off, a.in_w, a.Hin = lin.col_off.data_ptr(), lin.widths.data_ptr(), lin.H
        a.out_off, a.out_w, a.Hout = lout.col_off.data_ptr(), lout.widths.data_ptr(), lout.H
        a.B, a.C, a.max_out, a.n_out = lin.B, C, lout.max_cols, lout.N
        return a

    # 0: dwconv 3x3 s2 (mel tower), image only
    lin, lout = lay_pair([25, 8, 13], 10, True)
    X0 = torch.randn(20, lin.N, generator=g).to(cuda)
    w0, b0 = torch.randn(20, 9, generator=g).to(cuda), torch.randn(20, generator=g).to(cuda)
    a = base(0, X0, lin, lout, 20)
    img0 = ops.new_image(20, lout.N, cuda)
    a.w, a.bias, a.kh, a.lrelu, a.yh = w0.data_ptr(), b0.data_ptr(), 3, 1, img0.data_ptr()
    probs.append(a)
    # 1: dwconv 1x3 s(1,2) (TV tower), fp32 + no image
    lin1, lout1 = lay_pair([199, 66], 10, False)
    X1 = torch.randn(64, lin1.N, generator=g).to(cuda)
    w1, b1 = torch.randn(64, 3, generator=g).to(cuda), torch.randn(64, generator=g).to(cuda)
    y1 = lout1.new(64)
    a = base(0, X1, lin1, lout1, 64)
    a.w, a.bias, a.kh, a.lrelu, a.y, a.ldy = w1.data_ptr(), b1.data_ptr(), 1, 0, y1.data_ptr(), y1.stride(0)
    probs.append(a)
    # 2: avgpool 2x2 + residual merge, fp32 + LeakyReLU image
    r2 = torch.randn(20, lout.N, generator=g).to(cuda)
    y2, img2 = lout.new(20), ops.new_image(20, lout.N, cuda)
    a = base(1, X0, lin, lout, 20)
    a.pool_h, a.res, a.ldr, a.y, a.ldy, a.yh, a.lrelu = 2, r2.data_ptr(), r2.stride(0), y2.data_ptr(), y2.stride(0), img2.data_ptr(), 1
    probs.append(a)
    # 3: 1-D avgpool, image only
    lin3, lout3 = lay_pair([199, 7, 66], 1, False)
    X3 = torch.randn(130, lin3.N, generator=g).to(cuda)
    img3 = ops.new_image(130, lout3.N, cuda)
    a = base(1, X3, lin3, lout3, 130)
    a.pool_h, a.yh = 1, img3.data_ptr()
    probs.append(a)
    arr = (DownArgs * len(probs))(*probs)
    ops.check(L.as_down_multi_f32(arr, len(probs), ops.stream()), "as_down_multi_f32")
    torch.cuda.synchronize()
    assert torch.equal(img0, ops.dwconv_down_image(X0, lin, lout, w0, b0, 3, True))
    assert torch.equal(y1, ops.dwconv_down(X1, lin1, lout1.new(64), lout1, w1, b1, 1, False))
    y2s = lout.new(20)
    img2s = ops.avgpool_down_image(X0, lin, y2s, lout, 2, res=r2, yh_lrelu=True)
    assert torch.equal(y2, y2s) and torch.equal(img2, img2s)
    assert torch.equal(img3, ops.avgpool_down_image(X3, lin3, None, lout3, 1))
    with pytest.raises(ops._lib.HipLibraryError):
        ops.check(L.as_down_multi_f32(arr, 7, ops.stream()), "as_down_multi_f32")


def test_conv_gemm_multi_direct_stems(cuda):
    """as_conv_gemm_multi_f32 on a set of Cin = 1 stem convs (the towers' first convs: the direct kernel): one launch, every output --
    fp32 rows and the LeakyReLU operand image -- bitwise equal to the single launches; a set that mixes direct and tiled problems is
    refused."""
    g = torch.Generator().manual_seed(23)
    specs = [(64, 3, 80, [199, 66, 150]), (64, 3, 10, [199, 7]), (48, 1, 1, [300, 5, 64])]      # (Cout, kh, H, widths): 2-D 3x3 / 1-D k3
    deferred, outs, singles = [], [], []
    for cout, kh, H, widths in specs:
        lay = Layout(widths, cuda, H=H)
        w = torch.randn(cout, 1, kh, 3, generator=g) if kh > 1 else torch.randn(cout, 1, 3, generator=g)
        b = torch.randn(cout, generator=g).to(cuda)
        x = torch.randn(1, lay.N, generator=g).to(cuda)
        taps = taps_2d(3, 3) if kh > 1 else taps_1d(3)
        wt = ops.prep_weight(w, cuda)
        y, yh = lay.new(cout), ops.new_image(cout, lay.N, cuda)
        ops.conv_gemm(wt, x, lay, y, taps, bias=b, yh=yh, yh_lrelu=True, defer=deferred)
        y1, yh1 = lay.new(cout), ops.new_image(cout, lay.N, cuda)
        plan = {}
        ops.conv_gemm(wt, x, lay, y1, taps, bias=b, yh=yh1, yh_lrelu=True, plan_out=plan)
        assert plan["kind"] == 0
        outs.append((y, yh))
        singles.append((y1, yh1))
    ops.conv_gemm_multi(deferred)
    torch.cuda.synchronize()
    for (y, yh), (y1, yh1) in zip(outs, singles):
        assert torch.equal(y, y1) and torch.equal(yh, yh1)
    lay = Layout([50], cuda)
    x = torch.randn(32, 50, generator=g).to(cuda)
    ops.conv_gemm(ops.prep_weight(torch.randn(64, 32, 3, generator=g), cuda), None, lay, lay.new(64), taps_1d(3), xs=ops.split_act(x, lay), K=32,
                  defer=deferred)
    with pytest.raises(ops._lib.HipLibraryError):
        ops.conv_gemm_multi(deferred[-2:])


@pytest.mark.parametrize("C,k,dil,lens", [(32, 3, 1, [700, 3, 241, 1]), (32, 7, 5, [500, 17]), (32, 11, 5, [960, 40, 240]), (32, 11, 3, [239]),
                                          (64, 3, 3, [481, 5]), (64, 7, 1, [300, 300]), (64, 11, 5, [721, 26]), (64, 11, 1, [250])])
def test_respair_equals_two_convs(cuda, monkeypatch, C, k, dil, lens):
    """One residual step of the vocoder's ResBlock1 as one launch (ops.respair) against fp32 torch and against the two conv GEMM launches
    it replaces: ragged utterances (shorter than a halo, one column, exactly one tile, several tiles), every (k, dilation) of the config."""
    g = torch.Generator().manual_seed(C + 13 * k + dil)
    w1 = torch.randn(C, C, k, generator=g) * (0.6 / np.sqrt(C * k))
    w2 = torch.randn(C, C, k, generator=g) * (0.6 / np.sqrt(C * k))
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    xs = [torch.randn(C, L, generator=g) for L in lens]

    def step(x):
        t = F.conv1d(F.leaky_relu(x[None], 0.1), w1, b1, padding=dil * (k // 2), dilation=dil)
        return (F.conv1d(F.leaky_relu(t, 0.1), w2, b2, padding=k // 2) + x[None])[0]

    want = packed([step(x) for x in xs])
    lay = Layout(lens, cuda)
    W1, W2 = ops.prep_weight(w1, cuda), ops.prep_weight(w2, cuda)
    X = packed(xs).to(cuda)
    y = ops.respair(X, lay, W1, b1.to(cuda), W2, b2.to(cuda), k, dil, 0.1)
    torch.cuda.synchronize()
    d = float((y.cpu() - want).abs().max())
    assert d <= 2e-6, d
    # the two launches
    t1 = [(0, dil * (t - k // 2)) for t in range(k)]
    xh = ops.split_act(X, lay, in_act=ops.ACT_LRELU, in_slope=0.1)
    xth = ops.new_image(C, lay.N, cuda)
    ops.conv_gemm(W1, None, lay, None, t1, bias=b1.to(cuda), xs=xh, K=C, yh=xth, yh_lrelu=True, in_slope=0.1)
    y2 = ops.conv_gemm(W2, None, lay, lay.new(C), taps_1d(k), bias=b2.to(cuda), res=X, xs=xth, K=C)
    assert float((y - y2).abs().max()) <= 2e-6
    if True:                                                                 # both workgroup widths (the library picks by LDS footprint)
        for nw in ("4", "8"):
            monkeypatch.setenv("AS_RESPAIR_NW", nw)
            yn = ops.respair(X, lay, W1, b1.to(cuda), W2, b2.to(cuda), k, dil, 0.1)
            assert float((yn.cpu() - want).abs().max()) <= 2e-6, nw
        monkeypatch.delenv("AS_RESPAIR_NW")
    # x given as the phase-major output of the up-sampling conv (interleave_phases folded into the read)
    for u in (2, 3):
        lens_in = [max(1, L // u) for L in lens]
        lay_u = Layout([u * L for L in lens_in], cuda)
        xb = torch.randn(C, generator=g) * 0.2
        Z = torch.randn(u * C, sum(lens_in), generator=g)
        xu = ops.interleave_phases(Z.to(cuda), xb.to(cuda), C, u, sum(lens_in), lay_u.new(C))
        ya = ops.respair(xu, lay_u, W1, b1.to(cuda), W2, b2.to(cuda), k, dil, 0.1)
        yb = ops.respair((Z.to(cuda), xb.to(cuda), u), lay_u, W1, b1.to(cuda), W2, b2.to(cuda), k, dil, 0.1)
        assert torch.equal(ya, yb), u
    # the result as the next conv's operand image instead of fp32
    yi = ops.respair(X, lay, W1, b1.to(cuda), W2, b2.to(cuda), k, dil, 0.1, image_slope=0.1)
    parts = image_parts(yi.cpu(), C, lay.N)                                  # [2][KBx * 16][N + 1]
    act = F.leaky_relu(y.cpu(), 0.1)
    assert float((parts[0, :C, :-1] + parts[1, :C, :-1] - act).abs().max()) <= 1e-6
    assert torch.equal(parts[0, :C, :-1], act.half().float())
    assert (C == parts.shape[1] or float(parts[:, C:, :].abs().max()) == 0.0) and float(parts[:, :, -1].abs().max()) == 0.0
    # the stage's mean folded into the step
    A, B = torch.randn(C, lay.N, generator=g), torch.randn(C, lay.N, generator=g)
    y3 = ops.respair(X, lay, W1, b1.to(cuda), W2, b2.to(cuda), k, dil, 0.1, add=(A.to(cuda), B.to(cuda)))
    assert float((y3.cpu() - ((A + B) + y.cpu()) / 3.0).abs().max()) <= 5e-7


def test_respair_rejects(cuda):
    from artspeech_amd import _lib
    lay = Layout([100], cuda)
    w = ops.prep_weight(torch.randn(32, 32, 3), cuda)
    X = torch.randn(32, 100, device=cuda)
    with pytest.raises(_lib.HipLibraryError):
        ops.respair(X, lay, w, None, w, None, 3, 1, 0.1, Y=X)                  # in place
    w48 = ops.prep_weight(torch.randn(48, 48, 3), cuda)
    with pytest.raises(_lib.HipLibraryError):
        ops.respair(torch.randn(48, 100, device=cuda), lay, w48, None, w48, None, 3, 1, 0.1)
    with pytest.raises(_lib.HipLibraryError):
        ops.respair(X, lay, w, None, w, None, 4, 1, 0.1)                       # even k


def test_conv_walls_beyond_65535_columns(cuda):
    """An utterance longer than 2^16 columns (218 mel frames at the vocoder's last stage): the column descriptors carry 22-bit
    positions, so the conv's zero padding sits at the utterance's own ends and nowhere else."""
    g = torch.Generator().manual_seed(5)
    lens = [70001, 9, 131075]
    w = torch.randn(16, 8, 7, generator=g) / np.sqrt(56)
    xs = [torch.randn(8, L, generator=g) for L in lens]
    want = packed([F.conv1d(x[None], w, padding=9, dilation=3)[0] for x in xs])
    lay = Layout(lens, cuda)
    y = ops.conv_gemm(ops.prep_weight(w, cuda), packed(xs).to(cuda), lay, lay.new(16), [(0, 3 * (t - 3)) for t in range(7)])
    assert float((y.cpu() - want).abs().max()) <= 2e-6


@pytest.mark.parametrize("C,k,lens", [(32, 7, [1500, 2, 1, 64, 700]), (2, 7, [300, 301]), (32, 3, [5])])
def test_conv_post(cuda, C, k, lens):
    """The vocoder's one-row last conv as plain fp32 FMAs (ops.conv_post) against torch."""
    g = torch.Generator().manual_seed(C + k)
    w = torch.randn(1, C, k, generator=g) / np.sqrt(C * k)
    b = torch.randn(1, generator=g) * 0.1
    xs = [torch.randn(C, L, generator=g) for L in lens]
    want = packed([torch.tanh(F.conv1d(F.leaky_relu(x[None], 0.01), w, b, padding=k // 2))[0] for x in xs])
    lay = Layout(lens, cuda)
    y = ops.conv_post(packed(xs).to(cuda), lay, w[0].contiguous().to(cuda), b.to(cuda), 0.01)
    assert float((y.cpu() - want).abs().max()) <= 1e-6


def test_mean3_image(cuda):
    g = torch.Generator().manual_seed(3)
    C, N = 40, 1500
    A, B, Cc = (torch.randn(C, N, generator=g) for _ in range(3))
    xh = ops.mean3_image(A.to(cuda), B.to(cuda), Cc.to(cuda), N, 0.1)
    parts = image_parts(xh.cpu(), C, N)
    want = F.leaky_relu(((A + B) + Cc) / 3.0, 0.1)
    assert float((parts[0, :C, :-1] + parts[1, :C, :-1] - want).abs().max()) <= 1e-6
    assert float(parts[:, C:, :].abs().max()) == 0.0 and float(parts[:, :, -1].abs().max()) == 0.0


@pytest.mark.parametrize("lens", [[200, 31, 1], [32, 63, 2, 97], [333], [2048, 5]])
def test_xl_attention_image_equals_exact(cuda, lens):
    """The Transformer-XL attention of the EMA predictor on the matrix cores (operand images in, f16x3 products) against the exact fp32
    kernel: utterances shorter than / equal to / one past a wave's 31 queries, several key tiles, one frame."""
    g = torch.Generator().manual_seed(sum(lens))
    C, heads = 256, 4
    lay = Layout(lens, cuda)
    N = lay.N
    x = torch.randn(C, N, generator=g)
    wq, wk, wv, wp = (torch.randn(C, C, 1, generator=g) / 16 for _ in range(4))
    bq, bk, bv = (torch.randn(C, generator=g) * 0.1 for _ in range(3))
    u, v = torch.randn(heads, 64, generator=g) * 0.3, torch.randn(heads, 64, generator=g) * 0.3
    pe = torch.randn(C, N, generator=g)
    one = [(0, 0)]
    X, PE = x.to(cuda), pe.to(cuda)
    qkv = ops.conv_gemm(ops.prep_weight(torch.cat([wq, wk, wv], 0), cuda), X, lay, lay.new(3 * C), one, bias=torch.cat([bq, bk, bv]).to(cuda))
    pos = ops.conv_gemm(ops.prep_weight(wp, cuda), PE, lay, lay.new(C), one)
    want = ops.xl_attention(qkv, C, heads, pos, u.to(cuda), v.to(cuda), 1.0 / 16, lay, lay.new(C))
    qh, ph = ops.new_image(4 * C, N, cuda), ops.new_image(C, N, cuda)
    b4 = torch.cat([bq + u.reshape(-1), bq + v.reshape(-1), bk, bv]).to(cuda)
    qkv4 = ops.conv_gemm(ops.prep_weight(torch.cat([wq, wq, wk, wv], 0), cuda), X, lay, lay.new(4 * C), one, bias=b4, yh=qh)
    ops.conv_gemm(ops.prep_weight(wp, cuda), PE, lay, None, one, yh=ph)
    got = ops.xl_attention_image(qkv4, qh, ph, C, heads, 1.0 / 16, lay, torch.full((C, N), float("nan"), device=cuda))
    d = float((got - want).abs().max())
    assert d <= 2e-5, d
    # the result as the out-projection's operand image
    oh = ops.xl_attention_image(qkv4, qh, ph, C, heads, 1.0 / 16, lay, image=True)
    parts = image_parts(oh.cpu(), C, N)
    assert float((parts[0, :C, :-1] + parts[1, :C, :-1] - got.cpu()).abs().max()) <= 1e-6
    assert torch.equal(parts[0, :C, :-1], got.cpu().half().float()) and float(parts[:, :, -1].abs().max()) == 0.0


def test_respair_fuzz(cuda):
    """Random shapes of the fused residual step against fp32 torch: every odd k the library accepts (1 .. 17), dilations up to the halo
    limit, ragged batches around the tile widths (240 / 496 columns) and the halo."""
    rng = np.random.default_rng(20261004)
    g = torch.Generator().manual_seed(99)
    for case in range(14):
        C = int(rng.choice([32, 64]))
        k = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 15, 17]))
        dil = int(rng.integers(1, max(2, min(6, 40 // max(k // 2, 1)) + 1)))
        B = int(rng.integers(1, 5))
        lens = [int(rng.choice([1, 2, 7, 239, 240, 241, 480, 495, 496, 497, int(rng.integers(1, 1300))])) for _ in range(B)]
        w1 = torch.randn(C, C, k, generator=g) * (0.7 / np.sqrt(C * k))
        w2 = torch.randn(C, C, k, generator=g) * (0.7 / np.sqrt(C * k))
        b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
        xs = [torch.randn(C, L, generator=g) for L in lens]
        want = packed([(F.conv1d(F.leaky_relu(F.conv1d(F.leaky_relu(x[None], 0.1), w1, b1, padding=dil * (k // 2), dilation=dil), 0.1),
                                 w2, b2, padding=k // 2) + x[None])[0] for x in xs])
        lay = Layout(lens, cuda)
        y = ops.respair(packed(xs).to(cuda), lay, ops.prep_weight(w1, cuda), b1.to(cuda), ops.prep_weight(w2, cuda), b2.to(cuda), k, dil, 0.1)
        d = float((y.cpu() - want).abs().max())
        assert d <= 3e-6, (case, C, k, dil, lens, d)


@pytest.mark.parametrize("cin,cout,u,lens", [(64, 32, 2, [300, 7, 1]), (128, 64, 3, [100, 33]), (96, 128, 5, [40]), (48, 256, 10, [9, 20])])
@pytest.mark.parametrize("tile", ["", "22", "21", "12", "14", "2"])
def test_conv_gemm_interleaved_store(cuda, monkeypatch, cin, cout, u, lens, tile):
    """ConvGemmArgs.ileave_u: the (phase, channel) rows of a ConvTranspose1d-as-conv stored in time order by the epilogue, against the
    same conv followed by interleave_phases."""
    if tile:
        monkeypatch.setenv("AS_GEMM_TILE", tile)
    g = torch.Generator().manual_seed(cin + cout + u)
    w = torch.randn(u * cout, cin, 3, generator=g) / np.sqrt(3 * cin)
    b = torch.randn(cout, generator=g)
    lay = Layout(lens, cuda)
    X = torch.randn(cin, lay.N, generator=g).to(cuda)
    wt = ops.prep_weight(w, cuda)
    z = ops.conv_gemm(wt, X, lay, lay.new(u * cout), taps_1d(3))
    lay_up = lay.scaled(u)
    want = ops.interleave_phases(z, b.to(cuda), cout, u, lay.N, lay_up.new(cout))
    got = ops.conv_gemm(wt, X, lay, torch.full((cout, lay_up.N), float("nan"), device=cuda), taps_1d(3), bias=b.repeat(u).to(cuda), ileave=u)
    assert float((got - want).abs().max()) <= 1e-6
