"""Thin torch-tensor wrappers over the C ABI (include/artspeech_hip.h).  PyTorch is used only for
device memory and the current HIP stream; every computation below is a HIP kernel of this repo.
No fallbacks: a missing library or a failing call raises."""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, ptr, stream

AS_MAX_TAPS = _lib.AS_MAX_TAPS
META_MAX_H, META_MAX_W = 1023, 4194303      # include/artspeech_hip.h AS_META_MAX_H / AS_META_MAX_W
KTILE = 16                      # the GEMM k-tile (BK in csrc/conv_gemm.hip): weights are zero-padded to it
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_ABS, ACT_SWISH = 0, 1, 2, 3, 4, 5


ConvGemmArgs = _lib.ConvGemmArgs


def _ld(t):
    """row stride (in elements) of a 2-D row-major tensor / row-slice view."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise _lib.HipLibraryError("expected a 2-D tensor with contiguous rows")
    return t.stride(0)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.HipLibraryError("expected a tensor in GPU memory (the HIP path has no CPU fallback)")
    # launches go to the CURRENT device's current stream (_lib.stream): memory of another GPU would be used from the wrong stream
    if t.device.index != torch.cuda.current_device():
        raise _lib.HipLibraryError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                                   "run the call under `with torch.cuda.device(tensor.device)` (models.py / mas.py / pipeline.py do)")
    return t.data_ptr()


def _settle(device):
    """Layouts are cached and shared by branches running on different HIP streams: what one branch creates lazily on
    its stream (offset tables, the column descriptors) must be complete before another stream can pick it up.  Created
    once per geometry, so a host-side wait is cheap; inside a graph capture nothing new is created (warm-up did)."""
    if isinstance(device, torch.device) and device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(device).synchronize()


class Layout:
    """Packed-frames geometry: B utterances, utterance b is an H x widths[b] image (H = 1: a sequence)."""

    def __init__(self, widths, device, H=1):
        self.widths_host = [int(w) for w in widths]
        self.B = len(self.widths_host)
        self.H = int(H)
        if self.H > META_MAX_H or any(w > META_MAX_W for w in self.widths_host):
            raise ValueError(f"an utterance of more than {META_MAX_W} columns (or {META_MAX_H} rows) does not fit the column descriptors "
                             "(include/artspeech_hip.h AS_META_PACK): split it")
        off = [0]
        for w in self.widths_host:
            off.append(off[-1] + self.H * w)
        self.off_host = off
        self.N = off[-1]
        self.max_w = max(self.widths_host) if self.widths_host else 0
        self.max_cols = self.H * self.max_w
        self.device = device
        self.widths = torch.tensor(self.widths_host, dtype=torch.int32, device=device)
        self.col_off = torch.tensor(off, dtype=torch.int32, device=device)
        self._meta = None
        _settle(device)

    @property
    def meta(self):
        if self._meta is None:
            m = torch.empty(max(self.N, 1), dtype=torch.int64, device=self.device)
            check(_lib.lib().as_make_meta(_p(self.widths), _p(self.col_off), self.B, self.H, self.N, _p(m), stream()),
                  "as_make_meta")
            _settle(self.device)
            self._meta = m
        return self._meta

    def new(self, C):
        """[C][N] fp32 activation"""
        n = max(self.N, 1)
        return torch.empty((C, n), dtype=torch.float32, device=self.device)

    def scaled(self, k):
        return layout([w * k for w in self.widths_host], self.device, self.H)

    def halved(self, h_too):
        """geometry after a 2x down-sampling (W -> ceil(W/2); H -> H/2 when h_too)."""
        H = self.H // 2 if h_too else self.H
        return layout([(w + 1) // 2 for w in self.widths_host], self.device, H)

    def valid_conv(self, K, stride):
        H = (self.H - K) // stride + 1
        return layout([(w - K) // stride + 1 for w in self.widths_host], self.device, H)


_LAYOUTS = {}


def layout(widths, device, H=1):
    """Cached Layout: geometry objects own small device arrays (offsets, descriptors); steady-state
    batches of the same shape must not re-upload them every call."""
    key = (tuple(int(w) for w in widths), int(H), str(device))
    lay = _LAYOUTS.get(key)
    if lay is None:
        if len(_LAYOUTS) > 4096:
            _LAYOUTS.clear()
        lay = _LAYOUTS[key] = Layout(widths, device, H)
    return lay


# arithmetic of the conv GEMM: "h3" = fp16 matrix cores, two-way split operands, three products, fp32-accurate (default);
# "h1" = plain fp16 operands (the h parts only, one product): the 16-bit-operand mode BASELINE.md names for config C2.
GEMM_IMPL = os.environ.get("AS_GEMM_IMPL", "h3")


def kbx(K):
    """k-blocks of a split image: ceil(K / 16) rounded up to a multiple of 4."""
    return (((K + 15) >> 4) + 3) & ~3


class GemmWeight:
    """A conv / linear weight prepared for the GEMM: `wh` = the split fp16 image [G][T][KBx][4][M][8] (int16 bit patterns) on
    the device, `scale` = the power of two it was multiplied by, `w32` = the fp32 [T][Kp][M] image (Cin = 1 only: the
    direct kernel reads it), `shape` = (T, Kp, M) of one weight set, `G` = weight sets stacked for a grouped launch."""

    def __init__(self, wh, scale, shape, G, w32=None):
        self.wh, self.scale, self.shape, self.G, self.w32 = wh, scale, shape, G, w32


def split_f16x2_weight(w4):
    """w4 fp32 [G][Cout][Cin][T] (CPU) -> (image int16 [G][T][KBx][4][Cout][8], scale).  The same arithmetic as the library's
    as_prep_weight_f16x2_host (tests/test_abi_cpu.py compares them bit for bit): scale = the power of two that puts max |w| in
    [2^13, 2^14); h = fp16(w * scale), l = fp16(w * scale - h); plane p*2 + kh of k-block kb holds part p of k = 16 kb + 8 kh + 0..7."""
    G, M, K, T = w4.shape
    mx = float(w4.abs().max())
    scale = 1.0
    if mx > 0:
        import math
        _, e = math.frexp(mx)
        scale = math.ldexp(1.0, 14 - e)
    ws = w4.float() * scale
    h = ws.half()
    l = (ws - h.float()).half()
    kx = kbx(K) * 16
    parts = torch.stack([h, l], 0)                                          # [2][G][M][K][T]
    if kx != K:
        parts = torch.cat([parts, parts.new_zeros(2, G, M, kx - K, T)], dim=3)
    img = parts.reshape(2, G, M, kx // 16, 2, 8, T).permute(1, 6, 3, 0, 4, 2, 5)   # [G][T][kb][p][kh][M][8]
    return img.contiguous().view(torch.int16).reshape(G, T, kx // 16, 4, M, 8), scale


def prep_weight(w, device=None, stack=None, sc=None):
    """conv / linear weight [Cout, Cin, *kernel] -> GemmWeight.  stack: further weights of the same shape (a grouped launch:
    weight set g serves the columns [g * group_cols, (g+1) * group_cols)).  sc: the weight(s) [Cout, Cin2] of a 1x1 conv on a second
    operand (conv_gemm(..., x2s=, K2=): a block's learned shortcut summed by the same launch), one per weight set."""
    ws = [w] + list(stack or [])
    cout, cin = w.shape[0], w.shape[1]
    w4 = torch.stack([x.reshape(cout, cin, -1).float() for x in ws], 0).cpu()
    T = w4.shape[3]
    kp = (cin + KTILE - 1) // KTILE * KTILE
    if sc is not None:
        scs = list(sc) if isinstance(sc, (list, tuple)) else [sc]
        w2 = torch.stack([x.reshape(cout, -1).float() for x in scs], 0).cpu().contiguous()      # [G][Cout][Cin2]
        G, cin2 = len(ws), w2.shape[2]
        assert w2.shape[0] == G
        L = _lib.lib()
        nbytes = L.as_prep_weight_f16x2_sc_bytes(G, cout, cin, T, cin2)
        img = torch.empty(nbytes // 2, dtype=torch.int16)
        scale = ctypes.c_float()
        w4c = w4.contiguous()
        check(L.as_prep_weight_f16x2_sc_host(w4c.data_ptr(), w2.data_ptr(), G, cout, cin, T, cin2, img.data_ptr(), ctypes.byref(scale)),
              "as_prep_weight_f16x2_sc_host")
        return GemmWeight(img.to(device) if device is not None else img, scale.value, (T, kp, cout), G, None)
    img, scale = split_f16x2_weight(w4)
    w32 = None
    if cin == 1 and len(ws) == 1:
        w32 = torch.cat([w4[0].permute(2, 1, 0), w4.new_zeros(T, kp - cin, cout)], dim=1).contiguous()   # [T][Kp][M]
    if device is not None:
        img = img.to(device)
        w32 = w32.to(device) if w32 is not None else None
    return GemmWeight(img, scale, (T, kp, cout), len(ws), w32)


def strided_source(lay_in, lay_out, stride, device):
    """(src_col int32 [N_out], src_meta int64 [N_out]) of a valid conv with taps (a, d) >= 0 from lay_in to lay_out: output (ho, wo)
    of utterance b reads input position (ho * stride + a, wo * stride + d); descriptor = AS_META_PACK(h, w, H, W) of the INPUT."""
    cols, metas = [], []
    for b in range(lay_out.B):
        Wi, Wo = lay_in.widths_host[b], lay_out.widths_host[b]
        for ho in range(lay_out.H):
            for wo in range(Wo):
                h, w = ho * stride, wo * stride
                cols.append(lay_in.off_host[b] + h * Wi + w)
                metas.append(h | (lay_in.H << 10) | (w << 20) | (Wi << 42))               # AS_META_PACK
    return torch.tensor(cols, dtype=torch.int32, device=device), torch.tensor(metas, dtype=torch.int64, device=device)


def taps_1d(k):
    return [(0, t - k // 2) for t in range(k)]


def taps_2d(kh, kw):
    return [(a - kh // 2, d - kw // 2) for a in range(kh) for d in range(kw)]


def new_image(K, N, device):
    """uninitialised split image for K channels x N columns"""
    return torch.empty(max(_lib.lib().as_split_f16x2_bytes(K, N) // 2, 8), dtype=torch.int16, device=device)


def split_act(X, lay, in_act=0, in_slope=0.2):
    """X [K][*] fp32 -> the GEMM's split activation image (int16 [KBx][4][N+1][8], LeakyReLU applied first when
    in_act = ACT_LRELU): pass it as conv_gemm(..., xs=) to every conv that reads the same activations."""
    K, N = X.shape[0], lay.N
    xs = new_image(K, N, X.device)
    check(_lib.lib().as_split_f16x2_f32(_p(X), _ld(X), K, N, in_act, in_slope, _p(xs), stream()), "as_split_f16x2_f32")
    return xs


def adain_split(X, gb, lay, lrelu=True):
    """AdaIN1d + LeakyReLU of X [C][N], stored only as the split operand image of the conv that follows (conv_gemm(Wt, None,
    ..., xs=, K=C))."""
    C = X.shape[0]
    xs = new_image(C, lay.N, X.device)
    check(_lib.lib().as_adain_split_f32(_p(X), _ld(X), C, _p(gb), _ld(gb), _p(lay.col_off), lay.B, lay.N, int(lrelu), _p(xs), stream()),
          "as_adain_split_f32")
    return xs


def adain_image(X, lay, gb, gb_sc, N_out, ldgb=1, gb_off=None, src_off=None, pool_w=None, pool_b=None, x_up=None):
    """as_adain_image_f32: AdaIN1d + LeakyReLU(0.2) (+ the fused x2 up-sampler) of X [C][*] as an operand image over N_out columns;
    gamma(u, c) = gb[gb_off[u] + c * gb_sc] (gb_off None: u * ldgb), utterance u reads X at src_off[u] (None: its own columns)."""
    C = X.shape[0]
    xs = new_image(C, N_out, X.device)
    a = _lib.AdainArgs()
    a.x, a.ldx, a.C = _p(X), _ld(X), C
    a.gb, a.gb_off, a.ldgb, a.gb_sc = _p(gb), _p(gb_off), ldgb, gb_sc
    a.col_off, a.src_off, a.U, a.N, a.lrelu, a.yh = _p(lay.col_off), _p(src_off), lay.B, N_out, 1, _p(xs)
    a.pool_w, a.pool_b, a.x_up, a.ld_up = _p(pool_w), _p(pool_b), _p(x_up), (_ld(x_up) if x_up is not None else 0)
    check(_lib.lib().as_adain_image_f32(ctypes.byref(a), stream()), "as_adain_image_f32")
    return xs


def rows_image(x):
    """x [B][K] -> the operand image of its transpose [K][B] (as_rows_image_f32)"""
    B, K = x.shape
    xh = new_image(K, B, x.device)
    check(_lib.lib().as_rows_image_f32(_p(x), _ld(x), K, B, _p(xh), stream()), "as_rows_image_f32")
    return xh


def project_cols(X, N, w, bias, Y):
    """Y[m][j] = bias[m] + sum_k w[m][k] X[k][j], M <= 16 (as_project_cols_f32)"""
    M, K = w.shape
    check(_lib.lib().as_project_cols_f32(_p(X), _ld(X), K, N, _p(w), _p(bias), M, _p(Y), _ld(Y), stream()), "as_project_cols_f32")
    return Y


def conv_gemm(Wt, X, lay, Y, taps, bias=None, res=None, act=0, div_sqrt2=False, in_act=0, transpose_out=False,
              use_meta=True, in_slope=0.2, act_slope=0.2, xs=None, K=None, group_cols=0, yh=None, yh_lrelu=False, n_prod=None, plan_out=None,
              x2s=None, K2=0, src_col=None, src_meta=None, N_in=0, defer=None, ileave=0):
    """Y = epi(sum_t Wt[t]^T X shifted by tap t).  defer: a list -- the call is not launched but appended to it (conv_gemm_multi
    launches the whole list as ONE kernel: as_conv_gemm_multi_f32).  Wt: prep_weight(...); X [K][*] fp32 or None with xs= (the split image of
    X: split_act / adain_split / channel_layernorm_split / another conv's yh=) and K=; Y [M][*] (or [N][*] transposed) or None
    when only yh (the output as the next conv's split image, new_image(M, N)) is wanted.  group_cols: Wt holds Wt.G weight
    sets (and bias [G][M]); columns [g * group_cols, (g+1) * group_cols) use set g.  x2s / K2: the split image of a second operand
    whose 1x1 conv (weights: prep_weight(..., sc=)) is summed into the same accumulators (needs xs=).  src_col / src_meta / N_in: a
    strided or valid conv -- `lay` is the OUTPUT layout, xs an image over N_in input columns, output column j reads input column
    src_col[j] + dh * W_in + dw and src_meta[j] (strided_source) describes that input position.  ileave = u: the M = u C rows are
    (phase, channel) and Y is [C][u N] with Y[m][u j + r] = row r C + m at column j (ConvGemmArgs.ileave_u; bias per row)."""
    T, Kp, M = Wt.shape
    if X is None:
        if xs is None or K is None:
            raise ValueError("conv_gemm: X may be omitted only with xs= and K=")
    else:
        K = X.shape[0]
    if Kp % KTILE or not (Kp - KTILE < K <= Kp):
        raise ValueError(f"conv_gemm: weight rows {Kp} do not match input channels {K} (use ops.prep_weight)")
    if (Wt.G > 1) != (group_cols > 0):
        raise ValueError("conv_gemm: stacked weights need group_cols (and only they)")
    a = ConvGemmArgs()
    a.Kp = Kp
    a.Wh, a.W, a.X, a.Xh, a.Y, a.Yh, a.bias, a.res = _p(Wt.wh), _p(Wt.w32), _p(X), _p(xs), _p(Y), _p(yh), _p(bias), _p(res)
    a.acc_scale = 1.0 / Wt.scale
    a.n_groups, a.group_cols = Wt.G, group_cols
    a.meta = _p(lay.meta) if (use_meta and not (T == 1 and taps[0] == (0, 0))) else None
    a.M, a.N, a.K, a.T = M, lay.N, K, T
    a.ldx, a.ldy = (_ld(X) if X is not None else lay.N), (_ld(Y) if Y is not None else lay.N)
    a.ldr = _ld(res) if res is not None else 0
    a.act, a.div_sqrt2, a.in_act, a.transpose_out = act, int(div_sqrt2), in_act, int(transpose_out)
    a.yh_lrelu = int(yh_lrelu)
    a.Xh2, a.K2 = _p(x2s), K2
    if src_col is not None:
        a.src_col, a.N_in, a.meta = _p(src_col), N_in, _p(src_meta)
    a.n_prod = n_prod if n_prod is not None else (1 if GEMM_IMPL == "h1" else 3)
    a.ileave_u = int(ileave)
    a.in_slope, a.act_slope = in_slope, act_slope          # used as given (the acoustic path's LeakyReLU slope is 0.2)
    assert len(taps) == T
    for i, (dh, dw) in enumerate(taps):
        a.dh[i], a.dw[i] = dh, dw
    L = _lib.lib()
    nbytes = (L.as_conv_gemm_multi_workspace_bytes if defer is not None else L.as_conv_gemm_workspace_bytes)(ctypes.byref(a))
    if nbytes:                                       # split-K partial slabs, the split image of X (caller-owned scratch)
        ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=(Y if Y is not None else yh).device)
        a.ws, a.ws_bytes = ws.data_ptr(), nbytes
    if plan_out is not None:                         # (tests: which kernel ran)
        k, t, sl = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        check(L.as_conv_gemm_plan(ctypes.byref(a), ctypes.byref(k), ctypes.byref(t), ctypes.byref(sl)), "as_conv_gemm_plan")
        plan_out.update(kind=k.value, tile=t.value, slices=sl.value)
    if defer is not None:
        defer.append((a, [Wt, X, xs, Y, yh, bias, res, x2s, src_col, src_meta, lay, (ws if nbytes else None)]))
        return Y if Y is not None else yh
    check(L.as_conv_gemm_f32(ctypes.byref(a), stream()), "as_conv_gemm_f32")
    return Y if Y is not None else yh


def conv_gemm_multi(deferred, tile_out=None):
    """the conv_gemm(..., defer=list) calls collected in `deferred` (independent problems) as one launch"""
    n = len(deferred)
    arr = (ConvGemmArgs * n)(*[d[0] for d in deferred])
    L = _lib.lib()
    if tile_out is not None:
        tile_out.append(L.as_conv_gemm_multi_tile(arr, n))
    check(L.as_conv_gemm_multi_f32(arr, n, stream()), "as_conv_gemm_multi_f32")


def conv_gemm_multi_post(deferred, posts, lns=None):
    """as_conv_gemm_multi_post_f32: the deferred convs as one launch; posts[i] = None or (gb, gb_sc, lay, image[, gb_off]) -- the AdaIN1d +
    LeakyReLU that reads conv i's result, written as `image` by the same call (by the launch's reduction kernel when it is K-sliced and no
    utterance is wider than 256 columns); lns[i] = None or (gamma, beta, image, relu[, gamma2, beta2, n_split]) -- the channel LayerNorm
    (eps 1e-4) that does."""
    n = len(deferred)
    arr = (ConvGemmArgs * n)(*[d[0] for d in deferred])
    pa = (_lib.AdainArgs * n)()
    la = (_lib.LnArgs * n)()
    mw = (ctypes.c_int32 * n)()
    for i, q in enumerate(lns or []):
        if q is None:
            continue
        la[i].gamma, la[i].beta, la[i].yh, la[i].relu, la[i].eps = _p(q[0]), _p(q[1]), _p(q[2]), int(q[3]), 1e-4
        if len(q) > 4:
            la[i].gamma2, la[i].beta2, la[i].n_split = _p(q[4]), _p(q[5]), int(q[6])
    for i, q in enumerate(posts or [None] * n):
        if q is None:
            continue
        gb, gb_sc, lay, img = q[:4]
        gb_off = q[4] if len(q) > 4 else None
        pa[i].gb, pa[i].gb_off, pa[i].ldgb, pa[i].gb_sc = _p(gb), _p(gb_off), 1, gb_sc
        pa[i].col_off, pa[i].U, pa[i].lrelu, pa[i].yh = _p(lay.col_off), lay.B, 1, _p(img)
        mw[i] = q[5] if len(q) > 5 else (max(lay.widths_host) if hasattr(lay, "widths_host") else int(lay.widths.max()))   # (q[5]: tests)
    check(_lib.lib().as_conv_gemm_multi_post_f32(arr, pa, mw, la, n, stream()), "as_conv_gemm_multi_post_f32")


def embed(tokens_i32, emb, scale, Y, group2=None, n_cols=None):
    """group2 = (emb2, n_split): token columns >= n_split look their rows up in the second table.  n_cols > tokens: the token list is
    read twice (columns [tokens, n_split) are filler, column n_split + j is token j again)."""
    V, C = emb.shape
    emb2, n_split = group2 if group2 is not None else (None, 0)
    n_tok = tokens_i32.numel()
    check(_lib.lib().as_embed_groups_f32(_p(tokens_i32), n_tok, _p(emb), _p(emb2), n_split, C, n_tok if n_cols is None else n_cols, V, scale,
                                         _p(Y), _ld(Y), stream()), "as_embed_groups_f32")
    return Y


def channel_layernorm(X, N, gamma, beta, Y, relu=False, eps=1e-4, group2=None):
    """group2 = (gamma2, beta2, n_split): columns >= n_split use the second affine pair."""
    g2, b2, n_split = group2 if group2 is not None else (None, None, 0)
    check(_lib.lib().as_channel_layernorm_groups_f32(_p(X), _ld(X), X.shape[0], N, _p(gamma), _p(beta), _p(g2), _p(b2), n_split, eps,
                                                     int(relu), _p(Y), _ld(Y), stream()), "as_channel_layernorm_groups_f32")
    return Y


def adain(X, gb, lay, Y, lrelu=True, pool_w=None, pool_b=None, x_up=None):
    C = X.shape[0]
    check(_lib.lib().as_adain_f32(_p(X), _ld(X), C, _p(gb), _ld(gb), _p(lay.col_off), lay.B, _p(Y), _ld(Y), int(lrelu),
                                  _p(pool_w), _p(pool_b), _p(x_up), _ld(x_up) if x_up is not None else 0, stream()),
          "as_adain_f32")
    return Y


def linear_rows(x, w, bias, y=None):
    B, K = x.shape
    M = w.shape[0]
    if y is None:
        y = torch.empty((B, M), dtype=torch.float32, device=x.device)
    check(_lib.lib().as_linear_rows_f32(_p(x), _ld(x), _p(w), _p(bias), B, M, K, _p(y), _ld(y), stream()),
          "as_linear_rows_f32")
    return y


def durations(dur_f32, forced, tok_lay, max_frames):
    dev = tok_lay.device
    dur_i = torch.empty(max(tok_lay.N, 1), dtype=torch.int32, device=dev)
    frame_off = torch.empty(tok_lay.B + 1, dtype=torch.int32, device=dev)
    tof = torch.empty(max(max_frames, 1), dtype=torch.int32, device=dev) if max_frames > 0 else None
    check(_lib.lib().as_durations_f32(_p(dur_f32), _p(forced), _p(tok_lay.col_off), tok_lay.B, _p(dur_i), _p(frame_off),
                                      _p(tof), max_frames, stream()), "as_durations_f32")
    return dur_i, frame_off, tof


def expand(X, tok_of_frame, n_frames, repeat, Y):
    check(_lib.lib().as_expand_f32(_p(X), _ld(X), X.shape[0], _p(tok_of_frame), n_frames, repeat, _p(Y), _ld(Y), stream()),
          "as_expand_f32")
    return Y


def ref_features(mel, f0_raw, ema_raw, N, stats24, feat):
    check(_lib.lib().as_ref_features_f32(_p(mel), _ld(mel), mel.shape[0], _p(f0_raw), _p(ema_raw), _ld(ema_raw), N,
                                         _p(stats24), _p(feat), _ld(feat), stream()), "as_ref_features_f32")
    return feat


def crop(src, src_lay, start, dst, dst_lay):
    check(_lib.lib().as_crop_f32(_p(src), _ld(src), _p(src_lay.col_off), start, _p(dst), _ld(dst), _p(dst_lay.col_off),
                                 dst_lay.B, src.shape[0], dst_lay.max_cols, stream()), "as_crop_f32")
    return dst


def bn_lrelu_maxpool_rows(X, tok_lay, H, k, scale, shift, slope, Y, to_channels=False):
    """BatchNorm(eval) -> LeakyReLU -> max over k consecutive image rows; X [C][H * frames] images of tok_lay's utterances."""
    check(_lib.lib().as_bn_lrelu_maxpool_rows_f32(_p(X), _ld(X), _p(tok_lay.col_off), tok_lay.B, X.shape[0], H, k, _p(scale), _p(shift),
                                                  slope, _p(Y), _ld(Y), int(to_channels), tok_lay.N, stream()),
          "as_bn_lrelu_maxpool_rows_f32")
    return Y


def rows_to_images(src, src_lay, start, H, img_lay):
    """rows [H][sum L] (a row-slice view is fine) -> [1][sum H*L] images in `img_lay` (same utterance widths, H rows)."""
    dst = img_lay.new(1)
    check(_lib.lib().as_rows_to_images_f32(_p(src), _ld(src), _p(src_lay.col_off), start, H, _p(dst), _p(img_lay.col_off),
                                           img_lay.B, src_lay.max_cols, stream()), "as_rows_to_images_f32")
    return dst


def interleave_phases(Z, bias, C, u, n_in, Y):
    check(_lib.lib().as_interleave_phases_f32(_p(Z), _ld(Z), _p(bias), C, u, n_in, _p(Y), _ld(Y), stream()),
          "as_interleave_phases_f32")
    return Y


def mean3(A, B, C3, n, Y):
    check(_lib.lib().as_mean3_f32(_p(A), _p(B), _p(C3), _ld(A), A.shape[0], n, _p(Y), _ld(Y), stream()), "as_mean3_f32")
    return Y


def conv_post(X, lay, w, bias, in_slope, tanh_out=True):
    """y [1][N] = tanh(conv1d(LeakyReLU(X [C][N], in_slope), w fp32 [C][k]) + bias): the vocoder's last conv (vocoder.py:111-113)."""
    Y = lay.new(1)
    check(_lib.lib().as_conv_post_f32(_p(X), _ld(X), X.shape[0], lay.N, _p(w), _p(bias), w.shape[1], float(in_slope), int(tanh_out),
                                      _p(lay.meta), _p(Y), stream()), "as_conv_post_f32")
    return Y


def respair(X, lay, w1, b1, w2, b2, k, dil, slope, Y=None, add=None, image_slope=None):
    """One residual step of ResBlock1 (Vocoder/vocoder.py:35-42) as one launch: Y = X + conv2(lrelu(conv1(lrelu(X)))), conv1 with
    dilation `dil`, both with k taps; X [C][N] fp32 with C = 32 or 64, w1 / w2 = prep_weight of the [C][C][k] weights.
    add = (A, B): Y = ((A + B) + Y) / 3 (the mean of a stage's three stacks).  Y must not be X.
    image_slope: return LeakyReLU(Y, image_slope) as a split operand image (conv_gemm(..., xs=)) INSTEAD of the fp32 Y.
    X may be (Z, bias, u): the phase-major output Z [u C][N / u] of the ConvTranspose1d-as-conv before interleave_phases, read in place."""
    a = _lib.ResPairArgs()
    if isinstance(X, tuple):
        X, xb, u = X
        C = xb.shape[0]
        a.x_u, a.x_bias = int(u), _p(xb)
    else:
        C = X.shape[0]
    if image_slope is not None:
        yh = new_image(C, lay.N, X.device)
        a.yh, a.yh_slope = _p(yh), float(image_slope)
    else:
        if Y is None:
            Y = lay.new(C)
        a.y, a.ldy = _p(Y), _ld(Y)
    a.x, a.ldx = _p(X), _ld(X)
    a.w1, a.w2, a.b1, a.b2 = _p(w1.wh), _p(w2.wh), _p(b1), _p(b2)
    a.scale1, a.scale2 = 1.0 / w1.scale, 1.0 / w2.scale
    a.C, a.N, a.k, a.dil, a.slope = C, lay.N, int(k), int(dil), float(slope)
    a.col_off, a.B, a.max_w = _p(lay.col_off), lay.B, lay.max_cols
    if add is not None:
        a.add1, a.add2, a.ld_add, a.out_div = _p(add[0]), _p(add[1]), _ld(add[0]), 3.0
    check(_lib.lib().as_respair_f32(ctypes.byref(a), stream()), "as_respair_f32")
    return Y if image_slope is None else yh


def mean3_image(A, B, C3, n, slope):
    """LeakyReLU((A + B + C3) / 3, slope) as a split operand image (conv_gemm(..., xs=))"""
    xh = new_image(A.shape[0], n, A.device)
    check(_lib.lib().as_mean3_image_f32(_p(A), _p(B), _p(C3), _ld(A), A.shape[0], n, float(slope), _p(xh), stream()), "as_mean3_image_f32")
    return xh


def dwconv_down(X, lin, Y, lout, w, bias, kh, lrelu):
    check(_lib.lib().as_dwconv_down_f32(_p(X), _ld(X), _p(lin.col_off), _p(lin.widths), lin.H, _p(Y), _ld(Y),
                                        _p(lout.col_off), _p(lout.widths), lout.H, _p(w), _p(bias), kh, lin.B,
                                        X.shape[0], lout.max_cols, int(lrelu), stream()), "as_dwconv_down_f32")
    return Y


def avgpool_down(X, lin, Y, lout, pool_h, res=None):
    check(_lib.lib().as_avgpool_down_f32(_p(X), _ld(X), _p(lin.col_off), _p(lin.widths), lin.H, _p(Y), _ld(Y),
                                         _p(lout.col_off), _p(lout.widths), lout.H, pool_h, _p(res),
                                         _ld(res) if res is not None else 0, lin.B, X.shape[0], lout.max_cols, stream()),
          "as_avgpool_down_f32")
    return Y


def dwconv_down_image(X, lin, lout, w, bias, kh, lrelu):
    """dwconv_down written only as the consumer conv's operand image (as_dwconv_down_image_f32)"""
    yh = new_image(X.shape[0], lout.N, X.device)
    check(_lib.lib().as_dwconv_down_image_f32(_p(X), _ld(X), _p(lin.col_off), _p(lin.widths), lin.H, _p(lout.col_off), _p(lout.widths), lout.H,
                                              _p(w), _p(bias), kh, lin.B, X.shape[0], lout.max_cols, int(lrelu), _p(yh), lout.N, stream()),
          "as_dwconv_down_image_f32")
    return yh


def avgpool_down_image(X, lin, Y, lout, pool_h, res=None, yh_lrelu=False):
    """avgpool_down with the result (also) as an operand image; Y may be None (as_avgpool_down_image_f32)"""
    yh = new_image(X.shape[0], lout.N, X.device)
    check(_lib.lib().as_avgpool_down_image_f32(_p(X), _ld(X), _p(lin.col_off), _p(lin.widths), lin.H, _p(Y), _ld(Y) if Y is not None else 0,
                                               _p(lout.col_off), _p(lout.widths), lout.H, pool_h, _p(res), _ld(res) if res is not None else 0,
                                               lin.B, X.shape[0], lout.max_cols, _p(yh), lout.N, int(yh_lrelu), stream()),
          "as_avgpool_down_image_f32")
    return yh


def stem_pool_image(x1, lin, lout, pool_h, Wt, bias, kh):
    """avgpool_down(stem conv(x1)) of a one-channel input x1 [N_in] as the operand image over lout (as_stem_pool_image_f32); Wt =
    prep_weight of the Cin = 1 stem ([C][1][kh][3])"""
    C = Wt.shape[2]
    yh = new_image(C, lout.N, x1.device)
    check(_lib.lib().as_stem_pool_image_f32(_p(x1), _p(lin.col_off), _p(lin.widths), lin.H, _p(lout.col_off), _p(lout.widths), lout.H, pool_h,
                                            _p(Wt.w32), Wt.shape[1], _p(bias), kh, lin.B, C, lout.max_cols, _p(yh), lout.N, stream()),
          "as_stem_pool_image_f32")
    return yh


def im2col_valid_image(X, lin, lout, K, stride, lrelu):
    yh = new_image(X.shape[0] * K * K, lout.N, X.device)
    check(_lib.lib().as_im2col_valid_image_f32(_p(X), _ld(X), _p(lin.col_off), _p(lin.widths), _p(lout.col_off), _p(lout.widths), K, stride,
                                               int(lrelu), lin.B, X.shape[0], _p(yh), stream()), "as_im2col_valid_image_f32")
    return yh


def im2col_valid(X, lin, col, lout, K, stride, lrelu):
    check(_lib.lib().as_im2col_valid_f32(_p(X), _ld(X), _p(lin.col_off), _p(lin.widths), lin.H, _p(col), _ld(col),
                                         _p(lout.col_off), _p(lout.widths), lout.H, K, stride, int(lrelu), lin.B,
                                         X.shape[0], lout.max_cols, stream()), "as_im2col_valid_f32")
    return col


def mean_pool(X, lay, lrelu, y=None):
    C = X.shape[0]
    if y is None:
        y = torch.empty((lay.B, C), dtype=torch.float32, device=X.device)
    check(_lib.lib().as_mean_pool_f32(_p(X), _ld(X), _p(lay.col_off), lay.B, C, int(lrelu), _p(y), _ld(y), stream()),
          "as_mean_pool_f32")
    return y


def channel_layernorm_split(X, lay, gamma, beta, relu=False, eps=1e-4, group2=None):
    """channel LayerNorm (+ReLU) of X [C][N] stored only as the pre-split operand image of the conv that follows
    (conv_gemm(Wt, None, ..., xs=, K=C))."""
    L = _lib.lib()
    C = X.shape[0]
    g2, b2, n_split = group2 if group2 is not None else (None, None, 0)
    xs = new_image(C, lay.N, X.device)
    check(L.as_channel_layernorm_split_f32(_p(X), _ld(X), C, lay.N, _p(gamma), _p(beta), _p(g2), _p(b2), n_split, eps, int(relu), _p(xs),
                                           stream()), "as_channel_layernorm_split_f32")
    return xs


def relpos_attention(qkv, C, heads, window, ek, ev, lay, out, group2=None):
    """group2 = (ek2, ev2, b_split): utterances >= b_split use the second pair of relative-position tables."""
    ek2, ev2, b_split = group2 if group2 is not None else (None, None, 0)
    check(_lib.lib().as_relpos_attention_groups_f32(_p(qkv), _ld(qkv), C, heads, window, _p(ek), _p(ev), _p(ek2), _p(ev2), b_split,
                                                    _p(lay.col_off), lay.B, lay.max_w, _p(out), _ld(out), stream()),
          "as_relpos_attention_groups_f32")
    return out


def relpos_attention_image(qkv, qkv_h, C, heads, window, ek, ev, lay, out=None, out_h=None, group2=None):
    """the attention fed by the q/k/v GEMM's operand image (qkv_h = conv_gemm(..., yh=...) over the same lay.N columns); writes fp32 `out`
    and / or `out_h`, the o-projection's operand image"""
    ek2, ev2, b_split = group2 if group2 is not None else (None, None, 0)
    check(_lib.lib().as_relpos_attention_image_f32(_p(qkv), _ld(qkv), _p(qkv_h), lay.N, C, heads, window, _p(ek), _p(ev), _p(ek2), _p(ev2), b_split,
                                                   _p(lay.col_off), lay.B, lay.max_w, _p(out), _ld(out) if out is not None else 0, _p(out_h),
                                                   stream()), "as_relpos_attention_image_f32")
    return out if out is not None else out_h


def xl_attention(qkv, C, heads, pos, u_bias, v_bias, inv_scale, lay, out):
    check(_lib.lib().as_xl_attention_f32(_p(qkv), _ld(qkv), C, heads, _p(pos), _ld(pos), _p(u_bias), _p(v_bias), inv_scale,
                                         _p(lay.col_off), lay.B, lay.max_w, _p(out), _ld(out), stream()), "as_xl_attention_f32")
    return out


def xl_attention_image(qkv4, qkv4_h, pos_h, C, heads, inv_scale, lay, out=None, image=False):
    """as_xl_attention_image_f32: qkv4 fp32 [4C][N] = rows q + u, q + v, k, v and its operand image, pos_h = the image of pos [C][N].
    image=True: the result as the operand image of the GEMM that follows (conv_gemm(..., xs=, K=C)) instead of fp32 `out`."""
    oh = new_image(C, lay.N, qkv4.device) if image else None
    check(_lib.lib().as_xl_attention_image_f32(_p(qkv4), _ld(qkv4), _p(qkv4_h), _p(pos_h), lay.N, C, heads, inv_scale, _p(lay.col_off), lay.B,
                                               lay.max_w, _p(out), _ld(out) if out is not None else 0, _p(oh), stream()),
          "as_xl_attention_image_f32")
    return oh if image else out


def glu_dwconv_bn_swish(A, C, w, scale, shift, lay, Y):
    check(_lib.lib().as_glu_dwconv_bn_swish_f32(_p(A), _ld(A), C, _p(w), w.shape[1], _p(scale), _p(shift), _p(lay.col_off), lay.B,
                                                _p(Y), _ld(Y), stream()), "as_glu_dwconv_bn_swish_f32")
    return Y


def lstm_step0(gx, H, N, out):
    check(_lib.lib().as_lstm_step0_f32(_p(gx), _ld(gx), H, N, _p(out), _ld(out), stream()), "as_lstm_step0_f32")
    return out


def bilstm(jobs, lay, H, xchg=None):
    """jobs: list of (gx_tm [N][8H], whh_t [2][H][4H], out [2H][N]) -- independent LSTMs sharing one launch.
    xchg: bilstm_exchange_buffer(...) -> H = 256 recurrences run split over clusters of four workgroups."""
    arr = (_lib.BiLstmJob * len(jobs))()
    for i, (gx_tm, whh_t, out) in enumerate(jobs):
        arr[i].gx_tm, arr[i].whh_t, arr[i].out = _p(gx_tm), _p(whh_t), _p(out)
        arr[i].ldg, arr[i].ldo = _ld(gx_tm), _ld(out)
    if xchg is not None:
        check(_lib.lib().as_bilstm_cluster_f32(arr, len(jobs), _p(lay.col_off), lay.B, H, lay.max_w, _p(xchg), xchg.numel() * xchg.element_size(),
                                               stream()), "as_bilstm_cluster_f32")
    else:
        check(_lib.lib().as_bilstm_f32(arr, len(jobs), _p(lay.col_off), lay.B, H, stream()), "as_bilstm_f32")
    return [j[2] for j in jobs]


def bilstm_exchange_buffer(n_jobs, B, device):
    """the zero-filled exchange buffer as_bilstm_cluster_f32 wants (allocate once, reuse for every launch)"""
    return torch.zeros(_lib.lib().as_bilstm_cluster_bytes(n_jobs, B) // 8, dtype=torch.int64, device=device)
