"""Randomised check of the conv GEMM: random shapes / taps / epilogues, every tile (and split-K) against a float64 convolution
computed by torch on the same data, and the operand image a launch writes against as_split_f16x2_f32 of its own fp32 output.
python scripts/exp/gemm_fuzz.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    M = int(rng.choice([1, 4, 10, 16, 31, 32, 33, 48, 64, 80, 96, 128, 129, 200, 256, 300, 512]))
    K = int(rng.choice([1, 3, 12, 16, 17, 32, 33, 48, 64, 65, 100, 128, 200, 256]))
    T = int(rng.choice([1, 3, 5, 7, 9]))
    dil = int(rng.choice([1, 1, 2, 3]))
    B = int(rng.integers(1, 6))
    lens = [int(rng.integers(1, 300)) for _ in range(B)]
    lay = ops.Layout(lens, dev)
    g = torch.Generator().manual_seed(case)
    w = torch.randn(M, K, T, generator=g) / np.sqrt(K * T)
    X = torch.randn(K, lay.N, generator=g).to(dev)
    bias = torch.randn(M, generator=g).to(dev) if rng.random() < 0.7 else None
    res = torch.randn(M, lay.N, generator=g).to(dev) if rng.random() < 0.4 else None
    act = int(rng.choice([0, 0, 1, 2]))
    div = bool(rng.random() < 0.3) and res is not None
    lrelu_in = bool(rng.random() < 0.3)
    taps = [(0, dil * (t - T // 2)) for t in range(T)]
    # float64 reference: per utterance, zero padding at its own ends
    xin = X.double().cpu()
    if lrelu_in:
        xin = torch.where(xin > 0, xin, 0.2 * xin)
    ref = torch.zeros(M, lay.N, dtype=torch.float64)
    o = 0
    for L in lens:
        xu = xin[:, o:o + L]
        for t, (_, dw) in enumerate(taps):
            lo, hi = max(0, -dw), min(L, L - dw)
            if hi > lo:
                ref[:, o + lo:o + hi] += w[:, :, t].double() @ xu[:, lo + dw:hi + dw]
        o += L
    if bias is not None: ref += bias.double().cpu()[:, None]
    if res is not None: ref += res.double().cpu()
    if div: ref /= np.sqrt(2.0)
    if act == 1: ref = ref.clamp(min=0)
    if act == 2: ref = torch.where(ref > 0, ref, 0.2 * ref)
    wt = ops.prep_weight(w, dev)
    scale = float(ref.abs().max()) + 1.0
    for tile in ("", "11", "12", "14", "21", "22", "2"):
        if tile == "2" and M > 32: continue
        if tile in ("21", "22") and M <= 64: continue
        for ks in ("", "3"):
            os.environ.pop("AS_GEMM_TILE", None); os.environ.pop("AS_GEMM_KSPLIT", None)
            if tile: os.environ["AS_GEMM_TILE"] = tile
            if ks: os.environ["AS_GEMM_KSPLIT"] = ks
            if tile == "11" and K <= 32: continue            # (an image of <= 32 channels is never given to that tile: conv_gemm.hip)
            Y = lay.new(M); Y.fill_(float("nan"))
            yh = ops.new_image(M, lay.N, dev)
            ops.conv_gemm(wt, X, lay, Y, taps, bias=bias, res=res, act=act, div_sqrt2=div, in_act=ops.ACT_LRELU if lrelu_in else 0, yh=yh)
            err = float((Y.double().cpu() - ref).abs().max())
            want = ops.split_act(Y, lay)
            rows = 32 if (M <= 32 and (tile in ("", "2"))) else None
            img_ok = torch.equal(yh, want) if rows is None else torch.equal(yh.view(-1)[: yh.numel() // 2], want.view(-1)[: want.numel() // 2])
            if not (err <= 2e-5 * scale) or not img_ok:
                bad += 1
                print(f"case {case}: M{M} K{K} T{T} dil{dil} lens{lens} bias={bias is not None} res={res is not None} act={act} div={div} "
                      f"lrelu_in={lrelu_in} tile={tile or 'auto'} ksplit={ks or 'auto'}: err {err:.2e} (scale {scale:.1f}) image {'ok' if img_ok else 'DIFFERS'}", flush=True)
os.environ.pop("AS_GEMM_TILE", None); os.environ.pop("AS_GEMM_KSPLIT", None)
print(f"{n_cases} cases: {bad} failures")
sys.exit(1 if bad else 0)
