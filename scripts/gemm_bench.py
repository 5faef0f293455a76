"""Micro-benchmark of the conv GEMM on the path's shapes (tuning aid).  AS_LIB_PATH selects an experiment build.
usage: gemm_bench.py [M,N,K,T,L ...]   env: IMPLS=x6,x6d,x6ds,f32  TILES=,22,21,12,11  KSPLITS=,1,2,4
x6 = activations split in the k loop; x6d = split kernel + pre-split GEMM (what a call costs); x6ds = pre-split GEMM alone"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd import ops, _lib
dev = torch.device("cuda:0")
SHAPES = [  # M, N(total cols), K, taps, per-utt length (1-D)
    (1024, 6400, 1024, 3, 200), (512, 6400, 512, 3, 200), (1024, 6400, 1216, 3, 200), (1024, 1280, 512, 9, 40),
    (512, 1280, 512, 5, 40), (512, 1280, 512, 1, 40), (128, 128000, 128, 3, 4000), (256, 32000, 256, 3, 1000), (1536, 1280, 512, 1, 40),
    (512, 8000, 512, 9, 250), (64, 509440, 64, 9, 15920), (512, 1280, 1024, 1, 40), (256, 6400, 512, 3, 200), (128, 1600, 128, 3, 50),
]
args = [a for a in sys.argv[1:]]
if args:
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in args]
IMPLS = os.environ.get("IMPLS", "x6,x6d,x6ds,f32").split(",")
TILES = os.environ.get("TILES", "").split(",")
KSPLITS = os.environ.get("KSPLITS", "").split(",")
print("lib:", _lib.LIB_PATH)
for (M, N, K, T, L) in SHAPES:
    lay = ops.layout([L] * (N // L), dev)
    w = torch.randn(M, K, T) / (K * T) ** 0.5
    wt = ops.prep_weight(w, dev)
    X = lay.new(K); X.copy_(torch.randn(K, lay.N, device=dev))
    b = torch.randn(M, device=dev)
    taps = ops.taps_1d(T)
    ref = None
    for impl in IMPLS:
        ops.GEMM_IMPL = "f32" if impl == "f32" else "x6"
        os.environ["AS_GEMM_X6D"] = "1" if impl == "x6d" else "0"
        xs = ops.split_act(X, lay) if impl == "x6ds" else None
        for tile in TILES:
            for ks in KSPLITS:
                os.environ.pop("AS_GEMM_TILE", None); os.environ.pop("AS_GEMM_KSPLIT", None)
                if tile: os.environ["AS_GEMM_TILE"] = tile
                if ks: os.environ["AS_GEMM_KSPLIT"] = ks
                Y = lay.new(M)
                for _ in range(3):
                    ops.conv_gemm(wt, X, lay, Y, taps, bias=b, xs=xs)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 20
                e0.record()
                for _ in range(n):
                    ops.conv_gemm(wt, X, lay, Y, taps, bias=b, xs=xs)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
                if ref is None:
                    ref = Y.clone()
                d = float((Y - ref).abs().max())
                print(f"M{M} N{lay.N} K{K} T{T} {impl:4s} tile={tile or 'auto':4s} S={ks or 'auto':4s}: {ms*1e3:8.1f} us  "
                      f"{2.0*M*lay.N*K*T/ms/1e9:6.1f} TF/s  maxdiff vs first {d:.2e}", flush=True)
