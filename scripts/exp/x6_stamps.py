"""Timing experiment (needs the X6_EXP_STAMPS build: scripts/build_exp.sh x6_STAMPS -DX6_EXP_STAMPS, run with
AS_LIB_PATH=.../exp_x6_STAMPS.so): per-workgroup s_memtime segments prologue / k loop / epilogue of the bf16x6 GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
SHAPES = [(512, 6400, 512, 3, 200, "22"), (512, 6400, 512, 3, 200, "21"), (1024, 6400, 1024, 3, 200, "22"), (256, 32000, 256, 9, 1000, "22"),
          (512, 1280, 512, 1, 40, "21")]
os.environ["AS_GEMM_KSPLIT"] = "1"
for (M, N, K, T, L, tile) in SHAPES:
    os.environ["AS_GEMM_TILE"] = tile
    bm = 128
    bn = 128 if tile == "22" else 64
    lay = ops.layout([L] * (N // L), dev)
    wt = ops.prep_weight(torch.randn(M, K, T) / (K * T) ** 0.5, dev)
    X = lay.new(K); X.copy_(torch.randn(K, lay.N, device=dev))
    b = torch.randn(M, device=dev)
    Y = lay.new(M)
    for _ in range(3):
        ops.conv_gemm(wt, X, lay, Y, ops.taps_1d(T), bias=b)
    torch.cuda.synchronize()
    y = Y.cpu()
    rows = torch.arange(0, M, bm)
    cols = torch.arange(0, lay.N, bn)
    seg = [y[rows + i][:, cols] for i in range(4)]
    print(f"M{M} N{N} K{K} T{T} tile{tile}: tiles {len(rows) * len(cols)}  prologue {seg[0].mean():.0f}  loop {seg[1].mean():.0f} "
          f"(min {seg[1].min():.0f} max {seg[1].max():.0f})  epilogue {seg[2].mean():.0f} ticks;  in end-of-tile wait+barrier {seg[3].mean():.0f}")
