"""Per-section s_memtime ticks of the tap-shared kernel (X6_EXP_STAMPS build, AS_LIB_PATH=...exp_x6_STAMPS.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
os.environ["AS_GEMM_KSPLIT"] = "1"; os.environ["AS_GEMM_TILE"] = "223"
for (M, N, K, T, L) in [(512, 6400, 512, 3, 200), (1024, 6400, 1024, 3, 200), (256, 32000, 256, 9, 1000)]:
    lay = ops.layout([L] * (N // L), dev)
    wt = ops.prep_weight(torch.randn(M, K, T) / (K * T) ** 0.5, dev)
    X = lay.new(K); X.copy_(torch.randn(K, lay.N, device=dev))
    Y = lay.new(M)
    for _ in range(3):
        ops.conv_gemm(wt, X, lay, Y, ops.taps_1d(T), bias=torch.randn(M, device=dev))
    torch.cuda.synchronize()
    y = Y.cpu()
    rows, cols = torch.arange(0, M, 128), torch.arange(0, lay.N, 128)
    seg = [y[rows + i][:, cols] for i in range(4)]
    print(f"M{M} N{N} K{K} T{T}: tiles {len(rows) * len(cols)}  per super-iteration: T0 {seg[0].mean():.0f}  T1 {seg[1].mean():.0f}  barrier {seg[2].mean():.0f}  T2 {seg[3].mean():.0f} ticks"
          f"  (T0 min {seg[0].min():.0f} max {seg[0].max():.0f})")
