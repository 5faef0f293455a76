"""Micro-benchmark of the conv GEMM on the path's shapes (tuning aid).  AS_LIB_PATH selects an experiment build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd import ops, _lib
dev = torch.device("cuda:0")
SHAPES = [  # M, N(total cols), K, taps, per-utt length (1-D)
    (1024, 6400, 1024, 3, 200), (512, 6400, 512, 3, 200), (1024, 6400, 1216, 3, 200), (1024, 1280, 512, 9, 40),
    (512, 1280, 512, 5, 40), (512, 1280, 512, 1, 40), (128, 128000, 128, 3, 4000), (256, 32000, 256, 3, 1000), (1536, 1280, 512, 1, 40),
]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]
print("lib:", _lib.LIB_PATH)
for (M, N, K, T, L) in SHAPES:
    lay = ops.layout([L] * (N // L), dev)
    wt = ops.prep_weight(torch.randn(M, K, T)).to(dev)
    X = lay.new(K); X.copy_(torch.randn(K, lay.N, device=dev))
    Y = lay.new(M)
    b = torch.randn(M, device=dev)
    taps = ops.taps_1d(T)
    for _ in range(3):
        ops.conv_gemm(wt, X, lay, Y, taps, bias=b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.conv_gemm(wt, X, lay, Y, taps, bias=b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"M{M} N{lay.N} K{K} T{T}: {ms*1e3:8.1f} us  {2.0*M*lay.N*K*T/ms/1e9:6.1f} TF/s")
