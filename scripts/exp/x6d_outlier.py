"""Per-call timing of the pre-split GEMM path (looks for sporadic slow calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
os.environ["AS_GEMM_X6D"] = "1"
for rep in range(3):
    for (M, N, K, T, L) in [(1024, 1280, 512, 9, 40), (512, 1280, 512, 5, 40), (512, 6400, 512, 3, 200), (512, 1280, 512, 1, 40)]:
        lay = ops.layout([L] * (N // L), dev)
        wt = ops.prep_weight(torch.randn(M, K, T) / (K * T) ** 0.5, dev)
        X = lay.new(K); X.copy_(torch.randn(K, lay.N, device=dev))
        Y = lay.new(M)
        ev, host = [], []
        for i in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            ops.conv_gemm(wt, X, lay, Y, ops.taps_1d(T))
            e1.record()
            host.append((time.perf_counter() - t0) * 1e6)
            torch.cuda.synchronize()
            ev.append(e0.elapsed_time(e1) * 1e3)
        print(f"rep{rep} M{M} N{N} K{K} T{T}: event us min {min(ev):.1f} med {sorted(ev)[20]:.1f} max {max(ev):.1f} (call {ev.index(max(ev))});  host us med {sorted(host)[20]:.1f} max {max(host):.1f} (call {host.index(max(host))})", flush=True)
