"""Does a conv GEMM inside the step lose time to COLD weights?  Ten launches in a hipGraph with the same weights / with ten different
weight tensors (and a 64 MB scrub of L2 in between, optionally):  python scripts/exp/gemm_cold.py [M,N,K,T,L ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
SHAPES = [(512, 6400, 512, 3, 200), (1024, 6400, 1024, 3, 200), (512, 3840, 512, 5, 40), (512, 1280, 512, 3, 40), (512, 2560, 1024, 1, 40)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]
REP = 10
for (M, N, K, T, L) in SHAPES:
    lay = ops.layout([L] * (N // L), dev)
    wts = [ops.prep_weight(torch.randn(M, K, T) / (K * T) ** 0.5, dev) for _ in range(REP)]
    Xs = [torch.randn(K, lay.N, device=dev) for _ in range(REP)]
    xss = [ops.split_act(X, lay) for X in Xs]
    b = torch.randn(M, device=dev)
    taps = ops.taps_1d(T)
    Y = lay.new(M)
    scrub = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for mode in ("same w, same x", "10 w, same x", "10 w, 10 x", "10 w, 10 x, scrub"):
        def body():
            for r in range(REP):
                wi = 0 if mode.startswith("same") else r
                xi = r if "10 x" in mode else 0
                if "scrub" in mode:
                    scrub.zero_()
                ops.conv_gemm(wts[wi], None, lay, Y, taps, bias=b, xs=xss[xi], K=K)
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                body()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (5 * REP) * 1e3
        print(f"M{M} N{lay.N} K{K} T{T}  {mode:22s} {us:8.1f} us per launch" + ("  (incl. the scrub)" if "scrub" in mode else ""), flush=True)
    # the scrub alone
    def sb():
        for r in range(REP): scrub.zero_()
    sb(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sb(); e1.record(); torch.cuda.synchronize()
    print(f"   scrub alone {e0.elapsed_time(e1) / REP * 1e3:8.1f} us")
