"""Do two kernels of different streams overlap on this chip?  A latency-bound recurrence (64 workgroups) or a bandwidth kernel (AdaIN) on one
stream, the big conv GEMM on another: time of each alone and of both together (20 launches each, wall clock around a synchronise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
layg = Layout([200] * 32, dev)
wg = ops.prep_weight(torch.randn(1024, 1024, 3, generator=g) / 55.0, dev)
xsg = ops.split_act(R(1024, layg.N), layg)
Yg = layg.new(1024)
gemm = lambda: ops.conv_gemm(wg, None, layg, Yg, ops.taps_1d(3), xs=xsg, K=1024)
H = 128
layl = Layout([200] * 32, dev)
jobs = [(R(layl.N, 8 * H) * 0.1, R(2, H, 4 * H) * 0.05, layl.new(2 * H))]
lstm = lambda: ops.bilstm(jobs, layl, H, None)
B, L, C = 32, 200, 1024
laya = Layout([L] * B, dev)
Xa, gb = R(C, laya.N), R(B, 2 * C)
adain = lambda: ops.adain_image(Xa, laya, gb, 1, laya.N, ldgb=2 * C)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def graph_of(fn, n, stream):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=stream):
            for _ in range(n): fn()
    return gr
def timed(graphs):
    for gr, st in graphs:
        with torch.cuda.stream(st): gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        for gr, st in graphs:
            with torch.cuda.stream(st): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3
N = 20
gg = graph_of(gemm, N, sa)
for name, fn in (("lstm H128 (64 workgroups)", lstm), ("adain 1024 ch x 6400", adain), ("gemm (second copy)", gemm)):
    go = graph_of(fn, N, sb)
    a, b, ab = timed([(gg, sa)]), timed([(go, sb)]), timed([(gg, sa), (go, sb)])
    print(f"{name:28s}: gemm alone {a:.3f} ms, it alone {b:.3f} ms, together {ab:.3f} ms  (sum {a + b:.3f}, max {max(a, b):.3f})")
