"""Round 5: does a kernel of another stream start while a conv GEMM with MORE workgroups than the chip has slots (512) is still being
dispatched?  AdaIN launches on one stream, the conv GEMM M1024 K1024 T3 over 32 k utterances x 200 columns (200 k tiles of 128 x 128 ... ) on
another, each alone and together.  (the 32-utterance GEMM -- 400 workgroups, all resident at once -- is round 3's overlap_probe.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
wg = ops.prep_weight(torch.randn(1024, 1024, 3, generator=g) / 55.0, dev)
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
for k in (1, 2, 4, 8):
    B, L, C = 32 * k, 200, 1024
    lay = Layout([L] * B, dev)
    xs = ops.split_act(R(1024, lay.N), lay)
    Y = lay.new(1024)
    gemm = lambda: ops.conv_gemm(wg, None, lay, Y, ops.taps_1d(3), xs=xs, K=1024)
    Xa, gb = R(C, lay.N), R(B, 2 * C)
    adain = lambda: ops.adain_image(Xa, lay, gb, 1, lay.N, ldgb=2 * C)
    N = max(20 // k, 3)
    gg, ga = graph_of(gemm, N, sa), graph_of(adain, N, sb)
    a, b, ab = timed([(gg, sa)]), timed([(ga, sb)]), timed([(gg, sa), (ga, sb)])
    print(f"{B:4d} utterances ({8 * 50 * k} GEMM workgroups) x {N}: gemm alone {a:.3f} ms, adain alone {b:.3f}, together {ab:.3f}  (sum {a + b:.3f}; hidden {(a + b - ab) / b:.2f} of the adain)")
    del gg, ga
