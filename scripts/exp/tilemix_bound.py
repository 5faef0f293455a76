"""Round 6: the fallback of VERDICT r5 item 1 -- ONE launch that mixes 128 x 128 and 128 x 64 tiles of one problem, so that the tail round of
the 400 / 800-tile decoder launches (512 workgroup slots) is made of smaller tiles.  Bounded before it is built: the conv over the first u1
utterances with the 128 x 128 tile on one stream and over the remaining ones with the 128 x 64 tile on another, free-running (no edges)
-- what a mixed launch could reach at best -- against the one launch of today.  AS_GEMM_TILE is read per call: set while capturing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
KEEP = []      # (graphs stay alive: a CUDAGraph that is destroyed takes its private pool with it)

def graph_of(fn, stream, n=10):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=stream):
            for _ in range(n): fn()
    return gr

def timed(graphs, reps=10, n=10):
    for gr, st in graphs:
        with torch.cuda.stream(st): gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            for gr, st in graphs:
                with torch.cuda.stream(st): gr.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps / n * 1e6)
    return best

def conv_on(wt, K, M, U, tile, L=200):
    lay = Layout([L] * U, dev)
    xs = ops.split_act(R(K, lay.N), lay)
    Y = lay.new(M)
    def f():
        os.environ["AS_GEMM_TILE"] = str(tile)
        ops.conv_gemm(wt, None, lay, Y, ops.taps_1d(3), xs=xs, K=K)
        os.environ.pop("AS_GEMM_TILE", None)
    KEEP.append(f)                                  # (a replayed graph writes the closure's tensors: they must outlive it)
    return f

for (M, K, U) in ((1024, 1024, 64), (1024, 1024, 32), (512, 512, 64), (512, 512, 32), (1024, 1216, 64)):
    wt = ops.prep_weight(torch.randn(M, K, 3, generator=g) / 55.0, dev)
    whole = timed([(graph_of(conv_on(wt, K, M, U, 22), sa), sa)])
    whole21 = timed([(graph_of(conv_on(wt, K, M, U, 21), sa), sa)])
    best = (1e9, None)
    rows = []
    for u1 in sorted(set(int(U * f) for f in (0.5, 0.5625, 0.625, 0.6875, 0.75, 0.8125, 0.875))):
        g1 = graph_of(conv_on(wt, K, M, u1, 22), sa)
        g2 = graph_of(conv_on(wt, K, M, U - u1, 21), sb)
        KEEP.extend([g1, g2])
        t = timed([(g1, sa), (g2, sb)])
        rows.append(f"{u1}+{U - u1}: {t:.1f}")
        best = min(best, (t, u1))
    print(f"M{M} N{U * 200} K{K} T3: one launch 128x128 {whole:.1f} us, 128x64 {whole21:.1f} | split (first part 128x128 + rest 128x64, two free streams): "
          + ", ".join(rows) + f" | best {best[0]:.1f} us = {best[0] / whole:.3f} of today's", flush=True)
