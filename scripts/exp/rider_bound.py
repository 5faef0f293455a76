"""Round 6: what could RIDERS buy on the decoder phase?  (the AdaIN image writers of one half-batch as extra workgroups
of the OTHER half's conv launch.)  Measured before anything is built, with the library's own launches: the decoder's conv / AdaIN
sequence (models.py:497-517: encode 640 -> 1024, decode.0-1 1216 -> 1024, decode.2 1216 -> 512, decode.3-5 512 -> 512; k = 3; per block
AdaIN -> conv1 -> AdaIN -> conv2 with the learned shortcut folded into conv2's reduction) at 200 mel frames per utterance, as hipGraphs:

  A  one stream, every launch over the whole call (today's chain)
  B  one stream, every launch twice over half the call (what splitting the call into halves costs, nothing hidden)
  C  two streams, no edges: the convs of both halves on one, the AdaINs of both halves on the other -- every AdaIN is free to run beside
     any conv: an UPPER bound of what riders at this call width can hide (riders add the data dependencies back)
  D  C at full width: the bound for a call of TWICE the width split into halves of this width

per call of U utterances; riders at U per call can gain at most A - C, riders at 2 U per call at most A(2 U) / 2 - D(U)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
# (Cout, K of the 3-tap part, K2 of the folded shortcut) and the AdaIN in front of each conv
CONVS = [(1024, 640, 0), (1024, 1024, 640), (1024, 1216, 0), (1024, 1024, 1216), (1024, 1216, 0), (1024, 1024, 1216),
         (512, 1216, 0), (512, 512, 1216)] + [(512, 512, 0), (512, 512, 0)] * 3
ADAIN = [640, 1024, 1216, 1024, 1216, 1024, 1216, 512] + [512, 512] * 3
W = {}
def weight(M, K, K2):
    key = (M, K, K2)
    if key not in W:
        # (the shortcut's K2 channels are one more tap's worth of reduction: priced as K2 / 3 more channels of the 3-tap conv)
        Ke = K + (K2 + 2) // 3
        Ke = (Ke + 15) // 16 * 16
        W[key] = (ops.prep_weight(torch.randn(M, Ke, 3, generator=g) / 55.0, dev), Ke)
    return W[key]

def build(U, L=200):
    lay = Layout([L] * U, dev)
    ops_c, ops_a = [], []
    for (M, K, K2), C in zip(CONVS, ADAIN):
        wt, Ke = weight(M, K, K2)
        xs = ops.split_act(R(Ke, lay.N), lay)
        Y = lay.new(M)
        ops_c.append((lambda wt=wt, xs=xs, Y=Y, Ke=Ke: ops.conv_gemm(wt, None, lay, Y, ops.taps_1d(3), xs=xs, K=Ke)))
        Xa, gb = R(C, lay.N), R(U, 2 * C)
        ops_a.append((lambda Xa=Xa, gb=gb, C=C: ops.adain_image(Xa, lay, gb, 1, lay.N, ldgb=2 * C)))
    return ops_c, ops_a

def graph_of(fns, stream):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for f in fns: f()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=stream):
            for f in fns: f()
    return gr

def timed(graphs, reps=20):
    for _ in range(3):
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
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
res = {}
for U in (32, 64, 128):
    cf, af = build(U)
    ch, ah = build(U // 2)
    inter = lambda c, a: [f for pair in zip(a, c) for f in pair]
    A = timed([(graph_of(inter(cf, af), sa), sa)])
    conv_only = timed([(graph_of(cf, sa), sa)])
    adain_only = timed([(graph_of(af, sb), sb)])
    B = timed([(graph_of([f for a_, c_ in zip(ah, ch) for f in (a_, a_, c_, c_)], sa), sa)])
    C = timed([(graph_of([f for c_ in ch for f in (c_, c_)], sa), sa), (graph_of([f for a_ in ah for f in (a_, a_)], sb), sb)])
    D = timed([(graph_of(cf, sa), sa), (graph_of(af, sb), sb)])
    res[U] = dict(A=A, B=B, C=C, D=D, conv=conv_only, adain=adain_only)
    print(f"{U:4d} utterances per call: A chain {A:.3f} ms (convs alone {conv_only:.3f}, AdaINs alone {adain_only:.3f}) | B halves, one stream {B:.3f} | "
          f"C halves, AdaINs free beside the convs {C:.3f} | D full width, AdaINs free {D:.3f}", flush=True)
for U in (32, 64):
    r, r2 = res[U], res[2 * U]
    print(f"riders at {U:3d} per call: at most {r['A'] - r['C']:+.3f} ms per call = {(r['A'] - r['C']) * 32 / U:+.3f} ms per 32 utterances "
          f"(A - C); at {2 * U} per call in halves of {U}: at most {(r2['A'] / 2 - r['D']) * 32 / U:+.3f} ms per 32 utterances (A({2 * U}) / 2 - D({U}))")
