"""Micro-benchmark of the multi-problem conv GEMM launch (as_conv_gemm_multi_f32) on sets of the path's shapes: every set one by one
(as_conv_gemm_f32 per problem) against ONE launch, each form captured into a hipGraph of REP repetitions.
usage: multi_bench.py            env: AS_GEMM_TILE as for the library"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from artspeech_amd import ops, _lib
dev = torch.device("cuda:0")
# M, N, K, T, per-utterance length
SETS = {
    "enc L3 ffn1 + dur blk": [(1024, 2560, 512, 9, 40), (512, 1280, 512, 3, 40)],
    "enc L3 qkv + dur blk": [(1536, 2560, 512, 1, 40), (512, 1280, 512, 3, 40)],
    "enc pre T5 + mel b2c1 + ema b2c1 + dur b2c1": [(512, 3840, 512, 5, 40), (128, 128000, 128, 9, 4000), (128, 32000, 128, 9, 1000), (128, 32000, 128, 9, 1000)],
    "enc L1 ffn1 + mel b3c1 + towers b3c1 + enf0": [(1024, 3840, 512, 9, 40), (256, 32000, 256, 9, 1000), (256, 16000, 256, 9, 500), (256, 16000, 256, 9, 500), (256, 1600, 256, 3, 50)],
    "enc o-proj + mel b4c1 + enf0": [(512, 3840, 512, 1, 40), (512, 8000, 512, 9, 250), (256, 800, 256, 3, 25)],
    "two decoder convs (cannot merge in the path; reference)": [(512, 6400, 512, 3, 200), (512, 6400, 512, 3, 200)],
    "M64 stems: mel + ema + dur": [(64, 509440, 64, 9, 15920), (64, 63680, 64, 9, 1990), (64, 64000, 64, 9, 2000)],
}
REP = 5
L = _lib.lib()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(REP):
                fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3


for name, shapes in SETS.items():
    deferred, flops = [], 0.0
    for (M, N, K, T, Lu) in shapes:
        lay = ops.layout([Lu] * (N // Lu), dev)
        wt = ops.prep_weight(torch.randn(M, K, T) / (K * T) ** 0.5, dev)
        xs = ops.split_act(torch.randn(K, lay.N, device=dev), lay)
        ops.conv_gemm(wt, None, lay, lay.new(M), ops.taps_1d(T), bias=torch.randn(M, device=dev), xs=xs, K=K, defer=deferred)
        flops += 2.0 * M * lay.N * K * T
    n = len(deferred)
    arr = (ops.ConvGemmArgs * n)(*[d[0] for d in deferred])
    each = []
    for i in range(n):
        a = deferred[i][0]
        each.append(timed(lambda: ops.check(L.as_conv_gemm_f32(ctypes.byref(a), ops.stream()), "gemm")))
    one_by_one = timed(lambda: [ops.check(L.as_conv_gemm_f32(ctypes.byref(d[0]), ops.stream()), "gemm") for d in deferred])
    tile = L.as_conv_gemm_multi_tile(arr, n)
    merged = timed(lambda: ops.check(L.as_conv_gemm_multi_f32(arr, n, ops.stream()), "multi"))
    print(f"{name}: alone {' + '.join(f'{t:.1f}' for t in each)} = {sum(each):.1f} us; back to back {one_by_one:.1f} us ({flops / one_by_one / 1e6:.0f} TF/s); "
          f"ONE launch (tile {tile}) {merged:.1f} us ({flops / merged / 1e6:.0f} TF/s)", flush=True)
