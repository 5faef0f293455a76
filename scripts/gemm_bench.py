"""Micro-benchmark of the conv GEMM on the path's shapes (tuning aid).  AS_LIB_PATH selects an experiment build.
usage: gemm_bench.py [M,N,K,T,L ...]   env: PIPES=13,12,23 (KT NS)  TILES=,22,21,12,11  KSPLITS=,1,2,4  SPLIT=0,1
Every configuration is captured into a hipGraph of REP launches and replayed, so the time is the device's, not the host's.
SPLIT=1 includes the standalone split of the fp32 activations (what a call without an operand image costs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd import ops, _lib
dev = torch.device("cuda:0")
SHAPES = [  # M, N(total cols), K, taps, per-utt length (1-D)
    (1024, 6400, 1024, 3, 200), (512, 6400, 512, 3, 200), (1024, 6400, 1216, 3, 200), (1024, 2560, 512, 9, 40),
    (512, 2560, 512, 5, 40), (512, 2560, 512, 1, 40), (1536, 2560, 512, 1, 40), (512, 2560, 1024, 1, 40),
    (512, 19200, 512, 3, 200), (256, 19200, 512, 3, 200), (128, 19200, 256, 3, 200),
    (128, 128000, 128, 9, 4000), (256, 32000, 256, 9, 1000), (512, 8000, 512, 9, 250), (64, 509440, 64, 9, 15920),
    (256, 6400, 512, 3, 200), (128, 1600, 128, 3, 50),
]
args = [a for a in sys.argv[1:]]
if args:
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in args]
PIPES = os.environ.get("PIPES", "13").split(",")
TILES = os.environ.get("TILES", "").split(",")
KSPLITS = os.environ.get("KSPLITS", "").split(",")
SPLITS = os.environ.get("SPLIT", "0").split(",")
REP = 10
print("lib:", _lib.LIB_PATH)
for (M, N, K, T, L) in SHAPES:
    lay = ops.layout([L] * (N // L), dev)
    ZERO = os.environ.get("ZERO", "0")              # 1: all-zero operands (the chip holds a higher clock: is the loop power-limited?); 2: zero activations only
    w = torch.randn(M, K, T) / (K * T) ** 0.5
    if ZERO == "1": w = w * 0 + 1e-3 * (torch.arange(M * K * T).reshape(M, K, T) % 2 == 3)
    wt = ops.prep_weight(w + (1e-30 if ZERO == "1" else 0), dev)
    X = torch.randn(K, lay.N, device=dev)
    if ZERO in ("1", "2"): X.zero_()
    b = torch.randn(M, device=dev)
    taps = ops.taps_1d(T)
    xs = ops.split_act(X, lay)
    ref = None
    for pipe in PIPES:
        os.environ["AS_H3_KT"], os.environ["AS_H3_NS"] = pipe[0], pipe[1]
        for tile in TILES:
            for ks in KSPLITS:
                for sp in SPLITS:
                    os.environ.pop("AS_GEMM_TILE", None); os.environ.pop("AS_GEMM_KSPLIT", None)
                    if tile: os.environ["AS_GEMM_TILE"] = tile
                    if ks: os.environ["AS_GEMM_KSPLIT"] = ks
                    Y = lay.new(M)
                    call = (lambda: ops.conv_gemm(wt, X, lay, Y, taps, bias=b)) if sp == "1" else \
                           (lambda: ops.conv_gemm(wt, None, lay, Y, taps, bias=b, xs=xs, K=K))
                    call()
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    s = torch.cuda.Stream()
                    with torch.cuda.stream(s):
                        call()
                        torch.cuda.synchronize()
                        with torch.cuda.graph(graph, stream=s):
                            for _ in range(REP):
                                call()
                    graph.replay()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        graph.replay()
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / (5 * REP)
                    if ref is None:
                        ref = Y.clone()
                    d = float((Y - ref).abs().max())
                    print(f"M{M} N{lay.N} K{K} T{T} kt{pipe[0]}ns{pipe[1]} tile={tile or 'auto':4s} S={ks or 'auto':4s} split={sp}: {ms*1e3:8.1f} us  "
                          f"{2.0*M*lay.N*K*T/ms/1e9:6.1f} TF/s  maxdiff {d:.1e}", flush=True)
