"""How much does a conv GEMM lose when its WEIGHTS come from HBM (as inside the step: 650 MB of weights and GBs of activations pass between
two uses of a layer's weights -- more than the 256 MB Infinity Cache holds)?  Each launch is bracketed by HIP events (the bracket's own
~4 us is the same in every mode); between launches a 512 MB buffer is rewritten (cold), or only the activations are refreshed (warm).
python scripts/exp/gemm_hbm_cold.py [M,N,K,T,L ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
SHAPES = [(512, 6400, 512, 3, 200), (1024, 6400, 1024, 3, 200), (512, 3840, 512, 5, 40), (512, 1280, 512, 3, 40), (512, 2560, 1024, 1, 40),
          (1536, 3840, 512, 1, 40), (1024, 3840, 512, 9, 40), (256, 19200, 512, 3, 200)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]
REP = 12
scrub = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
warmer = torch.empty(1, device=dev)
for (M, N, K, T, L) in SHAPES:
    lay = ops.layout([L] * (N // L), dev)
    wt = ops.prep_weight(torch.randn(M, K, T) / (K * T) ** 0.5, dev)
    X = torch.randn(K, lay.N, device=dev)
    b = torch.randn(M, device=dev)
    taps = ops.taps_1d(T)
    Y = lay.new(M)
    res = {}
    for mode in ("warm", "cold all", "cold weights only", "cold + weights re-read just before"):
        ts = []
        for r in range(REP):
            xs = None
            if mode != "warm":
                scrub.zero_()                                       # 512 MB written: nothing older survives in the Infinity Cache
            xs = ops.split_act(X, lay)                              # the producer just wrote the activations (as in the step)
            if mode == "cold all":
                scrub.zero_()
            if mode.endswith("just before"):
                warmer = wt.wh.view(torch.int16).to(torch.int32).sum()   # a pass over the weight image: it is in the Infinity Cache again
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv_gemm(wt, None, lay, Y, taps, bias=b, xs=xs, K=K)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        res[mode] = ts[len(ts) // 2]
    print(f"M{M} N{lay.N} K{K} T{T}: " + "  ".join(f"{k}: {v:6.1f} us" for k, v in res.items()), flush=True)
