"""tower down-sampling kernels alone (hipGraph of REP launches): dwconv_down_image / avgpool_down_image / stem_pool at the C3 towers' shapes.
A/B: AS_LIB_PATH=artspeech_amd/lib/exp_head.so python scripts/exp/down_bench.py   against   python scripts/exp/down_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops, _lib
dev = torch.device("cuda:0")
REP = 10


def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3


print("lib:", _lib.LIB_PATH)
for name, C, H, W, kh, half in [("mel b1", 64, 80, 199, 3, True), ("mel b2", 128, 40, 100, 3, True), ("ema b1", 64, 10, 199, 1, False), ("enf0 b1", 128, 1, 199, 1, False)]:
    lin = ops.layout([W] * 32, dev, H=H)
    lout = ops.layout([(W + 1) // 2] * 32, dev, H=H // 2 if half else H)
    X = torch.randn(C, lin.N, device=dev)
    w = torch.randn(C, kh * 3, device=dev); b = torch.randn(C, device=dev)
    t1 = timed(lambda: ops.dwconv_down_image(X, lin, lout, w, b, kh, True))
    t2 = timed(lambda: ops.avgpool_down_image(X, lin, None, lout, 2 if half else 1))
    print(f"{name}: C{C} {H}x{W}: dwconv_down_image {t1:.1f} us, avgpool_down_image {t2:.1f} us")
