"""the Cin = 1 direct conv alone (graph of 10 launches): time per launch for the towers' stems"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for H, W, B, yout in ((80, 199, 32, False), (10, 199, 32, False), (10, 200, 32, False), (10, 199, 32, True), (1, 199, 32, False)):
    lay = Layout([W] * B, dev, H=H)
    w = ops.prep_weight(torch.randn(64, 1, 3, 3, generator=g) / 3, dev)
    b = torch.randn(64, generator=g).to(dev)
    X = torch.randn(1, lay.N, generator=g).to(dev)
    yh = ops.new_image(64, lay.N, dev)
    Y = lay.new(64) if yout else None
    fn = lambda: ops.conv_gemm(w, X, lay, Y, ops.taps_2d(3, 3), bias=b, yh=yh, yh_lrelu=True)
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for _ in range(10): fn()
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 200 * 1e6
    mb = (64 * lay.N * 4 * (2 if yout else 1)) / 1e6
    print(f"H{H} W{W} B{B} N{lay.N} y={yout}: {us:.1f} us per launch, {mb:.0f} MB written, {mb / us / 1e3 * 1e3:.0f} GB/s")
