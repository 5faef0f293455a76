"""hipGraph replay cost per dependent kernel node (tuning aid): a chain of n tiny launches of the library, captured and replayed;
also the same chain launched eagerly from Python and a chain of torch element-wise kernels for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
lay = ops.layout([40] * 32, dev)
X = lay.new(64); X.normal_()
g = torch.ones(64, device=dev); b = torch.zeros(64, device=dev)
def chain_lib(n):
    y = X
    for _ in range(n):
        y = ops.channel_layernorm(y, lay.N, g, b, lay.new(64))
    return y
def chain_torch(n):
    y = X
    for _ in range(n):
        y = y * 1.0001
    return y
for name, fn in (("library kernel (channel_layernorm, ~5 us)", chain_lib), ("torch element-wise", chain_torch)):
    for n in (50, 200):
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            fn(n); torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=s):
                keep = fn(n)
        torch.cuda.current_stream().wait_stream(s)
        for _ in range(3): gr.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn(n)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
        print(f"{name:45s} n={n:4d}: graph replay {tg / n * 1e6:6.2f} us per node, eager {te / n * 1e6:6.2f} us per launch")
