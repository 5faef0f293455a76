"""One fused residual step of the vocoder (ops.respair) at a stage's shape, as a hipGraph of repetitions: us per launch, the bytes it must
move (x in, y out) and the flop it must do.  C, K, D, N (columns), B (utterances) from the environment; AS_LIB_PATH picks a build."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 32))
for C, k, d in [tuple(int(v) for v in c.split(",")) for c in os.environ.get("CASES", "32,3,1 32,7,3 32,11,5 64,3,1 64,7,3 64,11,3 64,11,5").split()]:
    per = int(os.environ.get("PER", 60000 if C == 32 else 30000))
    lay = ops.layout([per] * B, dev)
    g = torch.Generator().manual_seed(1)
    w1 = ops.prep_weight(torch.randn(C, C, k, generator=g) / (C * k) ** 0.5, dev)
    w2 = ops.prep_weight(torch.randn(C, C, k, generator=g) / (C * k) ** 0.5, dev)
    b = torch.zeros(C, device=dev)
    X = torch.randn(C, lay.N, device=dev)
    Y = lay.new(C)
    for _ in range(2):
        ops.respair(X, lay, w1, b, w2, b, k, d, 0.1, Y=Y)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    REP = 10
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for _ in range(REP):
                ops.respair(X, lay, w1, b, w2, b, k, d, 0.1, Y=Y)
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * REP) * 1e3
    fl = 2 * 2.0 * C * C * k * lay.N
    print(f"C{C} k{k} d{d} N{lay.N}: {us:7.1f} us  {8.0 * C * lay.N / us / 1e6:5.2f} TB/s (x in, y out)  {fl / us / 1e6:6.1f} TFLOP/s")
    try:
        import ctypes
        from artspeech_amd import _lib
        L = ctypes.CDLL(os.environ["AS_LIB_PATH"]) if os.environ.get("AS_LIB_PATH") else None
        if L is not None and hasattr(L, "as_respair_debug_times"):
            buf = (ctypes.c_ulonglong * 8)()
            L.as_respair_debug_times(buf, 1)
            ops.respair(X, lay, w1, b, w2, b, k, d, 0.1, Y=Y); torch.cuda.synchronize()
            L.as_respair_debug_times(buf, 1)
            n = max(buf[6], 1)
            print("   per workgroup (cycles of wave 0): fill %d conv1 %d mid %d conv2 %d epilogue %d total %d  (%d workgroups)" % tuple([buf[i] // n for i in range(6)] + [buf[6]]))
    except Exception as e:
        print("timing:", e)
