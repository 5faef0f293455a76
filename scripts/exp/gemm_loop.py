"""the conv GEMM back to back for N seconds (scripts/exp/power_watch.sh samples the chip's power / clock meanwhile); prints its rate"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5
M, N, K, T, L = 1024, 6400, 1024, 3, 200
lay = ops.layout([L] * (N // L), dev)
w = torch.randn(M, K, T) / (K * T) ** 0.5
X = torch.randn(K, lay.N, device=dev)
if os.environ.get("ZERO") == "1":
    w = w * 0 + 1e-30
    X.zero_()
wt = ops.prep_weight(w, dev)
xs = ops.split_act(X, lay)
Y = lay.new(M)
taps = ops.taps_1d(T)
call = lambda: ops.conv_gemm(wt, None, lay, Y, taps, xs=xs, K=K)
call(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    call(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(50): call()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); n += 1000
dt = time.time() - t0
print(f"ZERO={os.environ.get('ZERO','0')}: {dt / n * 1e6:.1f} us per launch, {2.0 * M * N * K * T * n / dt / 1e12:.0f} TFLOP/s (f16x3 terms)")
