"""EMA_Predictor on the HIP path: frames/s at the headline batch (32 utterances x 200 frames), synthetic weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd import ema as E, ops, synth
dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 200))
net = E.EMA_Predictor(device=dev).load_state_dict(E.synth_ema_state_dict(seed=3407))
lay = ops.layout([T] * B, dev)
mel, f0, n = lay.new(80), lay.new(1), lay.new(1)
mel.copy_(torch.from_numpy(synth.hash_tensor("ema/bench", (80, B * T), 1, 1.0)))
f0.copy_(torch.from_numpy(synth.hash_tensor("ema/bench/f0", (1, B * T), 1, 1.0)))
n.copy_(torch.from_numpy(synth.hash_tensor("ema/bench/n", (1, B * T), 1, 1.0)))
for _ in range(3):
    out = net.forward_packed(f0, n, mel, lay)
torch.cuda.synchronize()
import time
k = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
t0 = time.perf_counter()
for _ in range(k):
    out = net.forward_packed(f0, n, mel, lay)
host_ms = (time.perf_counter() - t0) / k * 1e3
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / k
# the same launches as a hipGraph: what the device needs when the host is not in the way (eager, ~70 launches per batch are issued by
# Python at 10-60 us each depending on the box's host load)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    net.forward_packed(f0, n, mel, lay); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out_g = net.forward_packed(f0, n, mel, lay)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(k): g.replay()
torch.cuda.synchronize()
gms = (time.perf_counter() - t0) / k * 1e3
print(f"EMA_Predictor B={B} T={T}: {gms:.3f} ms per batch by graph replay ({B * T / gms * 1e3:,.0f} frames/s), {ms:.3f} eager "
      f"(host issues a batch in {host_ms:.3f} ms), finite={bool(torch.isfinite(out).all())}, graph == eager: {bool(torch.equal(out, out_g))}")
