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
k = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(k):
    out = net.forward_packed(f0, n, mel, lay)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / k
print(f"EMA_Predictor B={B} T={T}: {ms:.3f} ms per batch, {B * T / ms * 1e3:,.0f} frames/s, finite={bool(torch.isfinite(out).all())}")
