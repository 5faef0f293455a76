"""JDCNet on the HIP path: reference-mel frames/s at the headline batch (32 utterances x 200 frames),
synthetic weights.  PROFILE=1 adds the library's per-class HIP-event breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd import jdc as J, ops, synth
dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 200))
net = J.JDCNet(device=dev).load_state_dict(J.synth_jdc_state_dict(1, seed=3407))
lay = ops.layout([T] * B, dev)
mel = lay.new(80)
mel.copy_(torch.from_numpy(synth.hash_tensor("jdc/bench", (80, B * T), 1, 1.0)))
for _ in range(3):
    out = net.forward_packed(mel, lay)
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    out = net.forward_packed(mel, lay)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flop = 2.0 * B * T * (80 * (9 * 64 + 9 * 64 * 64) + 40 * (9 * 64 * 128 + 9 * 128 * 128 + 64 * 128) + 20 * (9 * 128 * 192 + 9 * 192 * 192 + 128 * 192)
                      + 10 * (9 * 192 * 256 + 9 * 256 * 256 + 192 * 256) + 512 * 2048 + 256 * 2048 + 512)
print(f"JDCNet B={B} T={T}: {ms:.3f} ms per batch, {B * T / ms * 1e3:,.0f} frames/s, {flop / ms / 1e9:.1f} TFLOP/s (conv + LSTM algorithmic flop "
      f"{flop / B / 1e9:.2f} GFLOP per utterance), finite={bool(torch.isfinite(out).all())}")
