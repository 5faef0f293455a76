"""JDCNet on the HIP path: reference-mel frames/s at the headline batch (32 utterances x 200 frames),
synthetic weights; hipGraph replay.  scripts/exp/jdc_prof.py lists the launches."""
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
# a hipGraph of the forward, replayed: the device's time (called eagerly from Python the ~25 launches are host-paced)
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    net.forward_packed(mel, lay)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = net.forward_packed(mel, lay)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
flop = 2.0 * B * T * (80 * (9 * 64 + 9 * 64 * 64) + 40 * (9 * 64 * 128 + 9 * 128 * 128 + 64 * 128) + 20 * (9 * 128 * 192 + 9 * 192 * 192 + 128 * 192)
                      + 10 * (9 * 192 * 256 + 9 * 256 * 256 + 192 * 256) + 512 * 2048 + 256 * 2048 + 512)
print(f"JDCNet B={B} T={T}: {ms:.3f} ms per batch, {B * T / ms * 1e3:,.0f} frames/s, {flop / ms / 1e9:.1f} TFLOP/s (conv + LSTM algorithmic flop "
      f"{flop / B / 1e9:.2f} GFLOP per utterance), finite={bool(torch.isfinite(out).all())}")
