"""per-launch HIP-event times of one JDCNet forward at the C3 batch (AS_PROF_CSV)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
csv_path = os.environ.setdefault("AS_PROF_CSV", "/tmp/jdc_prof.csv")
if os.path.exists(csv_path): os.remove(csv_path)
import torch
from artspeech_amd import jdc as J, ops, synth, _lib
dev = torch.device("cuda:0")
net = J.JDCNet(device=dev).load_state_dict(J.synth_jdc_state_dict(1, seed=3407))
lay = ops.layout([200] * 32, dev)
mel = lay.new(80); mel.copy_(torch.from_numpy(synth.hash_tensor("jdc/bench", (80, 6400), 1, 1.0)))
for _ in range(3): net.forward_packed(mel, lay)
torch.cuda.synchronize()
L = _lib.lib(); L.as_prof_enable(1)
net.forward_packed(mel, lay); torch.cuda.synchronize()
n = 7
ms, fl, by, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int32 * n)()
L.as_prof_collect(ms, fl, by, cnt, n); L.as_prof_enable(0)
print("per class ms", [round(v, 3) for v in ms], "launches", list(cnt))
for ln in open(csv_path): print(ln.strip()[:120])
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): net.forward_packed(mel, lay)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: host {1e3*(t1-t0)/20:.3f} ms per forward to enqueue, {1e3*(t2-t0)/20:.3f} ms per forward in all")
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    net.forward_packed(mel, lay); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = net.forward_packed(mel, lay)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print(f"graph replay: {1e3*(time.perf_counter()-t0)/20:.3f} ms per forward")
