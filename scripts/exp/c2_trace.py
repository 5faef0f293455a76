"""C2 (batch 1, 30 tokens -> 150 frames) run eagerly N times, for rocprofv3 --kernel-trace --stats: which kernels make up the 2.5 ms"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from artspeech_amd import models, synth
from artspeech_amd.weights import DEFAULT_STATS, load_distribution
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
net = model.ArtsSpeech
n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 1
host, g = bench.make_inputs(dev, n_utt, 30, 75, 150, seed0=bench.DATA_SEED + 1000)
r = bench.Runner(net, g)
net.rt.set_serial(True)
for _ in range(22):
    r.step()
torch.cuda.synchronize()
net.rt.set_serial(False)
run = r.capture()
el = r.timed(run, 50, 5)
print(f"C2 x{n_utt}: graph replay {el / 50 * 1e3:.3f} ms per step")
