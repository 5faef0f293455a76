"""experiment: the C3 step replayed from ONE hipGraph on one stream, against TWO independent graphs (two plans, two workspaces) replayed
alternately on two streams -- consecutive batches overlap, a kernel's tail round is filled by the other batch's kernels"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from artspeech_amd import models, synth
from artspeech_amd.weights import DEFAULT_STATS, load_distribution

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
nets = []
NF = int(os.environ.get('NF', '2'))
for i in range(NF):
    model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
    models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
    nets.append(model.ArtsSpeech)
host, g = bench.make_inputs(dev)
runners = [bench.Runner(n, g) for n in nets]
runs = [r.capture() for r in runners]
streams = [torch.cuda.Stream() for _ in range(NF)]
K = 100

def timed(fn):
    for _ in range(10):
        for j in range(NF): fn(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

one = timed(lambda i: runs[0]())
def two(i):
    with torch.cuda.stream(streams[i % NF]):
        runs[i % NF]()
both = timed(two)
print(f"one graph, one stream: {one:.3f} ms per step; {NF} graphs on {NF} streams: {both:.3f} ms per step")
