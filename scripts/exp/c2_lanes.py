"""C2 (one utterance per request) with n requests in flight through as_lanes: ms per utterance against the number of lanes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from artspeech_amd import models, synth
from artspeech_amd.weights import DEFAULT_STATS, load_distribution
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
net = model.ArtsSpeech
for cfg, args in (("C2", (1, 30, 75, 150)), ("C3", (32, 40, 100, 200))):
    for n in [int(v) for v in os.environ.get("NS", "1,2,3,4,6,8,12,16").split(",")]:
        batches = [bench.make_inputs(dev, *args, seed0=bench.DATA_SEED + 1000 + 100 * i)[1] for i in range(n)]
        firsts = [bench.Runner(net, b).step()["mel"].clone() for b in batches]
        r = bench.bench_native_lanes(net, batches, firsts, 400 if cfg == "C2" else 100, 0)
        print(cfg, "lanes", n, "ms per batch", round(r["ms_per_step"], 3), r["results_bitwise_equal"], flush=True)
