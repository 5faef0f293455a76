"""print the order in which a merging serial plan plays the recorded branches of one C3 step out (AS_DEBUG_SCHED=1)"""
import os, sys
os.environ["AS_DEBUG_SCHED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from artspeech_amd import models, synth
from artspeech_amd.weights import DEFAULT_STATS, load_distribution
dev = torch.device("cuda:0")
sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
net = model.ArtsSpeech.replica()
net.rt.set_serial(True)
cfg = os.environ.get("CONFIG", "C3")
_, g = bench.make_inputs(dev, 1, 30, 75, 150, seed0=bench.DATA_SEED + 1000) if cfg == "C2" else (
    bench.make_inputs(dev, 8, 1024, 1024, 200, seed0=bench.DATA_SEED + 1000) if cfg == "C5" else bench.make_inputs(dev))
r = bench.Runner(net, g)
r.step()
torch.cuda.synchronize()
if os.environ.get("AS_PROF_CSV"):
    import ctypes
    from artspeech_amd import _lib
    L = _lib.lib()
    L.as_prof_enable(1)
    r.step()
    n = 7
    ms, fl, by, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int32 * n)()
    L.as_prof_collect(ms, fl, by, cnt, n)
    L.as_prof_enable(0)
    print("per class ms", [round(v, 3) for v in ms], "launches", list(cnt), file=sys.stderr)
print("---- second step", file=sys.stderr)
r.step()
torch.cuda.synchronize()
