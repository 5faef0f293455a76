import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from artspeech_amd import models, synth
from artspeech_amd.weights import DEFAULT_STATS, load_distribution
sys.path.insert(0, "tests")
dev = torch.device("cuda:0")
sd = synth.synth_state_dict(512, 64, seed=3407)
m = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80, n_token=178, n_layer=3, max_conv_dim=512, dropout=0.2),
                       None, stage="second", distribution=load_distribution(DEFAULT_STATS), device=dev)
models.load_checkpoint(m, None, {"net": {"ArtsSpeech": sd}})
net = m.ArtsSpeech
B, N, T = int(os.environ.get("B", 8)), int(os.environ.get("N", 1024)), 200
def raw(t, seed):
    mel, f0, ema = synth.synth_ref_features(t, seed)
    f0r = (f0 * np.float32(DEFAULT_STATS["pitch"][3]) + np.float32(DEFAULT_STATS["pitch"][2])).astype(np.float32)
    emar = (ema * np.asarray(DEFAULT_STATS["EMA"][3], np.float32)[:, None] + np.asarray(DEFAULT_STATS["EMA"][2], np.float32)[:, None]).astype(np.float32)
    return mel, f0r, emar
toks = [synth.synth_tokens(N, 100 + b) for b in range(B)]
feats = [raw(T, 100 + b) for b in range(B)]
texts = torch.from_numpy(np.stack(toks)); mels = torch.from_numpy(np.stack([f[0] for f in feats]))
f0s = torch.from_numpy(np.stack([f[1] for f in feats])); emas = torch.from_numpy(np.stack([f[2] for f in feats]))
forced = [np.ones(N, np.int64)] * B
def run():
    out, aux = net([texts, torch.full((B,), N), mels, torch.full((B,), T)], None, None, step="test", features=(f0s, emas),
                   forced_durations=forced, return_aux=True)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in aux.items() if isinstance(v, torch.Tensor)}
models.CONCURRENT = False
ref = run()
for trial in range(3):
    models.CONCURRENT = True
    got = run()
    print("trial", trial, {k: float((got[k].float() - ref[k].float()).abs().max()) for k in ref})
