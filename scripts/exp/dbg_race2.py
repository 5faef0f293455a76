import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from artspeech_amd import models, synth, ops
from artspeech_amd.weights import DEFAULT_STATS, load_distribution
dev = torch.device("cuda:0")
sd = synth.synth_state_dict(512, 64, seed=3407)
m = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80, n_token=178, n_layer=3, max_conv_dim=512, dropout=0.2),
                       None, stage="second", distribution=load_distribution(DEFAULT_STATS), device=dev)
models.load_checkpoint(m, None, {"net": {"ArtsSpeech": sd}})
net = m.ArtsSpeech
B, N = 8, 1024
toks = torch.from_numpy(np.stack([synth.synth_tokens(N, 100 + b) for b in range(B)])).reshape(-1).to(device=dev, dtype=torch.int32)
lay = ops.layout([N] * B, dev)
ref = net.arts_encoder.forward_packed(toks, lay).clone(); torch.cuda.synchronize()
again = net.arts_encoder.forward_packed(toks, lay).clone(); torch.cuda.synchronize()
print("same stream, repeat:", float((again - ref).abs().max()))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    a = net.arts_encoder.forward_packed(toks, lay).clone()
torch.cuda.synchronize()
print("other stream alone:", float((a - ref).abs().max()))
with torch.cuda.stream(s1):
    a = net.arts_encoder.forward_packed(toks, lay)
with torch.cuda.stream(s2):
    b = net.text_encoder.forward_packed(toks, lay)
torch.cuda.synchronize()
print("two encoders concurrently:", float((a - ref).abs().max()))
# hog: a long unrelated kernel stream beside it
x = torch.randn(8192, 8192, device=dev)
with torch.cuda.stream(s2):
    for _ in range(10): y = x @ x
with torch.cuda.stream(s1):
    a = net.arts_encoder.forward_packed(toks, lay)
torch.cuda.synchronize()
print("beside a big torch matmul:", float((a - ref).abs().max()))

# ---- which op diverges first
rec = None
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        out = orig(*a, **k)
        if rec is not None and torch.cuda.current_stream() == s1:
            rec.append((name, tuple(out.shape), out.clone()))
        return out
    setattr(ops, name, f)
for n in ("conv_gemm", "channel_layernorm", "relpos_attention", "embed"):
    wrap(n)
rec = []
with torch.cuda.stream(s1):
    a = net.arts_encoder.forward_packed(toks, lay)
torch.cuda.synchronize()
serial = rec
rec = []
with torch.cuda.stream(s1):
    a = net.arts_encoder.forward_packed(toks, lay)
with torch.cuda.stream(s2):
    b = net.text_encoder.forward_packed(toks, lay)
torch.cuda.synchronize()
conc = rec
rec = None
for i, ((n1, sh, t1), (n2, _, t2)) in enumerate(zip(serial, conc)):
    d = float((t1 - t2).abs().max())
    if d > 0 or i < 3:
        bad = (t1 != t2)
        rows = bad.any(1).nonzero().flatten().tolist()
        cols = bad.any(0).nonzero().flatten().tolist()
        print(i, n1, sh, "maxdiff", d, "bad rows", len(rows), rows[:8], "bad cols", len(cols), cols[:12])
        if d > 0: break
