import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
B, N, C = 8, 1024, 512
lay = ops.layout([N] * B, dev)
g = torch.Generator(device="cpu").manual_seed(1)
qkv = lay.new(3 * C); qkv.copy_(torch.randn(3 * C, lay.N, generator=g))
ek = torch.randn(9, 128, generator=g).to(dev) * 0.1; ev = torch.randn(9, 128, generator=g).to(dev) * 0.1
ref = ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, lay.new(C)).clone(); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
w = ops.prep_weight(torch.randn(1024, 512, 9) / 68, dev)
X = lay.new(512); X.copy_(torch.randn(512, lay.N))
def other_op(other):
    if other == "attention":
        return ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, lay.new(C))
    if other.startswith("gemm"):
        ops.GEMM_IMPL = "f32" if other.endswith("f") else "x6"
        os.environ["AS_GEMM_TILE"] = other[4:6]
        return ops.conv_gemm(w, X, lay, lay.new(1024), ops.taps_1d(9))
    return ops.channel_layernorm(X, lay.N, torch.ones(512, device=dev), torch.zeros(512, device=dev), lay.new(512))
for other in ("attention", "gemm22", "gemm21", "gemm22f", "ln"):
    bref = other_op(other).clone(); torch.cuda.synchronize()
    worst = 0.0; worst_b = 0.0
    for trial in range(5):
        with torch.cuda.stream(s1):
            a = ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, lay.new(C))
        with torch.cuda.stream(s2):
            b = other_op(other)
        torch.cuda.synchronize()
        worst = max(worst, float((a - ref).abs().max())); worst_b = max(worst_b, float((b - bref).abs().max()))
    print("attention beside", other, "max diff vs alone:", worst, " other op diff vs alone:", worst_b)

# ---- is it the input, the output, or the computation?
ops.GEMM_IMPL = "x6"; os.environ["AS_GEMM_TILE"] = "22"
qkv0 = qkv.clone(); torch.cuda.synchronize()
with torch.cuda.stream(s1):
    out = lay.new(C)
    a = ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, out)
with torch.cuda.stream(s2):
    y = lay.new(1024)
    b = ops.conv_gemm(w, X, lay, y, ops.taps_1d(9))
torch.cuda.synchronize()
print("qkv changed:", float((qkv - qkv0).abs().max()), " out ptr", hex(out.data_ptr()), "size", out.numel() * 4, " y ptr", hex(y.data_ptr()), "size", y.numel() * 4,
      " qkv ptr", hex(qkv.data_ptr()), " X ptr", hex(X.data_ptr()))
d = (a - ref).abs()
bad = d > 1e-4
print("bad elements", int(bad.sum()), "rows", bad.any(1).nonzero().flatten().tolist()[:10], "cols", bad.any(0).nonzero().flatten().tolist()[:16])
a2 = ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, lay.new(C)); torch.cuda.synchronize()
print("attention alone afterwards:", float((a2 - ref).abs().max()))
# values at bad places: are they GEMM outputs?
idx = bad.nonzero()[:5]
for r, c in idx.tolist():
    print("bad at", r, c, "got", float(a[r, c]), "want", float(ref[r, c]))
