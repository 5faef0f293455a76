import os, sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from artspeech_amd import ops
from artspeech_amd.ops import Layout, taps_1d
cuda = torch.device("cuda:0")
cin, cout, k, lens = 96, 80, 5, [40, 41]
g = torch.Generator().manual_seed(cin * 7 + cout + k)
w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
b = torch.randn(cout, generator=g)
xs = [torch.randn(cin, L, generator=g) for L in lens]
res = [torch.randn(cout, L, generator=g) for L in lens]
want = torch.cat([(F.conv1d(x[None], w, b, padding=k // 2)[0] + r) / np.sqrt(2) for x, r in zip(xs, res)], 1)
lay = Layout(lens, cuda)
wt = ops.prep_weight(w, cuda)
X = lay.new(cin); X.copy_(torch.cat(xs, 1))
for ks in ["1", "2", ""]:
    if ks: os.environ["AS_GEMM_KSPLIT"] = ks
    else: os.environ.pop("AS_GEMM_KSPLIT", None)
    y = ops.conv_gemm(wt, X, lay, lay.new(cout), taps_1d(k), bias=b.to(cuda), res=torch.cat(res, 1).to(cuda), div_sqrt2=True)
    d = (y.cpu() - want).abs()
    print("ksplit", ks, "max", float(d.max()), "bad rows", (d.max(1).values > 1e-4).nonzero().flatten().tolist()[:20], "bad cols", (d.max(0).values > 1e-4).nonzero().flatten().tolist()[:20])
