import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout, taps_1d
cuda = torch.device("cuda:0")
os.environ["AS_GEMM_TILE"] = "22"
M, K, lens = 128, 64, [50, 13, 1, 200]
g = torch.Generator().manual_seed(M + K)
lay = Layout(lens, cuda)
w = ops.prep_weight(torch.randn(M, K, 3, generator=g) / np.sqrt(3 * K), cuda)
b = torch.randn(M, generator=g).to(cuda)
X = torch.randn(K, lay.N, generator=g).to(cuda)
y = ops.conv_gemm(w, X, lay, lay.new(M), taps_1d(3), bias=b, act=ops.ACT_RELU)
want = ops.split_act(y, lay)
yh = ops.new_image(M, lay.N, cuda); yh.fill_(0x3c00)
ops.conv_gemm(w, X, lay, lay.new(M), taps_1d(3), bias=b, act=ops.ACT_RELU, yh=yh)
kbx, nx = ops.kbx(M), lay.N + 1
def parts(t):
    img = t[: kbx * 4 * nx * 8].view(torch.float16).reshape(kbx, 2, 2, nx, 8).float()
    return img.permute(1, 0, 2, 4, 3).reshape(2, kbx * 16, nx).cpu()
pg, pw = parts(yh), parts(want)
bad = (pg != pw)
print("mismatches", int(bad.sum()), "of", bad.numel(), "untouched(1.0)", int((pg == 1.0).sum()))
idx = bad.nonzero()
print("parts", idx[:, 0].unique().tolist(), "rows", idx[:, 1].unique().tolist()[:40], "cols", idx[:, 2].unique().tolist()[:40])
for p_, r_, c_ in idx[:12].tolist():
    print(p_, r_, c_, "got", float(pg[p_, r_, c_]), "want", float(pw[p_, r_, c_]), "y", float(y[r_, c_]) if c_ < lay.N else None)
# which source row does each got row correspond to?
yc = y.cpu()
for r_ in range(0, 16):
    src = [(rr) for rr in range(M) if torch.equal(pg[0, r_, :lay.N], yc[rr].half().float())]
    print("image row", r_, "holds y row", src)
