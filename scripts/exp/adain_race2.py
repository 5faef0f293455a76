import os, sys
sys.path.insert(0, "/root/repo")
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, L, C = 32, 100, 512
lay = Layout([L] * B, dev); lay2 = Layout([2 * L] * B, dev)
X = torch.randn(C, lay.N, generator=g).to(dev); gb = torch.randn(B, 2 * C, generator=g).to(dev)
pw, pb = torch.randn(C, 3, generator=g).to(dev), torch.randn(C, generator=g).to(dev); xup = lay2.new(C)
run_adain = lambda: ops.adain_image(X, lay, gb, 1, lay2.N, ldgb=2 * C, pool_w=pw, pool_b=pb, x_up=xup)
ref = run_adain().clone(); torch.cuda.synchronize()
M, K = 1024, 1024
layg = Layout([200] * 32, dev)
w = ops.prep_weight(torch.randn(M, K, 3, generator=g) / 55.0, dev)
xsg = ops.split_act(torch.randn(K, layg.N, generator=g).to(dev), layg)
Yg = layg.new(M)
ops.conv_gemm(w, None, layg, Yg, ops.taps_1d(3), xs=xsg, K=K); torch.cuda.synchronize(); yref = Yg.clone()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad_a = bad_g = 0
for rnd in range(30):
    ys = []
    for i in range(10):
        with torch.cuda.stream(sb):
            Yi = layg.new(M)
            ops.conv_gemm(w, None, layg, Yi, ops.taps_1d(3), xs=xsg, K=K); ys.append(Yi)
        with torch.cuda.stream(sa):
            o = run_adain()
            bad_a += 0
            ys.append(o)
    torch.cuda.synchronize()
    for t in ys:
        if t.dtype == torch.float32: bad_g += int(not torch.equal(t, yref))
        else: bad_a += int(not torch.equal(t, ref))
print("adain outputs wrong:", bad_a, "of 300; GEMM outputs wrong:", bad_g, "of 300")
import ctypes
from artspeech_amd import _lib
L_ = _lib.lib()
if hasattr(L_, "as_debug_adain_counter"):
    buf = (ctypes.c_uint * 4)()
    L_.as_debug_adain_counter(buf)
    print("LDS-vs-global checks:", buf[0], "mismatches at first read:", buf[1], "still wrong on re-read:", buf[2])
