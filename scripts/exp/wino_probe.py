"""Round 5, VERDICT item 8: Winograd F(2,3) for the k = 3 convs of the AdaIN blocks (models.py:176-202, 474-517) -- the kernel-level question
first.  A k = 3 conv over N columns = 3 N column-taps of K channels; F(2,3) makes it FOUR pointwise (T = 1) convs over N / 2 tile columns each,
i.e. 2 N column-taps (2/3 of the matrix-core products), one weight set per transform component (the grouped launch the encoders use:
ConvGemmArgs.n_groups = 4), with the input transform B^T d in front (the AdaIN image writer would do it) and the output transform A^T m behind
(4 fp32 planes of N / 2 columns in, N columns out):
    V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3       (d_k = a[2t - 1 + k], zero outside the utterance)
    U0 = w0, U1 = (w0 + w1 + w2) / 2, U2 = (w0 - w1 + w2) / 2, U3 = w2
    y[2t] = m0 + m1 + m2, y[2t+1] = m1 - m2 - m3
This script: (a) the grouped pointwise launch against the direct conv, both replayed from a hipGraph; (b) the error of the Winograd form
through the real f16x3 kernel against float64, beside the direct conv's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from artspeech_amd import ops
dev = torch.device("cuda:0")
REP = 10


def timed(call):
    call(); torch.cuda.synchronize()
    graph, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        call(); torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(REP): call()
    graph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): graph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3


for (M, K, B, L) in [(1024, 1024, 32, 200), (1024, 1216, 32, 200), (512, 512, 32, 200), (512, 512, 96, 200), (1024, 1024, 128, 200)]:
    g = torch.Generator().manual_seed(M + K + B)
    lay = ops.layout([L] * B, dev)
    N = lay.N
    w = torch.randn(M, K, 3, generator=g) / (3 * K) ** 0.5
    x = torch.randn(K, N, generator=g)
    bias = torch.randn(M, generator=g)
    want = torch.cat([F.conv1d(x[:, b * L:(b + 1) * L][None].double(), w.double(), bias.double(), padding=1)[0] for b in range(B)], 1)
    xd = x.to(dev)
    xs = ops.split_act(xd, lay)
    wt = ops.prep_weight(w, dev)
    Y = lay.new(M)
    bd = bias.to(dev)
    direct = lambda: ops.conv_gemm(wt, None, lay, Y, ops.taps_1d(3), bias=bd, xs=xs, K=K)
    t_direct = timed(direct)
    err_direct = float((Y.double().cpu() - want).abs().max())
    # Winograd form: the transforms by torch (what the AdaIN image writer / the consumer would do), the four pointwise convs by the library
    NT = (L + 1) // 2
    a = F.pad(xd.view(K, B, L), (1, 2 * NT - L + 1))                      # a[-1] .. a[2 NT]: zero outside the utterance
    d = [a[:, :, k:k + 2 * NT:2] for k in range(4)]                       # d_k[t] = a[2t - 1 + k]
    V = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
    gc = (B * NT + 127) // 128 * 128
    XW = torch.zeros(K, 4 * gc, device=dev)
    for i in range(4):
        XW[:, i * gc:i * gc + B * NT] = V[i].reshape(K, B * NT)
    w64 = w.double()
    U = [w64[:, :, 0], (w64[:, :, 0] + w64[:, :, 1] + w64[:, :, 2]) / 2, (w64[:, :, 0] - w64[:, :, 1] + w64[:, :, 2]) / 2, w64[:, :, 2]]
    U = [u.float()[:, :, None].contiguous() for u in U]
    wt4 = ops.prep_weight(U[0], dev, stack=U[1:])
    layw = ops.layout([gc] * 4, dev)
    xws = ops.split_act(XW, layw)
    YW = layw.new(M)
    wino = lambda: ops.conv_gemm(wt4, None, layw, YW, ops.taps_1d(1), xs=xws, K=K, group_cols=gc)
    t_wino = timed(wino)
    m = [YW[:, i * gc:i * gc + B * NT].view(M, B, NT) for i in range(4)]
    y = torch.stack([m[0] + m[1] + m[2], m[1] - m[2] - m[3]], 3).reshape(M, B, 2 * NT)[:, :, :L].reshape(M, N) + bd[:, None]
    err_wino = float((y.double().cpu() - want).abs().max())
    fl = 2.0 * M * N * K * 3
    print(f"M{M} K{K} N{N}: direct T3 {t_direct:7.1f} us ({fl / t_direct / 1e6:5.0f} TF/s, err {err_direct:.1e})   "
          f"F(2,3) as 4 pointwise groups over {4 * gc} columns {t_wino:7.1f} us ({fl / t_wino / 1e6:5.0f} algorithmic TF/s, err {err_wino:.1e})   "
          f"ratio {t_wino / t_direct:.2f}; extra traffic of the transform domain {M * N * 4 * 2 / 1e6 + K * N * 4 / 1e6:.0f} MB", flush=True)
