"""times as_relpos_attention_groups_f32 at the C5 and C3 shapes (AS_ATTN selects the kernel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout

dev = torch.device("cuda:0")
for name, lens in (("C5 16x1024", [1024] * 16), ("C3 64x40", [40] * 64), ("200 x 32", [200] * 32)):
    C = 512
    g = torch.Generator().manual_seed(1)
    N = sum(lens)
    qkv = torch.randn(3 * C, N, generator=g).to(dev)
    ek = (torch.randn(9, 128, generator=g) * 0.1).to(dev)
    ev = (torch.randn(9, 128, generator=g) * 0.1).to(dev)
    lay = Layout(lens, dev)
    out = lay.new(C)
    res = {}
    for mode in ("valu", "mfma"):
        os.environ["AS_ATTN"] = mode
        for _ in range(3):
            ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, out)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / n * 1e6
        res[mode + "_out"] = out.clone()
    flop = sum(4 * L * L * 128 * 4 for L in lens)
    print(f"{name}: valu {res['valu']:.1f} us, mfma {res['mfma']:.1f} us ({flop / res['mfma'] / 1e6:.1f} TF/s fp32-equivalent), "
          f"max diff {float((res['valu_out'] - res['mfma_out']).abs().max()):.2e}")

if os.environ.get("ATTN_DBG"):
    # experiment build with -DATTN_DBG: cycle marks of workgroup (0, 1, 1), thread 0, in row 255 of `out`
    lens = [40] * 64
    lay = Layout(lens, dev)
    g = torch.Generator().manual_seed(1)
    C = 512
    X = torch.randn(C, lay.N, generator=g).to(dev)
    w = ops.prep_weight(torch.randn(3 * C, C, 1, generator=g) / C ** 0.5, dev)
    qkv, qkv_h = lay.new(3 * C), ops.new_image(3 * C, lay.N, dev)
    ops.conv_gemm(w, X, lay, qkv, ops.taps_1d(1), yh=qkv_h)
    out = lay.new(C)
    for _ in range(3):
        ops.relpos_attention_image(qkv, qkv_h, C, 4, 4, ek, ev, lay, out=out)
    torch.cuda.synchronize()
    row = out[255].cpu()
    n = int(row[0])
    print("marks (cycles of the 100 MHz counter? raw):", [int(v) for v in row[1:n]])
    t0 = time.perf_counter()
    for _ in range(20):
        ops.relpos_attention_image(qkv, qkv_h, C, 4, 4, ek, ev, lay, out=out)
    torch.cuda.synchronize()
    print("image kernel, 64 x 40:", (time.perf_counter() - t0) / 20 * 1e6, "us")
