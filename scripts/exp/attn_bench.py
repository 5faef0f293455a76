"""times the two attention entry points at the C5 and C3 shapes: as_relpos_attention_groups_f32 (exact fp32, vector ALU) and
as_relpos_attention_image_f32 (the path: matrix cores, operands from the q/k/v GEMM's image)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout

dev = torch.device("cuda:0")
for name, lens in (("C5 24x1024", [1024] * 24), ("C3 96x40", [40] * 96), ("200 x 32", [200] * 32)):
    C = 512
    g = torch.Generator().manual_seed(1)
    lay = Layout(lens, dev)
    X = torch.randn(C, lay.N, generator=g).to(dev)
    w = ops.prep_weight(torch.randn(3 * C, C, 1, generator=g) / C ** 0.5, dev)
    qkv, qkv_h = lay.new(3 * C), ops.new_image(3 * C, lay.N, dev)
    ops.conv_gemm(w, X, lay, qkv, ops.taps_1d(1), yh=qkv_h)
    ek = (torch.randn(9, 128, generator=g) * 0.1).to(dev)
    ev = (torch.randn(9, 128, generator=g) * 0.1).to(dev)
    out = lay.new(C)
    res = {}
    for mode, fn in (("exact", lambda: ops.relpos_attention(qkv, C, 4, 4, ek, ev, lay, out)),
                     ("image", lambda: ops.relpos_attention_image(qkv, qkv_h, C, 4, 4, ek, ev, lay, out=out))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / n * 1e6
        res[mode + "_out"] = out.clone()
    flop = sum(4 * L * L * 128 * 4 for L in lens)
    print(f"{name}: exact fp32 {res['exact']:.1f} us, from the image {res['image']:.1f} us ({flop / res['image'] / 1e6:.1f} TF/s fp32-equivalent), "
          f"max diff {float((res['exact_out'] - res['image_out']).abs().max()):.2e}")

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
