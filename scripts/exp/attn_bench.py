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
