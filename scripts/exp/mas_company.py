"""MAS (banded kernel: workgroups that poll each other) while another stream keeps the chip full of conv GEMMs; every result compared
with the first.  python scripts/exp/mas_company.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import mas, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [(8, 1024, 2000), (32, 300, 500), (3, 2000, 1000)]
lay = ops.layout([200] * 32, dev)
wt = ops.prep_weight(torch.randn(1024, 1024, 3) / 55.0, dev)
X = torch.randn(1024, lay.N, device=dev); xs = ops.split_act(X, lay); Y = lay.new(1024)
side = torch.cuda.Stream()
bad = 0
for (B, Tx, Ty) in cases:
    v = torch.rand(B, Tx, Ty, device=dev)
    xl = torch.randint(Tx // 2, Tx + 1, (B,), device=dev); yl = torch.maximum(xl, torch.randint(Ty // 2, Ty + 1, (B,), device=dev))
    ref = mas.maximum_path_lens(v, xl, yl, want=("dur", "rows"))
    torch.cuda.synchronize()
    for it in range(60):
        with torch.cuda.stream(side):
            for _ in range(6):
                ops.conv_gemm(wt, None, lay, Y, ops.taps_1d(3), xs=xs, K=1024)
        out = mas.maximum_path_lens(v, xl, yl, want=("dur", "rows"))
        torch.cuda.synchronize()
        if not (torch.equal(out["dur"], ref["dur"]) and torch.equal(out["rows"], ref["rows"])):
            bad += 1
    print(f"[{B},{Tx},{Ty}] 60 launches beside the GEMM stream: {'all equal' if not bad else str(bad) + ' DIFFER'}", flush=True)
sys.exit(1 if bad else 0)
