"""One MAS shape, a few launches (for rocprofv3 --kernel-trace --stats): python scripts/exp/mas_one.py [B Tx Ty]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from artspeech_amd import mas
dev = torch.device("cuda:0")
B, Tx, Ty = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 1024, 2000)
value = torch.rand(B, Tx, Ty, device=dev)
xl = torch.full((B,), Tx); yl = torch.full((B,), Ty)
for _ in range(10): mas.maximum_path_lens(value, xl, yl, want=("dur",))
torch.cuda.synchronize()
