"""MAS geometry sweep (AS_MAS_R = rows per lane): python scripts/exp/mas_r.py B Tx Ty"""
import time, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from artspeech_amd import mas
dev = torch.device("cuda:0")
B, Tx, Ty = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 1024, 2000)
value = torch.rand(B, Tx, Ty, device=dev)
xl = torch.full((B,), Tx); yl = torch.full((B,), Ty)
for _ in range(3): mas.maximum_path_lens(value, xl, yl, want=("dur",))
torch.cuda.synchronize(); t0 = time.time()
n = 20
for _ in range(n): mas.maximum_path_lens(value, xl, yl, want=("dur",))
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"[{B},{Tx},{Ty}] R={os.environ.get('AS_MAS_R')}: {dt*1e6:.1f} us")
