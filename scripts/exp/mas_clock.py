"""Where the DP wave of a MAS band spends its cycles (-DAS_EXPERIMENTS build): python scripts/exp/mas_clock.py [B Tx Ty]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from artspeech_amd import mas, _lib
dev = torch.device("cuda:0")
B, Tx, Ty = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 1024, 2000)
value = torch.rand(B, Tx, Ty, device=dev)
xl = torch.full((B,), Tx); yl = torch.full((B,), Ty)
for _ in range(5): mas.maximum_path_lens(value, xl, yl, want=("dur",))
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
assert L.as_mas_debug(out) == 0
names = ("wait for the loader", "poll the band above", "dp block", "all")
for n, v in zip(names, out[:4]):
    print(f"[{B},{Tx},{Ty}] {n:22s} {v:9d} cycles  {v / Ty:7.1f} / column")
