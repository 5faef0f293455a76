"""When each DP wave of utterance 0 starts / ends its first block and ends its last (-DAS_EXPERIMENTS build): the pipeline's fill.
python scripts/exp/mas_hops.py [B Tx Ty]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from artspeech_amd import mas, _lib
dev = torch.device("cuda:0")
B, Tx, Ty = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 1024, 2000)
value = torch.rand(B, Tx, Ty, device=dev)
xl = torch.full((B,), Tx, device=dev); yl = torch.full((B,), Ty, device=dev)
for _ in range(5): mas.maximum_path_lens(value, xl, yl, want=("dur",))
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 128)()
assert L.as_mas_debug2(out) == 0
n = (Tx + 63) // 64
t0 = out[0]
prev = None
for i in range(n):
    a, b_, c = [(out[4 * i + k] - t0) / 100.0 for k in range(3)]
    print(f"wave {i:2d} (band {i // 2}): block 0 starts {a:7.2f} us, done {b_:7.2f}, last block done {c:7.2f}" + (f"   start lag {a - prev:5.2f}" if prev is not None else ""))
    prev = a
