import time, sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from artspeech_amd import mas, _lib
dev = torch.device("cuda:0")
for (B, Tx, Ty) in ((32, 40, 100), (8, 1024, 2000), (32, 128, 500)):
    value = torch.rand(B, Tx, Ty, device=dev)
    xl = torch.full((B,), Tx, device=dev); yl = torch.full((B,), Ty, device=dev)
    for want in (("dur",), ("path",)):
        for _ in range(3): mas.maximum_path_lens(value, xl, yl, want=want)
        torch.cuda.synchronize(); t0 = time.time()
        n = 20
        for _ in range(n): mas.maximum_path_lens(value, xl, yl, want=want)
        torch.cuda.synchronize(); dt = (time.time() - t0) / n
        print(f"MAS {B}x{Tx}x{Ty} want={want}: {dt*1e6:.1f} us  {B*Tx*Ty/dt/1e9:.2f} Gcell/s  {4*B*Tx*Ty/dt/1e9:.1f} GB/s read")
