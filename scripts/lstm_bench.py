import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
for (H, B, L, J) in ((128, 32, 200, 3), (128, 32, 200, 1), (256, 32, 40, 1), (128, 2, 200, 1)):
    lay = ops.layout([L] * B, dev)
    jobs = [(torch.randn(lay.N, 8 * H, device=dev) * 0.1, torch.randn(2, H, 4 * H, device=dev) * 0.05, lay.new(2 * H)) for _ in range(J)]
    for _ in range(3): ops.bilstm(jobs, lay, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.bilstm(jobs, lay, H)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"H{H} B{B} L{L} jobs{J}: {ms*1e3:8.1f} us   {ms*1e3/L:6.2f} us/step")
