"""us per step of the recurrence kernels (as_bilstm_f32 / as_bilstm_cluster_f32) at the shapes of the path"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd import ops
dev = torch.device("cuda:0")
cases = [(128, 32, 200, 3, None), (128, 32, 200, 1, None), (128, 2, 200, 1, None),
         (256, 32, 40, 1, None), (256, 32, 40, 1, "2"), (256, 16, 40, 1, "1"), (256, 8, 1024, 1, None), (256, 8, 1024, 1, "1"), (256, 8, 1024, 1, "2"),
         (256, 1, 200, 1, None), (256, 1, 200, 1, "1")]
for (H, B, L, J, cluster) in cases:
    lay = ops.Layout([L] * B, dev)
    jobs = [(torch.randn(lay.N, 8 * H, device=dev) * 0.1, torch.randn(2, H, 4 * H, device=dev) * 0.05, lay.new(2 * H)) for _ in range(J)]
    xchg = None
    if cluster:
        os.environ["AS_LSTM_CLUSTER"] = cluster
        xchg = ops.bilstm_exchange_buffer(J, B, dev)
    for _ in range(3): ops.bilstm(jobs, lay, H, xchg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.bilstm(jobs, lay, H, xchg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"H{H} B{B} L{L} jobs{J} cluster={cluster}: {ms*1e3:8.1f} us   {ms*1e3/L:6.2f} us/step")
