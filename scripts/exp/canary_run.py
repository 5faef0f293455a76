import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout
dev = torch.device("cuda:0")
can = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "artspeech_amd", "lib", "exp_canary.so"))
can.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
g = torch.Generator().manual_seed(0)
layg = Layout([200] * 32, dev)
w = ops.prep_weight(torch.randn(1024, 1024, 3, generator=g) / 55.0, dev)
xsg = ops.split_act(torch.randn(1024, layg.N, generator=g).to(dev), layg); Yg = layg.new(1024)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for lds_kb in (0, 4, 32):
    for company in (False, True):
        err = torch.zeros(4, dtype=torch.int32, device=dev)
        for i in range(200):
            if company:
                with torch.cuda.stream(sb):
                    ops.conv_gemm(w, None, layg, Yg, ops.taps_1d(3), xs=xsg, K=1024)
            with torch.cuda.stream(sa):
                can.canary_launch(err.data_ptr(), 2048, 20, lds_kb, sa.cuda_stream)
        torch.cuda.synchronize()
        e = err.tolist()
        print(f"canary LDS {lds_kb:2d} KB, GEMM beside it: {company}:  register mismatches {e[0]}, LDS mismatches {e[1]}, workgroups-threads done {e[2]}")
