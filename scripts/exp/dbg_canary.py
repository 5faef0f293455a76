import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
L.as_exp_lds_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
B, N = 8, 1024
lay = ops.layout([N] * B, dev)
w = ops.prep_weight(torch.randn(1024, 512, 9) / 68, dev)
X = lay.new(512); X.copy_(torch.randn(512, lay.N))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for impl, tile in (("x6", "22"), ("x6", "21"), ("f32", "22")):
    ops.GEMM_IMPL = impl; os.environ["AS_GEMM_TILE"] = tile
    for kb in (40, 86, 100):
        bad = torch.zeros(1, dtype=torch.int32, device=dev); first = torch.full((1,), 2**31 - 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            L.as_exp_lds_canary(kb * 1024, 512, 400000, bad.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream)
        with torch.cuda.stream(s2):
            for _ in range(3):
                ops.conv_gemm(w, X, lay, lay.new(1024), ops.taps_1d(9))
        torch.cuda.synchronize()
        print(impl, tile, "canary", kb, "KB: changed words", int(bad), "first", int(first) if int(bad) else None)
