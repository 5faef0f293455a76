#!/bin/bash
# batch-1 shapes: 64x64 tile with EIGHT waves splitting K inside the workgroup (no slabs, no reduce launch) against the 4-wave tile + K slices
# (needs a build with the tile: conv_gemm_h3.hip `case 18: return launch_h3<1, 1, 8, 1, 2, 3>(a, S, stream);`, the kernel's static_assert
#  relaxed to NT == 512, and `wk = 8` for it in gemm_ksplit; the shipped library does not carry it -- it lost on every shape)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3p; mkdir -p $O
TILES=,11,18 KSPLITS=,1,2 python3 $R/scripts/gemm_bench.py 512,150,512,3,150 1024,150,1024,3,150 1024,150,1216,3,150 1024,286,512,9,143 512,30,512,3,30 512,286,512,5,143 256,450,512,3,150 512,190,512,9,190 2>&1 | grep "us " | tee $O/t18.log
