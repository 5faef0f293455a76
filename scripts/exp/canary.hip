// debug aid: a kernel that parks known values in many VGPRs (and optionally LDS), idles, and checks them -- run beside the conv GEMM
// to see whether a co-resident kernel's registers or LDS are disturbed.   hipcc --offload-arch=gfx950 -shared -fPIC -o canary.so canary.hip
#include <hip/hip_runtime.h>
template <int LDSF>
__global__ void __launch_bounds__(256) canary_kernel(unsigned* err, int iters)
{
    __shared__ float lds[LDSF > 0 ? LDSF : 1];
    const int tid = threadIdx.x + blockIdx.x * 256;
    float v[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) { v[k] = (float)(tid * 3 + k); asm volatile("" : "+v"(v[k])); }
    if (LDSF > 0)
        for (int i = threadIdx.x; i < LDSF; i += 256) lds[i] = (float)(i * 7 + blockIdx.x);
    __syncthreads();
    unsigned bad = 0, badl = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_sleep(64);
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            asm volatile("" : "+v"(v[k]));
            bad += v[k] != (float)(tid * 3 + k);
        }
        if (LDSF > 0)
            for (int i = threadIdx.x; i < LDSF; i += 256) badl += lds[i] != (float)(i * 7 + blockIdx.x);
    }
    if (bad) atomicAdd(&err[0], bad);
    if (badl) atomicAdd(&err[1], badl);
    atomicAdd(&err[2], 1u);
}
extern "C" int canary_launch(unsigned* err, int blocks, int iters, int lds_kb, void* stream)
{
    if (lds_kb >= 32) hipLaunchKernelGGL(canary_kernel<8192>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, err, iters);
    else if (lds_kb > 0) hipLaunchKernelGGL(canary_kernel<1024>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, err, iters);
    else hipLaunchKernelGGL(canary_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, err, iters);
    return (int)hipGetLastError();
}
