// calibration: fp32 MFMA issue rate with and without LDS fragment reads (tuning aid, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    __shared__ float As[2][16][128], Bs[2][16][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1, l31 = lane & 31, lk = lane >> 5;
    for (int i = tid; i < 2 * 16 * 128; i += 256) { (&As[0][0][0])[i] = i * 1e-6f; (&Bs[0][0][0])[i] = i * 2e-6f; }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float a0 = lane * 1e-3f, a1 = a0 + 1.f, b0 = a0 + 2.f, b1 = a0 + 3.f;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE == 1) {
                a0 = As[buf][2 * s + lk][wm * 64 + l31]; a1 = As[buf][2 * s + lk][wm * 64 + 32 + l31];
                b0 = Bs[buf][2 * s + lk][wn * 64 + l31]; b1 = Bs[buf][2 * s + lk][wn * 64 + 32 + l31];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            if (NACC > 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            if (NACC > 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            if (NACC > 3) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + tid] = s;
}
template <int MODE, int NACC> void run(const char* name, int blocks)
{
    float* out; hipMalloc(&out, blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("%-28s blocks %4d: %8.3f ms  %7.1f TF/s\n", name, blocks, ms, fl / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<0, 4>("regs, 4 acc", 256); run<0, 4>("regs, 4 acc", 512); run<0, 4>("regs, 4 acc", 768);
    run<0, 1>("regs, 1 acc", 256); run<0, 2>("regs, 2 acc", 256);
    run<1, 4>("lds frags, 4 acc", 256); run<1, 4>("lds frags, 4 acc", 512); run<1, 4>("lds frags, 4 acc", 768);
    return 0;
}
