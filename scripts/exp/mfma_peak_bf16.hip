// calibration: what one MI355X actually sustains on v_mfma_f32_32x32x16_bf16 (tuning aid, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* ticks)
{
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    bf16x8 a0, a1, b0, b1;
    for (int e = 0; e < 8; ++e) { a0[e] = (__bf16)(lane * 1e-3f + e); a1[e] = (__bf16)(1.f + e); b0[e] = (__bf16)(2.f - e); b1[e] = (__bf16)(0.5f * e); }
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
            if (NACC > 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
            if (NACC > 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
            if (NACC > 3) acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
        }
    }
    unsigned long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) *ticks = t1 - t0;
}
template <int NACC> void run(const char* name, int blocks)
{
    float* out; hipMalloc(&out, blocks * 256 * 4);
    unsigned long long* tk; hipMalloc(&tk, 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, 10, tk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, tk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 6 * NACC;
    double fl = (double)blocks * 4 * nm * 2.0 * 32 * 32 * 16;
    printf("%-16s blocks %4d: %8.3f ms  %8.1f TF/s bf16 (= %6.1f fp32-equivalent with 6 products)  %.1f clock64 ticks per MFMA per wave, %.2f ticks/ns\n", name, blocks, ms,
           fl / ms / 1e9, fl / ms / 1e9 / 6, (double)h / nm, (double)h / (ms * 1e6));
    hipFree(out); hipFree(tk);
}
int main()
{
    run<4>("4 acc", 256); run<4>("4 acc", 512); run<1>("1 acc", 256); run<2>("2 acc", 256); run<4>("4 acc", 64);
    return 0;
}
