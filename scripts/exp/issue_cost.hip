// calibration: cycles that wave-instructions of each kind add to a stream of v_mfma_f32_32x32x16_bf16 (tuning aid).
// One workgroup of 4 waves per CU, 24 MFMAs per iteration with NF filler instructions of one kind dealt out behind them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
// KIND: 0 none, 1 buffer_load_dword, 2 buffer_load_dwordx4, 3 LDS-DMA dwordx4, 4 ds_read_b128, 5 ds_write_b128, 6 v_fma, 7 s_add
template <int KIND, int NF>
__global__ void __launch_bounds__(256) k(const float* src, float* out, int iters, unsigned long long* ticks)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 26, 0x00020000);
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    bf16x8 a0, b0;
    for (int e = 0; e < 8; ++e) { a0[e] = (__bf16)(lane * 1e-3f + e); b0[e] = (__bf16)(2.f - e); }
    float sink = 0.f; u32x4 sinkv = {0, 0, 0, 0}; int ssink = iters;
    constexpr int NR = NF > 24 ? 24 : (NF > 0 ? NF : 1);     // results stay in registers until the end of the iteration (as in the
    float lf[NR]; u32x4 lv[NR];                              // GEMM: loaded now, consumed an iteration later)
    unsigned voff = (blockIdx.x * 256 + tid) * 16;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 24; ++n) {
            acc[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[n & 3], 0, 0, 0);
            if (n * NF / 24 != (n + 1) * NF / 24 || (NF >= 24)) {
#pragma unroll
                for (int r = 0; r < (NF >= 24 ? NF / 24 : 1); ++r) {
                    if (KIND == 1) lf[(n * NR / 24) % NR] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (it * 24 + n) * 64 & 0xffff, 0));
                    if (KIND == 2) lv[(n * NR / 24) % NR] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (it * 24 + n) * 64 & 0xffff, 0);
#if __HIP_DEVICE_COMPILE__
                    if (KIND == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + wave * 1024 + (n & 7) * 4096), 16, voff, (it * 24 + n) * 64 & 0xffff, 0, 0);
#endif
                    if (KIND == 4) lv[(n * NR / 24) % NR] = *reinterpret_cast<u32x4*>(lds + ((n + r) & 7) * 4096 + tid * 16);
                    if (KIND == 5) *reinterpret_cast<u32x4*>(lds + ((n + r) & 7) * 4096 + tid * 16) = sinkv;
                    if (KIND == 6) sink = __builtin_fmaf(sink, 1.0001f, 0.5f);
                    if (KIND == 7) ssink = ssink * 3 + n;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (KIND == 1) for (int q = 0; q < NR; ++q) sink += lf[q];
        if (KIND == 2 || KIND == 4) for (int q = 0; q < NR; ++q) sinkv[q & 3] ^= lv[q][q & 3];
    }
    unsigned long long t1 = clock64();
    float s = sink + sinkv[0] + sinkv[1] + ssink;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) *ticks = t1 - t0;
}
template <int KIND, int NF> void run(const char* name)
{
    const int blocks = 256, iters = 500;
    float *out, *src; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&src, 1 << 27); hipMemset(src, 0, 1 << 27);
    unsigned long long* tk; hipMalloc(&tk, 8);
    hipLaunchKernelGGL((k<KIND, NF>), dim3(blocks), dim3(256), 0, 0, src, out, 10, tk);
    hipLaunchKernelGGL((k<KIND, NF>), dim3(blocks), dim3(256), 0, 0, src, out, iters, tk);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
    printf("%-22s x%2d per 24 MFMAs: %7.1f ticks per iteration (768 = MFMA-bound)\n", name, NF, (double)h / iters);
    hipFree(out); hipFree(src); hipFree(tk);
}
int main()
{
    run<0, 0>("none");
    run<1, 8>("buffer_load_dword"); run<1, 24>("buffer_load_dword");
    run<2, 8>("buffer_load_dwordx4"); run<2, 24>("buffer_load_dwordx4");
    run<3, 3>("LDS-DMA dwordx4"); run<3, 8>("LDS-DMA dwordx4"); run<3, 24>("LDS-DMA dwordx4");
    run<4, 12>("ds_read_b128"); run<4, 24>("ds_read_b128"); run<4, 48>("ds_read_b128");
    run<5, 3>("ds_write_b128"); run<5, 12>("ds_write_b128"); run<5, 24>("ds_write_b128");
    run<6, 48>("v_fma_f32"); run<6, 96>("v_fma_f32"); run<6, 144>("v_fma_f32");
    run<7, 48>("s_mul/add"); run<7, 96>("s_mul/add");
    return 0;
}
