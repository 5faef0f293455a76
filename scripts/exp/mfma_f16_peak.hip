// calibration (tuning aid, not part of the library): what a bare fp16 MFMA loop sustains on this chip on RANDOM operands, for the two
// MFMA shapes, with the operands held in registers or re-read from LDS at the conv GEMM's ratio (8 ds_read_b128 per 12 MFMAs of
// 32x32x16 = per 16-deep k-block of a 64x64 wave tile in f16x3 arithmetic).  The chip lowers its clock under matrix-core load
// (MI355X_MICROARCH.md, DVFS give-back): this is the ceiling the GEMM kernel can be compared with, not 2516 TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: v_mfma_f32_32x32x16_f16, wave tile 64x64 = 2x2 accumulators, per k-block (A h, A l) x 2 row tiles, (B h, B l) x 2 column tiles,
//          12 MFMAs (h*l, l*h, h*h).   SHAPE 1: v_mfma_f32_16x16x32_f16, 4x4 accumulators, per 32-deep step 16 fragments, 48 MFMAs.
// LDSF 0: fragments stay in registers; 1: re-read from LDS every step (conflict-free 16-byte rows)
template <int SHAPE, int LDSF>
__global__ void __launch_bounds__(256) k(const f16x8* __restrict__ src, float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16x8* lds = reinterpret_cast<f16x8*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[(blockIdx.x * 2048 + i) % (1 << 16)];
    __syncthreads();
    constexpr int NF = SHAPE == 0 ? 8 : 16;
    f16x8 f[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) f[q] = lds[(q * 64 + lane + wave * 17) & 2047];
    if (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a >> 1][a & 1][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if (LDSF) {
#pragma unroll
                for (int q = 0; q < NF; ++q) f[q] = lds[((q + (it & 7) * 8) * 64 + lane) & 2047];
            }
            // f[0..1] A h, f[2..3] A l, f[4..5] B h, f[6..7] B l
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], f[6 + j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 + i], f[4 + j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], f[4 + j], acc[i][j], 0, 0, 0);
        }
        float s = 0.f;
        for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a >> 1][a & 1][e];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        f32x4 acc[4][4];
        for (int a = 0; a < 16; ++a) for (int e = 0; e < 4; ++e) acc[a >> 2][a & 3][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if (LDSF) {
#pragma unroll
                for (int q = 0; q < NF; ++q) f[q] = lds[((q + (it & 7) * 16) * 64 + lane) & 2047];
            }
            // f[0..3] A h, f[4..7] A l, f[8..11] B h, f[12..15] B l
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[i], f[12 + j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[4 + i], f[8 + j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[i], f[8 + j], acc[i][j], 0, 0, 0);
        }
        float s = 0.f;
        for (int a = 0; a < 16; ++a) for (int e = 0; e < 4; ++e) s += acc[a >> 2][a & 3][e];
        out[blockIdx.x * 256 + tid] = s;
    }
}

template <int SHAPE, int LDSF> void run(const char* name, int blocks, const f16x8* src, int zero)
{
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = SHAPE == 0 ? 40000 : 20000;          // same flop either way; ~several ms per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, LDSF>), dim3(blocks), dim3(256), 32768, 0, src, out, iters);   // warm: the clock settles
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, LDSF>), dim3(blocks), dim3(256), 32768, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * iters * (SHAPE == 0 ? 12 : 48);
    const double fl = mfma * (SHAPE == 0 ? 32768.0 : 16384.0);
    // cycles the matrix pipes need at 2.4 GHz: one SIMD runs its waves' MFMAs back to back (32 / 16 cycles each)
    const double pipe_ms = mfma * (SHAPE == 0 ? 32.0 : 16.0) / 1024.0 / 2.4e6;
    printf("%-34s %s blocks %4d: %8.3f ms  %7.1f TF/s fp16  (= %6.1f TF/s of f16x3 fp32 work)  pipe-bound at 2.4 GHz: %6.3f ms => clock x busy = %.2f GHz\n", name,
           zero ? "zeros " : "random", blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 3, pipe_ms, 2.4 * pipe_ms / ms);
    hipFree(out);
}

int main()
{
    const size_t n = (size_t)(1 << 16) * 8;
    _Float16* h = (_Float16*)malloc(n * 2);
    f16x8 *drand, *dzero;
    hipMalloc(&drand, n * 2);
    hipMalloc(&dzero, n * 2);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(drand, h, n * 2, hipMemcpyHostToDevice);
    hipMemset(dzero, 0, n * 2);
    for (int z = 0; z < 2; ++z) {
        const f16x8* s = z ? dzero : drand;
        for (int blocks : {256, 512}) {
            run<0, 0>("32x32x16 registers", blocks, s, z);
            run<1, 0>("16x16x32 registers", blocks, s, z);
            run<0, 1>("32x32x16 + 8 ds_read_b128 / 12", blocks, s, z);
            run<1, 1>("16x16x32 + 16 ds_read_b128 / 48", blocks, s, z);
        }
    }
    return 0;
}
