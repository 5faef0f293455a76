// power experiment (tuning aid): does the ORDER of the twelve f16x3 MFMAs of a k-block matter on a power-limited chip?  Consecutive MFMAs
// that share an operand register toggle fewer of the matrix core's input latches.  Orders of (A part / row tile, B part / column tile):
//   0  the kernel's: per product group (h,l) (l,h) (h,h), raster over (i, j): (0,0) (0,1) (1,0) (1,1)
//   1  snake inside a group: (0,0) (0,1) (1,1) (1,0) -- one operand changes per step inside a group
//   2  snake across the groups too: the first MFMA of a group shares an operand with the last of the one before
//   3  worst case: every consecutive pair changes both operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define M(A, B, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[A], f[B], acc[I][J], 0, 0, 0)
template <int ORDER, int LDSF>
__global__ void __launch_bounds__(256) k(const f16x8* __restrict__ src, float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16x8* lds = reinterpret_cast<f16x8*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[(blockIdx.x * 2048 + i) % (1 << 16)];
    __syncthreads();
    f16x8 f[8];      // f[0..1] A h, f[2..3] A l, f[4..5] B h, f[6..7] B l
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = lds[(q * 64 + lane + wave * 17) & 2047];
    f32x16 acc[2][2];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a >> 1][a & 1][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (LDSF) {
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = lds[((q + (it & 7) * 8) * 64 + lane) & 2047];
        }
        if (ORDER == 0) {
            M(0, 6, 0, 0); M(0, 7, 0, 1); M(1, 6, 1, 0); M(1, 7, 1, 1);
            M(2, 4, 0, 0); M(2, 5, 0, 1); M(3, 4, 1, 0); M(3, 5, 1, 1);
            M(0, 4, 0, 0); M(0, 5, 0, 1); M(1, 4, 1, 0); M(1, 5, 1, 1);
        } else if (ORDER == 1) {
            M(0, 6, 0, 0); M(0, 7, 0, 1); M(1, 7, 1, 1); M(1, 6, 1, 0);
            M(2, 4, 0, 0); M(2, 5, 0, 1); M(3, 5, 1, 1); M(3, 4, 1, 0);
            M(0, 4, 0, 0); M(0, 5, 0, 1); M(1, 5, 1, 1); M(1, 4, 1, 0);
        } else if (ORDER == 2) {
            // A h0: B l0, B l1 | A h1: B l1, B l0 ... then stay on a B: (A l0, B h0) shares nothing with (A h1, B l0) -> go through (A h1, B h0) first
            M(0, 6, 0, 0); M(0, 7, 0, 1); M(1, 7, 1, 1); M(1, 6, 1, 0);
            M(3, 4, 1, 0); M(3, 5, 1, 1); M(2, 5, 0, 1); M(2, 4, 0, 0);      // (l,h): A l1 B h0, A l1 B h1, A l0 B h1, A l0 B h0
            M(0, 4, 0, 0); M(0, 5, 0, 1); M(1, 5, 1, 1); M(1, 4, 1, 0);      // (h,h): A h0 B h0 shares B h0 with the one before
        } else {
            M(0, 6, 0, 0); M(3, 5, 1, 1); M(0, 7, 0, 1); M(2, 4, 0, 0); M(1, 6, 1, 0); M(2, 5, 0, 1);
            M(1, 7, 1, 1); M(3, 4, 1, 0); M(0, 5, 0, 1); M(1, 4, 1, 0); M(0, 4, 0, 0); M(1, 5, 1, 1);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a >> 1][a & 1][e];
    out[blockIdx.x * 256 + tid] = s;
}
template <int ORDER, int LDSF> void run(const char* name, const f16x8* src)
{
    float* out;
    const int blocks = 512, iters = 40000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<ORDER, LDSF>), dim3(blocks), dim3(256), 32768, 0, src, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ORDER, LDSF>), dim3(blocks), dim3(256), 32768, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 12 * 32768.0;
    printf("%-28s lds %d: %8.3f ms  %7.1f TF/s fp16\n", name, LDSF, ms, fl / ms / 1e9);
    hipFree(out);
}
int main()
{
    const size_t n = (size_t)(1 << 16) * 8;
    _Float16* h = (_Float16*)malloc(n * 2);
    f16x8* d;
    hipMalloc(&d, n * 2);
    srand(1);
    // h parts ~ U(-1, 1); (the l parts of real operands are 2^-11 of that: the generator's values serve for both here, as in mfma_f16_peak)
    for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("0 raster (the kernel's)", d); run<1, 0>("1 snake in group", d); run<2, 0>("2 snake across groups", d); run<3, 0>("3 scattered", d);
        run<0, 1>("0 raster (the kernel's)", d); run<1, 1>("1 snake in group", d); run<2, 1>("2 snake across groups", d); run<3, 1>("3 scattered", d);
    }
    return 0;
}
