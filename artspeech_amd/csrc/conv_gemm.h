// Internal pieces shared by the two conv-GEMM arithmetic paths (conv_gemm.hip: fp32 MFMA; conv_gemm_x6.hip: bf16x6).
#pragma once
#include "artspeech_hip.h"
#include "common.h"

#define OOB 0xFFFFFFFFu

// host: launch the bf16x6 kernel for tile `choice` (22 = 128x128, 21 = 128x64, 12 = 64x128, 11 = 64x64) on a grid of
// (tiles, S); returns AS_OK or a hipError_t.  (conv_gemm_x6.hip)
int as_conv_gemm_x6_launch(const ConvGemmArgs& a, int choice, int S, hipStream_t stream);

#ifdef __HIPCC__
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
static __device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// XCD-aware order (speed only): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
// run of logical tiles -- the output-channel tiles of one column range then share that XCD's L2 copy of the
// activation columns.  Bijective for any grid size.
static __device__ __forceinline__ int logical_tile()
{
    const int nb = gridDim.x, xcd = blockIdx.x & 7, q8 = nb >> 3, r8 = nb & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
}

// Accumulator tiles -> Y.  A 32x32 MFMA accumulator holds C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31];
// a wave owns TM x TN of them at rows m0 + wm*32*TM, columns n0 + wn*32*TN.  S > 1: raw partial sums into this
// slice's slab (splitk_reduce_kernel applies the epilogue).
template <int TM, int TN>
static __device__ __forceinline__ void epilogue(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm,
                                                int wn, int l31, int lk, int S)
{
    if (S > 1) {
        float* slab = a.ws + (size_t)blockIdx.y * a.M * a.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const int col = n0 + wn * 32 * TN + jn * 32 + l31;
                if (col >= a.N) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                    if (row < a.M) slab[(size_t)row * a.N + col] = acc[i][jn][e];
                }
            }
        return;
    }
    // Loads first, stores after: Y may alias res, so the compiler keeps program order, and a load issued after a
    // store waits out its whole latency alone (measured: 64 such round trips = 40k cycles per 128x128 tile).
    float bv[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
            bv[i][e] = (a.bias && row < a.M) ? a.bias[row] : 0.f;
        }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = n0 + wn * 32 * TN + jn * 32 + l31;
            if (col >= a.N) continue;
            const int row0 = m0 + wm * 32 * TM + i * 32 + 4 * lk;
            float rv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + (e & 3) + 8 * (e >> 2);
                rv[e] = (a.res && row < a.M) ? a.res[(size_t)row * a.ldr + col] : 0.f;
            }
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x = acc[i][jn][e] + bv[i][e];
                x += rv[e];
                if (a.div_sqrt2) x = x / 1.41421356237309504880f;
                if (a.act == 1) x = x > 0.f ? x : 0.f;
                else if (a.act == 2) x = x > 0.f ? x : 0.2f * x;
                v[e] = x;
            }
            if (a.transpose_out) {                           // time-major output for the LSTM: 4 consecutive rows = 16 bytes
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = row0 + 8 * q;
                    float* y = a.Y + (size_t)col * a.ldy + row;
                    if (row + 3 < a.M && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0) {
                        *reinterpret_cast<f32x4*>(y) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (row + c < a.M) y[c] = v[4 * q + c];
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    if (row < a.M) a.Y[(size_t)row * a.ldy + col] = v[e];
                }
            }
        }
    }
}
#endif
