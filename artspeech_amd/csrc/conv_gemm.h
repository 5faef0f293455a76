// Internal pieces shared by the two conv-GEMM arithmetic paths (conv_gemm.hip: fp32 MFMA; conv_gemm_x6.hip: bf16x6).
#pragma once
#include "artspeech_hip.h"
#include "common.h"

#define OOB 0xFFFFFFFFu
#define OOBH 0x80000000u   /* epilogue: out of range for every descriptor (< 2 GiB, host-checked) even after adding an in-range offset */

// host: launch the bf16x6 kernel for tile `choice` (22 = 128x128, 21 = 128x64, 12 = 64x128, 11 = 64x64) on a grid of
// (tiles, S); returns AS_OK or a hipError_t.  (conv_gemm_x6.hip)
int as_conv_gemm_x6_launch(const ConvGemmArgs& a, int choice, int S, hipStream_t stream);
// host: the tap-shared bf16x6 kernel (conv_gemm_x6t.hip, tile 128x128): k-tiles it would run (0 = taps not in groups
// of three consecutive column offsets), and its launch on a grid of (tiles, S)
int as_conv_gemm_x6t_ktiles(const ConvGemmArgs& a);
int as_conv_gemm_x6t_launch(const ConvGemmArgs& a, int S, hipStream_t stream);

// host: the bf16x6 kernel reading BOTH operands by LDS-DMA (conv_gemm_x6d.hip; a.Xs = pre-split activations; tiles 22, 21,
// 12), and the launch of the kernel that makes that image (no profiling scope of its own)
int as_conv_gemm_x6d_launch(const ConvGemmArgs& a, int choice, int S, hipStream_t stream);
int as_split_bf16x3_launch(const float* x, int ldx, int K, int N, int lrelu, float slope, uint16_t* xs, hipStream_t stream);

#ifdef __HIPCC__
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

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
// Branch-free: every access is a raw BUFFER access whose per-lane offset carries the whole (row, column) position, so
// rows >= M fall past the descriptor's end (loads return 0, stores are dropped by the range check) and columns >= N
// get an out-of-range offset -- no exec-mask branches, no 64-bit address arithmetic per element (the branchy
// version of this function took 20-40k cycles per 128x128 tile, a third of a small GEMM).
static __device__ __forceinline__ float div_sqrt2f(float x)
{
    // x / sqrt(2), correctly rounded without the division sequence: q = x*(1/c), one Newton correction in fma
    const float c = 1.41421356237309504880f, rc = 0.70710678118654752440f;
    const float q = x * rc;
    return __builtin_fmaf(__builtin_fmaf(-q, c, x), rc, q);
}

template <int TM, int TN, bool DIV, int ACT, bool TR>
static __device__ __forceinline__ void epilogue_tiles(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], int rbase, int cbase,
                                                      int l31)
{
    // (cbase is 64-aligned inside a tile that starts at a multiple of 64 and n_split is a multiple of 128: a wave's columns are all on one side)
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>((a.n_split > 0 && cbase >= a.n_split) ? a.bias2 : a.bias), 0, a.bias ? a.M * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res), 0, a.res ? (int)((unsigned)a.M * a.ldr * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY =
        __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, (int)((unsigned)(TR ? a.N : a.M) * a.ldy * 4u), 0x00020000);
    float bv[TM][16];                                      // loads first: Y may alias res, so program order is kept
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[i][e] = 0.f;
    if (a.bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                bv[i][e] = buf_load1(rsB, (unsigned)(rbase + i * 32 + (e & 3) + 8 * (e >> 2)) * 4u, 0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = cbase + jn * 32 + l31;
            const int row0 = rbase + i * 32;
            const unsigned r_off = col < a.N ? (unsigned)(row0 * a.ldr + col) * 4u : OOBH;
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = 0.f;
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = buf_load1(rsR, r_off + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ldr * 4), 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x = acc[i][jn][e] + bv[i][e];
                x += v[e];
                if (DIV) x = div_sqrt2f(x);
                if (ACT == 1) x = x > 0.f ? x : 0.f;
                if (ACT == 2) x = x > 0.f ? x : a.act_slope * x;
                if (ACT == 3) x = tanhf(x);
                if (ACT == 4) x = fabsf(x);
                if (ACT == 5) x = x / (1.0f + expf(-x));
                v[e] = x;
            }
            if (TR) {                                        // time-major output for the LSTM: Y[col][row], 4 rows = 16 bytes
                const bool quad = (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 && row0 + 28 <= a.M;
                const unsigned off = col < a.N ? (unsigned)(col * a.ldy + row0) * 4u : OOBH;
                if (quad) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, t), rsY, off + 32u * q, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int dr = (e & 3) + 8 * (e >> 2);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rsY,
                                                              row0 + dr < a.M ? off + 4u * dr : OOBH, 0, 0);
                    }
                }
            } else {
                const unsigned off = col < a.N ? (unsigned)(row0 * a.ldy + col) * 4u : OOBH;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rsY,
                                                          off + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ldy * 4), 0, 0);
            }
        }
    }
}

template <int TM, int TN, int AUX>
static __device__ __forceinline__ void slab_store(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], __amdgpu_buffer_rsrc_t rs,
                                                  int rbase, int cbase, int l31)
{
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = cbase + jn * 32 + l31;
            const unsigned v0 = col < a.N ? (unsigned)((rbase + i * 32) * a.N + col) * 4u : OOBH;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                // through a VGPR: storing element e of an accumulator tuple directly, hipcc 7.2 emits the tuple's
                // first register for every e
                float t = acc[i][jn][e];
                asm volatile("" : "+v"(t));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), rs,
                                                      v0 + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.N * 4), 0, AUX);
            }
        }
}

template <int TM, int TN>
static __device__ __forceinline__ void epilogue_dispatch(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], int rbase, int cbase,
                                                         int l31)
{
    // one lean copy per (divide, activation, transposed) combination the path uses
    if (a.transpose_out) epilogue_tiles<TM, TN, false, 0, true>(a, acc, rbase, cbase, l31);
    else if (a.div_sqrt2 && a.act == 0) epilogue_tiles<TM, TN, true, 0, false>(a, acc, rbase, cbase, l31);
    else if (a.div_sqrt2 && a.act == 2) epilogue_tiles<TM, TN, true, 2, false>(a, acc, rbase, cbase, l31);
    else if (a.div_sqrt2) epilogue_tiles<TM, TN, true, 1, false>(a, acc, rbase, cbase, l31);
    else if (a.act == 0) epilogue_tiles<TM, TN, false, 0, false>(a, acc, rbase, cbase, l31);
    else if (a.act == 1) epilogue_tiles<TM, TN, false, 1, false>(a, acc, rbase, cbase, l31);
    else if (a.act == 3) epilogue_tiles<TM, TN, false, 3, false>(a, acc, rbase, cbase, l31);
    else if (a.act == 4) epilogue_tiles<TM, TN, false, 4, false>(a, acc, rbase, cbase, l31);
    else if (a.act == 5) epilogue_tiles<TM, TN, false, 5, false>(a, acc, rbase, cbase, l31);
    else epilogue_tiles<TM, TN, false, 2, false>(a, acc, rbase, cbase, l31);
}

// S > 1 (split-K): this slice's raw partial sums go to its slab; splitk_reduce_kernel sums the slabs in a fixed order
// and applies the epilogue.  (Combining inside this kernel -- per-tile arrival counters, the last slice reduces -- was
// built and measured on MI355X: slower.  With agent-scope release/acquire fences the whole step went from 9.9 to
// 12.5 ms (every fence writes back / invalidates an L2), with write-through slabs and sc1 loads to 11.5 ms (the last
// arriver reads S slabs serially while the reduce kernel is wide and short), and keeping the accumulators live across
// the hand-off cost 56 VGPRs = one wave per SIMD for every launch.)
// `active` = this wave holds a result (false for the waves of a K group that already folded theirs into group 0).
template <int TM, int TN>
static __device__ __forceinline__ void epilogue(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm,
                                                int wn, int l31, int lk, int S, bool active = true)
{
    if (!active) return;
    const int rbase = m0 + wm * 32 * TM + 4 * lk;          // this lane's first row; element e adds i*32 + (e&3) + 8*(e>>2)
    const int cbase = n0 + wn * 32 * TN;
    if (S == 1) {
        epilogue_dispatch<TM, TN>(a, acc, rbase, cbase, l31);
        return;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.ws + (size_t)blockIdx.y * a.M * a.N, 0,
                                                                       (int)((unsigned)a.M * a.N * 4u), 0x00020000);
    slab_store<TM, TN, 0>(a, acc, rs, rbase, cbase, l31);
}
// (Built and measured on MI355X, not kept: issuing MFMA(activation fragment, weight fragment) so that a lane owns four
// consecutive COLUMNS of one row -- 16-byte residual loads and output stores, 34 memory instructions per 64x64 block
// instead of 160.  Each store instruction then touches 32 rows x 32 bytes instead of 2 rows x 128 bytes and the
// output-heavy shapes lost more (M128 N128000 K128: 72 -> 85 us, M64 N509440 K64 T9: 379 -> 411 us) than the small
// ones gained (M512 N1280 K512: 18.0 -> 17.4 us).)
#endif
