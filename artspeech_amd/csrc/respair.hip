// N2 (HiFi-GAN generator), the stages with 32 / 64 channels: ONE residual step of ResBlock1 (Vocoder/vocoder.py:35-42)
//
//     y = x + conv2( lrelu( conv1( lrelu(x) ) ) )          conv1: k taps, dilation d;  conv2: k taps, dilation 1
//
// as ONE launch.  As two conv GEMM launches the step moves six [C][N] tensors through HBM (operand image in, image out, image in,
// residual in, fp32 out, image out: 1.5 GB at 32 channels x 1.92 M samples) for 2 x 2 C C k N flop -- 46-164 TFLOP/s on the 32-row tile,
// bound by workgroup turnover and bytes.  Here a workgroup keeps a column tile on chip: it reads x once, writes y once.
//
//   workgroup = NW waves (4, or 8), 64 NW - 16 output columns of ONE utterance (tiles never straddle an utterance wall: the zero padding
//   of both convs is then a property of the tile's edge columns, not of (column, tap) pairs)
//   1. x[C][t0 - 8 - h1 .. t0 + 64 NW - 8 + h1) -> LeakyReLU -> (h, l) fp16 split -> LDS, in the conv GEMM's operand order
//      [k-block][part, k-half][column][8]: a tap is a column offset of a ds_read_b128, as in the images of conv_gemm_h3.hip
//   2. conv1 over the 64 NW columns from t0 - 8 (8 >= (k-1)/2 columns of lead for conv2): f16x3 products on v_mfma_f32_32x32x16_f16,
//      each wave 64 columns x all C rows; the weights are read from the conv GEMM's weight image (no second weight format), a granule
//      (a tap's two k-blocks at C = 32, one k-block of a tap at C = 64: 12 MFMAs per wave) at a time: into registers one granule ahead
//      at C = 32, through two 4 KB LDS buffers by LDS-DMA at C = 64 (see `dma` / `lda` below)
//   3. + bias, LeakyReLU, zero outside the utterance, split -> LDS over the x tile (v_permlane32_swap gives every lane whole 16-byte rows)
//   4. conv2 over 64 NW columns (the last 16 are not stored), + bias + x (+ the two other stacks' results, / 3: the stage's mean,
//      vocoder.py:104-110) -> y fp32 and / or LeakyReLU(y) as the operand image of the conv that follows
// Same arithmetic as the two launches (same split, same three products, fp32 accumulation, smallest terms first inside a k-block);
// the order of the fp32 partial sums inside a tap differs, so results agree to fp32 rounding, not bit for bit.
//
// Measured (scripts/exp/respair_bench.py, 32 x 200 frames): the conv phases run at ~80 % of what the matrix cores sustain on this
// arithmetic (-DRP_EXP_TIMING: 88 granules of 8 waves in 112 k cycles at C = 64, k = 11); what is left is that a workgroup's fill and
// epilogue (HBM) and its convs (MFMA) follow each other, and two to four workgroups per CU overlap them only in part: k = 3 steps run
// at 2.0-2.6 TB/s of x-in + y-out (the residual is read a second time, the halo columns twice), k = 11 ones at 260-320 TFLOP/s.
// Tried, same times within 3 %: three weight buffers with the DMA two granules ahead and a vmcnt(1) wait before the barrier, fragments
// double-buffered in registers, all loads of the fill in flight at once, staggering the first wave of workgroups.  Tried and SLOWER
// (scripts/exp/records/respair_persistent.diff, results equal): one persistent eight-wave workgroup per CU that loads the next tile's
// columns (and, at 32 channels, the residual) into registers while it multiplies the current one -- 237 / 307 / 375 us against 184 / 241 /
// 309 at 32 channels (k = 3 / 7 / 11), 365 / 498 / 630 against 220 / 356 / 494 at 64: its loads do run under its MFMAs, but its conv1-to-conv2
// hand-over, epilogue and tile commit no longer run under ANOTHER workgroup's MFMAs, and one workgroup's loads per CU are fewer bytes in
// flight than four workgroups' (256 registers with 12 / 81 spills).
#include "conv_gemm.h"
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void rp_lds_void;

// knock-out switch of experiment builds only (scripts/build_exp.sh NAME -DRP_EXP_NOMFMA): what bounds a step
#ifdef RP_EXP_NOMFMA
#define RP_MFMA(A, B, C) (C)
#else
#define RP_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#endif

#ifdef RP_EXP_TIMING
__device__ unsigned long long rp_times[8];     // sum over workgroups (wave 0): fill, conv1, mid, conv2, epilogue, total; [6] = workgroups
extern "C" int as_respair_debug_times(unsigned long long* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rp_times), sizeof(rp_times)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rp_times), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define RP_T(i) const unsigned long long tm##i = __builtin_readcyclecounter();
#else
#define RP_T(i)
#endif

namespace {
constexpr int RP_ML = 8;         // conv1's lead over the first output column ((k-1)/2 <= 8)
constexpr int RP_ABUF = 4096;    // one weight granule

// NW waves: 64 NW conv1 columns, 64 NW - 16 output columns per workgroup.  Four waves and two to four workgroups per CU wherever the tile
// fits twice into the 160 KB of LDS; eight waves, one workgroup per CU (152 KB) for 64 channels with a halo over 19 columns (k = 11, d = 5:
// as four waves alone on a CU that step took 879 us against 540 for d = 1, 3)
template <int C, int NW>
// (C = 32: at most 128 registers, so that four workgroups share a CU -- 33-40 KB of LDS each; with three, 144 registers: 6-8 % slower)
__global__ void __launch_bounds__(64 * NW, C == 32 && NW == 4 ? 4 : (C == 32 ? 2 : 1)) respair_kernel(const AsResPairArgs a)
{
    constexpr int KB = C / 16, MB = C / 32, PL = KB * 4;
    constexpr int NT = 64 * NW;
    constexpr int RP_MW = 64 * NW;                                       // conv1 columns of a workgroup
    constexpr int RP_OW = RP_MW - 16;                                    // output columns of a workgroup
    constexpr int RP_MWP = RP_MW + 16;                                   // columns of the conv1 tile as conv2 addresses it: MW - 1 + 8 + 8 < MWP
    constexpr int GPT = C == 32 ? 1 : KB;                               // weight granules per tap
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool WLDS = C == 64;                                       // weights through LDS (below)
    unsigned char* const abuf = smem;                                    // WLDS: two 4 KB weight granules
    unsigned char* const tile = smem + (WLDS ? 2 * RP_ABUF : 0);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int n_lo = a.col_off[b], n_hi = a.col_off[b + 1];
    const int t0 = n_lo + logical_of((int)blockIdx.x, (int)gridDim.x) * RP_OW;
    if (t0 >= n_hi) return;
    RP_T(0)
    const int half = a.k >> 1, h1 = a.dil * half;
    const int XW = RP_MW + 2 * h1;                                       // columns of the x tile
    const int X0 = t0 - RP_ML - h1;                                      // its first column
    const int G = a.k * GPT;                                             // granules per conv
    constexpr unsigned TAPB = 4u * 4u * C * 16u;                         // bytes per tap of a weight image (as_kbx(C) = 4 k-blocks)
    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w1), 0, (int)(a.k * TAPB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w2), 0, (int)(a.k * TAPB), 0x00020000);
    // The weights of a conv = G granules of 4 KB (a tap's two k-blocks at C = 32, one k-block of a tap at C = 64; four fragments per lane),
    // read from the conv GEMM's weight image.  All workgroups read the same 45-180 KB per conv: cache hits.
    //   C = 32: straight into registers, one granule ahead of the MFMAs that use them; no barrier inside a conv (the tile is read-only
    //           there) and no LDS for weights -- four workgroups (33-40 KB each) share a CU.
    //   C = 64: by LDS-DMA into two 4 KB buffers, a granule per barrier, then four ds_read_b128 per wave: every wave of a workgroup
    //           needs the whole granule, and eight waves per CU loading it separately cost 10 % (600 against 540 us at k = 11).
    auto dma = [&](int gg) {                                             // granule gg of the step (conv1's G, then conv2's G) -> buffer gg & 1
#if __HIP_DEVICE_COMPILE__
        const int g = gg < G ? gg : gg - G;
        if (NW == 4 || wave < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(gg < G ? rsW1 : rsW2, (rp_lds_void*)(abuf + (gg & 1) * RP_ABUF + wave * 1024), 16,
                                                     (unsigned)tid * 16u, g * RP_ABUF, 0, 0);
#endif
    };
    (void)dma;
    if constexpr (WLDS) dma(0);
    // offset of fragment q inside a granule.  C = 32: q = (k-block, part); C = 64: q = (32-row block, part)
    auto frag_off = [&](int q) {
        return C == 32 ? (unsigned)((((q >> 1) * 4 + (q & 1) * 2 + lk) * 32 + l31) * 16)
                       : (unsigned)((((q & 1) * 2 + lk) * C + (q >> 1) * 32 + l31) * 16);
    };
    auto lda = [&](f16x8 (&w)[4], const __amdgpu_buffer_rsrc_t& rs, int g) {
        const int soff = C == 32 ? g * (int)TAPB : g * RP_ABUF;          // (C = 64: a tap is exactly four granules)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, frag_off(q), soff, 0));
    };

    // 1. the x tile
    // x_u >= 2: x is the ConvTranspose1d's phase-major output z [u C][N / u] (vocoder.py of this package: one 3-tap conv GEMM with
    // u x C rows) and x[ch][col] = z[(col % u) C + ch][col / u] + x_bias[ch] -- as_interleave_phases_f32 folded into the six reads of
    // a stage's input (fill and residual of the three stacks' first steps); consecutive lanes then read u runs of 64 / u floats.
    const int xu = a.x_u >= 2 ? a.x_u : 1;
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((unsigned)(xu * C) * a.ldx * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsXB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x_bias), 0, (a.x_u >= 2 && a.x_bias) ? C * 4 : 0, 0x00020000);
    // byte offset of x[ch][col] (col inside the tensor)
    auto x_off = [&](int ch, int col) {
        if (xu == 1) return (unsigned)(ch * a.ldx + col) * 4u;
        const int q = col / xu, r = col - q * xu;
        return (unsigned)((r * C + ch) * a.ldx + q) * 4u;
    };
    {
        // every load of a batch of six items (48 per thread) is in flight before the first is converted: the fill is one or two memory
        // round trips of the workgroup, not one per item
        const int total = (C / 8) * XW;
        constexpr int BI = 6;
        for (int it0 = tid; it0 < total; it0 += BI * NT) {
            float v[BI][8];
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int it = it0 + i * NT;
                const int g = it / XW, c = it - g * XW;
                const int col = X0 + c;
                const bool in = it < total && col >= n_lo && col < n_hi;
                const unsigned off = in ? x_off(8 * g, col) : OOBH;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = buf_load1(rsX, off + (unsigned)(e * a.ldx) * 4u, 0);
                if (xu > 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[i][e] += buf_load1(rsXB, in ? (unsigned)(8 * g + e) * 4u : OOBH, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int it = it0 + i * NT;
                if (it < total) {
                    const int g = it / XW, c = it - g * XW;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[i][e] = v[i][e] > 0.f ? v[i][e] : a.slope * v[i][e];
                    u32x4_t h, l;
                    split2(v[i], h, l);
                    const int pl = (g >> 1) * 4 + (g & 1);
                    *reinterpret_cast<u32x4_t*>(tile + ((size_t)pl * XW + c) * 16) = h;
                    *reinterpret_cast<u32x4_t*>(tile + ((size_t)(pl + 2) * XW + c) * 16) = l;
                }
            }
        }
    }
    __syncthreads();
    RP_T(1)

    f32x16 acc[MB][2];
    auto zero = [&]() {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][cb][e] = 0.f;
    };
    // one granule: weight fragments w, activations from `tl` (TW columns per plane) at this lane's column ci (+ 32 per block), k-block kb (C = 64)
    auto mac = [&](const f16x8 (&w)[4], const unsigned char* tl, int TW, int ci, int kb64) {
        if constexpr (C == 32) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f16x8 bh[2], bl[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    bh[cb] = *reinterpret_cast<const f16x8*>(tl + ((kb * 4 + lk) * TW + ci + cb * 32) * 16);
                    bl[cb] = *reinterpret_cast<const f16x8*>(tl + ((kb * 4 + 2 + lk) * TW + ci + cb * 32) * 16);
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[0][cb] = RP_MFMA(w[2 * kb], bl[cb], acc[0][cb]);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[0][cb] = RP_MFMA(w[2 * kb + 1], bh[cb], acc[0][cb]);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[0][cb] = RP_MFMA(w[2 * kb], bh[cb], acc[0][cb]);
            }
        } else {
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                bh[cb] = *reinterpret_cast<const f16x8*>(tl + ((kb64 * 4 + lk) * TW + ci + cb * 32) * 16);
                bl[cb] = *reinterpret_cast<const f16x8*>(tl + ((kb64 * 4 + 2 + lk) * TW + ci + cb * 32) * 16);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[m][cb] = RP_MFMA(w[2 * m], bl[cb], acc[m][cb]);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[m][cb] = RP_MFMA(w[2 * m + 1], bh[cb], acc[m][cb]);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[m][cb] = RP_MFMA(w[2 * m], bh[cb], acc[m][cb]);
        }
    };
    // one conv (granules base .. base + G - 1 of the step) over the tile `tl`: lane column c0 at shift 0, `dil` columns per tap
    auto conv = [&](const __amdgpu_buffer_rsrc_t& rs, int base, const unsigned char* tl, int TW, int c0, int dil) {
        if constexpr (WLDS) {
            for (int g = 0; g < G; ++g) {
                const int gg = base + g;
                if (gg + 1 < 2 * G) dma(gg + 1);                         // (conv2's first granule behind conv1's last)
                const unsigned char* ab = abuf + (gg & 1) * RP_ABUF;
                f16x8 w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const f16x8*>(ab + frag_off(q));
                mac(w, tl, TW, c0 + dil * (g / GPT - half), g & (KB - 1));
                if (gg + 1 < 2 * G) __syncthreads();                     // granule gg + 1 has landed; buffer gg & 1 is free
            }
        } else {                                                         // two weight register sets, granules in pairs
            f16x8 w0[4], w1[4];
            lda(w0, rs, 0);
            for (int g = 0; g < G; g += 2) {
                if (g + 1 < G) lda(w1, rs, g + 1);
                mac(w0, tl, TW, c0 + dil * (g / GPT - half), g & (KB - 1));
                if (g + 1 < G) {
                    if (g + 2 < G) lda(w0, rs, g + 2);
                    mac(w1, tl, TW, c0 + dil * ((g + 1) / GPT - half), (g + 1) & (KB - 1));
                }
            }
        }
    };

    // 2. conv1: this wave's columns 64 wave .. 64 wave + 63 of the conv1 tile; column m of it reads x tile column m + h1 + shift
    zero();
    conv(rsW1, 0, tile, XW, wave * 64 + l31 + h1, a.dil);
    __syncthreads();                                                     // every wave is past its last read of the x tile

    RP_T(2)
    // 3. conv1's result as conv2's operand, over the x tile (every wave is past its last read of it)
    {
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b1), 0, a.b1 ? C * 4 : 0, 0x00020000);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float bv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) bv[e] = buf_load1(rsB, (unsigned)(m * 32 + 4 * lk + (e & 3) + 8 * (e >> 2)) * 4u, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int mc = wave * 64 + cb * 32 + l31;                // column of the conv1 tile
                const int col = t0 - RP_ML + mc;
                const bool in = col >= n_lo && col < n_hi;               // outside the utterance conv2 sees its zero padding
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float x = __builtin_fmaf(acc[m][cb][e], a.scale1, bv[e]);
                    x = x > 0.f ? x : a.slope * x;
                    v[e] = in ? x : 0.f;
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float t[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x0 = v[8 * pr + r], x1 = v[8 * pr + 4 + r];
                        // (inline asm and the s_nop: see yh_store_tile in conv_gemm.h)
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
                        t[r] = x0;
                        t[4 + r] = x1;
                    }
                    u32x4_t h, l;
                    split2(t, h, l);
                    const int g = m * 4 + 2 * pr + lk;                   // 8-row group: k-block g / 2, k-half g % 2
                    const int pl = (g >> 1) * 4 + (g & 1);
                    *reinterpret_cast<u32x4_t*>(tile + (pl * RP_MWP + mc) * 16) = h;
                    *reinterpret_cast<u32x4_t*>(tile + ((pl + 2) * RP_MWP + mc) * 16) = l;
                }
            }
        }
        if (tid < PL * (RP_MWP - RP_MW)) {                               // the columns behind conv1's: only unused outputs read them
            const int pl = tid / (RP_MWP - RP_MW), c = RP_MW + tid % (RP_MWP - RP_MW);
            const u32x4_t z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4_t*>(tile + (pl * RP_MWP + c) * 16) = z;
        }
    }
    __syncthreads();
    RP_T(3)

    // 4. conv2: output column o of the workgroup reads conv1 tile column o + 8 + shift
    zero();
    conv(rsW2, G, tile, RP_MWP, wave * 64 + l31 + RP_ML, 1);
    RP_T(4)
    {
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b2), 0, a.b2 ? C * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y ? (int)((unsigned)C * a.ldy * 4u) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(
            a.yh, 0, a.yh ? (int)(4u * 4u * ((unsigned)a.N + 1u) * 16u) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.add1), 0, a.add1 ? (int)((unsigned)C * a.ld_add * 4u) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.add2), 0, a.add2 ? (int)((unsigned)C * a.ld_add * 4u) : 0, 0x00020000);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float bv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) bv[e] = buf_load1(rsB, (unsigned)(m * 32 + 4 * lk + (e & 3) + 8 * (e >> 2)) * 4u, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int oc = wave * 64 + cb * 32 + l31, col = t0 + oc;
                const bool ok = oc < RP_OW && col < n_hi;
                const int row0 = m * 32 + 4 * lk;
                const unsigned xo = ok ? x_off(row0, col) : OOBH;
                const unsigned yo = ok ? (unsigned)(row0 * a.ldy + col) * 4u : OOBH;
                const unsigned po = ok ? (unsigned)(row0 * a.ld_add + col) * 4u : OOBH;
                float r[16], p[16], q[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) r[e] = buf_load1(rsX, xo + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ldx) * 4u, 0);
                if (xu > 1) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) r[e] += buf_load1(rsXB, ok ? (unsigned)(row0 + (e & 3) + 8 * (e >> 2)) * 4u : OOBH, 0);
                }
                if (a.add1) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        p[e] = buf_load1(rsP, po + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ld_add) * 4u, 0);
                        q[e] = buf_load1(rsQ, po + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ld_add) * 4u, 0);
                    }
                }
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float x = __builtin_fmaf(acc[m][cb][e], a.scale2, bv[e]);
                    x += r[e];
                    if (a.add1) x = ((p[e] + q[e]) + x) / a.out_div;
                    v[e] = x;
                }
                if (a.y) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rsY,
                                                              yo + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ldy) * 4u, 0, 0);
                }
                if (a.yh) {                                              // LeakyReLU(y) as the next conv's operand image (conv_gemm.h yh_store_tile)
                    const unsigned NXy = (unsigned)a.N + 1u;
                    const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float t[8];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            float x0 = v[8 * pr + rr], x1 = v[8 * pr + 4 + rr];
                            x0 = x0 > 0.f ? x0 : a.yh_slope * x0;
                            x1 = x1 > 0.f ? x1 : a.yh_slope * x1;
                            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
                            t[rr] = x0;
                            t[4 + rr] = x1;
                        }
                        u32x4_t h, l;
                        split2(t, h, l);
                        const int g = m * 4 + 2 * pr + lk;
                        const unsigned pl = (unsigned)((g >> 1) * 4 + (g & 1));
                        const unsigned off = ok ? (pl * NXy + (unsigned)col) * 16u : OOBH;
                        __builtin_amdgcn_raw_buffer_store_b128(h, rsH, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(l, rsH, off + 2u * NXy * 16u, 0, 0);
                        if (C == 32) {                                   // k-blocks 2, 3 of the image (as_kbx(32) = 4) are zero rows
                            __builtin_amdgcn_raw_buffer_store_b128(z, rsH, off + 8u * NXy * 16u, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(z, rsH, off + 10u * NXy * 16u, 0, 0);
                        }
                        if (ok && col == 0) {                            // the zero column N, once per (group, part)
                            const unsigned offz = (pl * NXy + (unsigned)a.N) * 16u;
                            __builtin_amdgcn_raw_buffer_store_b128(z, rsH, offz, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(z, rsH, offz + 2u * NXy * 16u, 0, 0);
                            if (C == 32) {
                                __builtin_amdgcn_raw_buffer_store_b128(z, rsH, offz + 8u * NXy * 16u, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(z, rsH, offz + 10u * NXy * 16u, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }
#ifdef RP_EXP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RP_T(5)
    if (tid == 0) {
        atomicAdd(&rp_times[0], tm1 - tm0); atomicAdd(&rp_times[1], tm2 - tm1); atomicAdd(&rp_times[2], tm3 - tm2);
        atomicAdd(&rp_times[3], tm4 - tm3); atomicAdd(&rp_times[4], tm5 - tm4); atomicAdd(&rp_times[5], tm5 - tm0); atomicAdd(&rp_times[6], 1ull);
    }
#endif
}

template <int C>
int lds_bytes(const AsResPairArgs& a, int nw)
{
    const int h1 = a.dil * (a.k / 2), mw = 64 * nw;
    const int cols = mw + 2 * h1 > mw + 16 ? mw + 2 * h1 : mw + 16;
    return (C == 64 ? 2 * RP_ABUF : 0) + (C / 16) * 4 * cols * 16;
}

template <int C, int NW>
int launch(const AsResPairArgs& a, hipStream_t stream)
{
    AS_LDS_OPT_IN((respair_kernel<C, NW>), 160 * 1024);
    const int tiles = as_cdiv(a.max_w, 64 * NW - 16);
    hipLaunchKernelGGL((respair_kernel<C, NW>), dim3((unsigned)((tiles + 7) & ~7), (unsigned)a.B), dim3(64 * NW), lds_bytes<C>(a, NW), stream, a);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
}  // namespace

extern "C" int as_respair_f32(const AsResPairArgs* ap, as_stream_t stream_)
{
    if (!ap) return AS_EINVAL;
    const AsResPairArgs& a = *ap;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!a.x || (!a.y && !a.yh) || a.x == a.y || !a.w1 || !a.w2 || !a.col_off) return AS_EINVAL;
    if (a.yh && ((reinterpret_cast<uintptr_t>(a.yh) & 15) != 0 || 256.0 * ((double)a.N + 1.0) >= 2147483648.0)) return AS_EINVAL;
    if (a.C != 32 && a.C != 64) return AS_EINVAL;
    if (a.k < 1 || !(a.k & 1) || a.k / 2 > RP_ML || a.dil < 1 || a.dil * (a.k / 2) > 40) return AS_EINVAL;
    const int xu = a.x_u >= 2 ? a.x_u : 1;
    if (a.x_u < 0 || a.x_u == 1 || (long)a.ldx * xu < (long)a.N || (xu > 1 && a.N % xu != 0)) return AS_EINVAL;
    if (a.B <= 0 || a.B > 65535 || a.N < 0 || a.max_w < 0 || (a.y && a.ldy < a.N)) return AS_EINVAL;
    if ((a.add1 == nullptr) != (a.add2 == nullptr) || (a.add1 && (a.ld_add < a.N || !(a.out_div > 0.f)))) return AS_EINVAL;
    // 32-bit byte offsets inside every tensor (raw buffer accesses)
    const double lim = 2147483648.0;
    if ((double)a.C * xu * a.ldx * 4.0 >= lim || (a.y && (double)a.C * a.ldy * 4.0 >= lim) || (a.add1 && (double)a.C * a.ld_add * 4.0 >= lim)) return AS_EINVAL;
    if (a.N == 0 || a.max_w == 0) return AS_OK;
    char tag[96];
    snprintf(tag, sizeof(tag), "respair C%d N%d k%d d%d%s", a.C, a.N, a.k, a.dil, a.add1 ? " mean3" : "");
    AsProfScope prof__(AS_CLS_GEMM, 2.0 * 2.0 * a.C * (double)a.C * a.k * (double)a.N, 8.0 * a.C * (double)a.N, stream, tag);
    int nw = a.C == 32 || lds_bytes<64>(a, 4) <= 80 * 1024 ? 4 : 8;      // C = 64: twice into the LDS, or one wide workgroup
    if (const char* e = getenv("AS_RESPAIR_NW")) nw = atoi(e) == 8 ? 8 : (atoi(e) == 4 ? 4 : nw);
    if (a.C == 32) return nw == 4 ? launch<32, 4>(a, stream) : launch<32, 8>(a, stream);
    return nw == 4 ? launch<64, 4>(a, stream) : launch<64, 8>(a, stream);
}
