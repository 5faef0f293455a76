// N2 (HiFi-GAN generator), the stages with 32 / 64 channels: ONE residual step of ResBlock1 (Vocoder/vocoder.py:35-42)
//
//     y = x + conv2( lrelu( conv1( lrelu(x) ) ) )          conv1: k taps, dilation d;  conv2: k taps, dilation 1
//
// as ONE launch.  As two conv GEMM launches the step moves six [C][N] tensors through HBM (operand image in, image out, image in,
// residual in, fp32 out, image out: 1.5 GB at 32 channels x 1.92 M samples) for 2 x 2 C C k N flop -- 46-164 TFLOP/s on the 32-row tile,
// bound by workgroup turnover and bytes.  Here a workgroup keeps a column tile on chip: it reads x once, writes y once.
//
//   workgroup = 4 waves, 240 output columns of one utterance (tiles never straddle an utterance wall: the zero padding of both convs is
//   then a property of the tile's edge columns, not of (column, tap) pairs)
//   1. x[C][t0 - 8 - h1 .. t0 + 248 + h1) -> LeakyReLU -> (h, l) fp16 split -> LDS, in the conv GEMM's operand order
//      [k-block][part, k-half][column][8]: a tap is a column offset of a ds_read_b128, as in the images of conv_gemm_h3.hip
//   2. conv1 over the 256 columns t0 - 8 .. t0 + 248 (8 >= (k-1)/2 columns of lead for conv2): f16x3 products on
//      v_mfma_f32_32x32x16_f16, each wave 64 columns x all C rows; the weights stream through two 4 KB LDS buffers by LDS-DMA, one
//      granule (a tap's k-blocks at C = 32, one k-block of a tap at C = 64) per barrier = 12 MFMAs per wave, read straight from the conv
//      GEMM's weight image (no second weight format)
//   3. + bias, LeakyReLU, zero outside the utterance, split -> LDS over the x tile (v_permlane32_swap gives every lane whole 16-byte rows)
//   4. conv2 over 256 columns (the last 16 are not stored), + bias + x (+ the two other stacks' results, / 3: the stage's mean,
//      vocoder.py:104-110) -> y
// Same arithmetic as the two launches (same split, same three products, fp32 accumulation, smallest terms first inside a k-block);
// the order of the fp32 partial sums inside a tap differs, so results agree to fp32 rounding, not bit for bit.
#include "conv_gemm.h"
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void rp_lds_void;

namespace {
constexpr int RP_ML = 8;         // conv1's lead over the first output column ((k-1)/2 <= 8)
constexpr int RP_ABUF = 4096;    // one weight granule

// NW waves: 64 NW conv1 columns, 64 NW - 16 output columns per workgroup.  Four waves and two to four workgroups per CU wherever the tile
// fits twice into the 160 KB of LDS; eight waves, one workgroup per CU (152 KB) for 64 channels with a halo over 19 columns (k = 11, d = 5:
// as four waves alone on a CU that step took 879 us against 540 for d = 1, 3)
template <int C, int NW>
__global__ void __launch_bounds__(64 * NW) respair_kernel(const AsResPairArgs a)
{
    constexpr int KB = C / 16, MB = C / 32, PL = KB * 4;
    constexpr int NT = 64 * NW;
    constexpr int RP_MW = 64 * NW;                                       // conv1 columns of a workgroup
    constexpr int RP_OW = RP_MW - 16;                                    // output columns of a workgroup
    constexpr int RP_MWP = RP_MW + 16;                                   // columns of the conv1 tile as conv2 addresses it: MW - 1 + 8 + 8 < MWP
    constexpr int GPT = C == 32 ? 1 : KB;                               // weight granules per tap
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const abuf = smem;
    unsigned char* const tile = smem + 2 * RP_ABUF;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int n_lo = a.col_off[b], n_hi = a.col_off[b + 1];
    const int t0 = n_lo + logical_of((int)blockIdx.x, (int)gridDim.x) * RP_OW;
    if (t0 >= n_hi) return;
    const int half = a.k >> 1, h1 = a.dil * half;
    const int XW = RP_MW + 2 * h1;                                       // columns of the x tile
    const int X0 = t0 - RP_ML - h1;                                      // its first column
    const int G = a.k * GPT;                                             // granules per conv
    constexpr unsigned TAPB = 4u * 4u * C * 16u;                         // bytes per tap of a weight image (as_kbx(C) = 4 k-blocks)
    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w1), 0, (int)(a.k * TAPB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w2), 0, (int)(a.k * TAPB), 0x00020000);
    (void)rsW1;
    (void)rsW2;
    // granule gg of the step (conv1's G, then conv2's G) -> weight buffer gg & 1: 16 bytes per thread, the image's own order
    auto dma = [&](int gg) {
#if __HIP_DEVICE_COMPILE__
        const int g = gg < G ? gg : gg - G;
        const int soff = C == 32 ? g * (int)TAPB : g * RP_ABUF;          // (C = 64: a tap is exactly four granules)
        if (NW == 4 || wave < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(gg < G ? rsW1 : rsW2, (rp_lds_void*)(abuf + (gg & 1) * RP_ABUF + wave * 1024), 16,
                                                     (unsigned)tid * 16u, soff, 0, 0);
#endif
    };
    dma(0);

    // 1. the x tile
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((unsigned)C * a.ldx * 4u), 0x00020000);
    {
        const int total = (C / 8) * XW;
#pragma unroll 2
        for (int it = tid; it < total; it += NT) {
            const int g = it / XW, c = it - g * XW;
            const int col = X0 + c;
            const unsigned off = (col >= n_lo && col < n_hi) ? (unsigned)(8 * g * a.ldx + col) * 4u : OOBH;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = buf_load1(rsX, off + (unsigned)(e * a.ldx) * 4u, 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : a.slope * v[e];
            u32x4_t h, l;
            split2(v, h, l);
            const int pl = (g >> 1) * 4 + (g & 1);
            *reinterpret_cast<u32x4_t*>(tile + ((size_t)pl * XW + c) * 16) = h;
            *reinterpret_cast<u32x4_t*>(tile + ((size_t)(pl + 2) * XW + c) * 16) = l;
        }
    }
    __syncthreads();

    f32x16 acc[MB][2];
    auto zero = [&]() {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][cb][e] = 0.f;
    };
    // one granule: weights from buffer gg & 1, activations from `tl` (TW columns per plane) at this lane's column ci (+ 32 per block)
    auto mac = [&](int gg, const unsigned char* tl, int TW, int ci) {
        const unsigned char* ab = abuf + (gg & 1) * RP_ABUF;
        if constexpr (C == 32) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(ab + ((kb * 4 + lk) * 32 + l31) * 16);
                const f16x8 al = *reinterpret_cast<const f16x8*>(ab + ((kb * 4 + 2 + lk) * 32 + l31) * 16);
                f16x8 bh[2], bl[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    bh[cb] = *reinterpret_cast<const f16x8*>(tl + ((kb * 4 + lk) * TW + ci + cb * 32) * 16);
                    bl[cb] = *reinterpret_cast<const f16x8*>(tl + ((kb * 4 + 2 + lk) * TW + ci + cb * 32) * 16);
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[cb], acc[0][cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[cb], acc[0][cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[cb], acc[0][cb], 0, 0, 0);
            }
        } else {
            const int g = gg < G ? gg : gg - G;
            const int kb = g & (KB - 1);
            f16x8 ah[MB], al[MB], bh[2], bl[2];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                ah[m] = *reinterpret_cast<const f16x8*>(ab + (lk * C + m * 32 + l31) * 16);
                al[m] = *reinterpret_cast<const f16x8*>(ab + ((2 + lk) * C + m * 32 + l31) * 16);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                bh[cb] = *reinterpret_cast<const f16x8*>(tl + ((kb * 4 + lk) * TW + ci + cb * 32) * 16);
                bl[cb] = *reinterpret_cast<const f16x8*>(tl + ((kb * 4 + 2 + lk) * TW + ci + cb * 32) * 16);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[m][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[cb], acc[m][cb], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[m][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[cb], acc[m][cb], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[m][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[cb], acc[m][cb], 0, 0, 0);
        }
    };

    // 2. conv1: this wave's columns 64 wave .. 64 wave + 63 of the conv1 tile; column m of it reads x tile column m + h1 + shift
    zero();
    const int c1 = wave * 64 + l31 + h1;
    for (int gg = 0; gg < G; ++gg) {
        dma(gg + 1);                                                     // (gg + 1 <= G: conv2's first granule behind conv1's last)
        mac(gg, tile, XW, c1 + a.dil * (gg / GPT - half));
        __syncthreads();                                                 // the granule gg + 1 has landed; buffer gg & 1 is free
    }

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

    // 4. conv2: output column o of the workgroup reads conv1 tile column o + 8 + shift
    zero();
    const int c2 = wave * 64 + l31 + RP_ML;
    for (int gg = G; gg < 2 * G; ++gg) {
        if (gg + 1 < 2 * G) dma(gg + 1);
        mac(gg, tile, RP_MWP, c2 + ((gg - G) / GPT - half));
        if (gg + 1 < 2 * G) __syncthreads();
    }
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
                const unsigned xo = ok ? (unsigned)(row0 * a.ldx + col) * 4u : OOBH;
                const unsigned yo = ok ? (unsigned)(row0 * a.ldy + col) * 4u : OOBH;
                const unsigned po = ok ? (unsigned)(row0 * a.ld_add + col) * 4u : OOBH;
                float r[16], p[16], q[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) r[e] = buf_load1(rsX, xo + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ldx) * 4u, 0);
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
}

template <int C>
int lds_bytes(const AsResPairArgs& a, int nw)
{
    const int h1 = a.dil * (a.k / 2), mw = 64 * nw;
    const int cols = mw + 2 * h1 > mw + 16 ? mw + 2 * h1 : mw + 16;
    return 2 * RP_ABUF + (C / 16) * 4 * cols * 16;
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
    if (a.B <= 0 || a.B > 65535 || a.N < 0 || a.max_w < 0 || a.ldx < a.N || (a.y && a.ldy < a.N)) return AS_EINVAL;
    if ((a.add1 == nullptr) != (a.add2 == nullptr) || (a.add1 && (a.ld_add < a.N || !(a.out_div > 0.f)))) return AS_EINVAL;
    // 32-bit byte offsets inside every tensor (raw buffer accesses)
    const double lim = 2147483648.0;
    if ((double)a.C * a.ldx * 4.0 >= lim || (a.y && (double)a.C * a.ldy * 4.0 >= lim) || (a.add1 && (double)a.C * a.ld_add * 4.0 >= lim)) return AS_EINVAL;
    if (a.N == 0 || a.max_w == 0) return AS_OK;
    char tag[96];
    snprintf(tag, sizeof(tag), "respair C%d N%d k%d d%d%s", a.C, a.N, a.k, a.dil, a.add1 ? " mean3" : "");
    AsProfScope prof__(AS_CLS_GEMM, 2.0 * 2.0 * a.C * (double)a.C * a.k * (double)a.N, 8.0 * a.C * (double)a.N, stream, tag);
    if (a.C == 32) return launch<32, 4>(a, stream);
    int nw = lds_bytes<64>(a, 4) <= 80 * 1024 ? 4 : 8;                   // twice into the LDS, or one wide workgroup
    if (const char* e = getenv("AS_RESPAIR_NW")) nw = atoi(e) == 8 ? 8 : (atoi(e) == 4 ? 4 : nw);
    return nw == 4 ? launch<64, 4>(a, stream) : launch<64, 8>(a, stream);
}
