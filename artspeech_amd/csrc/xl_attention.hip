// EMA_Predictor: RelativeMultiHeadAttention.forward (Utils/EMA/conformer/attention.py:77-119) on the matrix
// cores.  conformer.hip's xl_attention_kernel is the same attention in exact fp32 on the vector ALU (the reference this kernel is tested
// against, and what a head width other than 64 would need): 335 us per call at 32 x 200 frames -- 2.6 GFLOP at 7.8 TFLOP/s, its 16-query
// workgroups re-stage every key / value / position tile 13 times.  Here, with e = T - 1 - i + j (T frames, query i, key j):
//
//   score[i][j] = ( (q_i + u) . k_j  +  pos[i][j] ) / sqrt(d_model)
//   pos[i][j]   = (q_i     + v) . p_e            j <= i        (keys at or before the query)
//               = 0                              j == i + 1
//               = (q_{i+1} + v) . p_{j-i-2}      j >= i + 2    (the rows the reference's _relative_shift wraps into, :111-119)
//
// TRANSPOSED, keys as rows: a wave owns 31 queries (columns i0 .. i0 + 30 of its 32-column tiles; column 31 = query i0 + 31 is there only
// because the wrapped entries of query i0 + 30 need it) and walks the utterance's keys 32 at a time:
//   S^T  [32 keys][32 queries]  = K_tile . (q + u)        K rows and q columns are 16-byte rows of the projection GEMM's OPERAND IMAGE
//   Ra^T [64 e   ][32 queries]  = P[e0 ..] . (q + v)      (conv_gemm_h3.hip), read from global memory straight into MFMA fragments:
//   Rb^T [64 c   ][32 queries]  = P[c0 ..] . (q + v)      f16x3 products, fp32-accurate, no staging and no conversion
//   pos^T[j][i] = Ra^T[j - i + 31][i]  or  Rb^T[j - i + 30][i + 1]: a per-column shift, through 8 KB of LDS per wave (the read of lane i
//   is 31 floats from lane i - 1's: conflict-free)
// With queries as COLUMNS the softmax statistics of a query live in one lane (rows = registers; the other 16 keys of the tile sit in lane
// + 32: one shuffle), the running output O^T [64 d][32 queries] is rescaled by a per-lane scalar, and the probabilities become the B
// operand of O^T += V^T . P^T by the v_permlane32_swap that conv_gemm.h's yh_store_tile uses.  V^T rows (8 consecutive keys of a
// channel) come from the fp32 copy of the projection.  u and v are folded into the projection's bias (ema.py stacks the query weights
// twice: rows q + u, q + v, k, v), so (q + u) . k is computed as the reference computes it.  No barrier: waves are independent.
#include "conv_gemm.h"
#define AS_FILE_CLS AS_CLS_ATTN

namespace {
#define XM_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)

struct XlArgs {
    const uint16_t* qh;     // operand image of the [4 C][N] projection (rows q + u, q + v, k, v)
    const uint16_t* ph;     // operand image of pos [C][N]
    const float* qkv;       // the projection's fp32 copy (its value rows are read)
    int ld, C, N;
    float inv_scale;
    const int* col_off;
    float* out;             // fp32 [C][N], or NULL
    int ldo;
    uint16_t* oh;           // the result as the operand image of the out-projection GEMM, or NULL
};

static __device__ __forceinline__ f16x8 ld_frag(__amdgpu_buffer_rsrc_t rs, unsigned off)
{
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}

__global__ void __launch_bounds__(256) xl_attention_mfma_kernel(const XlArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float xsm[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int o0 = a.col_off[b], T = a.col_off[b + 1] - o0;
    const int i0 = 31 * ((int)blockIdx.x * 4 + wave);
    if (i0 >= T) return;                                                 // (no barrier below: a wave leaves alone)
    float* sc = xsm + wave * (64 * 32);                                  // this wave's shift scratch [64][32]
    const unsigned NX = (unsigned)a.N + 1u;                              // columns of an image plane (the last one is zero)
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.qh), 0, (int)((unsigned)as_kbx(4 * a.C) * 4u * NX * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.ph), 0, (int)((unsigned)as_kbx(a.C) * 4u * NX * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.qkv), 0, (int)((unsigned)(4 * a.C) * a.ld * 4u), 0x00020000);
    // image offset of (k-block kb of a 64-channel group starting at channel c0, part p, this lane's k-half, column col)
    auto img = [&](int c0, int kb, int p, unsigned col) { return ((unsigned)(((c0 >> 4) + kb) * 4 + p * 2 + lk) * NX + col) * 16u; };
    const unsigned zc = (unsigned)a.N;                                   // the zero column

    // the wave's queries as B operands: columns i0 + l31 of the rows q + u (content) and q + v (positions)
    f16x8 qu[4][2], qv[4][2];
    {
        const unsigned col = i0 + l31 < T ? (unsigned)(o0 + i0 + l31) : zc;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                qu[kb][p] = ld_frag(rsQ, img(h * 64, kb, p, col));
                qv[kb][p] = ld_frag(rsQ, img(a.C + h * 64, kb, p, col));
            }
    }
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int ii = l31, i = i0 + ii;

    // A (rows of the image `rs` at channel group c0, row r = column `col` of the image) . B (q) over the 64 channels of the head
    auto prod = [&](__amdgpu_buffer_rsrc_t rs, int c0, unsigned col, const f16x8 (&q)[4][2]) {
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        f16x8 ah[4], al[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            ah[kb] = ld_frag(rs, img(c0, kb, 0, col));
            al[kb] = ld_frag(rs, img(c0, kb, 1, col));
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {                                 // smallest terms first, as the conv GEMM
            s = XM_MFMA(ah[kb], q[kb][1], s);
            s = XM_MFMA(al[kb], q[kb][0], s);
            s = XM_MFMA(ah[kb], q[kb][0], s);
        }
        return s;
    };

    for (int j0 = 0; j0 < T; j0 += 32) {
        // content: rows = keys j0 + l31
        const f32x16 s = prod(rsQ, 2 * a.C + h * 64, j0 + l31 < T ? (unsigned)(o0 + j0 + l31) : zc, qu);
        float pa[16], pb[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) pa[e] = pb[e] = 0.f;
        if (j0 <= i0 + 30) {                                             // some key of the tile is at or before some query: e = T - 1 - i + j
            const int e0 = T - 1 - i0 - 31 + j0;                         // row 0 of the window; entry (i, j) sits in row jj - ii + 31
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int c = e0 + rb * 32 + l31;
                const f32x16 r = prod(rsP, h * 64, (c >= 0 && c < T) ? (unsigned)(o0 + c) : zc, qv);
#pragma unroll
                for (int e = 0; e < 16; ++e) sc[(rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + l31] = r[e];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; ++e) pa[e] = sc[((e & 3) + 8 * (e >> 2) + 4 * lk - ii + 31) * 32 + ii];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (j0 + 31 >= i0 + 2) {                                         // some key is two or more past some query: row c = j - i - 2 of query i + 1
            const int c0 = j0 - i0 - 32;                                 // row 0 of the window; entry (i, j) sits in row jj - ii + 30, column ii + 1
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int c = c0 + rb * 32 + l31;
                const f32x16 r = prod(rsP, h * 64, (c >= 0 && c < T) ? (unsigned)(o0 + c) : zc, qv);
#pragma unroll
                for (int e = 0; e < 16; ++e) sc[(rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + l31] = r[e];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int i1 = ii < 31 ? ii + 1 : 31;                        // (column 31's own entries are never stored)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lk - ii + 30;
                pb[e] = sc[(row < 0 ? 0 : row) * 32 + i1];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        // scores of this lane's query for its 16 keys; online softmax (the other 16 keys: lane ^ 32)
        float p[16], tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = j0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
            const float pos = j <= i ? pa[e] : (j == i + 1 ? 0.f : pb[e]);
            p[e] = j < T ? (s[e] + pos) * a.inv_scale : -INFINITY;
            tmax = fmaxf(tmax, p[e]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);                                 // (finite: key j0 exists)
        const float corr = expf(m - mn);                                 // exp(-inf) = 0 on the first tile
        float tsum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            p[e] = expf(p[e] - mn);                                      // exp(-inf) = 0 past the utterance
            tsum += p[e];
        }
        tsum += __shfl_xor(tsum, 32);
        l = l * corr + tsum;
        m = mn;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] *= corr;
        // the probabilities as B operand (k = keys): v_permlane32_swap hands every lane 8 consecutive keys (conv_gemm.h yh_store_tile)
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float t[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x0 = p[8 * pr + r], x1 = p[8 * pr + 4 + r];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
                t[r] = x0;
                t[4 + r] = x1;
            }
            u32x4_t hh, ll;
            split2(t, hh, ll);
            ph[pr] = __builtin_bit_cast(f16x8, hh);
            pl[pr] = __builtin_bit_cast(f16x8, ll);
        }
        // O^T += V^T . P^T: A rows = channels d, k = keys: 8 consecutive keys of a channel from the fp32 value rows
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const unsigned row = (unsigned)(3 * a.C + h * 64 + rb * 32 + l31);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int j = j0 + 16 * kb + 8 * lk;
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = buf_load1(rsV, j + r < T ? (row * (unsigned)a.ld + (unsigned)(o0 + j + r)) * 4u : OOBH, 0);
                u32x4_t hh, ll;
                split2(v, hh, ll);
                const f16x8 vh = __builtin_bit_cast(f16x8, hh), vl = __builtin_bit_cast(f16x8, ll);
                acc[rb] = XM_MFMA(vh, pl[kb], acc[rb]);
                acc[rb] = XM_MFMA(vl, ph[kb], acc[rb]);
                acc[rb] = XM_MFMA(vh, ph[kb], acc[rb]);
            }
        }
    }
    const bool mine = ii < 31 && i < T;
    const float inv = 1.0f / l;
    if (a.out && mine) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                a.out[(size_t)(h * 64 + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk) * a.ldo + o0 + i] = acc[rb][e] * inv;
    }
    if (a.oh) {                                                          // 16-byte rows of 8 consecutive channels (conv_gemm.h yh_store_tile)
        const __amdgpu_buffer_rsrc_t rsO =
            __builtin_amdgcn_make_buffer_rsrc(a.oh, 0, (int)((unsigned)as_kbx(a.C) * 4u * NX * 16u), 0x00020000);
        const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float t[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x0 = acc[rb][8 * pr + r] * inv, x1 = acc[rb][8 * pr + 4 + r] * inv;
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
                    t[r] = x0;
                    t[4 + r] = x1;
                }
                u32x4_t hh, ll;
                split2(t, hh, ll);
                const int ch = h * 64 + rb * 32 + 16 * pr + 8 * lk;      // first of this lane's 8 channels
                const unsigned pl = (unsigned)((ch >> 4) * 4 + ((ch >> 3) & 1));
                const unsigned off = mine ? (pl * NX + (unsigned)(o0 + i)) * 16u : OOBH;
                __builtin_amdgcn_raw_buffer_store_b128(hh, rsO, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(ll, rsO, off + 2u * NX * 16u, 0, 0);
                if (b == 0 && i == 0) {                                  // the image's zero column, once per (channel group, part)
                    __builtin_amdgcn_raw_buffer_store_b128(z, rsO, (pl * NX + zc) * 16u, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(z, rsO, ((pl + 2u) * NX + zc) * 16u, 0, 0);
                }
            }
    }
}
}  // namespace

extern "C" int as_xl_attention_image_f32(const float* qkv, int ld, const uint16_t* qkv_h, const uint16_t* pos_h, int n_total, int C, int heads,
                                         float inv_scale, const int32_t* col_off, int B, int max_len, float* out, int ldo, uint16_t* out_h,
                                         as_stream_t stream)
{
    if (!qkv || !qkv_h || !pos_h || !col_off || (!out && !out_h) || C <= 0 || heads <= 0 || C != heads * 64 || B < 0 || B > 65535 || n_total < 0) return AS_EINVAL;
    if (ld < n_total || (out && ldo < n_total) || (reinterpret_cast<uintptr_t>(out_h) & 15) != 0) return AS_EINVAL;
    if (16.0 * as_kbx(4 * C) * 4.0 * ((double)n_total + 1.0) >= 2147483648.0 || 16.0 * C * (double)ld >= 2147483648.0) return AS_EINVAL;
    if (B == 0 || max_len <= 0) return AS_OK;
    XlArgs a;
    a.qh = qkv_h; a.ph = pos_h; a.qkv = qkv; a.ld = ld; a.C = C; a.N = n_total; a.inv_scale = inv_scale;
    a.col_off = col_off; a.out = out; a.ldo = ldo; a.oh = out_h;
    AsProfScope prof__(AS_FILE_CLS, 2.0 * 4.0 * 64.0 * heads * (double)max_len * max_len * B, 0, (hipStream_t)stream, "xl_attention_image");
    const int waves = as_cdiv(max_len, 31);
    hipLaunchKernelGGL(xl_attention_mfma_kernel, dim3(as_cdiv(waves, 4), heads, B), dim3(256), 4 * 64 * 32 * sizeof(float), (hipStream_t)stream, a);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
