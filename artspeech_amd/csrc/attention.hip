// K5 -- windowed relative-position self-attention (Glow-TTS style), replacing
// MultiHeadAttention.attention + the pad/reshape "skew" helpers (RelTransformerEnc.py:138-233).
//
//   scores[i][j] = q_i.k_j / sqrt(dk) + [|j-i| <= w] q_i.Ek[j-i+w] / sqrt(dk)
//   p = softmax_j(scores)                    (keys of the SAME utterance only)
//   out_i = sum_j p_ij v_j + sum_{|j-i|<=w} p_ij Ev[j-i+w]
//
// The reference zero-pads the (2w+1)-row tables to 2N-1 rows and multiplies densely (O(N^2 d) of
// zeros) and materialises [B,h,N,N] scores; here the relative terms are the +-w band they equal
//  and the softmax is computed online over 64-key tiles staged in LDS, so
// nothing of size N^2 ever exists.  q/k/v come from one fused [3C][N] projection GEMM.
//
// Two kernels:
//   relpos_attention_kernel        exact fp32 on the vector ALU, any head width <= 128, fp32 q/k/v rows (as_relpos_attention_groups_f32:
//                                  what a configuration with other head widths runs, and the reference the matrix-core kernel is
//                                  tested against).  Workgroup = (utterance, head, 16-query tile), 4 waves x 4 queries; QK^T: lane = key
//                                  (K tile rows padded to 65 floats -> conflict-free), softmax statistics by wave shuffles; PV: lane = channel.
//   relpos_attention_image_kernel  the path (128-channel heads): matrix cores, operands straight from the q/k/v GEMM's operand image
//                                  (as_relpos_attention_image_f32; described above the kernel).
#include <cstring>
#include "common.h"
#include "artspeech_hip.h"
#define AS_FILE_CLS AS_CLS_ATTN

#define QT 16
#define KT 64
#define KPAD 65
#define MAXDK 128
#define MAXREL 9

static __device__ __forceinline__ float wmax(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
static __device__ __forceinline__ float wsum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256)
relpos_attention_kernel(const float* __restrict__ qkv, int ld, int C, int heads, int window,
                        const float* __restrict__ emb_k1, const float* __restrict__ emb_v1, const float* __restrict__ emb_k2,
                        const float* __restrict__ emb_v2, int b_split, const int* __restrict__ col_off, float* __restrict__ out, int ldo)
{
    const int grp = emb_k2 ? (int)blockIdx.z / b_split : 0;             // utterance group g: tables first + g (second - first)
    const float* emb_k = emb_k1 + (ptrdiff_t)grp * (emb_k2 - emb_k1);
    const float* emb_v = emb_v1 + (ptrdiff_t)grp * (emb_v2 - emb_v1);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dk = C / heads;
    const int nrel = 2 * window + 1;
    float* Ks = sm;                              // [dk][KPAD]
    float* Vs = Ks + dk * KPAD;                  // [dk][KPAD]
    float* Qs = Vs + dk * KPAD;                  // [QT][dk]
    float* Ps = Qs + QT * dk;                    // [4 waves][KT][4]
    float* Rk = Ps + 4 * KT * 4;                 // [QT][MAXREL]
    float* Ek = Rk + QT * MAXREL;                // [nrel][dk]
    float* Ev = Ek + MAXREL * dk;                // [nrel][dk]

    const int b = blockIdx.z, h = blockIdx.y;
    const int o0 = col_off[b], N = col_off[b + 1] - o0;
    const int q0 = blockIdx.x * QT;
    if (q0 >= N) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float scale = sqrtf((float)dk);

    const float* Qg = qkv + (size_t)(h * dk) * ld + o0;
    const float* Kg = qkv + (size_t)(C + h * dk) * ld + o0;
    const float* Vg = qkv + (size_t)(2 * C + h * dk) * ld + o0;

    for (int i = tid; i < QT * dk; i += 256) {
        const int q = i % QT, d = i / QT;
        const int qi = q0 + q;
        Qs[q * dk + d] = qi < N ? Qg[(size_t)d * ld + qi] : 0.f;
    }
    for (int i = tid; i < nrel * dk; i += 256) { Ek[i] = emb_k[i]; Ev[i] = emb_v[i]; }
    __syncthreads();
    for (int i = tid; i < QT * nrel; i += 256) {
        const int q = i / nrel, r = i - q * nrel;
        float s = 0.f;
        for (int d = 0; d < dk; ++d) s += Qs[q * dk + d] * Ek[r * dk + d];
        Rk[q * MAXREL + r] = s / scale;
    }

    float m[4], l[4], acc[4][2];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) { m[qq] = -INFINITY; l[qq] = 0.f; acc[qq][0] = 0.f; acc[qq][1] = 0.f; }
    float* Pw = Ps + wave * KT * 4;
    const int d0 = lane, d1 = lane + 64;

    for (int k0 = 0; k0 < N; k0 += KT) {
        __syncthreads();                                    // previous tile fully consumed (and Rk visible)
        for (int i = tid; i < dk * KT; i += 256) {
            const int jj = i % KT, d = i / KT;
            const int kj = k0 + jj;
            const bool ok = kj < N;
            Ks[d * KPAD + jj] = ok ? Kg[(size_t)d * ld + kj] : 0.f;
            Vs[d * KPAD + jj] = ok ? Vg[(size_t)d * ld + kj] : 0.f;
        }
        __syncthreads();
        // ---- scores: lane = key
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const float* qb = Qs + (wave * 4) * dk;
        for (int d = 0; d < dk; ++d) {
            const float kv = Ks[d * KPAD + lane];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) s[qq] += qb[qq * dk + d] * kv;
        }
        const int kj = k0 + lane;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int qi = q0 + wave * 4 + qq;
            float sc = s[qq] / scale;
            const int r = kj - qi + window;
            if (r >= 0 && r < nrel) sc += Rk[(wave * 4 + qq) * MAXREL + r];
            if (kj >= N) sc = -INFINITY;
            const float mn = fmaxf(m[qq], wmax(sc));
            const float p = (kj < N) ? expf(sc - mn) : 0.f;
            const float corr = expf(m[qq] - mn);           // exp(-inf) = 0 on the first tile
            l[qq] = l[qq] * corr + wsum(p);
            acc[qq][0] *= corr;
            acc[qq][1] *= corr;
            m[qq] = mn;
            Pw[lane * 4 + qq] = p;
        }
        __syncthreads();
        // ---- PV: lane = channel (d0, d1)
        const int jn = (N - k0) < KT ? (N - k0) : KT;
        if (d0 < dk) {
            const bool two = d1 < dk;
            for (int jj = 0; jj < jn; ++jj) {
                const float4 p4 = *reinterpret_cast<const float4*>(Pw + jj * 4);
                const float v0 = Vs[d0 * KPAD + jj];
                const float v1 = two ? Vs[d1 * KPAD + jj] : 0.f;
                acc[0][0] += p4.x * v0; acc[1][0] += p4.y * v0; acc[2][0] += p4.z * v0; acc[3][0] += p4.w * v0;
                acc[0][1] += p4.x * v1; acc[1][1] += p4.y * v1; acc[2][1] += p4.z * v1; acc[3][1] += p4.w * v1;
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int qi = q0 + wave * 4 + qq;
                for (int r = 0; r < nrel; ++r) {
                    const int jg = qi + r - window;
                    if (jg < k0 || jg >= k0 + jn || jg < 0) continue;
                    const float p = Pw[(jg - k0) * 4 + qq];
                    acc[qq][0] += p * Ev[r * dk + d0];
                    if (two) acc[qq][1] += p * Ev[r * dk + d1];
                }
            }
        }
    }
    __syncthreads();
    // ---- normalise, transpose through LDS (Ks is free now), store rows of 16 consecutive frames
    float* Os = Ks;                                         // [dk][QT]
    if (d0 < dk) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            Os[d0 * QT + wave * 4 + qq] = acc[qq][0] / l[qq];
            if (d1 < dk) Os[d1 * QT + wave * 4 + qq] = acc[qq][1] / l[qq];
        }
    }
    __syncthreads();
    for (int i = tid; i < dk * QT; i += 256) {
        const int q = i % QT, d = i / QT;
        if (q0 + q < N) out[(size_t)(h * dk + d) * ldo + o0 + q0 + q] = Os[d * QT + q];
    }
}

#include "conv_gemm.h"
#define FK 64                                   // keys per tile
// ---------------------------------------------------------------------------------------------------
// The same attention fed by the q/k/v GEMM's OPERAND IMAGE (include/artspeech_hip.h: [k-block][plane][column][8 fp16], x = h + l): the image
// is already in MFMA fragment order, so
//   Q fragments are 16-byte global loads straight into registers (nothing staged, nothing converted),
//   K tiles go global -> LDS by LDS-DMA (one 1 KB `buffer_load_dwordx4 ... lds` per plane and k-block, double buffered),
//   only V (contraction over keys: needs key-contiguous rows) is read from the fp32 result and split while it is staged,
// and the result can be written as the o-projection's operand image.  (A first matrix-core version staged Q, K, V as fp32 rows of one
// utterance -- at 40 tokens ~250 dependent 160-byte loads per thread and most of its 38 us; this one does 16 + 8 + 16 wide loads.)
// Same arithmetic as the conv GEMMs: every fp32 operand is h + l (two fp16 values, 22 bits), a product is h.l + l.h + h.h on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation.
//   workgroup = (utterance, head, NW x 32 queries); a wave owns 32 queries for the whole key loop, keys come in tiles of 64.
//   S^T = K . Q^T   (A = K tile [32 keys][16 d], B = Q^T [16 d][32 queries]): a lane ends up with ONE query (column lane & 31)
//                   and 16 of the tile's 32 keys -> the online softmax is per-lane arithmetic plus one exchange with lane ^ 32.
//   O^T += V^T . P^T (A = V^T [32 d][16 keys], B = P^T [16 keys][32 queries]): the S^T accumulator registers ARE the B operand
//                   (after exp and the h/l split); the contraction visits the keys in the accumulator's row order, so V is
//                   staged in LDS in exactly that order and nothing is shuffled between the two products.
//   The two 9-row relative-position tables are matrix-core operands too (built once per workgroup in LDS): q . Ek[r] is one more
//   "key" block, sum_r p_r Ev[r] one more 16-deep contraction block in the <= 3 key blocks that intersect a wave's band.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void attn_lds_void;
struct AttnImageArgs {
    const float* qkv;            // fp32 [3C][ld]  (rows 2C.. = V)
    const uint16_t* qkv_h;       // image of the same [3C][n_total] matrix
    const float *ek1, *ev1, *ek2, *ev2;
    const int* col_off;
    float* out;                  // fp32 [C][ldo] or null
    uint16_t* out_h;             // image [C][n_total] or null
    int ld, C, window, b_split, n_total, ldo;
};

template <int NW>
__global__ void __launch_bounds__(NW * 64)
relpos_attention_image_kernel(const AttnImageArgs a)
{
    constexpr int DK = 128, KB = DK / 16, NT = NW * 64, NQ = NW * 32;
#ifdef ATTN_DBG
    long long tmark[16];
    int nmark = 0;
#define ATTN_MARK() tmark[nmark++] = __builtin_readcyclecounter()
#else
#define ATTN_MARK()
#endif
    ATTN_MARK();
    const int grp = a.ek2 ? (int)blockIdx.z / a.b_split : 0;
    const float* emb_k = a.ek1 + (ptrdiff_t)grp * (a.ek2 - a.ek1);
    const float* emb_v = a.ev1 + (ptrdiff_t)grp * (a.ev2 - a.ev1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    u32x4_t* Kt = reinterpret_cast<u32x4_t*>(smraw);                    // [2 buffers][p][kb][g][key 64]   (2 x 32 KB)
    u32x4_t* Vt = Kt + 2 * (2 * KB * 2 * FK);                           // [p][c][g][d 128]                (32 KB)
    u32x4_t* EkI = Vt + 2 * 4 * 2 * DK;                                 // [p][kb][g][32 rows r][8 d]: Ek as an A operand (rows >= nrel zero)  16 KB
    u32x4_t* EvT = EkI + 2 * KB * 2 * 32;                               // [p][g][d 128][8 r]: Ev^T as an A operand                             8 KB
    float* Rk = reinterpret_cast<float*>(EvT + 2 * 2 * DK);             // [NQ][9]    q . Ek[r] * log2(e)/sqrt(dk)
    float* Pb = Rk + NQ * MAXREL;                                       // [NQ][16]   the band's probabilities of the current block

    const int b = blockIdx.z, h = blockIdx.y;
    const int o0 = a.col_off[b], N = a.col_off[b + 1] - o0;
    const int q0 = blockIdx.x * NQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5, lq = lane & 31;
    const int nrel = 2 * a.window + 1, window = a.window;
    const size_t NX = (size_t)a.n_total + 1;
    if (a.out_h && blockIdx.x == 0 && b == 0 && tid < KB * 4) {         // the image's zero column, this head's k-blocks
        reinterpret_cast<u32x4_t*>(a.out_h)[((size_t)(h * KB) * 4 + tid) * NX + a.n_total] = u32x4_t{0u, 0u, 0u, 0u};
    }
    if (q0 >= N) return;
    const float cs = 1.44269504088896340736f / sqrtf((float)DK);
    const u32x4_t* img = reinterpret_cast<const u32x4_t*>(a.qkv_h);
    const int qkb0 = (h * DK) / 16, kkb0 = (a.C + h * DK) / 16;         // first k-block of this head's Q / K rows
    const float* Vg = a.qkv + (size_t)(2 * a.C + h * DK) * a.ld + o0;

    // ---- Q fragments: lane = (query, k half), columns past the image's end read its zero column
    const int ql = wave * 32 + lq, qi = q0 + ql, qw0 = q0 + wave * 32;
    const size_t qcol = (size_t)min(o0 + qi, a.n_total);
    u32x4_t Qh[KB], Ql[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        Qh[kb] = img[((size_t)(qkb0 + kb) * 4 + g) * NX + qcol];
        Ql[kb] = img[((size_t)(qkb0 + kb) * 4 + 2 + g) * NX + qcol];
    }
    // K tile k0 -> buffer `buf`: chunk (p, kb, g') = 64 keys x 16 bytes, contiguous in the image and in LDS
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.qkv_h), 0,
                                                                         (int)((unsigned)(3 * a.C / 16) * 4u * (unsigned)NX * 16u), 0x00020000);
    (void)rsK;
    (void)kkb0;
    auto issue_k = [&](int k0, int buf) {
#if __HIP_DEVICE_COMPILE__
#pragma unroll
        for (int i = 0; i < 2 * KB * 2 / NW; ++i) {
            const int ch = wave + NW * i, pp = ch / (KB * 2), kb = (ch / 2) % KB, gg = ch & 1;
            const unsigned voff = (unsigned)((((kkb0 + kb) * 4 + pp * 2 + gg) * NX + (size_t)(o0 + k0 + lane)) * 16u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (attn_lds_void*)(smraw + (size_t)buf * (2 * KB * 2 * FK * 16) + ch * (FK * 16)), 16, voff, 0,
                                                     0, 0);
        }
#endif
    };
    // V tile: (d, 4 consecutive keys) per item, 16-byte loads where the four keys exist
    constexpr int VI = DK * (FK / 4) / NT;
    float vr[VI][4];
    auto load_v = [&](int k0) {
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            const int idx = tid + it * NT, kq = idx % (FK / 4), d = idx / (FK / 4), kk = k0 + kq * 4;
            const float* row = Vg + (size_t)d * a.ld;
            // loads only -- nothing here may consume a loaded value, or every item becomes its own memory round trip (measured: 16 x 460 cycles);
            // keys past the utterance are zeroed when the tile is stored
            if (o0 + k0 + FK + 3 < a.ld) {                              // (workgroup-uniform) every quad of the tile stays inside its row
                const f32x4 v4 = *reinterpret_cast<const f32x4 __attribute__((aligned(4)))*>(row + kk);
                vr[it][0] = v4[0]; vr[it][1] = v4[1]; vr[it][2] = v4[2]; vr[it][3] = v4[3];
            } else {                                                    // the batch's last columns: element loads, clamped to the utterance
#pragma unroll
                for (int e = 0; e < 4; ++e) vr[it][e] = row[min(kk + e, N - 1)];
            }
        }
    };
    auto store_v = [&](int k0) {
        u32x2_t* V2 = reinterpret_cast<u32x2_t*>(Vt);
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            const int idx = tid + it * NT, kq = idx % (FK / 4), d = idx / (FK / 4);
            const int kk = kq * 4, sb = kk >> 5, w32 = kk & 31, ig = w32 >> 3, gg = (w32 & 7) >> 2, c = sb * 2 + (ig >> 1);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = k0 + kk + e < N ? vr[it][e] : 0.f;
            unsigned h0, l0, h1, l1;
            split2_pair(v[0], v[1], h0, l0);
            split2_pair(v[2], v[3], h1, l1);
            const u32x2_t hh = {h0, h1}, ll = {l0, l1};
            V2[(((0 * 4 + c) * 2 + gg) * DK + d) * 2 + (ig & 1)] = hh;
            V2[(((1 * 4 + c) * 2 + gg) * DK + d) * 2 + (ig & 1)] = ll;
        }
    };
    ATTN_MARK();
    issue_k(0, 0);                                                      // (in flight while the tables are built)
    load_v(0);
    ATTN_MARK();
    // the two 9-row tables as matrix-core operands: q . Ek[r] is one more "key" block, sum_r p_r Ev[r] one more contraction block.
    // (all loads first, then the conversions: one memory round trip, shared with the K / V tile above)
    {
        constexpr int EI = 32 * (DK / 8) / NT, VI2 = 2 * DK / NT;
        typedef f32x4 __attribute__((aligned(4))) f32x4u;
        f32x4 xe[EI][2];
        float xv[VI2][8];
#pragma unroll
        for (int it = 0; it < EI; ++it) {
            const int idx = tid + it * NT, r = idx & 31, dg = idx >> 5;
            const int rc = r < nrel ? r : 0;
            xe[it][0] = *reinterpret_cast<const f32x4u*>(emb_k + rc * DK + dg * 8);
            xe[it][1] = *reinterpret_cast<const f32x4u*>(emb_k + rc * DK + dg * 8 + 4);
        }
#pragma unroll
        for (int it = 0; it < VI2; ++it) {
            const int idx = tid + it * NT, d = idx % DK, gg = idx / DK;
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[it][j] = emb_v[min(8 * gg + j, nrel - 1) * DK + d];
        }
        ATTN_MARK();
#ifdef ATTN_DBG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ATTN_MARK();
#endif
#pragma unroll
        for (int it = 0; it < EI; ++it) {
            const int idx = tid + it * NT, r = idx & 31, dg = idx >> 5;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = r < nrel ? xe[it][j >> 2][j & 3] : 0.f;
            u32x4_t hh, ll;
            split2(x, hh, ll);
            EkI[((0 * KB + (dg >> 1)) * 2 + (dg & 1)) * 32 + r] = hh;
            EkI[((1 * KB + (dg >> 1)) * 2 + (dg & 1)) * 32 + r] = ll;
        }
#pragma unroll
        for (int it = 0; it < VI2; ++it) {
            const int idx = tid + it * NT, d = idx % DK, gg = idx / DK;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = 8 * gg + j < nrel ? xv[it][j] : 0.f;
            u32x4_t hh, ll;
            split2(x, hh, ll);
            EvT[(0 * 2 + gg) * DK + d] = hh;
            EvT[(1 * 2 + gg) * DK + d] = ll;
        }
    }
    ATTN_MARK();
    __syncthreads();                                                    // the tables are visible
    ATTN_MARK();
    // q . Ek[r] for the wave's 32 queries: S_rel^T = Ek . Q^T, rows r = 8 (e >> 2) + 4 g + (e & 3)
    {
        f32x16 SR;
#pragma unroll
        for (int e = 0; e < 16; ++e) SR[e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const f16x8 ah = __builtin_bit_cast(f16x8, EkI[((0 * KB + kb) * 2 + g) * 32 + lq]);
            const f16x8 al = __builtin_bit_cast(f16x8, EkI[((1 * KB + kb) * 2 + g) * 32 + lq]);
            SR = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, Ql[kb]), SR, 0, 0, 0);
            SR = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, Qh[kb]), SR, 0, 0, 0);
            SR = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, Qh[kb]), SR, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 8 * (e >> 2) + 4 * g + (e & 3);
            if (r < nrel) Rk[ql * MAXREL + r] = SR[e] * cs;
        }
    }
    ATTN_MARK();
    float m = -INFINITY, lsum = 0.f;
    f32x16 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;

    int buf = 0;
    for (int k0 = 0; k0 < N; k0 += FK, buf ^= 1) {
        store_v(k0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's share of the K tile has landed
        __syncthreads();                                                // K tile k0 (DMA) and V tile k0 are in LDS (first time: Rk too)
        if (k0 == 0) ATTN_MARK();
        if (k0 + FK < N) {
            issue_k(k0 + FK, buf ^ 1);
            load_v(k0 + FK);
        }
        const u32x4_t* Kc = Kt + (size_t)buf * (2 * KB * 2 * FK);
        f32x16 S[2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[sb][e] = 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f16x8 ah = __builtin_bit_cast(f16x8, Kc[(0 * KB * 2 + kb * 2 + g) * FK + sb * 32 + lq]);
                const f16x8 al = __builtin_bit_cast(f16x8, Kc[(1 * KB * 2 + kb * 2 + g) * FK + sb * 32 + lq]);
                S[sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, Ql[kb]), S[sb], 0, 0, 0);
                S[sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, Qh[kb]), S[sb], 0, 0, 0);
                S[sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, Qh[kb]), S[sb], 0, 0, 0);
            }
        }
        if (k0 == 0) ATTN_MARK();
        bool band[2];
        float mx = -INFINITY;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const int kb0 = k0 + sb * 32;
            band[sb] = kb0 <= qw0 + 31 + window && kb0 + 31 >= qw0 - window;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kb0 + 8 * (e >> 2) + 4 * g + (e & 3);
                float sc = S[sb][e] * cs;
                if (band[sb]) {
                    const int r = key - qi + window;
                    if (r >= 0 && r < nrel) sc += Rk[ql * MAXREL + r];
                }
                if (key >= N) sc = -INFINITY;
                S[sb][e] = sc;
                mx = fmaxf(mx, sc);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);
        const float corr = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        lsum *= corr;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mb][e] *= corr;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = __builtin_amdgcn_exp2f(S[sb][e] - mn);
                S[sb][e] = pe;
                lsum += pe;
            }
        if (k0 == 0) ATTN_MARK();
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            if (!band[sb]) continue;
            const int kb0 = k0 + sb * 32;
            // this query's band probabilities of the block as a row Pb[q][r] (r = key - q + w; entries of keys outside the block stay zero),
            // then O^T += Ev^T . Pb^T: one contraction block of 16 (9 used)
            if (g == 0) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) *reinterpret_cast<f32x4*>(Pb + ql * 16 + e4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = kb0 + 8 * (e >> 2) + 4 * g + (e & 3) - qi + window;
                if (r >= 0 && r < nrel) Pb[ql * 16 + r] = S[sb][e];
            }
            __builtin_amdgcn_wave_barrier();
            float pb[8];
            {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(Pb + ql * 16 + 8 * g), p1 = *reinterpret_cast<const f32x4*>(Pb + ql * 16 + 8 * g + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { pb[e] = p0[e]; pb[4 + e] = p1[e]; }
            }
            u32x4_t bh, bl;
            split2(pb, bh, bl);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f16x8 eh = __builtin_bit_cast(f16x8, EvT[(0 * 2 + g) * DK + mb * 32 + lq]);
                const f16x8 el = __builtin_bit_cast(f16x8, EvT[(1 * 2 + g) * DK + mb * 32 + lq]);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh, __builtin_bit_cast(f16x8, bl), acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(el, __builtin_bit_cast(f16x8, bh), acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh, __builtin_bit_cast(f16x8, bh), acc[mb], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (k0 == 0) ATTN_MARK();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = S[c >> 1][(c & 1) * 8 + j];
            u32x4_t ph, pl;
            split2(pv, ph, pl);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f16x8 vh = __builtin_bit_cast(f16x8, Vt[((0 * 4 + c) * 2 + g) * DK + mb * 32 + lq]);
                const f16x8 vl = __builtin_bit_cast(f16x8, Vt[((1 * 4 + c) * 2 + g) * DK + mb * 32 + lq]);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, __builtin_bit_cast(f16x8, pl), acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, __builtin_bit_cast(f16x8, ph), acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, __builtin_bit_cast(f16x8, ph), acc[mb], 0, 0, 0);
            }
        }
        __syncthreads();                                                // everybody is done with Vt and this K buffer
        if (k0 == 0) ATTN_MARK();
    }
    const float ltot = lsum + __shfl_xor(lsum, 32);
#ifdef ATTN_DBG
    __syncthreads();
    ATTN_MARK();
    if (a.out && blockIdx.x == 0 && blockIdx.y == 1 && blockIdx.z == 1 && tid == 0) {
        for (int i = 1; i < nmark; ++i) a.out[(size_t)(DK * 2 - 1) * a.ldo + i] = (float)(tmark[i] - tmark[i - 1]);
        a.out[(size_t)(DK * 2 - 1) * a.ldo] = (float)nmark;
        return;
    }
#endif
    if (qi < N) {
        const float inv = 1.0f / ltot;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mb][e4 * 4 + e] * inv;
                const int d = mb * 32 + 8 * e4 + 4 * g;                  // channels d .. d + 3 of this head
                if (a.out) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a.out[(size_t)(h * DK + d + e) * a.ldo + o0 + qi] = v[e];
                }
                if (a.out_h) {                                          // k-block (h DK + d) / 16, k half e4 & 1, entries 4 g .. 4 g + 3
                    unsigned h0, l0, h1, l1;
                    split2_pair(v[0], v[1], h0, l0);
                    split2_pair(v[2], v[3], h1, l1);
                    const size_t kbo = (size_t)(h * KB + mb * 2 + (e4 >> 1));
                    u32x2_t* oh = reinterpret_cast<u32x2_t*>(a.out_h);
                    oh[((kbo * 4 + (e4 & 1)) * NX + o0 + qi) * 2 + g] = u32x2_t{h0, h1};
                    oh[((kbo * 4 + 2 + (e4 & 1)) * NX + o0 + qi) * 2 + g] = u32x2_t{l0, l1};
                }
            }
    }
}

template <int NW>
static int launch_attention_image(const AttnImageArgs& a, int heads, int B, int max_len, hipStream_t stream)
{
    constexpr int NQ = NW * 32;
    const size_t smem = (size_t)(2 * (2 * 8 * 2 * FK) + 2 * 4 * 2 * 128 + 2 * 8 * 2 * 32 + 2 * 2 * 128) * 16 + sizeof(float) * (NQ * MAXREL + NQ * 16);
    AS_LDS_OPT_IN(relpos_attention_image_kernel<NW>, 160 * 1024);
    AsProfScope prof__(AS_FILE_CLS, 0, 0, stream);
    hipLaunchKernelGGL(relpos_attention_image_kernel<NW>, dim3(as_cdiv(max_len, NQ), heads, B), dim3(NW * 64), smem, stream, a);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

extern "C" int as_relpos_attention_image_f32(const float* qkv, int ld, const uint16_t* qkv_h, int n_total, int C, int heads, int window,
                                             const float* emb_rel_k, const float* emb_rel_v, const float* emb_rel_k2, const float* emb_rel_v2,
                                             int b_split, const int32_t* col_off, int B, int max_len, float* out, int ldo, uint16_t* out_h,
                                             as_stream_t stream)
{
    if ((emb_rel_k2 == nullptr) != (emb_rel_v2 == nullptr) || (emb_rel_k2 && b_split <= 0)) return AS_EINVAL;
    if (!qkv || !qkv_h || !emb_rel_k || !emb_rel_v || !col_off || (!out && !out_h) || C <= 0 || heads <= 0 || C % heads || n_total < 0 ||
        ld < n_total || (out && ldo < n_total))
        return AS_EINVAL;
    if (C / heads != 128 || 2 * window + 1 > MAXREL || window < 0 || B < 0) return AS_EINVAL;   // 128-channel heads only (the path's)
    if (((reinterpret_cast<uintptr_t>(qkv_h) | reinterpret_cast<uintptr_t>(out_h)) & 15) != 0) return AS_EINVAL;
    if ((double)(3 * C / 16) * 64.0 * (n_total + 1.0) >= 2147483648.0) return AS_EINVAL;      // 32-bit offsets in the image descriptor
    if (B == 0 || max_len <= 0) return AS_OK;
    AttnImageArgs a;
    a.qkv = qkv; a.qkv_h = qkv_h; a.ek1 = emb_rel_k; a.ev1 = emb_rel_v; a.ek2 = emb_rel_k2; a.ev2 = emb_rel_v2; a.col_off = col_off;
    a.out = out; a.out_h = out_h; a.ld = ld; a.C = C; a.window = window; a.b_split = b_split; a.n_total = n_total; a.ldo = ldo;
    if (max_len <= 64) return launch_attention_image<2>(a, heads, B, max_len, (hipStream_t)stream);
    return launch_attention_image<4>(a, heads, B, max_len, (hipStream_t)stream);
}

extern "C" int as_relpos_attention_groups_f32(const float* qkv, int ld, int C, int heads, int window, const float* emb_rel_k,
                                              const float* emb_rel_v, const float* emb_rel_k2, const float* emb_rel_v2, int b_split,
                                              const int32_t* col_off, int B, int max_len, float* out, int ldo, as_stream_t stream)
{
    if ((emb_rel_k2 == nullptr) != (emb_rel_v2 == nullptr) || (emb_rel_k2 && b_split <= 0)) return AS_EINVAL;
    if (!qkv || !emb_rel_k || !emb_rel_v || !col_off || !out || C <= 0 || heads <= 0 || C % heads) return AS_EINVAL;
    const int dk = C / heads;
    if (dk > MAXDK || 2 * window + 1 > MAXREL || window < 0 || B < 0) return AS_EINVAL;
    if (B == 0 || max_len <= 0) return AS_OK;
    const size_t smem = sizeof(float) * ((size_t)2 * dk * KPAD + QT * dk + 4 * KT * 4 + QT * MAXREL + 2 * MAXREL * dk);
    AS_LDS_OPT_IN(relpos_attention_kernel, 160 * 1024);   // > 64 KiB of dynamic LDS needs an explicit opt-in
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(relpos_attention_kernel, dim3(as_cdiv(max_len, QT), heads, B), dim3(256), smem, (hipStream_t)stream,
                       qkv, ld, C, heads, window, emb_rel_k, emb_rel_v, emb_rel_k2, emb_rel_v2, b_split, col_off, out, ldo);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

extern "C" int as_relpos_attention_f32(const float* qkv, int ld, int C, int heads, int window, const float* emb_rel_k,
                                       const float* emb_rel_v, const int32_t* col_off, int B, int max_len, float* out,
                                       int ldo, as_stream_t stream)
{
    return as_relpos_attention_groups_f32(qkv, ld, C, heads, window, emb_rel_k, emb_rel_v, nullptr, nullptr, 0, col_off, B, max_len, out,
                                          ldo, stream);
}
