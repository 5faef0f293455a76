// Internal pieces of the conv GEMM (conv_gemm.hip: dispatch, split-K reduce, Cin = 1 direct kernel; conv_gemm_h3.hip: the
// f16x3 matrix-core kernel and the writers of its operand images).
#pragma once
#include "artspeech_hip.h"
#include "common.h"

#define OOB 0xFFFFFFFFu
#define OOBH 0x80000000u   /* epilogue: out of range for every descriptor (< 2 GiB, host-checked) even after adding an in-range offset */

// host: launch the f16x3 kernel for tile `choice` (22 = 128x128, 21 = 128x64, 12 = 64x128, 11 = 64x64, 14 = 64x256; 4 waves each) over the
// tiles of `n` independent problems (1 <= n <= H3_MAXP), problem i in S[i] K slices; returns AS_OK or a hipError_t.  (conv_gemm_h3.hip)
int as_conv_gemm_h3_launch(const ConvGemmArgs* const* a, const int* S, int n, int choice, hipStream_t stream);
// host: ONE conv (no K slices, one weight set, f16x3) as one launch of two tile shapes: columns [0, col_split) on 128 x 128 tiles, listed
// first, columns [col_split, N) on 128 x 64 tiles behind them -- the last round of a launch of 1.x rounds of the chip is then made of
// half-size tiles (scripts/exp/records/riders_r06.txt).  col_split: a multiple of 128 inside (0, N).
int as_conv_gemm_h3_launch_mix(const ConvGemmArgs* a, int col_split, hipStream_t stream);
// host: the kernel that writes the split image of X (no profiling scope of its own)
int as_split_f16x2_launch(const float* x, int ldx, int K, int N, int lrelu, float slope, uint16_t* xh, hipStream_t stream);

static inline __host__ __device__ int as_kbx(int K) { return (((K + 15) >> 4) + 3) & ~3; }      // k-blocks of a split image: a multiple of 4

// tap offsets packed one byte per tap, (dh+8) << 4 | (dw+8) (|dh|, |dw| <= 7, checked by the host), eight taps per
// word: the k loop then selects a tap with scalar ALU only (a scalar or scratch load inside it would stall the wave)
struct H3Taps {
    unsigned long long w0, w1, w2, w3;
    int wide;                             // 1: every dh = 0 and the byte is dw + 128 (dilated 1-D convs, |dw| <= 127)
};

// host: pack the tap offsets of `a` for the kernel (AS_EINVAL if they do not fit a byte)
static inline int h3_pack_taps(const ConvGemmArgs& a, H3Taps* out)
{
    H3Taps tp = {0, 0, 0, 0, 0};
    unsigned long long* w = &tp.w0;
    for (int t = 0; t < a.T; ++t)
        if (a.dh[t] < -7 || a.dh[t] > 7 || a.dw[t] < -7 || a.dw[t] > 7) tp.wide = 1;
    for (int t = 0; t < a.T; ++t) {
        if (tp.wide && (a.dh[t] != 0 || a.dw[t] < -127 || a.dw[t] > 127)) return AS_EINVAL;
        const int byte = tp.wide ? a.dw[t] + 128 : ((a.dh[t] + 8) << 4) | (a.dw[t] + 8);
        w[t >> 3] |= (unsigned long long)byte << ((t & 7) * 8);
    }
    *out = tp;
    return AS_OK;
}

// Byte of tap t.  Written with masks: as a select chain hipcc turns it into scalar BRANCHES inside the k loop.
#ifdef __HIPCC__
static __device__ __forceinline__ unsigned long long h3_tap_word(const H3Taps& tp, int t)
{
    const int s = t >> 3;
    const unsigned long long m0 = 0ull - (unsigned long long)(s == 0), m1 = 0ull - (unsigned long long)(s == 1),
                             m2 = 0ull - (unsigned long long)(s == 2), m3 = 0ull - (unsigned long long)(s == 3);
    return (tp.w0 & m0) | (tp.w1 & m1) | (tp.w2 & m2) | (tp.w3 & m3);
}

#endif

#ifdef __HIPCC__
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
static __device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// ONE launch, several independent problems ("tile list"): the grid is the concatenation of the problems' workgroups, problem i owns
// the physical ids [wg0, wg0 + wgs) -- wgs = tiles x K slices rounded up to a multiple of 8 (the spare ones exit at once).  A small
// problem that would own the chip alone at a fraction of its width rides in the tail of a large one: the launch costs what its busiest CU
// costs.  The host lists the problems longest tile first.  Passed BY VALUE (kernel argument segment: every field a wave reads is a scalar
// load from constant memory, for any problem index).
#define H3_MAXP 6
struct H3Prob {
    ConvGemmArgs a;
    H3Taps tp;
    int32_t wg0, wgs;       // first workgroup, workgroups (a multiple of 8)
    int32_t tiles, S;       // output tiles, K slices: workgroup (tile, slice) = logical id % tiles, / tiles
    int32_t col0, col1;     // the columns [col0, col1) of `a` this entry covers (0, N: all; a mixed-tile launch lists one conv twice)
    int32_t small, pad_;    // mixed-tile launch: 1 = this entry's tiles are the launch's SMALL tile
};
struct H3Multi {
    int32_t n, pad_;
    H3Prob p[H3_MAXP];
};

// XCD-aware order (speed only): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
// run of a problem's logical tiles -- the output-channel tiles of one column range then share that XCD's L2 copy of the
// activation columns.  local = physical id - wg0 (wg0 and wgs are multiples of 8: local % 8 is the XCD).
static __device__ __forceinline__ int logical_of(int local, int wgs)
{
    return (local & 7) * (wgs >> 3) + (local >> 3);
}

// two fp32 -> (h, l) fp16 pairs with x = h + l to 22 bits: v_cvt_pk_f16_f32 (RNE), the residual is exact in fp32
static __device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l)
{
    const f32x2 v = {x0, x1};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f32x2 r = v - __builtin_convertvector(hh, f32x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// 8 fp32 (consecutive k of one column) -> two 16-byte rows of 8 fp16
static __device__ __forceinline__ void split2(const float (&x)[8], u32x4_t& h, u32x4_t& l)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned hu, lu;
        split2_pair(x[2 * e], x[2 * e + 1], hu, lu);
        h[e] = hu;
        l[e] = lu;
    }
}

static __device__ __forceinline__ float div_sqrt2f(float x)
{
    // x / sqrt(2), correctly rounded without the division sequence: q = x*(1/c), one Newton correction in fma
    const float c = 1.41421356237309504880f, rc = 0.70710678118654752440f;
    const float q = x * rc;
    return __builtin_fmaf(__builtin_fmaf(-q, c, x), rc, q);
}

template <int ACT>
static __device__ __forceinline__ float epi_act(float x, float slope)
{
    if (ACT == 1) x = x < 0.f ? 0.f : x;                             // (NaN stays NaN, as torch.relu)
    if (ACT == 2) x = x > 0.f ? x : slope * x;
    if (ACT == 3) x = tanhf(x);
    if (ACT == 4) x = fabsf(x);
    if (ACT == 5) x = x / (1.0f + expf(-x));
    return x;
}

// One 32x32 result tile as rows of the consumer's split image.  A lane holds column `col`, rows row0 + (e&3) + 8(e>>2) + 4 lk:
// half of each of the tile's four 8-row groups.  v_permlane32_swap hands lane (col, lk = 0) the other half of groups 0 and 2
// and lane (col, lk = 1) that of groups 1 and 3, so every lane stores whole 16-byte rows: 8 swaps, 4 stores per tile.
static __device__ __forceinline__ void yh_store_tile(const ConvGemmArgs& a, __amdgpu_buffer_rsrc_t rsH, const float (&v)[16], int row0,
                                                     int col, int lk, int n_end)
{
    const unsigned NXy = (unsigned)a.N + 1u;
    const int groups = 2 * as_kbx(a.M);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        float t[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x0 = v[8 * pr + r], x1 = v[8 * pr + 4 + r];
            if (a.yh_lrelu) {
                x0 = x0 > 0.f ? x0 : a.in_slope * x0;
                x1 = x1 > 0.f ? x1 : a.in_slope * x1;
            }
            // x0.hi <-> x1.lo.  Inline asm: with __builtin_amdgcn_permlane32_swap hipcc 7.2 uses the FIRST result for both
            // elements of the returned pair in this loop (found in the ISA: rows 4..7 of every group repeated rows 0..3); the
            // s_nop covers the VALU-write -> permlane-read hazard the compiler would otherwise pad itself.
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
            t[r] = x0;
            t[4 + r] = x1;
        }
        u32x4_t h, l;
        split2(t, h, l);
        const int g = (row0 >> 3) + 2 * pr + lk;                        // 8-row group: k-block g / 2, k-half g % 2
        const bool ok = col < n_end && g < groups;
        const unsigned off = ok ? ((unsigned)((g >> 1) * 4 + (g & 1)) * NXy + (unsigned)col) * 16u : OOBH;
        __builtin_amdgcn_raw_buffer_store_b128(h, rsH, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(l, rsH, off + 2u * NXy * 16u, 0, 0);
        if (col == 0 && g < groups) {                                   // the zero column N, once per (group, part)
            const u32x4_t z = {0u, 0u, 0u, 0u};
            const unsigned offz = ((unsigned)((g >> 1) * 4 + (g & 1)) * NXy + (unsigned)a.N) * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(z, rsH, offz, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(z, rsH, offz + 2u * NXy * 16u, 0, 0);
        }
    }
}

// Accumulator tiles -> Y / Yh.  A 32x32 MFMA accumulator holds C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31];
// a wave owns TM x TN of them at rows m0 + wm*32*TM, columns n0 + wn*32*TN.
// Branch-free: every access is a raw BUFFER access whose per-lane offset carries the whole (row, column) position, so
// rows >= M fall past the descriptor's end (loads return 0, stores are dropped by the range check) and columns >= N
// get an out-of-range offset -- no exec-mask branches, no 64-bit address arithmetic per element.
template <int TM, int TN, bool DIV, int ACT, bool TR>
static __device__ __forceinline__ void epilogue_tiles(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], int rbase, int cbase,
                                                      int l31, int lk, int grp, int n_end)
{
    // grp: the weight set of this tile's columns; n_end: one past the tile's last valid column (the end of its group, or N)
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias + (size_t)grp * a.M : nullptr), 0, a.bias ? a.M * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res), 0, a.res ? (int)((unsigned)a.M * a.ldr * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        a.Y, 0, a.Y ? (int)((unsigned)(TR ? a.N : (a.ileave_u > 1 ? a.M / a.ileave_u : a.M)) * a.ldy * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(
        a.Yh, 0, a.Yh ? (int)((unsigned)as_kbx(a.M) * 4u * ((unsigned)a.N + 1u) * 16u) : 0, 0x00020000);
    const float sc = a.acc_scale;
    if (a.range_probe) {                                   // range probe (as_set_range_probe): a wave-uniform branch, off by default
        bool bad = false;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) bad |= !(fabsf(acc[i][jn][e]) <= 3.0e38f);
        if (bad) as_status_raise(a.status, AS_STATUS_F16_RANGE);
    }
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
            const unsigned r_off = col < n_end ? (unsigned)(row0 * a.ldr + col) * 4u : OOBH;
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = 0.f;
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = buf_load1(rsR, r_off + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ldr * 4), 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x = __builtin_fmaf(acc[i][jn][e], sc, bv[i][e]);
                x += v[e];
                if (DIV) x = div_sqrt2f(x);
                v[e] = epi_act<ACT>(x, a.act_slope);
            }
            if (TR) {                                        // time-major output for the LSTM: Y[col][row], 4 rows = 16 bytes
                const bool quad = (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0 && row0 + 28 <= a.M;
                const unsigned off = col < n_end ? (unsigned)(col * a.ldy + row0) * 4u : OOBH;
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
                if (a.Y) {
                    unsigned off = col < n_end ? (unsigned)(row0 * a.ldy + col) * 4u : OOBH;
                    if (a.ileave_u > 1) {                               // rows (phase r, channel m) -> Y[m][u col + r]; C % 32 == 0: one r per tile
                        const int Cc = a.M / a.ileave_u, r = (row0 - 4 * lk) / Cc;
                        off = (col < n_end && row0 - 4 * lk < a.M) ? (unsigned)((row0 - r * Cc) * a.ldy + a.ileave_u * col + r) * 4u : OOBH;
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rsY,
                                                              off + (unsigned)(((e & 3) + 8 * (e >> 2)) * a.ldy * 4), 0, 0);
                }
                if (a.Yh) yh_store_tile(a, rsH, v, row0 - 4 * lk, col, lk, n_end);
            }
        }
    }
}

// AUX: cache policy of the stores (0 = default; 16 = sc1, write-through to the memory side: what an in-launch hand-off to other
// workgroups wants -- no release fence, MI355X_MICROARCH.md publish-large)
template <int TM, int TN, int AUX = 0>
static __device__ __forceinline__ void slab_store(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], __amdgpu_buffer_rsrc_t rs,
                                                  int rbase, int cbase, int l31, int n_end)
{
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = cbase + jn * 32 + l31;
            const unsigned v0 = col < n_end ? (unsigned)((rbase + i * 32) * a.N + col) * 4u : OOBH;
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

// The same partial sums TIME-MAJOR: slab[col][row] (row stride M).  For a K-sliced conv whose reduction also computes the channel
// LayerNorm of every column (as_conv_gemm_multi_post_f32, post_ln): a column's M channels then lie together, and the reduction's wave
// = one column, lane = 8 consecutive channels, reads them as two 16-byte loads per slab.  A lane holds 4 consecutive rows per (e >> 2):
// 16-byte stores (the four (e >> 2) groups and the two lane halves fill a column's 128-byte lines between them).
template <int TM, int TN>
static __device__ __forceinline__ void slab_store_tr(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], __amdgpu_buffer_rsrc_t rs,
                                                     int rbase, int cbase, int l31, int n_end)
{
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = cbase + jn * 32 + l31;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = rbase + i * 32 + 8 * q;                  // rows row .. row + 3 (rbase carries the lane half's + 4)
                float t0 = acc[i][jn][4 * q], t1 = acc[i][jn][4 * q + 1], t2 = acc[i][jn][4 * q + 2], t3 = acc[i][jn][4 * q + 3];
                asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));     // (through VGPRs: see slab_store)
                const f32x4 t = {t0, t1, t2, t3};
                // M % 8 == 0 (host-checked for this form): a quad is inside the rows or outside as a whole
                const unsigned off = (col < n_end && row < a.M) ? (unsigned)(col * a.M + row) * 4u : OOBH;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, t), rs, off, 0, 0);
            }
        }
}

// Split-K tail: y = epi(sum_s slab[s]) for the 8 rows 8 g .. 8 g + 7 of column j, slabs summed in the fixed order s = 0 .. S-1
// (deterministic whatever order the slices finished in); slab s = fp32 [M][N] at a.ws + s M N.  j == N with Yh: the image's zero
// column.  One 16-byte row of the consumer's split image when Yh is wanted.  (A device function: scripts/exp/conv_gemm_sn.hip, the
// small-N kernel that was measured and not shipped, ran the same code inside its own launch.)
// the epilogue of ONE summed element (the K-sliced launches' reduction kernels: plain, and with the AdaIN behind it)
// (LEAN: activations 0 .. 2 only -- tanh and swish expand to hundreds of instructions per use, and the fused reduction + AdaIN kernel
// inlines this 8 x 4 times: 15 000 instructions, whose fetch alone took longer than the two launches the kernel replaces.  Explicit
// roundings: the same bits in both forms whatever the compiler would contract.)
template <bool LEAN = false>
static __device__ __forceinline__ float as_reduce_value(const ConvGemmArgs& a, float x, float bias, float res)
{
    if (a.range_probe && !(fabsf(x) <= 3.0e38f)) as_status_raise(a.status, AS_STATUS_F16_RANGE);
    x = __fmul_rn(x, a.acc_scale);
    if (a.bias) x = __fadd_rn(x, bias);
    if (a.res) x = __fadd_rn(x, res);
    if (a.div_sqrt2) x = __fdiv_rn(x, 1.41421356237309504880f);
    if (a.act == 1) x = x < 0.f ? 0.f : x;                      // (NaN stays NaN, as torch.relu: a non-finite value upstream reaches to_out)
    else if (a.act == 2) x = x > 0.f ? x : __fmul_rn(a.act_slope, x);
    else if (!LEAN) {
        if (a.act == 3) x = tanhf(x);
        else if (a.act == 4) x = fabsf(x);
        else if (a.act == 5) x = x / (1.0f + expf(-x));
    }
    return x;
}

static __device__ __forceinline__ void as_reduce_epilogue(const ConvGemmArgs& a, int S, int j, int g)
{
    if (j > a.N || (j == a.N && !a.Yh)) return;
    const size_t total = (size_t)a.M * a.N;
    const float* slab = reinterpret_cast<const float*>(a.ws);
    const int grp = a.n_groups > 1 ? (j < a.N ? j : a.N - 1) / a.group_cols : 0;
    // (capacity layouts: the slices stored nothing for the filler columns of a group, and nothing is made of what lies there)
    if (a.n_valid && j < a.N && j - (a.n_groups > 1 ? grp * a.group_cols : 0) >= *a.n_valid) return;
    // the slabs first, eight loads (the thread's rows of one slab) in flight at a time and none of them behind a branch: with the
    // loads inside the per-row `if` every one of the 8 S waited for the one before (12.8 us per launch at batch 1, as long as the GEMM
    // it follows).  Same order of additions per element: s ascending.
    float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
        const int jc = j < a.N ? j : a.N - 1;
        size_t idx[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = 8 * g + r;
            idx[r] = (size_t)(row < a.M ? row : a.M - 1) * a.N + jc;
        }
        // four slabs' loads in flight at a time (a dependent trip per slab was S L2 round trips: 15-30 us at S = 16-24)
        int s = 0;
        for (; s + 4 <= S; s += 4) {
            float t[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 8; ++r) t[u][r] = slab[(size_t)(s + u) * total + idx[r]];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc8[r] += t[u][r];
        }
        for (; s < S; ++s) {
            float t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = slab[(size_t)s * total + idx[r]];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc8[r] += t[r];
        }
    }
    float bias8[8], res8[8];                              // (likewise: loaded for all eight rows at once, used below)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = 8 * g + r < a.M ? 8 * g + r : a.M - 1;
        bias8[r] = a.bias ? a.bias[(size_t)grp * a.M + row] : 0.f;
        res8[r] = a.res ? a.res[(size_t)row * a.ldr + (j < a.N ? j : a.N - 1)] : 0.f;
    }
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = 8 * g + r;
        float x = 0.f;
        if (row < a.M && j < a.N) {
            x = as_reduce_value(a, acc8[r], bias8[r], res8[r]);
            if (a.Y) {
                if (a.transpose_out) a.Y[(size_t)j * a.ldy + row] = x;
                else a.Y[(size_t)row * a.ldy + j] = x;
            }
        }
        v[r] = x;
    }
    if (a.Yh && g < 2 * as_kbx(a.M)) {
        if (a.yh_lrelu) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : a.in_slope * v[r];
        }
        u32x4_t h, l;
        split2(v, h, l);
        const size_t NX = (size_t)a.N + 1;
        u32x4_t* yh = reinterpret_cast<u32x4_t*>(a.Yh) + ((size_t)(g >> 1) * 4 + (g & 1)) * NX + j;
        yh[0] = h;
        yh[2 * NX] = l;
    }
}

template <int TM, int TN>
static __device__ __forceinline__ void epilogue_dispatch(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], int rbase, int cbase,
                                                         int l31, int lk, int grp, int n_end)
{
    // one lean copy per (divide, activation, transposed) combination the path uses
    if (a.transpose_out) epilogue_tiles<TM, TN, false, 0, true>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else if (a.div_sqrt2 && a.act == 0) epilogue_tiles<TM, TN, true, 0, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else if (a.div_sqrt2 && a.act == 2) epilogue_tiles<TM, TN, true, 2, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else if (a.div_sqrt2) epilogue_tiles<TM, TN, true, 1, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else if (a.act == 0) epilogue_tiles<TM, TN, false, 0, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else if (a.act == 1) epilogue_tiles<TM, TN, false, 1, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else if (a.act == 3) epilogue_tiles<TM, TN, false, 3, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else if (a.act == 4) epilogue_tiles<TM, TN, false, 4, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else if (a.act == 5) epilogue_tiles<TM, TN, false, 5, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
    else epilogue_tiles<TM, TN, false, 2, false>(a, acc, rbase, cbase, l31, lk, grp, n_end);
}

// S > 1 (split-K): this slice's raw partial sums go to its slab; splitk_reduce_kernel sums the slabs in a fixed order
// and applies the epilogue.  (Combining inside this kernel -- per-tile arrival counters, the last slice reduces -- was
// built and measured on MI355X in round 1: slower, see DESIGN.md.)
// `active` = this wave holds a result (false for the waves of a K group that already folded theirs into group 0).
template <int TM, int TN>
static __device__ __forceinline__ void epilogue(const ConvGemmArgs& a, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm,
                                                int wn, int l31, int lk, int S, int slice, bool active, int grp, int n_end)
{
    if (!active) return;
    const int rbase = m0 + wm * 32 * TM + 4 * lk;          // this lane's first row; element e adds i*32 + (e&3) + 8*(e>>2)
    const int cbase = n0 + wn * 32 * TN;
    if (S == 1) {
        epilogue_dispatch<TM, TN>(a, acc, rbase, cbase, l31, lk, grp, n_end);
        return;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(a.ws) + (size_t)slice * a.M * a.N, 0, (int)((unsigned)a.M * a.N * 4u), 0x00020000);
    if (a.slab_tr) slab_store_tr<TM, TN>(a, acc, rs, rbase, cbase, l31, n_end);
    else slab_store<TM, TN>(a, acc, rs, rbase, cbase, l31, n_end);
}
#endif
