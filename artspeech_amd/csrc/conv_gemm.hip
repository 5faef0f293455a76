// Host side of the conv GEMM (include/artspeech_hip.h: as_conv_gemm_f32): argument checks, tile and split-K choice, the
// split of fp32 activations into the matrix-core kernel's operand image when the caller has none, the deterministic split-K
// reduction, and the direct kernel for Cin = 1.  The matrix-core kernel itself is conv_gemm_h3.hip.
//
// Activations use the "packed frames" layout: a tensor is [C][N] with all utterances of
// the batch concatenated along the contiguous column axis and NO padding; column j carries a 64-bit
// descriptor (h, w, H, W) of its position inside its own utterance, so a tap (dh, dw) is valid iff
// 0 <= h+dh < H and 0 <= w+dw < W -- that is the conv's zero padding, and it is also what keeps
// utterances from seeing each other.  1-D convs are the H = 1 case.  Replaces
//   nn.Conv1d k in {1,3,5,9}  (RelTransformerEnc.py:110-118,257-258,306-314; models.py:176-181,480-495,592-594)
//   nn.Conv2d 3x3 / 1x1       (models.py:71-77, 385-399, 530-535)
//   nn.Linear on [*, C] rows  (the LSTM input projections, hoisted out of the recurrence)
#include <algorithm>
#include "common.h"
#include "conv_gemm.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>

// Cin = 1 (the first conv of every style tower and of the decoder's F0 / energy inputs): 9 MACs per output on a
// 16-deep matrix-core k-block would be 2 % useful work, and the op is bound by writing Y anyway (M x N x 4 bytes: 130 MB
// for the mel tower) -- so a direct kernel: a thread owns one column, keeps its T input taps in registers and streams the
// M output channels (stores coalesced across the wave), weights and bias from LDS.
#define DIRECT_MAX_M 128
#define DIRECT_TP 12                                       // weights per channel in LDS: T (<= 9 on the fast path) padded to 3 x 16 bytes
// fast form (T <= 9): a thread owns FOUR consecutive columns (one 16-byte store per channel when Y allows), three
// 16-byte LDS reads feed 4 x T FMAs
// (several independent problems in one launch -- the stems of the mel tower, the TV tower and dur_block are ready together: blockIdx.x
// walks the problems' column blocks back to back)
#define DIRECT_MAXP 6
struct DirectMulti {
    int32_t n, pad_;
    int32_t blk0[DIRECT_MAXP + 2];       // first workgroup of problem i; its workgroups: (column block of nx, channel-group share of ny), column block fastest
    int32_t nx[DIRECT_MAXP], ny[DIRECT_MAXP];
    ConvGemmArgs a[DIRECT_MAXP];
};
template <bool VEC>
__global__ void __launch_bounds__(256)
conv_direct_cin1_x4_kernel(const DirectMulti dm)
{
    __shared__ __attribute__((aligned(16))) float ws[DIRECT_MAX_M * DIRECT_TP];   // [m][t]
    __shared__ float bs[DIRECT_MAX_M];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < DIRECT_MAXP; ++i) pi += (i < dm.n && (int)blockIdx.x >= dm.blk0[i]) ? 1 : 0;
    const ConvGemmArgs& a = dm.a[pi];
    const int local = (int)blockIdx.x - dm.blk0[pi], bx = local % dm.nx[pi], by = local / dm.nx[pi], nyp = dm.ny[pi];
    for (int i = threadIdx.x; i < a.M * DIRECT_TP; i += 256) {
        const int m = i / DIRECT_TP, t = i % DIRECT_TP;
        ws[i] = t < a.T ? a.W[(size_t)t * a.Kp * a.M + m] : 0.f;      // k = 0 rows of the [T][Kp][M] image
    }
    for (int i = threadIdx.x; i < a.M; i += 256) bs[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();
    // a wave owns 256 consecutive columns, a lane the columns lane, lane + 64, lane + 128, lane + 192 of them: every load and every
    // store of the wave is then one contiguous run (256 bytes of x, 1 KB of an image row).  With four CONSECUTIVE columns per lane a
    // store instruction wrote 16 bytes every 64 -- 32 lines a quarter full instead of 8 full ones, and the texture path takes ~4 cycles
    // a line: the mel tower's first conv (130 MB of image) took 86 us.
    const int j0 = (bx * 256 + (threadIdx.x & ~63)) * 4 + (threadIdx.x & 63);
    if ((bx * 256 + (threadIdx.x & ~63)) * 4 >= a.N) return;   // (the whole wave)
    float x[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = j0 + 64 * c;
        int h = 0, w = 0, H = 1, Wj = 0x7fffffff;
        if (a.meta && j < a.N) {
            const unsigned long long md = a.meta[j];
            h = AS_META_h(md), w = AS_META_w(md), H = AS_META_H(md), Wj = AS_META_W(md);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // branch-free: all 36 loads of the thread are in flight together (an invalid tap reads the column itself)
            const int tt = t < a.T ? t : 0;
            const bool ok = t < a.T && j < a.N &&
                            (!a.meta || ((unsigned)(h + a.dh[tt]) < (unsigned)H && (unsigned)(w + a.dw[tt]) < (unsigned)Wj));
            const int jj = j < a.N ? j : a.N - 1;
            float v = a.X[ok ? jj + a.dh[tt] * (a.meta ? Wj : 0) + a.dw[tt] : jj];
            if (a.in_act == 2) v = v > 0.f ? v : a.in_slope * v;
            x[c][t] = ok ? v : 0.f;
        }
    }
    // channels in groups of 8: a group of a column is one 16-byte row per part of the consumer's operand image (a.Yh)
    u32x4_t* yh = reinterpret_cast<u32x4_t*>(a.Yh);
    const size_t NX = (size_t)a.N + 1;
    const int groups = yh ? 2 * as_kbx(a.M) : (a.M + 7) / 8;
    // `by` (of the problem's ny) takes a share of the channel groups: few columns (the 1-D towers: 7 column blocks) would otherwise leave the chip to
    // 7 workgroups walking 64 channels each (47 us for 1.6 MB of output)
    const int gpb = (groups + nyp - 1) / nyp, g_lo = by * gpb, g_hi = min(groups, g_lo + gpb);
    if (yh && bx == 0 && threadIdx.x == 0)
        for (int g = g_lo; g < g_hi; ++g)
            for (int p = 0; p < 2; ++p) yh[((size_t)(g >> 1) * 4 + (g & 1) + 2 * p) * NX + a.N] = u32x4_t{0u, 0u, 0u, 0u};
    for (int g = g_lo; g < g_hi; ++g) {
        float t8[4][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = g * 8 + r;
            float y[4] = {0.f, 0.f, 0.f, 0.f};
            if (m < a.M) {
                float wt[DIRECT_TP];
#pragma unroll
                for (int q = 0; q < DIRECT_TP / 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&ws[m * DIRECT_TP + 4 * q]);
                    wt[4 * q] = v[0]; wt[4 * q + 1] = v[1]; wt[4 * q + 2] = v[2]; wt[4 * q + 3] = v[3];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) sum += wt[t] * x[c][t];
                    sum += bs[m];
                    if (a.act == 1) sum = sum < 0.f ? 0.f : sum;
                    else if (a.act == 2) sum = sum > 0.f ? sum : a.act_slope * sum;
                    else if (a.act == 3) sum = tanhf(sum);
                    else if (a.act == 4) sum = fabsf(sum);
                    else if (a.act == 5) sum = sum / (1.0f + expf(-sum));
                    y[c] = sum;
                }
                if (a.Y) {
                    float* yr = a.Y + (size_t)m * a.ldy + j0;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (j0 + 64 * c < a.N) yr[64 * c] = y[c];
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) t8[c][r] = (a.yh_lrelu && y[c] < 0.f) ? a.in_slope * y[c] : y[c];
        }
        if (yh) {
            const size_t plane = ((size_t)(g >> 1) * 4 + (g & 1)) * NX + j0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (j0 + 64 * c < a.N) {
                    u32x4_t h, l;
                    split2(t8[c], h, l);
                    yh[plane + 64 * c] = h;
                    yh[plane + 64 * c + 2 * NX] = l;
                }
        }
    }
}

// general form (any T <= AS_MAX_TAPS): one column per thread
__global__ void __launch_bounds__(256)
conv_direct_cin1_kernel(const ConvGemmArgs a)
{
    __shared__ float ws[AS_MAX_TAPS * DIRECT_MAX_M];       // [t][m]
    __shared__ float bs[DIRECT_MAX_M];
    for (int i = threadIdx.x; i < a.T * a.M; i += 256) ws[i] = a.W[(size_t)(i / a.M) * a.Kp * a.M + (i % a.M)];   // k = 0 rows
    for (int i = threadIdx.x; i < a.M; i += 256) bs[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.N) return;
    int h = 0, w = 0, H = 1, Wj = 0x7fffffff;
    if (a.meta) {
        const unsigned long long md = a.meta[j];
        h = AS_META_h(md), w = AS_META_w(md), H = AS_META_H(md), Wj = AS_META_W(md);
    }
    float x[AS_MAX_TAPS];
#pragma unroll
    for (int t = 0; t < AS_MAX_TAPS; ++t) {
        x[t] = 0.f;
        if (t < a.T) {
            const bool ok = !a.meta || ((unsigned)(h + a.dh[t]) < (unsigned)H && (unsigned)(w + a.dw[t]) < (unsigned)Wj);
            if (ok) {
                float v = a.X[j + a.dh[t] * (a.meta ? Wj : 0) + a.dw[t]];
                if (a.in_act == 2) v = v > 0.f ? v : a.in_slope * v;
                x[t] = v;
            }
        }
    }
    for (int m = 0; m < a.M; ++m) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < AS_MAX_TAPS; ++t)
            if (t < a.T) s += ws[t * a.M + m] * x[t];
        s += bs[m];
        if (a.act == 1) s = s < 0.f ? 0.f : s;
        else if (a.act == 2) s = s > 0.f ? s : a.act_slope * s;
        else if (a.act == 3) s = tanhf(s);
        else if (a.act == 4) s = fabsf(s);
        else if (a.act == 5) s = s / (1.0f + expf(-s));
        a.Y[(size_t)m * a.ldy + j] = s;
    }
}

// y = epi(sum_s slab[s]) in a fixed order (deterministic, unlike float atomics).  A thread owns 8 consecutive rows of one
// column -- one 16-byte row of the consumer's split image when Yh is wanted.  (The body is as_reduce_epilogue, conv_gemm.h.)
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const ConvGemmArgs a, int S)
{
    as_reduce_epilogue(a, S, blockIdx.x * 256 + threadIdx.x, blockIdx.y);
}

// the same for the K-sliced problems of ONE multi-problem launch: blockIdx.x walks the problems' column blocks back to back (a launch
// per problem was 3-5 launches of ~6 us behind every merged launch at batch 1)
struct ReduceMulti {
    int32_t n, pad_;
    int32_t blk0[H3_MAXP + 2];           // first column block of problem i (blk0[n] = all)
    int32_t S[H3_MAXP];
    ConvGemmArgs a[H3_MAXP];
};
__global__ void __launch_bounds__(256)
splitk_reduce_multi_kernel(const ReduceMulti rm)
{
    int pi = 0;
#pragma unroll
    for (int i = 1; i < H3_MAXP; ++i) pi += (i < rm.n && (int)blockIdx.x >= rm.blk0[i]) ? 1 : 0;
    as_reduce_epilogue(rm.a[pi], rm.S[pi], ((int)blockIdx.x - rm.blk0[pi]) * 256 + threadIdx.x, blockIdx.y);
}

// ----------------------------------------------------------------------------------------------------------------
// Reduction of a K-sliced launch WITH the AdaIN1d (+ LeakyReLU) that reads its result (as_conv_gemm_multi_post_f32): at batch 1 every
// conv of an AdainResBlk1d is cut into K slices, and the kernel that sums the slabs of 8 channels over an utterance of <= 256 columns
// holds everything the instance norm's statistics need -- so it writes the NEXT conv's operand image itself (models.py:189-197: conv ->
// norm -> actv -> conv; one dependent launch less per conv, ~7 us each in BASELINE config C2's chain of ~170).
// The mapping is adain_image_kernel's (conv_gemm_h3.hip): a WAVE owns one 16-byte row group of the image -- 8 channels (k-block kb,
// k-half kh) of one utterance --, lane = column (i = lane + 64 j, j < 4), values in registers from the slab loads to the last store; the
// slabs are summed in the plain reduction's order (s ascending), the value epilogue is the plain one (as_reduce_value) and the statistics
// are adain_image_kernel's, so the image is bit-identical to reduce -> AdaIN as two launches.
// ----------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ float rp_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// NJ = register columns in use (utterance of <= 64 NJ columns): no condition inside the load loops
template <int NJ>
static __device__ __forceinline__ void reduce_adain_body(const ConvGemmArgs& a, int S, const AsAdainArgs& n, int kb, int kh, int u, int o0, int L)
{
    const int lane = threadIdx.x & 63;
    const int M = a.M;
    const size_t NX = (size_t)a.N + 1;
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(n.yh);
    u32x4_t* yh = reinterpret_cast<u32x4_t*>(a.Yh);
    const size_t plane = ((size_t)kb * 4 + kh) * NX;                    // h part; the l part two planes on
    const int c0 = kb * 16 + kh * 8;
    const int grp = a.n_groups > 1 ? o0 / a.group_cols : 0;             // (an utterance lies inside one weight group)
    const size_t gbase = n.gb_off ? (size_t)n.gb_off[u] : (size_t)u * n.ldgb;
    float g1[8], bt[8], bias8[8];
    unsigned rowb[8];                                                    // byte offset of (row, first column) inside a slab
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int c = c0 + r < M ? c0 + r : M - 1;
        g1[r] = 1.0f + n.gb[gbase + (size_t)c * n.gb_sc];
        bt[r] = n.gb[gbase + (size_t)(M + c) * n.gb_sc];
        bias8[r] = a.bias ? a.bias[(size_t)grp * M + c] : 0.f;
        rowb[r] = (unsigned)(c * a.N + o0) * 4u;
    }
    int col[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) col[j] = min(lane + 64 * j, L - 1);
    // the slabs through ONE buffer descriptor (S M N 4 bytes < 2 GiB: the host falls back otherwise): a load is one instruction -- the
    // lane's (row, column) offset in a register, the slab's in a scalar -- instead of a 64-bit address computed per element
    const unsigned slab_bytes = (unsigned)M * (unsigned)a.N * 4u;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)((unsigned)S * slab_bytes), 0x00020000);
    // the residual first (Y may alias it), then the slabs: four slabs' loads in flight at a time, none behind a branch; additions in the
    // plain reduction's order s = 0 .. S-1
    float res[8][NJ];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = c0 + r < M ? c0 + r : M - 1;
            res[r][j] = a.res ? a.res[(size_t)c * a.ldr + o0 + col[j]] : 0.f;
        }
    float v[8][NJ];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[r][j] = 0.f;
    constexpr int Q = 4;                                                 // slabs per trip: up to 128 loads of a lane in flight
    int s = 0;
    for (; s + Q <= S; s += Q) {
        float t[Q][8][NJ];
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < NJ; ++j) t[q][r][j] = buf_load1(rsS, rowb[r] + (unsigned)col[j] * 4u, (int)((unsigned)(s + q) * slab_bytes));
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < NJ; ++j) v[r][j] += t[q][r][j];
    }
    for (; s < S; ++s) {
        float t[8][NJ];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < NJ; ++j) t[r][j] = buf_load1(rsS, rowb[r] + (unsigned)col[j] * 4u, (int)((unsigned)s * slab_bytes));
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[r][j] += t[r][j];
    }
    // the conv's own epilogue; its outputs (fp32 rows, plain image) as the plain reduction writes them
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool in = c0 + r < M && lane + 64 * j < L;
            const float x = in ? as_reduce_value<true>(a, v[r][j], bias8[r], res[r][j]) : 0.f;
            v[r][j] = x;
            if (in && a.Y) a.Y[(size_t)(c0 + r) * a.ldy + o0 + lane + 64 * j] = x;
        }
    if (yh) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = lane + 64 * j;
            float t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = (a.yh_lrelu && v[r][j] < 0.f) ? a.in_slope * v[r][j] : v[r][j];
            u32x4_t h, l;
            split2(t, h, l);
            if (i < L) {
                yh[plane + o0 + i] = h;
                yh[plane + o0 + i + 2 * NX] = l;
            }
        }
    }
    // AdaIN over the utterance (adain_image_kernel's arithmetic: a lane adds its elements in ascending order, then the wave's butterfly)
    float mean[8], rstd[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lane + 64 * j < L) sacc += v[r][j];
        mean[r] = rp_wave_sum(sacc) / (float)L;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float vacc = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lane + 64 * j < L) { const float d = __fsub_rn(v[r][j], mean[r]); vacc = __fmaf_rn(d, d, vacc); }
        rstd[r] = 1.0f / sqrtf(rp_wave_sum(vacc) / (float)L + 1e-5f);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = lane + 64 * j;
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) t[r] = (c0 + r < M && i < L) ? as_adain_val(v[r][j], mean[r], rstd[r], g1[r], bt[r], n.lrelu) : 0.f;
        u32x4_t h, l;
        split2(t, h, l);
        if (i < L) {
            xs[plane + o0 + i] = h;
            xs[plane + o0 + i + 2 * NX] = l;
        }
    }
}

// workgroup = ONE wave = one 16-byte row group (8 channels) of one utterance: `rg` = 2 kb + kh
template <int NJ>
static __device__ __forceinline__ void reduce_adain(const ConvGemmArgs& a, int S, const AsAdainArgs& n, int rg, int u)
{
    const int lane = threadIdx.x & 63;
    const int kb = rg >> 1, kh = rg & 1;
    const size_t NX = (size_t)a.N + 1;
    const size_t plane = ((size_t)kb * 4 + kh) * NX;
    if (u == 0 && lane == 0) {                                          // the zero columns
        u32x4_t* xs = reinterpret_cast<u32x4_t*>(n.yh);
        u32x4_t* yh = reinterpret_cast<u32x4_t*>(a.Yh);
        xs[plane + a.N] = u32x4_t{0u, 0u, 0u, 0u};
        xs[plane + 2 * NX + a.N] = u32x4_t{0u, 0u, 0u, 0u};
        if (yh) {
            yh[plane + a.N] = u32x4_t{0u, 0u, 0u, 0u};
            yh[plane + 2 * NX + a.N] = u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    // (one utterance -- batch 1 -- owns every column: no dependent load of its offsets in front of the slab loads; the offsets it was
    // given are still read -- nothing waits for them until the end -- and an utterance that is NOT columns [0, N) is reported: the
    // two-launch form, as_adain_image_f32, honours col_off, and the two must never differ silently)
    int o0 = 0, L = a.N, c_lo = 0, c_hi = a.N;
    // (explicit widths -- a capacity layout: the utterance is NOT all of the launch's columns even when it is the only one)
    if (n.U > 1 || n.col_w) { o0 = n.col_off[u]; L = n.col_w ? n.col_w[u] : n.col_off[u + 1] - o0; }
    else if (n.col_off) { c_lo = n.col_off[0]; c_hi = n.col_off[1]; }
    if (L <= 0) return;
    if (L > 64 * NJ) {                                                   // the caller's post_max_w was not the widest utterance: say so, write nothing wrong silently
        if (lane == 0) as_status_raise(a.status, AS_STATUS_BAD_LAYOUT);
        return;
    }
    reduce_adain_body<NJ>(a, S, n, kb, kh, u, o0, L);                    // (L <= 64 NJ: the host picks the instantiation from the widest utterance)
    if ((c_lo != 0 || c_hi != a.N) && lane == 0) as_status_raise(a.status, AS_STATUS_BAD_LAYOUT);
}

// ----------------------------------------------------------------------------------------------------------------
// Reduction of a K-sliced launch WITH the channel LayerNorm (+ ReLU) that reads its result (as_conv_gemm_multi_post_f32, post_ln): the
// encoders' conv -> residual add -> LayerNorm -> conv (RelTransformerEnc.py:72-87, 318-325) at batch-1 sizes.  The slices stored their
// partial sums TIME-MAJOR (slab_store_tr), so a column's M <= 512 channels lie together: a WAVE owns one column, lane g its channels
// 8 g .. 8 g + 7 -- one 16-byte row group of the image.  Slabs summed in the plain reduction's order, the plain value epilogue, and
// channel_ln_split_kernel's statistics in ITS order of additions (a thread there sums groups g and g + 32, the 32 partial sums are
// added in ascending order): the image is bit-identical to conv -> reduce -> as_channel_layernorm_split_f32.
// ----------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ float rp_ln_total(float own, bool has)
{
    // sum_{q = 0 .. 31} partial[q], ascending, partial[q] in lane q (channel_ln_split_kernel: tot += red[q][col])
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) tot += __shfl(own, q);
    (void)has;
    return tot;
}

static __device__ __forceinline__ void reduce_ln(const ConvGemmArgs& a, int S, const AsLnArgs& n, int j)
{
    const int g = threadIdx.x & 63;
    const int M = a.M, N = a.N;
    const int groups = 2 * as_kbx(M);                                    // row groups of the image (zero beyond M)
    const size_t NX = (size_t)N + 1;
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(n.yh);
    const size_t plane = ((size_t)(g >> 1) * 4 + (g & 1)) * NX;
    if (j == 0 && g < groups) {                                          // the zero column
        xs[plane + N] = u32x4_t{0u, 0u, 0u, 0u};
        xs[plane + 2 * NX + N] = u32x4_t{0u, 0u, 0u, 0u};
    }
    const bool live = 8 * g < M;                                         // (M % 8 == 0: a lane's eight channels exist or do not)
    const int c0 = live ? 8 * g : M - 8;
    const int grp = a.n_groups > 1 ? j / a.group_cols : 0;
    const int lg = n.gamma2 ? j / n.n_split : 0;
    const float* gam = n.gamma + (ptrdiff_t)lg * (n.gamma2 - n.gamma) + c0;
    const float* bet = n.beta + (ptrdiff_t)lg * (n.beta2 - n.beta) + c0;
    // everything this lane loads, issued before the first use: gamma / beta / bias (16-byte loads), the residual's eight rows, the slabs
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam), g1 = *reinterpret_cast<const f32x4*>(gam + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bet), b1 = *reinterpret_cast<const f32x4*>(bet + 4);
    f32x4 bi0 = {0.f, 0.f, 0.f, 0.f}, bi1 = bi0;
    if (a.bias) {
        bi0 = *reinterpret_cast<const f32x4*>(a.bias + (size_t)grp * M + c0);
        bi1 = *reinterpret_cast<const f32x4*>(a.bias + (size_t)grp * M + c0 + 4);
    }
    float res[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) res[r] = a.res ? a.res[(size_t)(c0 + r) * a.ldr + j] : 0.f;
    const float* slab = reinterpret_cast<const float*>(a.ws) + (size_t)j * M + c0;
    const size_t total = (size_t)M * N;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        f32x4 t[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t[q][0] = *reinterpret_cast<const f32x4*>(slab + (size_t)(s + q) * total);
            t[q][1] = *reinterpret_cast<const f32x4*>(slab + (size_t)(s + q) * total + 4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += t[q][r >> 2][r & 3];
    }
    for (; s < S; ++s) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(slab + (size_t)s * total), t1 = *reinterpret_cast<const f32x4*>(slab + (size_t)s * total + 4);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += r < 4 ? t0[r] : t1[r - 4];
    }
    const float bias8[8] = {bi0[0], bi0[1], bi0[2], bi0[3], bi1[0], bi1[1], bi1[2], bi1[3]};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float x = live ? as_reduce_value<true>(a, v[r], bias8[r], res[r]) : 0.f;
        v[r] = x;
        if (live && a.Y) a.Y[(size_t)(c0 + r) * a.ldy + j] = x;
    }
    // statistics over the M channels of the column, in channel_ln_split_kernel's order: lane p < 32 adds its group's eight values, then
    // group p + 32's (the groups beyond: zeros there too), and the 32 partial sums are added in ascending order
    float hi[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) hi[r] = __shfl_down(v[r], 32);
    float sp = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sp += v[r];
#pragma unroll
    for (int r = 0; r < 8; ++r) sp += hi[r];
    const float mean = rp_ln_total(sp, true) / (float)M;
    float q2 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float d = __fsub_rn(v[r], mean);
        if (8 * g + r < M) q2 = __fmaf_rn(d, d, q2);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float d = __fsub_rn(hi[r], mean);
        if (8 * (g + 32) + r < M) q2 = __fmaf_rn(d, d, q2);
    }
    const float rs = 1.0f / sqrtf(rp_ln_total(q2, true) / (float)M + n.eps);
    if (g >= groups) return;
    const float gam8[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
    const float bet8[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    float t[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float o = 0.f;
        if (live) {
            o = __fmaf_rn(__fmul_rn(__fsub_rn(v[r], mean), rs), gam8[r], bet8[r]);
            if (n.relu) o = o < 0.f ? 0.f : o;
        }
        t[r] = o;
    }
    u32x4_t h, l;
    split2(t, h, l);
    xs[plane + j] = h;
    xs[plane + 2 * NX + j] = l;
}

// the K-sliced problems of ONE (multi-problem) launch: problem i either summed plainly (mode 0: a workgroup = 64 columns of one 8-row
// group, as_reduce_epilogue) or summed + AdaIN'd (mode 1: a workgroup = one row group of one utterance); workgroups of ONE wave -- the
// few dozen waves of such a launch then sit on as many CUs, each with a texture path of its own (with four waves to a workgroup the
// launch took 21 us where reduce + AdaIN as two launches took 14) --, a 1-D grid of exactly the workgroups every problem needs
struct ReducePost {
    int32_t n, pad_;
    int32_t blk0[H3_MAXP + 2];           // first workgroup of problem i (blk0[n] = all)
    int32_t S[H3_MAXP], mode[H3_MAXP];
    int32_t nxb[H3_MAXP];                // mode 0: column blocks; mode 1: row groups -- the fast index of the problem's workgroups
    ConvGemmArgs a[H3_MAXP];
    AsAdainArgs p[H3_MAXP];
    AsLnArgs l[H3_MAXP];                 // mode 2: + channel LayerNorm (a workgroup = one column; the slabs are time-major)
};
template <int NJ>
__global__ void __launch_bounds__(64)
splitk_reduce_post_kernel(const ReducePost rm)
{
    int pi = 0;
#pragma unroll
    for (int i = 1; i < H3_MAXP; ++i) pi += (i < rm.n && (int)blockIdx.x >= rm.blk0[i]) ? 1 : 0;
    const int local = (int)blockIdx.x - rm.blk0[pi], nxb = rm.nxb[pi];
    if (rm.mode[pi] == 0) as_reduce_epilogue(rm.a[pi], rm.S[pi], (local % nxb) * 64 + (int)threadIdx.x, local / nxb);
    else if (rm.mode[pi] == 1) reduce_adain<NJ>(rm.a[pi], rm.S[pi], rm.p[pi], local % nxb, local / nxb);
    else reduce_ln(rm.a[pi], rm.S[pi], rm.l[pi], local);
}

__global__ void __launch_bounds__(64)
splitk_reduce_ln_kernel(const ConvGemmArgs a, int S, const AsLnArgs n)
{
    reduce_ln(a, S, n, (int)blockIdx.x);
}

// ONE problem: its arguments directly in the kernel argument segment (every scalar load of the wave's head is issued at once; with the
// problem list a wave first looks its problem up, then fetches that problem's fields: one more dependent round trip in a kernel that
// consists of four)
template <int NJ>
__global__ void __launch_bounds__(64)
splitk_reduce_adain_kernel(const ConvGemmArgs a, int S, const AsAdainArgs n, int nrg)
{
    reduce_adain<NJ>(a, S, n, (int)blockIdx.x % nrg, (int)blockIdx.x / nrg);
}

// as_set_range_probe: every launch tests its accumulators for inf / NaN (what an operand beyond fp16's range turns into)
static int g_range_probe = 0;
extern "C" int as_set_range_probe(int on)
{
    g_range_probe = on != 0;
    return AS_OK;
}

static void tile_dims(int choice, int* bm, int* bn)
{
    // 42 / 22 / 21 / 12 / 11 / 14 / 2: 256x128, 128x128, 128x64, 64x128, 64x64, 64x256, 32x128 (4 waves each)
    *bm = choice == 42 ? 256 : (choice == 22 || choice == 21) ? 128 : choice == 2 ? 32 : 64;
    *bn = choice == 14 ? 256 : (choice == 42 || choice == 22 || choice == 12 || choice == 2) ? 128 : 64;
}

// Tile choice, fitted to sweeps on MI355X (scripts/gemm_bench.py, round 2).  The time of a launch is the time of its busiest CU:
// a tile alone on a CU costs t1 (relative to 128x128: 128x64 / 64x128 0.78 -- they split K over wave pairs and stage more per flop --
// 64x64 0.59), n > 1 tiles sharing a CU cost 0.71 n t1 (two co-resident workgroups cover each other's waits).  Examples the fit
// reproduces: M512 N6400 K512 T3: 128x128 (200 tiles) 33.5 us, 128x64 (400) 37.1; M512 N3840 K512 T5: 44.8 (120 tiles) against 35.0
// (240); M512 N1280 K512 T3: 25.6 / 17.8 / 15.2 us for 40 / 80 / 160 tiles.
static int gemm_tile_choice(int M, int N, int n_prod, int Kp)
{
    const char* env = getenv("AS_GEMM_TILE");           // tuning/experiments only: 42, 22, 21, 12, 11, 14
    if (env && atoi(env) > 0) return atoi(env);
    if (M <= 32 && n_prod == 3) return 2;                              // a 64-row tile would be half empty (HiFi-GAN's last stage: 32 channels, 1.9 M columns)
    // <= 64 output channels over many columns (the towers' first convs, 64 x 509 440): the 64 x 128 tile splits K over wave pairs and
    // reads six fragments per six products; 64 x 256 gives every wave a 64 x 64 block over the whole k (eight per twelve, like 128 x 128):
    // M64 N509440 K64 T9 167 -> 139 us, N128000 41 -> 34, N63680 20.8 -> 19.7; below one round of the chip (N6400: 8.6 -> 10.8) it loses.
    if (M <= 64 && n_prod == 3 && as_cdiv(N, 256) >= 240) return 14;
    const bool tall = M > 64 && (M % 128 == 0 || M % 128 > 64 || M >= 512);   // a 128-row tile is not half empty
    static const int choices[5] = {42, 22, 21, 12, 11};
    // 256x128 (a wave owns 128 x 64): a third less LDS traffic per matrix-core product, but 364 registers: ONE workgroup per CU.
    // Ten identical launches back to back in a hipGraph: M1024 N6400 K1216 T3 127 us against 152 for 128x128, M256 N32000 K256 T9 94
    // against 106, M512 N19200 105 against 96.  Inside the step the same launches take the same time with either tile (C3: 5.08 against
    // 5.10 ms of GEMM per step, C5: 8.78 against 8.70), so the default stays with the tiles that share a CU; -DAS_EXPERIMENTS builds
    // carry it (AS_GEMM_TILE=42 / AS_GEMM_USE42=1).
    double t1[5] = {1.55, 1.0, 0.78, 0.78, 0.59};
#ifdef AS_EXPERIMENTS
    static const bool use42 = getenv("AS_GEMM_USE42") != nullptr;
    static const double t42 = getenv("AS_GEMM_T42") ? atof(getenv("AS_GEMM_T42")) : 1.55;   // (tuning: the 256 x 128 tile's cost alone on a CU)
    t1[0] = t42;
#else
    constexpr bool use42 = false;
#endif
    int best = 11;
    double best_cost = 1e30;
    for (int c = 0; c < 5; ++c) {
        int bm, bn;
        tile_dims(choices[c], &bm, &bn);
        if (bm >= 128 && !tall) continue;
        if (bm == 256 && (M % 256 != 0 || n_prod != 3 || !use42)) continue;
        const double tiles = (double)as_cdiv(M, bm) * as_cdiv(N, bn);
        const double n = ceil(tiles / 256.0);
        const double cost = t1[c] * (bm == 256 ? n : (n > 1.0 ? 0.71 * n : 1.0));
        if (cost < best_cost * 0.97) { best_cost = cost; best = choices[c]; }   // (ties go to the larger tile)
    }
    // An image of <= 32 channels may come from the 32-row tile, which leaves the upper half of the image's 64-row block unwritten (clearing
    // it would be 245 MB per launch at the vocoder's last stage): such a K is never given to the tile whose four waves read all four
    // k-blocks of the block (64 x 64); every other tile reads two at most, i.e. rows 0..31, which every producer writes.
    if (Kp <= 32 && best == 11) best = 12;
    return best;
}

// K slices: only where even the smallest tiles leave most CUs idle (the towers' last convs: a few hundred columns, K = 12800).
// Everywhere else one launch without the reduce pass is as fast or faster (M1024 N2560 K512 T9: 94 us unsplit, 100 + 17 in four
// slices; M512 N1280 K512 T3: 16 against 31).
static int gemm_ksplit(int M, int N, int Kp, int T, int choice, int K2 = 0)
{
    const char* env = getenv("AS_GEMM_KSPLIT");          // tuning/experiments only
    int bm, bn;
    tile_dims(choice, &bm, &bn);
    const long tiles = (long)as_cdiv(M, bm) * as_cdiv(N, bn);
    const int wk = choice == 2 ? 2 : bm * bn >= 4 * 64 * 64 ? 1 : 4 * 64 * 64 / (bm * bn);   // waves that split K inside the workgroup
    const int nkt = T * as_cdiv(Kp / 16, wk) + as_cdiv(as_cdiv(K2, 16), wk);   // iterations (k-tile = 16 * WK)
    int s = 1;
    if (env && atoi(env) > 0) s = atoi(env);
    else if (tiles < 64) {
        s = as_cdiv(192, tiles);
        if (s > 16) s = 16;
    } else if (tiles <= 128 && nkt * wk >= 64) {
        // 64 .. 128 workgroups, each alone on its CU: the k loop runs at the latency of its own staging (3 stages in flight per workgroup:
        // M1024 N286 K512 T9 at batch 1, 80 tiles: 49 us = 0.4 TB/s of weights), so K slices that fill the 512 workgroup slots pay even
        // with the reduction pass behind them
        s = 512 / (int)tiles;
    }
    const int env_min = getenv("AS_GEMM_MINKT") ? atoi(getenv("AS_GEMM_MINKT")) : 0;   // tuning/experiments only
    const int min_kt = env_min > 0 ? env_min : 24 / wk;   // a slice keeps >= 384 k
    if (s > nkt / min_kt) s = nkt / min_kt;
    return s < 1 ? 1 : s;
}

// What one call runs: tile, K slices, and whether the activations still have to be split into the operand image.
struct GemmPlan {
    int choice, S;
    size_t slab_bytes, xh_bytes;      // workspace: split-K slabs first, then (256-byte aligned) the split activations
};
static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
static GemmPlan gemm_plan(const ConvGemmArgs& a)
{
    GemmPlan p = {};
    p.choice = gemm_tile_choice(a.M, a.N, a.n_prod ? a.n_prod : 3, a.Kp);
    p.S = a.ileave_u > 1 ? 1 : gemm_ksplit(a.M, a.N, a.Kp, a.T, p.choice, a.K2);   // (the slab reduction writes plain rows)
    p.slab_bytes = p.S > 1 ? (size_t)p.S * a.M * a.N * sizeof(float) : 0;
    p.xh_bytes = a.Xh ? 0 : as_split_f16x2_bytes(a.K, a.N);
    return p;
}

// A launch of 1.x rounds of the chip: 513 .. 1024 tiles of 128 x 128 on 512 workgroup slots (the six 1 024-row decoder convs at 64
// utterances per call: 800 tiles).  Its second round leaves most CUs with one workgroup, at 0.71 of the paired rate.  The last sixth of
// the columns on 128 x 64 tiles instead -- listed last, so they fill the slots the big tiles leave -- levels it: M1024 N12800 K1024 T3
// 227 -> 200-203 us with 12-25 % of the columns small, K1216 270 -> 237-240 (scripts/exp/tilemix_bound.py, two free-running launches:
// the bound; scripts/exp/records/riders_r06.txt).  Launches of one round or less lose (M1024 N6400 + 6 %, M512 N12800 + 4 %): not them.
// Returns the first small column (a multiple of 128), or 0: no mixing.  Same products, and for the small tile's columns the order of
// partial sums of the 128 x 64 tile (two K halves summed through LDS).
// OFF unless AS_GEMM_MIX=1: inside the step the gain is not there -- two coalescing lanes, 40 steps, alternating runs on one box: 3.86-3.87
// ms per step mixed against 3.82-3.84 unmixed (the conv class by events 0.387-0.389 against 0.384-0.386 of the ceiling: the launches are
// a little faster, the step is not) -- like every tile experiment before it.  The kernel and the rule stay, tested.
static int gemm_mix_split(const ConvGemmArgs& a, int choice, int S)
{
    const char* on = getenv("AS_GEMM_MIX");                                // (read per call: tests, A/B runs)
    const bool off = !(on && *on == '1');
    if (off || choice != 22 || S != 1 || a.n_prod != 3 || a.n_groups > 1 || a.M % 128 != 0 || a.slab_tr) return 0;
    const int tn = as_cdiv(a.N, 128), tiles = (a.M / 128) * tn;
    if (tiles <= 512 + 64 || tiles > 1024) return 0;
    const char* fe = getenv("AS_GEMM_MIX_FRAC");
    const double frac = fe ? atof(fe) : 0.17;
    const int small_t = std::max(1, (int)(frac * tn + 0.5));
    const int split = (tn - small_t) * 128;
    return split > 0 && split < a.N ? split : 0;
}

static bool direct_cin1(const ConvGemmArgs& a)
{
    return a.K == 1 && !a.K2 && a.W && a.X && (!a.Yh || a.T <= 9) && !a.res && !a.div_sqrt2 && !a.transpose_out && a.ileave_u <= 1 && a.M <= DIRECT_MAX_M && a.n_groups <= 1 &&
           !getenv("AS_GEMM_NO_DIRECT");
}

// A conv with an AdaIN / LayerNorm behind it (as_conv_gemm_multi_post_f32) whose output is tiny (batch-1 sizes: <= 1 MB) is cut into TWO K
// slices even where the slicing rules would leave it whole: its reduction kernel then replaces the normalisation's launch (the slices
// themselves cost nothing there: the launch is a few microseconds of latency either way).  The workspace always has room for it.
static const size_t kPostSlabMax = (size_t)1 << 20;
static bool post_slice_ok(const ConvGemmArgs& a)
{
    return a.Xh && a.ileave_u <= 1 && !a.transpose_out && (size_t)a.M * a.N * sizeof(float) <= kPostSlabMax &&
           a.T * (a.Kp / 16) + as_cdiv(a.K2, 16) >= 8;
}

extern "C" size_t as_conv_gemm_workspace_bytes(const ConvGemmArgs* a)
{
    if (!a || a->M <= 0 || a->N <= 0 || a->Kp <= 0 || a->T <= 0) return 0;
    if (direct_cin1(*a)) return 0;
    const GemmPlan p = gemm_plan(*a);
    const size_t slabs = std::max(p.slab_bytes, post_slice_ok(*a) ? 2 * (size_t)a->M * a->N * sizeof(float) : (size_t)0);
    return p.xh_bytes ? align256(slabs) + p.xh_bytes : slabs;
}

// Room for the K slices a problem may be cut into inside a multi-problem launch (as_conv_gemm_multi_f32 takes no more slices than
// a.ws_bytes holds slabs for): up to 16 slabs of a small output (<= 4 MB: anything larger has tiles enough to never need them), each
// slice keeping >= 384 k; never less than the single launch wants.
extern "C" size_t as_conv_gemm_multi_workspace_bytes(const ConvGemmArgs* a)
{
    const size_t single = as_conv_gemm_workspace_bytes(a);
    if (!a || a->M <= 0 || a->N <= 0 || a->Kp <= 0 || a->T <= 0 || !a->Xh) return single;
    const size_t slab = (size_t)a->M * a->N * sizeof(float);
    if (slab > ((size_t)4 << 20)) return single;
    const int nkt = a->T * (a->Kp / 16) + as_cdiv(a->K2, 16);
    const int s = std::min(16, nkt / 24);
    return std::max(single, s > 1 ? (size_t)s * slab : (size_t)0);
}

// which kernel a call with these arguments runs (tests, tuning): kind 0 = direct Cin = 1, 1 = tiled (tile = 22 / 21 / 12 / 11 / 14 / 2)
extern "C" int as_conv_gemm_plan(const ConvGemmArgs* a, int32_t* kind, int32_t* tile, int32_t* slices)
{
    if (!a || !kind || !tile || !slices || a->M <= 0 || a->N <= 0 || a->Kp <= 0 || a->T <= 0) return AS_EINVAL;
    ConvGemmArgs n = *a;
    if (n.n_prod == 0) n.n_prod = 3;
    if (n.n_groups < 1) n.n_groups = 1;
    *tile = 0;
    *slices = 1;
    if (direct_cin1(n)) { *kind = 0; return AS_OK; }
    const GemmPlan p = gemm_plan(n);
    *kind = 1;
    *tile = p.choice;
    *slices = p.S;
    return AS_OK;
}

// argument checks and defaults shared by the single and the multi-problem entry point: `norm` becomes what the kernels are given
static int conv_gemm_normalise(const ConvGemmArgs* args_host, ConvGemmArgs& norm)
{
    if (!args_host) return AS_EINVAL;
    norm = *args_host;
    if (!(fabsf(norm.in_slope) <= 3.0e38f) || !(fabsf(norm.act_slope) <= 3.0e38f)) return AS_EINVAL;   // slopes are used as given
    if (norm.acc_scale == 0.f) norm.acc_scale = 1.0f;
    norm.status = as_status_words_device();
    norm.slab_tr = 0;
    norm.range_probe = (g_range_probe || args_host->range_probe == AS_PROBE_THIS) ? 1 : 0;
    if (norm.n_prod == 0) norm.n_prod = 3;
    if (norm.n_groups < 1) norm.n_groups = 1;
    // a 1x1 conv reads every column from itself: no tap can leave the utterance, so the kernels need not fetch the column descriptors
    // (one dependent global load at the head of every workgroup)
    if (norm.T == 1 && norm.dh[0] == 0 && norm.dw[0] == 0 && !norm.src_col) norm.meta = nullptr;
    const ConvGemmArgs& a = norm;
    if (a.act < 0 || a.act > 5 || (a.in_act != 0 && a.in_act != 2) || (a.n_prod != 1 && a.n_prod != 3)) return AS_EINVAL;
    if ((!a.Wh && !(a.W && a.K == 1)) || (!a.X && !a.Xh) || (!a.Y && !a.Yh) || a.M <= 0 || a.N < 0 || a.K <= 0 || a.T <= 0 || a.T > AS_MAX_TAPS)
        return AS_EINVAL;
    if (a.Kp < a.K || a.Kp % 16 || a.Kp - a.K >= 16) return AS_EINVAL;
    if (a.K2 < 0 || (a.K2 > 0 && (!a.Xh2 || !a.Xh || !a.Wh || (reinterpret_cast<uintptr_t>(a.Xh2) & 15) != 0 ||
                                  (double)as_kbx(a.K2) * 64.0 * (a.N + 1.0) >= 2147483648.0)))
        return AS_EINVAL;                                                // the second operand comes as an image, beside an image
    if (a.src_col && (!a.Xh || !a.meta || a.K2 || a.N_in <= 0)) return AS_EINVAL;   // own input layout: an image, with the input positions
    if (a.ileave_u < 0 || a.ileave_u == 1) return AS_EINVAL;
    if (a.ileave_u > 1 && (a.M % a.ileave_u || (a.M / a.ileave_u) % 32 || !a.Y || a.Yh || a.res || a.transpose_out || a.n_groups > 1 ||
                           (long)a.ldy < (long)a.ileave_u * a.N || (double)(a.M / a.ileave_u) * a.ldy * 4.0 >= 2147483648.0))
        return AS_EINVAL;
    if (a.n_groups > 1 && (a.group_cols <= 0 || (long)a.group_cols * a.n_groups < a.N)) return AS_EINVAL;
    if ((a.X && a.ldx < a.N) || (a.Y && a.ldy < (a.transpose_out ? a.M : a.N)) || (a.res && (a.ldr < a.N || a.transpose_out)) ||
        (a.Yh && a.transpose_out))
        return AS_EINVAL;
    if (((reinterpret_cast<uintptr_t>(a.Wh) | reinterpret_cast<uintptr_t>(a.Xh) | reinterpret_cast<uintptr_t>(a.Yh)) & 15) != 0) return AS_EINVAL;
    // 32-bit byte offsets inside the buffer descriptors
    if (((double)a.T * as_kbx(a.K) + (a.K2 ? as_kbx(a.K2) : 0)) * 64.0 * a.M >= 2147483648.0 || (a.X && (double)a.K * a.ldx * 4.0 + 16.0 >= 4294967296.0)) return AS_EINVAL;
    if ((a.Y && (double)(a.transpose_out ? a.N : a.M) * a.ldy * 4.0 >= 2147483648.0) || (double)a.M * a.ldr * 4.0 >= 2147483648.0 ||
        (double)a.M * a.N * 4.0 >= 2147483648.0 || (a.Yh && (double)as_kbx(a.M) * 64.0 * (a.N + 1.0) >= 2147483648.0))
        return AS_EINVAL;
    return AS_OK;
}

static void gemm_tag(const ConvGemmArgs& a, char* tag, size_t n)
{
    char sc[24] = "";
    if (a.K2) snprintf(sc, sizeof(sc), "+K%d", a.K2);
    snprintf(tag, n, "M%d N%d K%d T%d%s", a.M, a.N, a.K, a.T, sc);
}
static double gemm_flops(const ConvGemmArgs& a) { return 2.0 * a.M * a.N * ((double)a.K * a.T + a.K2); }
// algorithmic bytes: weights + inputs + output once (4 bytes per element)
static double gemm_bytes(const ConvGemmArgs& a)
{
    return 4.0 * (((double)a.T * a.K + a.K2) * a.M * a.n_groups + ((double)a.K + a.K2) * a.N + (double)a.M * a.N);
}
static int launch_reduce(const ConvGemmArgs& a, int S, hipStream_t stream)
{
    const int rows = a.Yh ? (16 * as_kbx(a.M) > a.M ? 16 * as_kbx(a.M) : a.M) : a.M;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(as_cdiv(a.N + 1, 256), as_cdiv(rows, 8)), dim3(256), 0, stream, a, S);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// the fast direct form (T <= 9) for n problems
static int launch_direct_x4(const ConvGemmArgs* a, int n, hipStream_t stream)
{
    DirectMulti dm;
    memset(&dm, 0, sizeof(dm));
    dm.n = n;
    bool vec = true;
    for (int i = 0; i < n; ++i) {
        dm.a[i] = a[i];
        const int nx = as_cdiv(a[i].N, 1024), groups = a[i].Yh ? 2 * as_kbx(a[i].M) : (a[i].M + 7) / 8;
        dm.nx[i] = nx;
        dm.ny[i] = std::max(1, std::min(groups, as_cdiv(1024, nx)));               // >= ~1024 workgroups when the channels allow it
        dm.blk0[i + 1] = dm.blk0[i] + nx * dm.ny[i];
        vec = vec && a[i].Y && (a[i].N & 3) == 0 && (a[i].ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a[i].Y) & 15) == 0;
    }
    if (vec) hipLaunchKernelGGL(conv_direct_cin1_x4_kernel<true>, dim3(dm.blk0[n]), dim3(256), 0, stream, dm);
    else hipLaunchKernelGGL(conv_direct_cin1_x4_kernel<false>, dim3(dm.blk0[n]), dim3(256), 0, stream, dm);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// the AdaIN that reads a conv's result (as_conv_gemm_multi_post_f32): its arguments completed from the conv's
static bool post_wanted(const AsAdainArgs* post) { return post && post->yh; }
static int post_normalise(const ConvGemmArgs& a, const AsAdainArgs& post_host, AsAdainArgs& n)
{
    n = post_host;
    if (!a.Y || a.transpose_out || a.ileave_u > 1 || !n.gb || !n.col_off || n.U <= 0 || n.gb_sc <= 0 || (!n.gb_off && n.ldgb <= 0) ||
        (reinterpret_cast<uintptr_t>(n.yh) & 15) != 0)
        return AS_EINVAL;
    n.x = a.Y; n.ldx = a.ldy; n.C = a.M; n.N = a.N;
    n.src_off = nullptr; n.pool_w = nullptr; n.pool_b = nullptr; n.x_up = nullptr; n.ld_up = 0;
    return AS_OK;
}
static bool ln_wanted(const AsLnArgs* l) { return l && l->yh; }
static int ln_check(const ConvGemmArgs& a, const AsLnArgs& n)
{
    if (!a.Y || a.transpose_out || a.ileave_u > 1 || !n.gamma || !n.beta || ((n.gamma2 == nullptr) != (n.beta2 == nullptr)) ||
        (n.gamma2 && n.n_split <= 0) || (reinterpret_cast<uintptr_t>(n.yh) & 15) != 0 || !(n.eps >= 0.f))
        return AS_EINVAL;
    return AS_OK;
}
// can the reduction kernel of this K-sliced problem write the LayerNorm image itself?  (a wave per column, a lane per 8 channels; 16-byte loads)
static bool ln_fusable(const ConvGemmArgs& a, const AsLnArgs& n)
{
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const ptrdiff_t gs = n.gamma2 ? n.gamma2 - n.gamma : 0, bs = n.beta2 ? n.beta2 - n.beta : 0;
    return a.M % 8 == 0 && a.M <= 512 && a.act <= 2 && !a.Yh && al16(n.gamma) && al16(n.beta) && gs % 4 == 0 && bs % 4 == 0 &&
           (!a.bias || al16(a.bias)) && (double)a.M * a.N * 4.0 * 16.0 < 2147483648.0 && !getenv("AS_NO_REDUCE_LN");
}
// can the reduction kernel of this K-sliced problem write the AdaIN image itself?
static bool post_fusable(const ConvGemmArgs& a, int max_w) { return max_w > 0 && max_w <= 256 && a.act <= 2 && (double)a.M * a.N * 4.0 * 16.0 < 2147483648.0 && !getenv("AS_NO_REDUCE_ADAIN"); }

// one launch for the reductions of the K-sliced problems ptr[k] (So[k] > 1) of a launch; np[k] non-NULL: with the AdaIN behind it
static int launch_reduce_post(const ConvGemmArgs* const* ptr, const int* So, const AsAdainArgs* const* np, const int* max_w,
                              const AsLnArgs* const* nl, int m, hipStream_t stream)
{
    ReducePost rm;
    memset(&rm, 0, sizeof(rm));
    int mw = 1;
    for (int k = 0; k < m; ++k)
        if (So[k] > 1 && np[k]) mw = std::max(mw, max_w[k]);
    const int nj = (mw + 63) / 64;                                       // register columns per lane (post_fusable: <= 4)
    for (int k = 0; k < m; ++k)
        if (So[k] > 1) {
            const ConvGemmArgs& a = *ptr[k];
            const int i = rm.n;
            rm.a[i] = a;
            rm.S[i] = So[k];
            int blocks;
            if (nl && nl[k]) {
                rm.mode[i] = 2;
                rm.l[i] = *nl[k];
                rm.nxb[i] = 1;
                blocks = a.N;
            } else if (np[k]) {
                rm.mode[i] = 1;
                rm.p[i] = *np[k];
                rm.nxb[i] = as_kbx(a.M) * 2;
                blocks = rm.nxb[i] * np[k]->U;
            } else {
                const int rows = a.Yh ? std::max(16 * as_kbx(a.M), a.M) : a.M;
                rm.nxb[i] = as_cdiv(a.N + 1, 64);
                blocks = rm.nxb[i] * as_cdiv(rows, 8);
            }
            rm.blk0[i + 1] = rm.blk0[i] + blocks;
            ++rm.n;
        }
    if (rm.n == 0) return AS_OK;
    if (rm.n == 1 && rm.mode[0] == 2) {
        hipLaunchKernelGGL(splitk_reduce_ln_kernel, dim3(rm.blk0[1]), dim3(64), 0, stream, rm.a[0], rm.S[0], rm.l[0]);
        AS_CHECK_LAUNCH();
        return AS_OK;
    }
    if (rm.n == 1 && rm.mode[0] == 1) {
        const dim3 g1(rm.blk0[1]), b1(64);
        if (nj == 1) hipLaunchKernelGGL(splitk_reduce_adain_kernel<1>, g1, b1, 0, stream, rm.a[0], rm.S[0], rm.p[0], rm.nxb[0]);
        else if (nj == 2) hipLaunchKernelGGL(splitk_reduce_adain_kernel<2>, g1, b1, 0, stream, rm.a[0], rm.S[0], rm.p[0], rm.nxb[0]);
        else if (nj == 3) hipLaunchKernelGGL(splitk_reduce_adain_kernel<3>, g1, b1, 0, stream, rm.a[0], rm.S[0], rm.p[0], rm.nxb[0]);
        else hipLaunchKernelGGL(splitk_reduce_adain_kernel<4>, g1, b1, 0, stream, rm.a[0], rm.S[0], rm.p[0], rm.nxb[0]);
        AS_CHECK_LAUNCH();
        return AS_OK;
    }
    const dim3 gm(rm.blk0[rm.n]), bm(64);
    if (nj == 1) hipLaunchKernelGGL(splitk_reduce_post_kernel<1>, gm, bm, 0, stream, rm);
    else if (nj == 2) hipLaunchKernelGGL(splitk_reduce_post_kernel<2>, gm, bm, 0, stream, rm);
    else if (nj == 3) hipLaunchKernelGGL(splitk_reduce_post_kernel<3>, gm, bm, 0, stream, rm);
    else hipLaunchKernelGGL(splitk_reduce_post_kernel<4>, gm, bm, 0, stream, rm);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

static int conv_gemm_one(const ConvGemmArgs* args_host, const AsAdainArgs* post_host, int post_max_w, const AsLnArgs* ln_host, hipStream_t stream)
{
    ConvGemmArgs norm;
    const int rn = conv_gemm_normalise(args_host, norm);
    if (rn != AS_OK) return rn;
    const ConvGemmArgs& a = norm;
    if (a.N == 0) return AS_OK;
    AsAdainArgs post;
    const bool want_post = post_wanted(post_host), want_ln = ln_wanted(ln_host);
    if (want_post && want_ln) return AS_EINVAL;
    if (want_post) {
        const int rp = post_normalise(a, *post_host, post);
        if (rp != AS_OK) return rp;
    }
    if (want_ln) {
        const int rl = ln_check(a, *ln_host);
        if (rl != AS_OK) return rl;
    }
    bool post_done = false;
    {
    const int rc1 = [&]() -> int {
    if (direct_cin1(a)) {
        char tag[64];
        snprintf(tag, sizeof(tag), "M%d N%d K1 T%d direct", a.M, a.N, a.T);
        AsProfScope prof__(AS_CLS_GEMM, 2.0 * a.M * a.N * (double)a.T, 4.0 * ((double)a.T * a.M + a.N + (double)a.M * a.N), stream, tag);
        if (a.T <= 9) {
            const int rd = launch_direct_x4(&a, 1, stream);
            if (rd != AS_OK) return rd;
        } else {
            hipLaunchKernelGGL(conv_direct_cin1_kernel, dim3(as_cdiv(a.N, 256)), dim3(256), 0, stream, a);
        }
        AS_CHECK_LAUNCH();
        return AS_OK;
    }
    if (!a.Wh) return AS_EINVAL;
    GemmPlan plan = gemm_plan(a);
    if (plan.xh_bytes) {                                   // no operand image from the caller: it needs room in the workspace
        if (!a.ws || a.ws_bytes < plan.xh_bytes) return AS_EINVAL;
        if (a.ws_bytes < align256(plan.slab_bytes) + plan.xh_bytes) { plan.S = 1; plan.slab_bytes = 0; }
    } else if (plan.S > 1 && (!a.ws || a.ws_bytes < plan.slab_bytes)) {
        plan.S = 1;                                        // no workspace: no K slices
        plan.slab_bytes = 0;
    }
    // (see post_slice_ok: whatever AS_NO_REDUCE_* says -- the switch changes who normalises, never the conv's own arithmetic)
    int S_ = plan.S;
    if (S_ == 1 && post_slice_ok(a) && a.ws && a.ws_bytes >= 2 * (size_t)a.M * a.N * sizeof(float) &&
        ((want_ln && a.M % 8 == 0 && a.M <= 512 && !a.Yh && a.act <= 2) || (want_post && post_max_w > 0 && post_max_w <= 256 && a.act <= 2)))
        S_ = 2;
    const int S = S_;
    char tag[96], shape[64];
    gemm_tag(a, shape, sizeof(shape));
    snprintf(tag, sizeof(tag), "%s tile%d S%d%s%s%s", shape, plan.choice, S, a.n_prod == 1 ? " h1" : "", a.Xh ? "" : " +split",
             a.Yh ? (a.Y ? " y+yh" : " yh") : "");
    // algorithmic work of this launch: 2*M*N*(K*T + K2) flop; bytes = weights + inputs + output once (4 bytes per element)
    AsProfScope prof__(AS_CLS_GEMM, gemm_flops(a), gemm_bytes(a), stream, tag);
    if (!a.Xh) {                                                        // split once, behind the K slabs in the workspace
        uint16_t* xh = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(a.ws) + align256(plan.slab_bytes));
        const int rc = as_split_f16x2_launch(a.X, a.ldx, a.K, a.N, a.in_act == 2, a.in_slope, xh, stream);
        if (rc != AS_OK) return rc;
        norm.Xh = xh;
    }
    const ConvGemmArgs* one = &a;
    const bool fuse_ln = S > 1 && want_ln && ln_fusable(a, *ln_host);
    norm.slab_tr = fuse_ln ? 1 : 0;                                     // (the slices store time-major for the reduction that normalises columns)
    const int mix = gemm_mix_split(a, plan.choice, S);
    const int rc = mix ? as_conv_gemm_h3_launch_mix(one, mix, stream) : as_conv_gemm_h3_launch(&one, &S, 1, plan.choice, stream);
    if (rc != AS_OK) return rc;
    if (fuse_ln) {
        const AsAdainArgs* np = nullptr;
        const AsLnArgs* nl = ln_host;
        post_done = true;
        return launch_reduce_post(&one, &S, &np, &post_max_w, &nl, 1, stream);
    }
    if (S > 1 && want_post && post_fusable(a, post_max_w)) {
        const AsAdainArgs* np = &post;
        post_done = true;
        return launch_reduce_post(&one, &S, &np, &post_max_w, nullptr, 1, stream);
    }
    if (S > 1) return launch_reduce(a, S, stream);
    return AS_OK;
    }();
    if (rc1 != AS_OK) return rc1;
    }
    if (want_post && !post_done) return as_adain_image_f32(&post, stream);   // (its own profiling scope, behind the conv's)
    if (want_ln && !post_done)
        return as_channel_layernorm_split_f32(a.Y, a.ldy, a.M, a.N, ln_host->gamma, ln_host->beta, ln_host->gamma2, ln_host->beta2, ln_host->n_split,
                                              ln_host->eps, ln_host->relu, ln_host->yh, stream);
    return AS_OK;
}

extern "C" int as_conv_gemm_f32(const ConvGemmArgs* args_host, as_stream_t stream_)
{
    return conv_gemm_one(args_host, nullptr, 0, nullptr, static_cast<hipStream_t>(stream_));
}

// ----------------------------------------------------------------------------------------------------------------
// Several independent convolutions as ONE launch (include/artspeech_hip.h: as_conv_gemm_multi_f32).
// ----------------------------------------------------------------------------------------------------------------
static bool multi_tall(int M) { return M > 64 && (M % 128 == 0 || M % 128 > 64 || M >= 512); }

// one tile for the whole set: the cost model of gemm_tile_choice on the SUM of the problems' tiles
static int multi_tile_choice(const ConvGemmArgs* a, int n)
{
    const char* env = getenv("AS_GEMM_TILE");           // tuning/experiments only
    if (env && atoi(env) > 0) return atoi(env);
    // 128-row tiles when the problems that can fill them carry (nearly) all of the set's work: a 64-channel conv that rides along -- the
    // decoder's asr_res beside a predictor block -- leaves the lower half of its few tiles empty, which costs less than a launch of its own
    bool all_short = true, small_k = false, all_32 = true;
    double work = 0, tall_work = 0;
    for (int i = 0; i < n; ++i) {
        const double wi = (double)a[i].M * a[i].N * ((double)a[i].K * a[i].T + a[i].K2);
        work += wi;
        if (multi_tall(a[i].M)) tall_work += wi;
        all_short = all_short && a[i].M <= 64;
        all_32 = all_32 && a[i].M <= 32;
        small_k = small_k || a[i].Kp <= 32;
    }
    const bool all_tall = tall_work >= 0.9 * work;
    if (all_32 && a[0].n_prod == 3) return 2;             // (as the single launch: a 64-row tile would be half empty)
    static const int choices[5] = {22, 21, 12, 11, 14};
    static const double t1[5] = {1.0, 0.78, 0.78, 0.59, 1.3};   // (64 x 256 from M64 N509440 K64 T9: 139 us against 167 with 64 x 128)
    int best = 11;
    double best_cost = 1e30;
    for (int c = 0; c < 5; ++c) {
        int bm, bn;
        tile_dims(choices[c], &bm, &bn);
        if (bm >= 128 && !all_tall) continue;
        if (choices[c] == 14 && (!all_short || a[0].n_prod != 3)) continue;
        double tiles = 0;
        for (int i = 0; i < n; ++i)
            tiles += (double)as_cdiv(a[i].M, bm) * (a[i].n_groups > 1 ? a[i].n_groups * as_cdiv(a[i].group_cols, bn) : as_cdiv(a[i].N, bn));
        if (choices[c] == 14 && tiles < 240) continue;
        const double r = ceil(tiles / 256.0);
        const double cost = t1[c] * (r > 1.0 ? 0.71 * r : 1.0);
        if (cost < best_cost * 0.97) { best_cost = cost; best = choices[c]; }
    }
    if (small_k && best == 11) best = 12;               // (see gemm_tile_choice: images written by the 32-row tile)
    return best;
}

extern "C" int as_conv_gemm_multi_tile(const ConvGemmArgs* list_host, int n)
{
    if (!list_host || n < 1 || n > H3_MAXP) return AS_EINVAL;
    ConvGemmArgs norm[H3_MAXP];
    for (int i = 0; i < n; ++i) {
        norm[i] = list_host[i];
        if (norm[i].n_prod == 0) norm[i].n_prod = 3;
        if (norm[i].n_groups < 1) norm[i].n_groups = 1;
    }
    return multi_tile_choice(norm, n);
}

static int conv_gemm_multi(const ConvGemmArgs* list_host, const AsAdainArgs* post_host, const int32_t* post_max_w, const AsLnArgs* ln_host, int n,
                           hipStream_t stream)
{
    if (!list_host || n < 1 || n > AS_MAX_MULTI || (post_host && !post_max_w)) return AS_EINVAL;
    static_assert(AS_MAX_MULTI == H3_MAXP, "header and kernel disagree");
    if (n == 1) return conv_gemm_one(list_host, post_host, post_host ? post_max_w[0] : 0, ln_host, stream);
    ConvGemmArgs norm[H3_MAXP];
    AsAdainArgs post[H3_MAXP];
    AsLnArgs lnp[H3_MAXP];
    bool has_ln[H3_MAXP];
    bool has_post[H3_MAXP], post_done[H3_MAXP];
    int pmw[H3_MAXP], src[H3_MAXP];
    int m = 0, n_direct = 0;
    for (int i = 0; i < n; ++i) {
        const int r = conv_gemm_normalise(&list_host[i], norm[m]);
        if (r != AS_OK) return r;
        if (norm[m].N == 0) continue;                                    // nothing to do for this one
        n_direct += direct_cin1(norm[m]) && norm[m].T <= 9 ? 1 : 0;
        has_post[m] = post_host && post_wanted(&post_host[i]);
        has_ln[m] = ln_host && ln_wanted(&ln_host[i]);
        if (has_post[m] && has_ln[m]) return AS_EINVAL;
        if (has_ln[m]) {
            lnp[m] = ln_host[i];
            const int rl = ln_check(norm[m], lnp[m]);
            if (rl != AS_OK) return rl;
        }
        post_done[m] = false;
        pmw[m] = has_post[m] ? post_max_w[i] : 0;
        src[m] = i;
        if (has_post[m]) {
            const int rp = post_normalise(norm[m], post_host[i], post[m]);
            if (rp != AS_OK) return rp;
        }
        ++m;
    }
    if (m == 0) return AS_OK;
    if (m == 1) return conv_gemm_one(&list_host[src[0]], post_host ? &post_host[src[0]] : nullptr, pmw[0], ln_host ? &ln_host[src[0]] : nullptr, stream);
    bool any_post = false;
    for (int i = 0; i < m; ++i) any_post = any_post || has_post[i] || has_ln[i];
    auto finish_posts = [&]() -> int {                                   // the AdaINs / LayerNorms no reduction kernel took along: one launch each, in list order
        for (int i = 0; i < m; ++i) {
            if (has_post[i] && !post_done[i]) {
                const int r = as_adain_image_f32(&post[i], stream);
                if (r != AS_OK) return r;
            }
            if (has_ln[i] && !post_done[i]) {
                const ConvGemmArgs& a = norm[i];
                const int r = as_channel_layernorm_split_f32(a.Y, a.ldy, a.M, a.N, lnp[i].gamma, lnp[i].beta, lnp[i].gamma2, lnp[i].beta2,
                                                             lnp[i].n_split, lnp[i].eps, lnp[i].relu, lnp[i].yh, stream);
                if (r != AS_OK) return r;
            }
        }
        return AS_OK;
    };
    if (n_direct == m) {                                                 // a set of Cin = 1 stems: the direct kernel, one launch
        static_assert(DIRECT_MAXP == H3_MAXP, "one list length");
        double fl = 0, by = 0;
        for (int i = 0; i < m; ++i) {
            fl += 2.0 * norm[i].M * norm[i].N * (double)norm[i].T;
            by += 4.0 * ((double)norm[i].T * norm[i].M + norm[i].N + (double)norm[i].M * norm[i].N);
        }
        char tag[64];
        snprintf(tag, sizeof(tag), "multi%d direct: M%d N%d K1 T%d | ...", m, norm[0].M, norm[0].N, norm[0].T);
        int rd;
        {
            AsProfScope prof__(AS_CLS_GEMM, fl, by, stream, tag);
            rd = launch_direct_x4(norm, m, stream);
        }
        return rd != AS_OK ? rd : finish_posts();
    }
    // otherwise, in one launch only what the tiled kernel runs from operand images with the same arithmetic
    for (int i = 0; i < m; ++i)
        if (direct_cin1(norm[i]) || !norm[i].Wh || !norm[i].Xh || norm[i].n_prod != norm[0].n_prod || norm[i].ileave_u > 1) return AS_EINVAL;
    const int choice = multi_tile_choice(norm, m);
    int bm, bn;
    tile_dims(choice, &bm, &bn);
    // K slices: a launch costs what its longest chain of k iterations on one CU costs.  With W = all problems' tiles x iterations spread
    // over the chip's 512 workgroup slots, a problem whose tile alone runs longer than that share (the towers' closing 5 x 5 convs: 800
    // k-blocks on a dozen tiles, beside convs of 30-300) is cut into slices of about that length -- as many as its workspace holds slabs
    // for, each keeping >= 384 k.  (That is also the rule of the single launch for grids that leave most of the chip idle.)
    const int wk = choice == 2 ? 2 : bm * bn >= 4 * 64 * 64 ? 1 : 4 * 64 * 64 / (bm * bn);
    int order[H3_MAXP], S[H3_MAXP], nkt[H3_MAXP];
    double len[H3_MAXP], W = 0;
    for (int i = 0; i < m; ++i) {
        const ConvGemmArgs& a = norm[i];
        nkt[i] = a.T * as_cdiv(a.Kp / 16, wk) + as_cdiv(as_cdiv(a.K2, 16), wk);
        W += (double)nkt[i] * as_cdiv(a.M, bm) * (a.n_groups > 1 ? a.n_groups * as_cdiv(a.group_cols, bn) : as_cdiv(a.N, bn));
    }
    const double share = std::max(W / 512.0, 1.0);
    for (int i = 0; i < m; ++i) {
        const ConvGemmArgs& a = norm[i];
        const int min_kt = std::max(1, 24 / wk);
        static const double thr = getenv("AS_MULTI_SLICE_THR") ? atof(getenv("AS_MULTI_SLICE_THR")) : 1.5;   // (tuning)
        int s = nkt[i] > thr * share ? (int)ceil(nkt[i] / share) : 1;
        s = std::min(std::min(s, 16), nkt[i] / min_kt);
        const size_t slab = (size_t)a.M * a.N * sizeof(float);
        if (s > 1 && (!a.ws || a.ws_bytes / slab < (size_t)s)) s = a.ws ? (int)std::min<size_t>(a.ws_bytes / slab, (size_t)s) : 1;
        S[i] = s < 1 ? 1 : s;
        // (post_slice_ok: a tiny conv with a normalisation behind it is always cut in two -- its reduction replaces the normalisation's launch)
        if (S[i] == 1 && post_slice_ok(a) && a.ws && a.ws_bytes >= 2 * slab && nkt[i] >= 2 &&
            ((has_ln[i] && a.M % 8 == 0 && a.M <= 512 && !a.Yh && a.act <= 2) || (has_post[i] && pmw[i] > 0 && pmw[i] <= 256 && a.act <= 2)))
            S[i] = 2;
        len[i] = (double)nkt[i] / S[i];
        order[i] = i;
    }
    std::sort(order, order + m, [&](int x, int y) { return len[x] > len[y]; });
    const ConvGemmArgs* ptr[H3_MAXP];
    int So[H3_MAXP];
    double flops = 0, bytes = 0;
    char tag[160];
    int at = snprintf(tag, sizeof(tag), "multi%d tile%d:", m, choice);
    for (int k = 0; k < m; ++k) {
        ptr[k] = &norm[order[k]];
        So[k] = S[order[k]];
        flops += gemm_flops(*ptr[k]);
        bytes += gemm_bytes(*ptr[k]);
        char shape[64], sl[8] = "";
        gemm_tag(*ptr[k], shape, sizeof(shape));
        if (So[k] > 1) snprintf(sl, sizeof(sl), " S%d", So[k]);
        if (at < (int)sizeof(tag) - 1) at += snprintf(tag + at, sizeof(tag) - at, " %s%s%s", shape, sl, k + 1 < m ? " |" : "");
    }
    // K-sliced problems with a LayerNorm behind them store their slabs time-major (the reduction normalises whole columns)
    const AsLnArgs* nl[H3_MAXP];
    bool any_ln_fused = false;
    for (int k = 0; k < m; ++k) {
        const int i = order[k];
        nl[k] = (has_ln[i] && So[k] > 1 && ln_fusable(norm[i], lnp[i])) ? &lnp[i] : nullptr;
        norm[i].slab_tr = nl[k] ? 1 : 0;
        any_ln_fused = any_ln_fused || nl[k];
    }
    {
        AsProfScope prof__(AS_CLS_GEMM, flops, bytes, stream, tag);
        const int rc = as_conv_gemm_h3_launch(ptr, So, m, choice, stream);
        if (rc != AS_OK) return rc;
        // K-sliced problems whose result is read through an AdaIN (few columns: the reduction holds a channel's whole time axis): the
        // reduction launch writes the AdaIN image too
        bool fused = any_ln_fused;
        const AsAdainArgs* np[H3_MAXP];
        for (int k = 0; k < m; ++k) {
            const int i = order[k];
            np[k] = (any_post && has_post[i] && So[k] > 1 && post_fusable(*ptr[k], pmw[i])) ? &post[i] : nullptr;
            fused = fused || np[k];
        }
        if (fused) {
            int mwo[H3_MAXP];
            for (int k = 0; k < m; ++k) mwo[k] = pmw[order[k]];
            const int rr = launch_reduce_post(ptr, So, np, mwo, nl, m, stream);
            if (rr != AS_OK) return rr;
            for (int k = 0; k < m; ++k)
                if (np[k] || nl[k]) post_done[order[k]] = true;
        }
        ReduceMulti rm;
        memset(&rm, 0, sizeof(rm));
        int rows_max = 0;
        for (int k = 0; k < m && !fused; ++k)
            if (So[k] > 1) {
                const ConvGemmArgs& a = *ptr[k];
                rm.a[rm.n] = a;
                rm.S[rm.n] = So[k];
                rm.blk0[rm.n + 1] = rm.blk0[rm.n] + as_cdiv(a.N + 1, 256);
                rows_max = std::max(rows_max, a.Yh ? std::max(16 * as_kbx(a.M), a.M) : a.M);
                ++rm.n;
            }
        if (rm.n == 1) {
            const int rr = launch_reduce(rm.a[0], rm.S[0], stream);
            if (rr != AS_OK) return rr;
        } else if (rm.n > 1) {
            hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(rm.blk0[rm.n], as_cdiv(rows_max, 8)), dim3(256), 0, stream, rm);
            AS_CHECK_LAUNCH();
        }
    }
    return finish_posts();
}

extern "C" int as_conv_gemm_multi_f32(const ConvGemmArgs* list_host, int n, as_stream_t stream_)
{
    return conv_gemm_multi(list_host, nullptr, nullptr, nullptr, n, static_cast<hipStream_t>(stream_));
}

extern "C" int as_conv_gemm_multi_post_f32(const ConvGemmArgs* list_host, const AsAdainArgs* post_host, const int32_t* post_max_w,
                                           const AsLnArgs* post_ln_host, int n, as_stream_t stream_)
{
    return conv_gemm_multi(list_host, post_host, post_max_w, post_ln_host, n, static_cast<hipStream_t>(stream_));
}
