// K3/K4/K8/K10 -- every dense convolution / linear layer of the path as ONE implicit-GEMM kernel on
// the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain, 157 TFLOP/s peak).
//
//   Y[m][j] = epilogue( sum_t sum_k  Wt[t][k][m] * X[k][ j + dh[t]*Wj + dw[t] ] )      m < M, j < N
//
// Activations use the "packed frames" layout: a tensor is [C][N] with all utterances of
// the batch concatenated along the contiguous column axis and NO padding; column j carries a 64-bit
// descriptor (h, w, H, W) of its position inside its own utterance, so a tap (dh, dw) is valid iff
// 0 <= h+dh < H and 0 <= w+dw < W -- that is the conv's zero padding, and it is also what keeps
// utterances from seeing each other.  1-D convs are the H = 1 case.  Replaces
//   nn.Conv1d k in {1,3,5,9}  (RelTransformerEnc.py:110-118,257-258,306-314; models.py:176-181,480-495,592-594)
//   nn.Conv2d 3x3 / 1x1       (models.py:71-77, 385-399, 530-535)
//   nn.Linear on [*, C] rows  (the LSTM input projections, hoisted out of the recurrence)
// Weights are pre-transposed at load time to [tap][Kp][Cout] (Kp = Cin rounded up to 16, zero rows) so both
// operands stage as coalesced rows.  Tile: (64*TM) x (64*TN) x 16 per 256-thread workgroup, 2x2 waves,
// TM x TN MFMA tiles of 32x32 per wave; double-buffered LDS, one barrier per k-tile; the fragments of k-step
// s+1 are requested before the MFMAs of step s issue.
//
// Staging uses raw BUFFER loads: the hardware range check returns 0 for any offset >= num_records, so
// "tap outside the utterance", "row >= Cin", "column >= Cout" become a select of an out-of-range offset --
// no divergent branches, no zero fill in the loop.  Two instantiations of the staging:
//   QUAD   (args.quad_ok): a thread stages 16-byte quads -- 4 output channels / 4 consecutive frame columns of
//          one k row -- with ONE buffer_load_dwordx4 whose offset is a per-thread VGPR (fixed per tap) plus a
//          wave-uniform SGPR, and one ds_write_b128.  Needs M % 4 == 0, 16 readable bytes in front of X and no
//          quad straddling images of different width (the caller knows its layout; ops.py decides).
//   SCALAR (always correct): one dword per load, validity per column.
#include "common.h"
#include "conv_gemm.h"
#include <cstdio>

#define BK 16

// one k-tile of MFMAs from LDS buffer `buf`
// `mid` runs once after the first k-step's MFMAs have been issued: the staging work of the NEXT tiles (LDS store,
// global loads) is issued in the shadow of those MFMAs instead of at the end of the k-tile.
template <int TM, int TN, int BM, int BN, typename Mid>
static __device__ __forceinline__ void mma_tile(const float (*As)[BK][BM], const float (*Bs)[BK][BN], int buf, int wm,
                                                int wn, int l31, int lk, f32x16 (&acc)[TM][TN], Mid&& mid)
{
    float af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = As[buf][lk][wm * 32 * TM + i * 32 + l31];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) bf[0][jn] = Bs[buf][lk][wn * 32 * TN + jn * 32 + l31];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
        if (s + 1 < BK / 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[(s + 1) & 1][i] = As[buf][2 * (s + 1) + lk][wm * 32 * TM + i * 32 + l31];
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) bf[(s + 1) & 1][jn] = Bs[buf][2 * (s + 1) + lk][wn * 32 * TN + jn * 32 + l31];
        }
        // sched_barrier pins "request next fragments, then MFMA": hipcc otherwise sinks each ds_read next to its use
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][i], bf[s & 1][jn], acc[i][jn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0) {
            mid();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ======================================================================================================
// QUAD staging
// ======================================================================================================
template <int TM, int TN>
__global__ void __launch_bounds__(256)
conv_gemm_quad_kernel(const ConvGemmArgs a)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int AQ = BK * BM / 4 / 256;     // quads per thread per k-tile
    constexpr int BQ = BK * BN / 4 / 256;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, lk = lane >> 5;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tile = logical_tile();
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((a.n_split > 0 && n0 >= a.n_split) ? a.W2 : a.W), 0, (int)((unsigned)a.T * a.Kp * a.M * 4u), 0x00020000);
    // X descriptor starts 16 bytes in front of X (quad_ok promises they are readable): a quad whose first column
    // is masked may start at column -1..-4 of row 0 without its offset wrapping
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X) - 4, 0, (int)((unsigned)a.K * a.ldx * 4u + 16u), 0x00020000);

    // thread -> quads q = tid + i*256: row = q / (B?/4), first column = (q % (B?/4)) * 4
    unsigned a_voff[AQ];
#pragma unroll
    for (int i = 0; i < AQ; ++i) {
        const int q = tid + i * 256, row = q / (BM / 4), col = (q % (BM / 4)) * 4;
        a_voff[i] = (m0 + col) < a.M ? (unsigned)((row * a.M + m0 + col) * 4) : OOB;
    }
    unsigned b_mask[BQ][4];                                // per column: bit t = tap t reads inside the utterance
    int b_W[BQ];
#pragma unroll
    for (int i = 0; i < BQ; ++i) {
        const int q = tid + i * 256, col = (q % (BN / 4)) * 4;
        b_W[i] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = n0 + col + c;
            unsigned m = 0;
            if (j < a.N) {
                if (a.meta) {
                    const unsigned long long md = a.meta[j];
                    const int h = (int)(md & 0xffff), w = (int)((md >> 16) & 0xffff);
                    const int H = (int)((md >> 32) & 0xffff), Wj = (int)(md >> 48);
                    for (int t = 0; t < a.T; ++t)
                        if ((unsigned)(h + a.dh[t]) < (unsigned)H && (unsigned)(w + a.dw[t]) < (unsigned)Wj) m |= 1u << t;
                    if (c == 0) b_W[i] = Wj;
                } else {
                    m = 0xffffffffu;
                }
            }
            b_mask[i][c] = m;
        }
    }

    const int kt_per_tap = a.Kp / BK;
    const int nkt_all = a.T * kt_per_tap;
    const int S = gridDim.y;                               // split-K: blockIdx.y owns a slice of the (tap, k-tile) sequence
    const int kt_lo = (int)((long)nkt_all * blockIdx.y / S);
    const int nkt = (int)((long)nkt_all * (blockIdx.y + 1) / S);
    const bool k_ragged = (a.K % BK) != 0;

    // next tile to load: (tap, k0) advance without a division; per tap: quad offsets + validity bits
    int ld_t = kt_lo / kt_per_tap, ld_k0 = (kt_lo - ld_t * kt_per_tap) * BK;
    unsigned b_voff[BQ], b_ok[BQ];
    auto tap_setup = [&]() {
        const int dh = a.dh[ld_t], dw = a.dw[ld_t];
#pragma unroll
        for (int i = 0; i < BQ; ++i) {
            const int q = tid + i * 256, row = q / (BN / 4), col = (q % (BN / 4)) * 4;
            unsigned ok = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) ok |= ((b_mask[i][c] >> ld_t) & 1u) << c;
            b_ok[i] = ok;
            b_voff[i] = ok ? (unsigned)((row * a.ldx + n0 + col + dh * b_W[i] + dw) * 4 + 16) : OOB;
        }
    };
    tap_setup();

    f32x4 ra[AQ], rb[BQ];
    unsigned rok[BQ];
    auto gload = [&]() {
        const int a_soff = (ld_t * a.Kp + ld_k0) * a.M * 4;
        const int b_soff = ld_k0 * a.ldx * 4;
#pragma unroll
        for (int i = 0; i < AQ; ++i) ra[i] = buf_load4(rsW, a_voff[i], a_soff);
        const bool ragged_tile = k_ragged && ld_k0 + BK > a.K;   // last k-tile of a tap with Cin % 16 != 0
#pragma unroll
        for (int i = 0; i < BQ; ++i) {
            const int row = (tid + i * 256) / (BN / 4);
            const bool row_ok = !ragged_tile || (ld_k0 + row) < a.K;
            rb[i] = buf_load4(rsX, row_ok ? b_voff[i] : OOB, b_soff);
            rok[i] = b_ok[i];
        }
        ld_k0 += BK;
        if (ld_k0 >= a.Kp) { ld_k0 = 0; ld_t += 1; if (ld_t < a.T) tap_setup(); }
    };
    const bool in_lrelu = a.in_act == 2;
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AQ; ++i) {
            const int q = tid + i * 256;
            *reinterpret_cast<f32x4*>(&As[buf][q / (BM / 4)][(q % (BM / 4)) * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BQ; ++i) {
            const int q = tid + i * 256;
            f32x4 v = rb[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = ((rok[i] >> c) & 1u) ? v[c] : 0.f;
                if (in_lrelu) x = x > 0.f ? x : a.in_slope * x;  // LeakyReLU fused on the operand (models.py:89,142)
                v[c] = x;
            }
            *reinterpret_cast<f32x4*>(&Bs[buf][q / (BN / 4)][(q % (BN / 4)) * 4]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // tile kt_lo -> LDS, tile kt_lo+1 -> registers.  In iteration kt the registers (tile kt+1, loaded one whole
    // iteration earlier) are written to the other LDS buffer and refilled with tile kt+2 right behind the first
    // MFMAs, so the only thing left at the end of a k-tile is the barrier.
    gload();
    lstore(kt_lo & 1);
    if (kt_lo + 1 < nkt) gload();
    __syncthreads();
    for (int kt = kt_lo; kt < nkt; ++kt) {
        const int buf = kt & 1;
        mma_tile<TM, TN, BM, BN>(As, Bs, buf, wm, wn, l31, lk, acc, [&]() {
            if (kt + 1 < nkt) lstore(buf ^ 1);
            if (kt + 2 < nkt) gload();
        });
        __syncthreads();
    }
    epilogue<TM, TN>(a, acc, m0, n0, wm, wn, l31, lk, S);
}

// ======================================================================================================
// SCALAR staging (general)
// ======================================================================================================
template <int TM, int TN>
__global__ void __launch_bounds__(256)
conv_gemm_kernel(const ConvGemmArgs a)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int AROWS = BK * BM / 256;      // elements per thread per k-tile (A)
    constexpr int BROWS = BK * BN / 256;
    constexpr int A_RSTEP = 256 / BM, B_RSTEP = 256 / BN;
    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, lk = lane >> 5;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tile = logical_tile();
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((a.n_split > 0 && n0 >= a.n_split) ? a.W2 : a.W), 0, (int)((unsigned)a.T * a.Kp * a.M * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((unsigned)a.K * a.ldx * 4u), 0x00020000);

    // thread -> one column (A: output channel, B: frame column), BK/stride rows
    const int a_c = tid % BM, a_r0 = tid / BM;
    const int b_c = tid % BN, b_r0 = tid / BN;
    const unsigned a_voff = (m0 + a_c) < a.M ? (unsigned)((a_r0 * a.M + m0 + a_c) * 4) : OOB;
    const int j = n0 + b_c;
    unsigned tapmask = 0;
    int Wj = 0;
    if (j < a.N) {
        if (a.meta) {
            const unsigned long long md = a.meta[j];
            const int h = (int)(md & 0xffff), w = (int)((md >> 16) & 0xffff);
            const int H = (int)((md >> 32) & 0xffff);
            Wj = (int)(md >> 48);
            for (int t = 0; t < a.T; ++t)
                if ((unsigned)(h + a.dh[t]) < (unsigned)H && (unsigned)(w + a.dw[t]) < (unsigned)Wj) tapmask |= 1u << t;
        } else {
            tapmask = 0xffffffffu;
        }
    }

    const int kt_per_tap = a.Kp / BK;
    const int nkt_all = a.T * kt_per_tap;
    const int S = gridDim.y;
    const int kt_lo = (int)((long)nkt_all * blockIdx.y / S);
    const int nkt = (int)((long)nkt_all * (blockIdx.y + 1) / S);
    const bool k_ragged = (a.K % BK) != 0;
    const int a_step = A_RSTEP * a.M * 4, b_step = B_RSTEP * a.ldx * 4;

    int ld_t = kt_lo / kt_per_tap, ld_k0 = (kt_lo - ld_t * kt_per_tap) * BK;
    unsigned b_voff = OOB;
    auto tap_setup = [&]() {
        const bool ok = (tapmask >> ld_t) & 1u;
        const int src = j + a.dh[ld_t] * Wj + a.dw[ld_t];
        b_voff = (ok && (b_r0 * a.ldx + src) >= 0) ? (unsigned)(b_r0 * a.ldx + src) * 4u : OOB;
    };
    tap_setup();
    float ra[AROWS], rb[BROWS];
    auto gload = [&]() {
        const int a_soff = (ld_t * a.Kp + ld_k0) * a.M * 4;
        const int b_soff = ld_k0 * a.ldx * 4;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) ra[i] = buf_load1(rsW, a_voff, a_soff + i * a_step);
        if (k_ragged && ld_k0 + BK > a.K) {                 // last k-tile of a tap with Cin % 16 != 0
#pragma unroll
            for (int i = 0; i < BROWS; ++i)
                rb[i] = buf_load1(rsX, (ld_k0 + b_r0 + i * B_RSTEP) < a.K ? b_voff : OOB, b_soff + i * b_step);
        } else {
#pragma unroll
            for (int i = 0; i < BROWS; ++i) rb[i] = buf_load1(rsX, b_voff, b_soff + i * b_step);
        }
        ld_k0 += BK;
        if (ld_k0 >= a.Kp) { ld_k0 = 0; ld_t += 1; if (ld_t < a.T) tap_setup(); }
    };
    const bool in_lrelu = a.in_act == 2;
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AROWS; ++i) As[buf][a_r0 + i * A_RSTEP][a_c] = ra[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            float v = rb[i];
            if (in_lrelu) v = v > 0.f ? v : a.in_slope * v;
            Bs[buf][b_r0 + i * B_RSTEP][b_c] = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // tile kt_lo -> LDS, tile kt_lo+1 -> registers.  In iteration kt the registers (tile kt+1, loaded one whole
    // iteration earlier) are written to the other LDS buffer and refilled with tile kt+2 right behind the first
    // MFMAs, so the only thing left at the end of a k-tile is the barrier.
    gload();
    lstore(kt_lo & 1);
    if (kt_lo + 1 < nkt) gload();
    __syncthreads();
    for (int kt = kt_lo; kt < nkt; ++kt) {
        const int buf = kt & 1;
        mma_tile<TM, TN, BM, BN>(As, Bs, buf, wm, wn, l31, lk, acc, [&]() {
            if (kt + 1 < nkt) lstore(buf ^ 1);
            if (kt + 2 < nkt) gload();
        });
        __syncthreads();
    }
    epilogue<TM, TN>(a, acc, m0, n0, wm, wn, l31, lk, S);
}

// Cin = 1 (the first conv of every style tower and of the decoder's F0 / energy inputs): 9 MACs per output on a
// 16-deep matrix-core k-block would be 2 % useful work, and the op is bound by writing Y anyway (M x N x 4 bytes: 130 MB
// for the mel tower) -- so a direct kernel: a thread owns one column, keeps its T input taps in registers and streams the
// M output channels (stores coalesced across the wave), weights and bias from LDS.
#define DIRECT_MAX_M 128
#define DIRECT_TP 12                                       // weights per channel in LDS: T (<= 9 on the fast path) padded to 3 x 16 bytes
// fast form (T <= 9): a thread owns FOUR consecutive columns (one 16-byte store per channel when Y allows), three
// 16-byte LDS reads feed 4 x T FMAs
template <bool VEC>
__global__ void __launch_bounds__(256)
conv_direct_cin1_x4_kernel(const ConvGemmArgs a)
{
    __shared__ __attribute__((aligned(16))) float ws[DIRECT_MAX_M * DIRECT_TP];   // [m][t]
    __shared__ float bs[DIRECT_MAX_M];
    for (int i = threadIdx.x; i < a.M * DIRECT_TP; i += 256) {
        const int m = i / DIRECT_TP, t = i % DIRECT_TP;
        ws[i] = t < a.T ? a.W[(size_t)t * a.Kp * a.M + m] : 0.f;      // k = 0 rows of the [T][Kp][M] image
    }
    for (int i = threadIdx.x; i < a.M; i += 256) bs[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j0 >= a.N) return;
    float x[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = j0 + c;
        int h = 0, w = 0, H = 1, Wj = 0x7fffffff;
        if (a.meta && j < a.N) {
            const unsigned long long md = a.meta[j];
            h = (int)(md & 0xffff), w = (int)((md >> 16) & 0xffff), H = (int)((md >> 32) & 0xffff), Wj = (int)(md >> 48);
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
    for (int m = 0; m < a.M; ++m) {
        float wt[DIRECT_TP];
#pragma unroll
        for (int q = 0; q < DIRECT_TP / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&ws[m * DIRECT_TP + 4 * q]);
            wt[4 * q] = v[0]; wt[4 * q + 1] = v[1]; wt[4 * q + 2] = v[2]; wt[4 * q + 3] = v[3];
        }
        float y[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) sum += wt[t] * x[c][t];
            sum += bs[m];
            if (a.act == 1) sum = sum > 0.f ? sum : 0.f;
            else if (a.act == 2) sum = sum > 0.f ? sum : a.act_slope * sum;
            else if (a.act == 3) sum = tanhf(sum);
            else if (a.act == 4) sum = fabsf(sum);
            else if (a.act == 5) sum = sum / (1.0f + expf(-sum));
            y[c] = sum;
        }
        float* yr = a.Y + (size_t)m * a.ldy + j0;
        if (VEC) *reinterpret_cast<f32x4*>(yr) = f32x4{y[0], y[1], y[2], y[3]};
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (j0 + c < a.N) yr[c] = y[c];
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
        h = (int)(md & 0xffff), w = (int)((md >> 16) & 0xffff), H = (int)((md >> 32) & 0xffff), Wj = (int)(md >> 48);
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
        if (a.act == 1) s = s > 0.f ? s : 0.f;
        else if (a.act == 2) s = s > 0.f ? s : a.act_slope * s;
        else if (a.act == 3) s = tanhf(s);
        else if (a.act == 4) s = fabsf(s);
        else if (a.act == 5) s = s / (1.0f + expf(-s));
        a.Y[(size_t)m * a.ldy + j] = s;
    }
}

// y = epi(sum_s slab[s]) in a fixed order (deterministic, unlike float atomics)
__global__ void splitk_reduce_kernel(const ConvGemmArgs a, int S)
{
    const long total = (long)a.M * a.N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / a.N), col = (int)(i - (long)row * a.N);
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += a.ws[(size_t)s * total + i];
        if (a.bias) v += ((a.n_split > 0 && col >= a.n_split) ? a.bias2 : a.bias)[row];
        if (a.res) v += a.res[(size_t)row * a.ldr + col];
        if (a.div_sqrt2) v = v / 1.41421356237309504880f;
        if (a.act == 1) v = v > 0.f ? v : 0.f;
        else if (a.act == 2) v = v > 0.f ? v : a.act_slope * v;
        else if (a.act == 3) v = tanhf(v);
        else if (a.act == 4) v = fabsf(v);
        else if (a.act == 5) v = v / (1.0f + expf(-v));
        if (a.transpose_out) a.Y[(size_t)col * a.ldy + row] = v;
        else a.Y[(size_t)row * a.ldy + col] = v;
    }
}

static void tile_dims(int choice, int* bm, int* bn)
{
    // 22 / 21 / 12 / 11: 128x128, 128x64, 64x128, 64x64 with 4 waves; 228 / 218 / 128 (bf16x6 only): the same tiles with
    // 8 waves (two per SIMD), the extra four splitting K
    // 223 (bf16x6 only): 128x128, 4 waves, the three taps of a group share one staged activation tile (conv_gemm_x6t.hip)
    *bm = (choice == 22 || choice == 21 || choice == 228 || choice == 218 || choice == 223) ? 128 : 64;
    *bn = (choice == 22 || choice == 12 || choice == 228 || choice == 128 || choice == 223) ? 128 : 64;
}

// which arithmetic: bf16x6 when the caller supplied the split weights (AS_GEMM_IMPL=f32 forces the fp32 MFMAs when
// the fp32 image is there too -- experiments only)
static bool use_x6(const ConvGemmArgs& a)
{
    if (!a.Wx) return false;
    // tap offsets the bf16x6 kernel can pack into a byte: |dh|, |dw| <= 7, or dh = 0 everywhere and |dw| <= 127 (dilated 1-D)
    bool small = true, flat = true;
    for (int t = 0; t < a.T; ++t) {
        small &= a.dh[t] >= -7 && a.dh[t] <= 7 && a.dw[t] >= -7 && a.dw[t] <= 7;
        flat &= a.dh[t] == 0 && a.dw[t] >= -127 && a.dw[t] <= 127;
    }
    if (!small && !flat) return a.W ? false : true;        // (no fp32 image to fall back to: the launch reports EINVAL)
    const char* env = getenv("AS_GEMM_IMPL");
    return !(env && env[0] == 'f' && a.W && a.X);
}

// Tile and split-K choice, from sweeps on MI355X (scripts/gemm_bench.py): the kernel wants >= ~1000 workgroups
// (4-5 per CU) so that tile quantisation over 256 CUs and the lock-step load/compute phases of co-resident
// workgroups average out; shapes with fewer tiles get 128x64 tiles and 2-4 K slices (16 for tiny outputs).
static int gemm_tile_choice(int M, int N, bool x6, bool taps3 = false)
{
    const char* env = getenv("AS_GEMM_TILE");           // tuning/experiments only: 22, 21, 12, 11
    if (env && atoi(env) > 0 && (atoi(env) != 223 || (x6 && taps3))) return atoi(env);
    if (env && atoi(env) == 223) env = nullptr;
    const bool tall = M > 64 && (M % 128 == 0 || M % 128 > 64 || M >= 512);   // a 128-row tile is not half empty
    if (x6) {
        // cost model fitted to sweeps on MI355X (scripts/gemm_bench.py): a CU runs its L = ceil(tiles / 256) tiles
        // together (they overlap each other's waits: x0.85 when L >= 2); relative speed per tile shape from the sweep
        // (the 64x64 tile stages twice the bytes per flop and is never the best)
        static const int choices[3] = {22, 21, 12};
        static const double eff[3] = {1.0, 0.95, 0.85};
        int best = 12;
        double best_cost = 1e30;
        for (int c = 0; c < 3; ++c) {
            int bm, bn;
            tile_dims(choices[c], &bm, &bn);
            if (bm == 128 && !tall) continue;
            const long tiles = (long)as_cdiv(M, bm) * as_cdiv(N, bn);
            const long L = (tiles + 255) / 256;
            const double cost = (double)L * bm * bn / eff[c] * (L >= 2 ? 0.85 : 1.0);
            if (cost < best_cost) { best_cost = cost; best = choices[c]; }
        }
        return best;
    }
    if (tall) {
        if ((long)as_cdiv(M, 128) * as_cdiv(N, 128) >= 1000) return 22;
        if ((long)as_cdiv(M, 128) * as_cdiv(N, 64) >= 300) return 21;
        return 11;
    }
    if ((long)as_cdiv(M, 64) * as_cdiv(N, 128) >= 1000) return 12;
    return 11;
}

static int gemm_ksplit(int M, int N, int Kp, int T, int choice, bool x6)
{
    const char* env = getenv("AS_GEMM_KSPLIT");          // tuning/experiments only
    int bm, bn;
    tile_dims(choice, &bm, &bn);
    const long tiles = (long)as_cdiv(M, bm) * as_cdiv(N, bn);
    int nkt = T * (Kp / BK);
    const int waves = (choice > 100 && choice != 223) ? 8 : 4;
    if (x6) nkt = T * as_cdiv(Kp / 16, waves * 64 * 64 / (bm * bn));   // k-tile = 16 * WK, WK = waves / (tile / 64x64)
    if (choice == 223) nkt = (T / 3) * (Kp / 16);                       // one k-tile = a 16-deep block of three taps
    int s = 1;
    if (env && atoi(env) > 0) s = atoi(env);
    else if (x6) {
        const double gflop = 2e-9 * M * N * (double)Kp * T;
        if (tiles < 256) {
            s = as_cdiv(512, tiles);
            const int cap = (long)M * N >= 262144 ? 4 : 16;
            if (s > cap) s = cap;
            // more than half the CUs busy already and a short GEMM: the reduce pass (~9 us + a launch) costs more than the
            // slices gain (M512 N2560 K1024: 28 us unsplit, 32 us in two slices)
            if (tiles >= 128 && gflop < 3.0) s = 1;
        } else if (tiles < 384 && gflop >= 12.0) {
            // 256-383 tiles leave the second round of workgroups mostly empty (two fit a CU: 512 slots); a long GEMM is
            // worth slicing for that alone (M1024 N2560 K512 T9, 320 tiles: 172 us unsplit, 145 us in four slices; with
            // 400 tiles and more the slices only add their overhead)
            s = 4;
        }
    } else if (tiles < 1000) {
        s = as_cdiv(1200, tiles);
        const int cap = (long)M * N >= 262144 ? 4 : 16;  // the reduce pass moves S*M*N*8 bytes
        if (s > cap) s = cap;
    }
    // k-tiles a slice keeps: a slice pays ~6 us of prologue + epilogue and the split a reduce launch (~8 us), against
    // ~0.65 us per 32-deep k-tile; swept on the whole step (bench.py): a slice of >= 384 k (12 k-tiles of 32) is best
    const int env_min = getenv("AS_GEMM_MINKT") ? atoi(getenv("AS_GEMM_MINKT")) : 0;   // tuning/experiments only
    const int min_kt = env_min > 0 ? env_min : choice == 223 ? 8 : x6 ? 12 * 32 / (16 * waves * 64 * 64 / (bm * bn)) : 8;
    if (s > nkt / min_kt) s = nkt / min_kt;
    return s < 1 ? 1 : s;
}

// What one call runs: arithmetic, tile, K slices, and whether the activations are split into bf16 parts ahead of the
// GEMM (conv_gemm_x6d.hip) instead of inside every tile's k loop.  The pre-split kernel is 5-25 % faster than the
// in-loop one (no split VALU, no activation registers, three LDS stages on the 128x128 tile), but the split pass costs
// 10 bytes of HBM traffic per input element plus a launch: with the whole step as the judge (bench.py, per-shape HIP
// events, scripts/exp/x6d_policy.py) it pays for itself only on the 64-row tiles with 9 taps (the 64-channel 3x3
// convolutions of the style towers: 413 -> 307 us); elsewhere it is break-even or worse.  So: those shapes, and any
// call whose caller supplies the image (ConvGemmArgs.Xs: one split shared by several convolutions, or written by the
// kernel that produces the activations).
struct GemmPlan {
    bool x6, x6d;
    int choice, S;
    size_t slab_bytes, xs_bytes;      // workspace: split-K slabs first, then (256-byte aligned) the split activations
};
static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
static GemmPlan gemm_plan(const ConvGemmArgs& a, bool have_ws_for_xs)
{
    GemmPlan p = {};
    p.x6 = use_x6(a);
    const bool taps3 = p.x6 && a.n_split == 0 && as_conv_gemm_x6t_ktiles(a) > 0;     // (the tap-shared kernel has no second weight set)
    if (p.x6) {
        const char* env = getenv("AS_GEMM_X6D");          // tuning/experiments only: 0 never, 1 whenever the kernel can
        const int mode = env ? atoi(env) : -1;
        int c = gemm_tile_choice(a.M, a.N, true, false);
        if (a.Xs && c != 22 && c != 21 && c != 12) c = a.M > 64 ? 21 : 12;   // (a forced experiment tile the kernel lacks)
        bool want = a.Xs != nullptr;
        if (!want && have_ws_for_xs && mode != 0) want = mode == 1 || (c == 12 && a.T >= 9 && a.N >= 4096);
        if (want && (c == 22 || c == 21 || c == 12)) {
            p.x6d = true;
            p.choice = c;
        }
    }
    if (!p.x6d) p.choice = gemm_tile_choice(a.M, a.N, p.x6, taps3);
    p.S = gemm_ksplit(a.M, a.N, a.Kp, a.T, p.choice, p.x6);
    p.slab_bytes = p.S > 1 ? (size_t)p.S * a.M * a.N * sizeof(float) : 0;
    p.xs_bytes = (p.x6d && !a.Xs) ? as_split_bf16x3_bytes(a.K, a.N) : 0;
    return p;
}

extern "C" size_t as_conv_gemm_workspace_bytes(const ConvGemmArgs* args_host)
{
    if (!args_host || args_host->M <= 0 || args_host->N <= 0 || args_host->Kp <= 0 || args_host->T <= 0) return 0;
    const GemmPlan p = gemm_plan(*args_host, true);
    return p.xs_bytes ? align256(p.slab_bytes) + p.xs_bytes : p.slab_bytes;
}

template <int TM, int TN>
static void launch(bool quad, dim3 grid, hipStream_t stream, const ConvGemmArgs& a)
{
    if (quad) hipLaunchKernelGGL((conv_gemm_quad_kernel<TM, TN>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<TM, TN>), grid, dim3(256), 0, stream, a);
}

extern "C" int as_conv_gemm_f32(const ConvGemmArgs* args_host, as_stream_t stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!args_host) return AS_EINVAL;
    ConvGemmArgs norm = *args_host;
    if (norm.in_slope == 0.f) norm.in_slope = 0.2f;
    if (norm.act_slope == 0.f) norm.act_slope = 0.2f;
    if (norm.act < 0 || norm.act > 5 || (norm.in_act != 0 && norm.in_act != 2)) return AS_EINVAL;
    const ConvGemmArgs& a = norm;
    if ((!a.W && !a.Wx) || (!a.X && !(a.Xs && a.Wx)) || !a.Y || a.M <= 0 || a.N < 0 || a.K <= 0 || a.T <= 0 || a.T > AS_MAX_TAPS) return AS_EINVAL;
    if (a.Kp < a.K || a.Kp % BK) return AS_EINVAL;
    if (a.n_split < 0 || (a.n_split > 0 && (a.n_split % 128 || (a.W && !a.W2) || (a.Wx && !a.Wx2) || (a.bias && !a.bias2) || a.K == 1))) return AS_EINVAL;
    if (a.ldx < a.N || a.ldy < (a.transpose_out ? a.M : a.N) || (a.res && (a.ldr < a.N || a.transpose_out))) return AS_EINVAL;
    // 32-bit byte offsets inside the buffer descriptors
    if ((double)a.T * (a.Kp + 48) * a.M * 6.0 >= 4294967296.0 || (double)a.K * a.ldx * 4.0 + 16.0 >= 4294967296.0) return AS_EINVAL;
    // the epilogue's buffer descriptors (output, residual, split-K slab) stay below 2 GiB
    if ((double)(a.transpose_out ? a.N : a.M) * a.ldy * 4.0 >= 2147483648.0 || (double)a.M * a.ldr * 4.0 >= 2147483648.0 ||
        (double)a.M * a.N * 4.0 >= 2147483648.0) return AS_EINVAL;
    if (a.N == 0) return AS_OK;
    if (a.K == 1 && a.W && a.X && !a.res && !a.div_sqrt2 && !a.transpose_out && a.M <= DIRECT_MAX_M && !getenv("AS_GEMM_NO_DIRECT")) {
        char tag[64];
        snprintf(tag, sizeof(tag), "M%d N%d K1 T%d direct", a.M, a.N, a.T);
        AsProfScope prof__(AS_CLS_GEMM, 2.0 * a.M * a.N * (double)a.T, 4.0 * ((double)a.T * a.M + a.N + (double)a.M * a.N), stream, tag);
        if (a.T <= 9) {
            const bool vec = (a.N & 3) == 0 && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0;
            if (vec) hipLaunchKernelGGL(conv_direct_cin1_x4_kernel<true>, dim3(as_cdiv(a.N, 1024)), dim3(256), 0, stream, a);
            else hipLaunchKernelGGL(conv_direct_cin1_x4_kernel<false>, dim3(as_cdiv(a.N, 1024)), dim3(256), 0, stream, a);
        } else {
            hipLaunchKernelGGL(conv_direct_cin1_kernel, dim3(as_cdiv(a.N, 256)), dim3(256), 0, stream, a);
        }
        AS_CHECK_LAUNCH();
        return AS_OK;
    }
    GemmPlan plan = gemm_plan(a, a.ws != nullptr);
    if (plan.xs_bytes && a.ws_bytes < align256(plan.slab_bytes) + plan.xs_bytes) plan = gemm_plan(a, false);   // no room: split in the k loop
    const bool x6 = plan.x6;
    const int choice = plan.choice;
    int S = plan.S;
    if (S > 1 && (!a.ws || a.ws_bytes < plan.slab_bytes)) S = 1;         // no workspace: no K slices
    const char* envq = getenv("AS_GEMM_QUAD");           // tuning/experiments only: 0 forces the scalar staging
    const bool quad = a.quad_ok && (a.M & 3) == 0 && !(envq && atoi(envq) == 0);
    char tag[64];
    snprintf(tag, sizeof(tag), "M%d N%d K%d T%d tile%d S%d %s", a.M, a.N, a.K, a.T, choice, S, plan.x6d ? "x6d" : x6 ? "x6" : quad ? "q" : "s");
    // algorithmic work of this launch: 2*M*N*K*T flop; bytes = weights + input + output once
    AsProfScope prof__(AS_CLS_GEMM, 2.0 * a.M * a.N * (double)a.K * a.T,
                       4.0 * ((double)a.T * a.K * a.M + (double)a.K * a.N + (double)a.M * a.N), stream, tag);
    if (plan.x6d) {
        if (!a.Xs) {                                                    // split once, behind the K slabs in the workspace
            uint16_t* xs = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(a.ws) + align256(plan.slab_bytes));
            const int rc = as_split_bf16x3_launch(a.X, a.ldx, a.K, a.N, a.in_act == 2, a.in_slope, xs, stream);
            if (rc != AS_OK) return rc;
            norm.Xs = xs;
        }
        const int rc = as_conv_gemm_x6d_launch(a, choice, S, stream);
        if (rc != AS_OK) return rc;
    } else if (x6) {
        const int rc = choice == 223 ? as_conv_gemm_x6t_launch(a, S, stream) : as_conv_gemm_x6_launch(a, choice, S, stream);
        if (rc != AS_OK) return rc;
    } else {
        int bm, bn;
        tile_dims(choice, &bm, &bn);
        const dim3 grid(as_cdiv(a.M, bm) * as_cdiv(a.N, bn), S);
        switch (choice) {
        case 22: launch<2, 2>(quad, grid, stream, a); break;
        case 21: launch<2, 1>(quad, grid, stream, a); break;
        case 12: launch<1, 2>(quad, grid, stream, a); break;
        default: launch<1, 1>(quad, grid, stream, a); break;
        }
        AS_CHECK_LAUNCH();
    }
    if (S > 1) {
        const long total = (long)a.M * a.N;
        int blocks = as_cdiv(total, 256 * 4);
        blocks = blocks > 2048 ? 2048 : blocks;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a, S);
        AS_CHECK_LAUNCH();
    }
    return AS_OK;
}
