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
// TM x TN MFMA tiles of 32x32 per wave.
//
// Staging uses raw BUFFER loads: the hardware range check returns 0 for any offset >= num_records, so
// "tap outside the utterance", "row >= Cin", "column >= Cout" all become one select of an out-of-range
// offset -- no divergent branches and no explicit zero fill in the loop.  The next k-tile's 16 loads per
// thread are issued before the current tile's MFMAs and written to the other LDS buffer after them (one
// barrier per k-tile); a k-tile's fragments are all read from LDS up front so the 8*TM*TN MFMAs issue
// back to back behind counted lgkmcnt waits.
#include "common.h"
#include "conv_gemm.h"
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 16
#define OOB 0xFFFFFFFFu

static __device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

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

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    const int tiles_m = (a.M + BM - 1) / BM;
    // XCD-aware order (speed only): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of logical tiles -- the output-channel tiles of one column range then share that XCD's L2
    // copy of the activation columns instead of fetching them 8 times.  Bijective for any grid size.
    const int nb = gridDim.x, xcd = blockIdx.x & 7, q8 = nb >> 3, r8 = nb & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int m0 = (tile % tiles_m) * BM;
    const int n0 = (tile / tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W), 0, (int)((unsigned)a.T * a.Kp * a.M * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((unsigned)a.K * a.ldx * 4u), 0x00020000);

    // ---- staging geometry: thread -> one column (A: output channel, B: frame column), BK/stride rows
    const int a_c = tid % BM, a_r0 = tid / BM;
    const int b_c = tid % BN, b_r0 = tid / BN;
    const bool a_ok = (m0 + a_c) < a.M;
    const unsigned a_base = (unsigned)((a_r0 * a.M + m0 + a_c) * 4);
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
    // split-K: blockIdx.y owns a contiguous slice of the (tap, k-tile) sequence and writes a raw partial slab
    const int S = gridDim.y;
    const int kt_lo = (int)((long)nkt_all * blockIdx.y / S);
    const int nkt = (int)((long)nkt_all * (blockIdx.y + 1) / S);

    float ra[AROWS], rb[BROWS];
    auto gload = [&](int kt) {
        const int t = kt / kt_per_tap;                     // wave-uniform
        const int k0 = (kt - t * kt_per_tap) * BK;
        const unsigned a_tile = (unsigned)((t * a.Kp + k0) * a.M) * 4u;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const unsigned off = a_base + a_tile + (unsigned)(i * A_RSTEP * a.M) * 4u;
            ra[i] = buf_load(rsW, a_ok ? off : OOB);
        }
        const bool ok = (tapmask >> t) & 1u;
        const int src = j + a.dh[t] * Wj + a.dw[t];
        const unsigned b_base = (unsigned)((b_r0 + k0) * a.ldx + src) * 4u;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const unsigned off = b_base + (unsigned)(i * B_RSTEP * a.ldx) * 4u;
            rb[i] = buf_load(rsX, ok ? off : OOB);         // rows >= K fall outside num_records -> 0
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AROWS; ++i) As[buf][a_r0 + i * A_RSTEP][a_c] = ra[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            float v = rb[i];
            if (a.in_act == 2) v = v > 0.f ? v : 0.2f * v;  // LeakyReLU fused on the operand (models.py:89,142)
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

    const int l31 = lane & 31, lk = lane >> 5;
    gload(kt_lo);
    lstore(kt_lo & 1);
    __syncthreads();
    for (int kt = kt_lo; kt < nkt; ++kt) {
#ifdef AS_EXP_NO_STAGE                      // experiment: MFMA + LDS reads only (results are garbage)
        const int buf = 0;
#else
        const int buf = kt & 1;
#endif
#if !defined(AS_EXP_NO_GLOAD) && !defined(AS_EXP_NO_STAGE)
        if (kt + 1 < nkt) gload(kt + 1);
#endif
        // fragments of k-step s+1 are requested BEFORE the MFMAs of step s issue (sched_barrier pins the order;
        // hipcc otherwise sinks each ds_read next to its use and exposes the LDS latency once per step)
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
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][i], bf[s & 1][jn], acc[i][jn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifndef AS_EXP_NO_STAGE
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
#endif
    }

    // ---- epilogue: C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31]
    if (S > 1) {                                           // raw partial sums; splitk_reduce_kernel finishes
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
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = n0 + wn * 32 * TN + jn * 32 + l31;
            if (col >= a.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (row >= a.M) continue;
                float v = acc[i][jn][e];
                if (a.bias) v += a.bias[row];
                if (a.res) v += a.res[(size_t)row * a.ldr + col];
                if (a.div_sqrt2) v = v / 1.41421356237309504880f;
                if (a.act == 1) v = v > 0.f ? v : 0.f;
                else if (a.act == 2) v = v > 0.f ? v : 0.2f * v;
                if (a.transpose_out) a.Y[(size_t)col * a.ldy + row] = v;   // time-major output for the LSTM
                else a.Y[(size_t)row * a.ldy + col] = v;
            }
        }
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
        if (a.bias) v += a.bias[row];
        if (a.res) v += a.res[(size_t)row * a.ldr + col];
        if (a.div_sqrt2) v = v / 1.41421356237309504880f;
        if (a.act == 1) v = v > 0.f ? v : 0.f;
        else if (a.act == 2) v = v > 0.f ? v : 0.2f * v;
        if (a.transpose_out) a.Y[(size_t)col * a.ldy + row] = v;
        else a.Y[(size_t)row * a.ldy + col] = v;
    }
}

static void tile_dims(int choice, int* bm, int* bn)
{
    *bm = (choice == 22 || choice == 21) ? 128 : 64;
    *bn = (choice == 22 || choice == 12) ? 128 : 64;
}

// number of K slices: only for grids that cannot fill 256 CUs, and only while a slice keeps >= 8 k-tiles
static int gemm_ksplit(int M, int N, int Kp, int T, int choice)
{
    const char* env = getenv("AS_GEMM_KSPLIT");          // tuning/experiments only
    int bm, bn;
    tile_dims(choice, &bm, &bn);
    const long tiles = (long)as_cdiv(M, bm) * as_cdiv(N, bn);
    const int nkt = T * (Kp / BK);
    int s = 1;
    if (env && atoi(env) > 0) s = atoi(env);
    else if (tiles < 384) s = as_cdiv(768, tiles);
    if (s > nkt / 8) s = nkt / 8;
    if (s > 16) s = 16;
    return s < 1 ? 1 : s;
}

static int gemm_tile_choice(int M, int N);

extern "C" size_t as_conv_gemm_workspace_bytes(const ConvGemmArgs* args_host)
{
    if (!args_host || args_host->M <= 0 || args_host->N <= 0 || args_host->Kp <= 0 || args_host->T <= 0) return 0;
    const ConvGemmArgs& a = *args_host;
    const int s = gemm_ksplit(a.M, a.N, a.Kp, a.T, gemm_tile_choice(a.M, a.N));
    return s > 1 ? (size_t)s * a.M * a.N * sizeof(float) : 0;
}

static int gemm_tile_choice(int M, int N)
{
    const char* env = getenv("AS_GEMM_TILE");           // tuning/experiments only: 22, 21, 12, 11
    if (env && atoi(env) > 0) return atoi(env);
    // rows: a 128-row tile only when it is not half empty; columns: the widest tile that still gives
    // >= 1.5 workgroups per CU (256 CUs)
    const bool tall = M > 64 && (M % 128 == 0 || M % 128 > 64 || M >= 512);
    if (tall) {
        if ((long)as_cdiv(M, 128) * as_cdiv(N, 128) >= 384) return 22;
        if ((long)as_cdiv(M, 128) * as_cdiv(N, 64) >= 384) return 21;
    } else {
        if ((long)as_cdiv(M, 64) * as_cdiv(N, 128) >= 384) return 12;
    }
    return 11;
}

extern "C" int as_conv_gemm_f32(const ConvGemmArgs* args_host, as_stream_t stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!args_host) return AS_EINVAL;
    const ConvGemmArgs& a = *args_host;
    if (!a.W || !a.X || !a.Y || a.M <= 0 || a.N < 0 || a.K <= 0 || a.T <= 0 || a.T > AS_MAX_TAPS) return AS_EINVAL;
    if (a.Kp < a.K || a.Kp % BK) return AS_EINVAL;
    if (a.ldx < a.N || a.ldy < (a.transpose_out ? a.M : a.N) || (a.res && (a.ldr < a.N || a.transpose_out))) return AS_EINVAL;
    // 32-bit byte offsets inside the buffer descriptors
    if ((double)a.T * a.Kp * a.M * 4.0 >= 4294967296.0 || (double)a.K * a.ldx * 4.0 >= 4294967296.0) return AS_EINVAL;
    if (a.N == 0) return AS_OK;
    const int choice = gemm_tile_choice(a.M, a.N);
    int S = gemm_ksplit(a.M, a.N, a.Kp, a.T, choice);
    if (S > 1 && (!a.ws || a.ws_bytes < (size_t)S * a.M * a.N * sizeof(float))) S = 1;   // no workspace: no split
    char tag[64];
    snprintf(tag, sizeof(tag), "M%d N%d K%d T%d tile%d S%d", a.M, a.N, a.K, a.T, choice, S);
    // algorithmic work of this launch: 2*M*N*K*T flop; bytes = weights + input + output once
    AsProfScope prof__(AS_CLS_GEMM, 2.0 * a.M * a.N * (double)a.K * a.T,
                       4.0 * ((double)a.T * a.K * a.M + (double)a.K * a.N + (double)a.M * a.N), stream, tag);
    switch (choice) {
    case 22: hipLaunchKernelGGL((conv_gemm_kernel<2, 2>), dim3(as_cdiv(a.M, 128) * as_cdiv(a.N, 128), S), dim3(256), 0, stream, a); break;
    case 21: hipLaunchKernelGGL((conv_gemm_kernel<2, 1>), dim3(as_cdiv(a.M, 128) * as_cdiv(a.N, 64), S), dim3(256), 0, stream, a); break;
    case 12: hipLaunchKernelGGL((conv_gemm_kernel<1, 2>), dim3(as_cdiv(a.M, 64) * as_cdiv(a.N, 128), S), dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL((conv_gemm_kernel<1, 1>), dim3(as_cdiv(a.M, 64) * as_cdiv(a.N, 64), S), dim3(256), 0, stream, a); break;
    }
    AS_CHECK_LAUNCH();
    if (S > 1) {
        const long total = (long)a.M * a.N;
        int blocks = as_cdiv(total, 256 * 4);
        blocks = blocks > 2048 ? 2048 : blocks;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a, S);
        AS_CHECK_LAUNCH();
    }
    return AS_OK;
}
