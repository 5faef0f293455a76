// K2, K6, K7, K11 and the style-tower helpers: the bandwidth-bound kernels of the path.
// All operate on the packed-frames layout ([C][N] fp32, utterances concatenated along columns;
// col_off int32 [B+1] gives each utterance's column range).
#include "common.h"
#include <algorithm>
#include <cstring>
#include "artspeech_hip.h"
#include "conv_gemm.h"
#define AS_FILE_CLS AS_CLS_OTHER

static __device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

static __device__ __forceinline__ float lrelu02(float v) { return v > 0.f ? v : 0.2f * v; }

// ---------------------------------------------------------------------------------------------------
// column descriptors
// ---------------------------------------------------------------------------------------------------
__global__ void make_meta_kernel(const int* __restrict__ widths, const int* __restrict__ col_off, int B, int H,
                                 unsigned long long* __restrict__ meta, unsigned* __restrict__ status)
{
    const int b = blockIdx.y;
    const int W = widths[b];
    const int base = col_off[b];
    if (W > AS_META_MAX_W) {                                 // (the field would wrap: every conv would see walls in the wrong places)
        if (blockIdx.x == 0 && threadIdx.x == 0) as_status_raise(status, AS_STATUS_BAD_LAYOUT);
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const unsigned long long h = i / W, w = i - (i / W) * W;
        meta[base + i] = AS_META_PACK(h, w, H, W);
    }
}

extern "C" int as_make_meta(const int32_t* widths, const int32_t* col_off, int B, int H, int n_cols_max,
                            uint64_t* meta, as_stream_t stream)
{
    if (!widths || !col_off || !meta || B < 0 || H <= 0 || H > AS_META_MAX_H) return AS_EINVAL;
    if (B == 0) return AS_OK;
    const int gx = n_cols_max > 0 ? as_cdiv(as_cdiv(n_cols_max, B > 0 ? B : 1), 256) : 1;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(make_meta_kernel, dim3(gx < 1 ? 1 : (gx > 64 ? 64 : gx), B), dim3(256), 0, (hipStream_t)stream,
                       widths, col_off, B, H, reinterpret_cast<unsigned long long*>(meta), as_status_words_device());
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Embedding * sqrt(C), transposed to [C][N]   (RelTransformerEnc.py:373-374)
// ---------------------------------------------------------------------------------------------------
// (emb2, n_split: columns >= n_split take their rows from a second table -- two encoders run as one double-width launch)
// (n_tok < N: the token list is read twice, columns [n_tok, n_split) are filler)
__global__ void embed_kernel(const int* __restrict__ tok, int n_tok, const float* __restrict__ emb, const float* __restrict__ emb2,
                             int n_split, int C, int N, int V, float scale, float* __restrict__ y, int ldy, unsigned* __restrict__ status)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= N) return;
    const int g = emb2 ? j / n_split : 0;                // column group: table g = emb + g (emb2 - emb), its columns re-read the token list
    const int src = j - g * n_split;
    int t = src < n_tok ? tok[src] : 0;
    // nn.Embedding raises on such an id (RelTransformerEnc.py:11-16); a kernel cannot: it reports (as_device_status) and clamps
    if ((t < 0 || t >= V) && c == 0) as_status_raise(status, AS_STATUS_BAD_TOKEN);
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const float* e = emb + (ptrdiff_t)g * (emb2 - emb);
    y[(size_t)c * ldy + j] = e[(size_t)t * C + c] * scale;
}

extern "C" int as_embed_groups_f32(const int32_t* tokens, int n_tok, const float* emb, const float* emb2, int n_split, int C, int N, int V,
                                   float scale, float* y, int ldy, as_stream_t stream)
{
    if (!tokens || !emb || !y || C <= 0 || N < 0 || V <= 0 || ldy < N || n_tok < 0 || n_tok > N) return AS_EINVAL;
    if ((emb2 && n_split <= 0) || (n_tok < N && (!emb2 || n_split < n_tok))) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(embed_kernel, dim3(as_cdiv(N, 64), C), dim3(64), 0, (hipStream_t)stream, tokens, n_tok, emb, emb2, n_split, C, N, V,
                       scale, y, ldy, as_status_words_device());
    AS_CHECK_LAUNCH();
    return AS_OK;
}

extern "C" int as_embed_f32(const int32_t* tokens, const float* emb, int C, int N, int V, float scale, float* y,
                            int ldy, as_stream_t stream)
{
    return as_embed_groups_f32(tokens, N, emb, nullptr, 0, C, N, V, scale, y, ldy, stream);
}

// ---------------------------------------------------------------------------------------------------
// Channel LayerNorm over dim 0 of [C][N]  (RelTransformerEnc.py:281-290), optional ReLU (prenet :323)
// block = 32 columns x 32 channel slices (1024 threads); a thread's <= 16 values stay in registers across the
// mean / variance / normalise passes (one HBM read, one write), two-pass statistics like the reference.
// ---------------------------------------------------------------------------------------------------
#define LN_COLS 32
#define LN_PARTS 32
#define LN_MAXV 16
__global__ void __launch_bounds__(1024)
channel_ln_kernel(const float* __restrict__ x, int ldx, int C, int N, const float* __restrict__ gamma1,
                  const float* __restrict__ beta1, const float* __restrict__ gamma2, const float* __restrict__ beta2, int n_split,
                  float eps, int relu, float* __restrict__ y, int ldy)
{
    __shared__ float red[LN_PARTS][LN_COLS + 1];
    const int col = threadIdx.x % LN_COLS, part = threadIdx.x / LN_COLS;
    const int j = blockIdx.x * LN_COLS + col;
    const bool ok = j < N;
    const int grp = gamma2 ? j / n_split : 0;            // column group g: affine parameters first + g (second - first)
    const float* gamma = gamma1 + (ptrdiff_t)grp * (gamma2 - gamma1);
    const float* beta = beta1 + (ptrdiff_t)grp * (beta2 - beta1);
    float v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = part + i * LN_PARTS;
        v[i] = (ok && c < C) ? x[(size_t)c * ldx + j] : 0.f;
        s += v[i];
    }
    for (int c = part + LN_MAXV * LN_PARTS; c < C; c += LN_PARTS) s += ok ? x[(size_t)c * ldx + j] : 0.f;
    red[part][col] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < LN_PARTS; ++q) tot += red[q][col];
    const float mean = tot / (float)C;
    __syncthreads();
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = part + i * LN_PARTS;
        const float d = v[i] - mean;
        if (c < C) q2 += d * d;
    }
    for (int c = part + LN_MAXV * LN_PARTS; c < C; c += LN_PARTS) {
        const float d = (ok ? x[(size_t)c * ldx + j] : 0.f) - mean;
        q2 += d * d;
    }
    red[part][col] = q2;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int q = 0; q < LN_PARTS; ++q) tot += red[q][col];
    const float rs = 1.0f / sqrtf(tot / (float)C + eps);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = part + i * LN_PARTS;
        if (c < C) {
            float o = (v[i] - mean) * rs * gamma[c] + beta[c];
            if (relu) o = o < 0.f ? 0.f : o;
            y[(size_t)c * ldy + j] = o;
        }
    }
    for (int c = part + LN_MAXV * LN_PARTS; c < C; c += LN_PARTS) {
        float o = (x[(size_t)c * ldx + j] - mean) * rs * gamma[c] + beta[c];
        if (relu) o = o < 0.f ? 0.f : o;
        y[(size_t)c * ldy + j] = o;
    }
}

extern "C" int as_channel_layernorm_groups_f32(const float* x, int ldx, int C, int N, const float* gamma, const float* beta,
                                               const float* gamma2, const float* beta2, int n_split, float eps, int relu, float* y,
                                               int ldy, as_stream_t stream)
{
    if (!x || !y || !gamma || !beta || C <= 0 || N < 0 || ldx < N || ldy < N || ((gamma2 == nullptr) != (beta2 == nullptr)) ||
        (gamma2 && n_split <= 0))
        return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_CLS_LN, 8.0 * C * N, 8.0 * C * N, (hipStream_t)stream);
    hipLaunchKernelGGL(channel_ln_kernel, dim3(as_cdiv(N, LN_COLS)), dim3(1024), 0, (hipStream_t)stream, x, ldx, C, N, gamma,
                       beta, gamma2, beta2, n_split, eps, relu, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

extern "C" int as_channel_layernorm_f32(const float* x, int ldx, int C, int N, const float* gamma, const float* beta,
                                        float eps, int relu, float* y, int ldy, as_stream_t stream)
{
    return as_channel_layernorm_groups_f32(x, ldx, C, N, gamma, beta, nullptr, nullptr, 0, eps, relu, y, ldy, stream);
}

// ---------------------------------------------------------------------------------------------------
// AdaIN + LeakyReLU (+ fused depthwise ConvTranspose1d x2 upsample)   models.py:189-197, 230-240, 172
// one wave per (channel, utterance): per-(b,c) statistics over the utterance's own frames only.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
adain_kernel(const float* __restrict__ x, int ldx, int C, const float* __restrict__ gb, int ldgb,
             const int* __restrict__ col_off, int B, float* __restrict__ y, int ldy, int act,
             const float* __restrict__ pool_w, const float* __restrict__ pool_b, float* __restrict__ xup, int ldup)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);     // (c, b) pair, b fastest
    if (row >= C * B) return;
    const int c = row / B, b = row - c * B;
    const int o0 = col_off[b], L = col_off[b + 1] - o0;
    if (L <= 0) return;
    const float* xr = x + (size_t)c * ldx + o0;
    float s = 0.f;
    for (int i = lane; i < L; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)L;
    float v = 0.f;
    for (int i = lane; i < L; i += 64) { const float d = __fsub_rn(xr[i], mean); v = __fmaf_rn(d, d, v); }   // (explicit: as adain_image_kernel)
    const float var = wave_sum(v) / (float)L;
    const float rs = 1.0f / sqrtf(var + 1e-5f);
    const float g = 1.0f + gb[(size_t)b * ldgb + c];
    const float be = gb[(size_t)b * ldgb + C + c];
    if (!pool_w) {
        float* yr = y + (size_t)c * ldy + o0;
        for (int i = lane; i < L; i += 64) yr[i] = as_adain_val(xr[i], mean, rs, g, be, act);
    } else {
        // depthwise ConvTranspose1d(k3,s2,p1,op1): out[2i] = a[i] w1 + b ; out[2i+1] = a[i] w2 + a[i+1] w0 + b
        const float w0 = pool_w[c * 3 + 0], w1 = pool_w[c * 3 + 1], w2 = pool_w[c * 3 + 2], pb = pool_b[c];
        float* yr = y + (size_t)c * ldy + 2 * o0;
        float* ur = xup ? xup + (size_t)c * ldup + 2 * o0 : nullptr;
        for (int i = lane; i < L; i += 64) {
            const float xi = xr[i];
            const float a0 = as_adain_val(xi, mean, rs, g, be, act);
            const float a1 = i + 1 < L ? as_adain_val(xr[i + 1], mean, rs, g, be, act) : 0.f;
            float e0, e1;
            as_convt_pair(a0, a1, w0, w1, w2, pb, &e0, &e1);
            yr[2 * i] = e0;
            yr[2 * i + 1] = e1;
            if (ur) { ur[2 * i] = xi; ur[2 * i + 1] = xi; }
        }
    }
}

extern "C" int as_adain_f32(const float* x, int ldx, int C, const float* gamma_beta, int ldgb, const int32_t* col_off,
                            int B, float* y, int ldy, int lrelu, const float* pool_w, const float* pool_b,
                            float* x_up, int ld_up, as_stream_t stream)
{
    if (!x || !y || !gamma_beta || !col_off || C <= 0 || B < 0 || ldgb < 2 * C) return AS_EINVAL;
    if ((pool_w == nullptr) != (pool_b == nullptr)) return AS_EINVAL;
    if (B == 0) return AS_OK;
    AsProfScope prof__(AS_CLS_ADAIN, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(adain_kernel, dim3(as_cdiv((long)C * B, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, C,
                       gamma_beta, ldgb, col_off, B, y, ldy, lrelu, pool_w, pool_b, x_up, ld_up);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// small dense layer on per-utterance vectors: y[b][m] = bias[m] + sum_k W[m][k] x[b][k]
// (AdaIN fc models.py:237, style linears models.py:412-415,538, duration_proj models.py:565)
// one wave per output element, lanes over k.
// ---------------------------------------------------------------------------------------------------
template <int KR>
__global__ void __launch_bounds__(256)
linear_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                   int B, int M, int K, float* __restrict__ y, int ldy)
{
    static_assert(KR == 1 || KR == 2 || KR == 4 || KR == 8, "quads of the weight row per lane");
    // one wave per output feature m, all B rows: the weight row is read from HBM exactly once (the batched AdaIN
    // projections are 64 MB of weights against 64 KB of styles) and stays in registers; x is cache resident
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* wr = w + (size_t)m * K;
    const float bm = bias ? bias[m] : 0.f;
    // KR float4 quads of the weight row per lane (K <= 256 KR) and 32 / KR rows of x per trip: every load of a trip is issued before
    // its first use and none sits behind a branch (a quad past the row's end re-reads quad 0 against a zero weight), so a trip is ONE
    // memory round trip.  With one row per trip and the loads inside `if (q < nq)` the 32 rows were 32 round trips; with a fixed KR = 8
    // and four rows per trip, eight (17.6 us for the towers' 512-wide linears, nearly all of it those eight latencies).
    const bool vec = (K & 3) == 0 && (ldx & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
    if (vec && K <= 256 * KR) {
        constexpr int RT = 32 / KR;
        const int nq = K >> 2;                             // float4 quads in a row
        float4 wq[KR];
#pragma unroll
        for (int c = 0; c < KR; ++c) {
            const int q = lane + 64 * c;
            wq[c] = q < nq ? reinterpret_cast<const float4*>(wr)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int b0 = 0; b0 < B; b0 += RT) {
            float4 v[RT][KR];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float4* xq = reinterpret_cast<const float4*>(x + (size_t)min(b0 + r, B - 1) * ldx);
#pragma unroll
                for (int c = 0; c < KR; ++c) {
                    const int q = lane + 64 * c;
                    v[r][c] = xq[q < nq ? q : 0];
                }
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < KR; ++c) s += wq[c].x * v[r][c].x + wq[c].y * v[r][c].y + wq[c].z * v[r][c].z + wq[c].w * v[r][c].w;
                s = wave_sum(s);
                if (lane == 0 && b0 + r < B) y[(size_t)(b0 + r) * ldy + m] = s + bm;
            }
        }
        return;
    }
    for (int b = 0; b < B; ++b) {
        const float* xr = x + (size_t)b * ldx;
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += wr[k] * xr[k];
        s = wave_sum(s);
        if (lane == 0) y[(size_t)b * ldy + m] = s + bm;
    }
}

extern "C" int as_linear_rows_f32(const float* x, int ldx, const float* w, const float* bias, int B, int M, int K,
                                  float* y, int ldy, as_stream_t stream)
{
    if (!x || !w || !y || B < 0 || M <= 0 || K <= 0 || ldx < K || ldy < M) return AS_EINVAL;
    if (B == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    const dim3 grid(as_cdiv(M, 4)), block(256);
    hipStream_t s_ = (hipStream_t)stream;
    if (K <= 256) hipLaunchKernelGGL(linear_rows_kernel<1>, grid, block, 0, s_, x, ldx, w, bias, B, M, K, y, ldy);
    else if (K <= 512) hipLaunchKernelGGL(linear_rows_kernel<2>, grid, block, 0, s_, x, ldx, w, bias, B, M, K, y, ldy);
    else if (K <= 1024) hipLaunchKernelGGL(linear_rows_kernel<4>, grid, block, 0, s_, x, ldx, w, bias, B, M, K, y, ldy);
    else hipLaunchKernelGGL(linear_rows_kernel<8>, grid, block, 0, s_, x, ldx, w, bias, B, M, K, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// A handful of output rows over many columns: Y[m][j] = bias[m] + sum_k W[m][k] X[k][j], M <= 16 (duration_proj, the
// F0 / energy / TV projections, models.py:565,619-621).  One column per thread: every X row is read once, coalesced;
// the weights are wave-uniform (scalar loads).
// ---------------------------------------------------------------------------------------------------
template <int MM>
__global__ void __launch_bounds__(256)
project_cols_kernel(const float* __restrict__ x, int ldx, int K, int N, const float* w, const float* __restrict__ bias, int M,
                    float* __restrict__ y, int ldy)
{
    // block = 64 columns x 4 k-slices (one wave each: rows k = wave, wave + 4, ...), partial sums meet in LDS; the weights are
    // staged in LDS once, k-major and zero-padded to MM outputs: the MM weights of a k are one or four 16-byte broadcasts and the
    // multiply-add loops carry no `m < M` test (with [M][K] and the test, ten outputs cost four times one: 31 against 7.5 us)
    __shared__ float part[3][MM][64];
    extern __shared__ __attribute__((aligned(16))) float wsm[];         // [K][MM]
    for (int i = threadIdx.x; i < MM * K; i += 256) {
        const int k = i / MM, m = i % MM;
        wsm[i] = m < M ? w[(size_t)m * K + k] : 0.f;
    }
    __syncthreads();
    w = wsm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const bool ok = j < N;
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    int k = wave;
    const float* xc = x + (ok ? j : 0);                                 // (columns past N read column 0 and are not stored)
    for (; k + 60 < K; k += 64) {                                       // sixteen rows in flight: the kernel is one memory round trip per batch
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = xc[(size_t)(k + 4 * u) * ldx];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int m = 0; m < MM; ++m)
                acc[m] += w[(k + 4 * u) * MM + m] * v[u];
    }
    for (; k + 12 < K; k += 16) {                                       // four rows in flight
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = xc[(size_t)(k + 4 * u) * ldx];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < MM; ++m)
                acc[m] += w[(k + 4 * u) * MM + m] * v[u];
    }
    for (; k < K; k += 4) {
        const float v = xc[(size_t)k * ldx];
#pragma unroll
        for (int m = 0; m < MM; ++m)
            acc[m] += w[k * MM + m] * v;
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MM; ++m) part[wave - 1][m][lane] = acc[m];
    }
    __syncthreads();
    if (wave == 0 && ok) {
#pragma unroll
        for (int m = 0; m < MM; ++m)
            if (m < M) y[(size_t)m * ldy + j] = ((acc[m] + part[0][m][lane]) + (part[1][m][lane] + part[2][m][lane])) + (bias ? bias[m] : 0.f);
    }
}

extern "C" int as_project_cols_f32(const float* x, int ldx, int K, int N, const float* w, const float* bias, int M, float* y, int ldy,
                                   as_stream_t stream)
{
    const int MM = M == 1 ? 1 : M <= 4 ? 4 : 16;
    if (!x || !w || !y || K <= 0 || N < 0 || M <= 0 || M > 16 || (size_t)MM * K * 4 > 48 * 1024 || ldx < N || ldy < N) return AS_EINVAL;
    if (N == 0) return AS_OK;
    const size_t wsz = (size_t)MM * K * sizeof(float);
    AsProfScope prof__(AS_FILE_CLS, 2.0 * M * K * (double)N, 4.0 * (K + M) * (double)N, (hipStream_t)stream);
    const dim3 grid(as_cdiv(N, 64)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (M == 1) hipLaunchKernelGGL(project_cols_kernel<1>, grid, block, wsz, s, x, ldx, K, N, w, bias, M, y, ldy);
    else if (M <= 4) hipLaunchKernelGGL(project_cols_kernel<4>, grid, block, wsz, s, x, ldx, K, N, w, bias, M, y, ldy);
    else hipLaunchKernelGGL(project_cols_kernel<16>, grid, block, wsz, s, x, ldx, K, N, w, bias, M, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// A 1x1 convolution of a handful of input channels (K <= 16) into up to 128 output channels, written to up to TWO destinations, each as
// fp32 rows and / or as rows of an operand image -- decoder.F0_conv / N_conv / EMA_conv (models.py:480-482, 503-505: 12 -> 128 channels,
// block diagonal) whose result is concatenated into two tensors (x0 and the decode blocks' concat buffer).  On the matrix cores this was
// two launches of a 16-deep k-block that is three quarters zeros, each behind a split pass of its own (2 x 20 us); here a thread owns 8
// output channels of one column: K loads, 8 K FMAs in fp32, one 16-byte image row per part and destination.
__global__ void __launch_bounds__(256)
pointwise_small_kernel(const float* __restrict__ x, int ldx, int K, int N, const float* __restrict__ w, const float* __restrict__ bias, int M,
                       float* y1, int ld1, u32x4_t* h1, float* y2, int ld2, u32x4_t* h2)
{
    const int j = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;       // column (j == N: the images' zero column); 8-row group
    if (j > N) return;
    float xv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xv[k] = (k < K && j < N) ? x[(size_t)k * ldx + j] : 0.f;
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int m = 8 * g + r, mm = m < M ? m : M - 1;
        float acc = bias ? bias[mm] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < K) acc = __builtin_fmaf(w[(size_t)mm * K + k], xv[k], acc);
        const bool ok = m < M && j < N;
        v[r] = ok ? acc : 0.f;
        if (ok && y1) y1[(size_t)m * ld1 + j] = acc;
        if (ok && y2) y2[(size_t)m * ld2 + j] = acc;
    }
    u32x4_t h, l;
    split2(v, h, l);
    const size_t NX = (size_t)N + 1, at = ((size_t)(g >> 1) * 4 + (g & 1)) * NX + j;
    if (h1) { h1[at] = h; h1[at + 2 * NX] = l; }
    if (h2) { h2[at] = h; h2[at + 2 * NX] = l; }
}

extern "C" int as_pointwise_small_f32(const float* x, int ldx, int K, int N, const float* w, const float* bias, int M, float* y1, int ld1,
                                      uint16_t* yh1, float* y2, int ld2, uint16_t* yh2, as_stream_t stream)
{
    if (!x || !w || K <= 0 || K > 16 || M <= 0 || N < 0 || ldx < N || (!y1 && !yh1 && !y2 && !yh2) || (y1 && ld1 < N) || (y2 && ld2 < N)) return AS_EINVAL;
    if (((reinterpret_cast<uintptr_t>(yh1) | reinterpret_cast<uintptr_t>(yh2)) & 15) != 0) return AS_EINVAL;
    if (N == 0) return AS_OK;
    const int groups = (yh1 || yh2) ? 2 * as_kbx(M) : (M + 7) / 8;     // an image's rows beyond M (up to its 64-row block) are zeros
    AsProfScope prof__(AS_FILE_CLS, 2.0 * M * K * (double)N, 4.0 * (K + 2.0 * M * ((y1 || yh1 ? 1 : 0) + (y2 || yh2 ? 1 : 0))) * (double)N, (hipStream_t)stream);
    hipLaunchKernelGGL(pointwise_small_kernel, dim3(as_cdiv(N + 1, 256), groups), dim3(256), 0, (hipStream_t)stream, x, ldx, K, N, w, bias, M, y1, ld1,
                       reinterpret_cast<u32x4_t*>(yh1), y2, ld2, reinterpret_cast<u32x4_t*>(yh2));
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// K2: durations -> integer alignment -> gather          models.py:361-368
// ---------------------------------------------------------------------------------------------------
// round-half-even, clamp(min=1) (torch.round + clamp, models.py:361); frame offsets per utterance;
// tok_of_frame[f] = packed token column that frame f repeats.  Single workgroup (B and N are small).
// One global exclusive prefix over the packed tokens IS the frame index (utterances are packed back to back in both layouts): thread t
// owns a contiguous chunk of tokens, chunk sums are scanned in LDS, then every thread writes its tokens' frames.  (One thread per
// UTTERANCE walking its tokens made the long-form batch -- 8 x 1024 tokens -- a 170 us serial loop.)
__global__ void __launch_bounds__(1024)
durations_kernel(const float* __restrict__ dur_f, const int* __restrict__ forced, const int* __restrict__ tok_off,
                 int B, int* __restrict__ dur_i, int* __restrict__ frame_off, int* __restrict__ tok_of_frame,
                 int max_frames, unsigned* __restrict__ status)
{
    __shared__ int sums[1024];
    const int ntok = tok_off[B];
    const int t = threadIdx.x;
    const int per = (ntok + 1023) / 1024, lo = min(t * per, ntok), hi = min(lo + per, ntok);
    int local = 0;
    for (int i = lo; i < hi; ++i) {
        int d;
        if (forced) d = forced[i];
        else {
            // (the reference's `int(pred_dur[i])` raises on NaN / inf, models.py:363-366: here the device status does -- the count
            //  becomes 1 -- and an absurd finite value is held at 16 384 frames per token so that the sums stay inside an int: the caller's
            //  AS_ENOSPC path meets it)
            const float v = dur_f[i];
            const bool fin = fabsf(v) <= 3.0e38f;
            if (!fin) as_status_raise(status, AS_STATUS_F16_RANGE);
            const float r = fin ? rintf(v) : 1.f;          // ties to even, like torch.round
            d = (int)(r < 1.f ? 1.f : fminf(r, 16384.f));
        }
        dur_i[i] = d;
        local += d;
    }
    sums[t] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {             // inclusive scan of the chunk sums
        const int v = t >= off ? sums[t - off] : 0;
        __syncthreads();
        sums[t] += v;
        __syncthreads();
    }
    int f = sums[t] - local;                               // first frame of this thread's first token
    // frame_off[b] = frame index of utterance b's first token: the thread whose chunk holds that token writes it
    if (t == 0) frame_off[B] = sums[1023];
    int b = 0;
    if (lo < hi || ntok == 0) {
        // utterances whose first token lies in [lo, hi): B is small, a linear scan per thread is fine
        for (b = 0; b < B; ++b) {
            const int first = tok_off[b];
            if (first >= lo && first < hi) {
                int fb = f;
                for (int i = lo; i < first; ++i) fb += dur_i[i];
                frame_off[b] = fb;
            }
        }
    }
    if (t == 0)                                            // empty utterances at the very end (first token == ntok)
        for (b = 0; b < B; ++b)
            if (tok_off[b] >= ntok) frame_off[b] = sums[1023];
    if (!tok_of_frame) return;
    for (int i = lo; i < hi; ++i) {
        const int d = dur_i[i];
        for (int r = 0; r < d; ++r, ++f)
            if (f < max_frames) tok_of_frame[f] = i;
    }
}

extern "C" int as_durations_f32(const float* dur_f32, const int32_t* forced_dur, const int32_t* tok_off, int B,
                                int32_t* dur_i32, int32_t* frame_off, int32_t* tok_of_frame, int max_frames,
                                as_stream_t stream)
{
    if ((!dur_f32 && !forced_dur) || !tok_off || !dur_i32 || !frame_off || B < 0 || B > 1024) return AS_EINVAL;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(durations_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dur_f32, forced_dur, tok_off, B,
                       dur_i32, frame_off, tok_of_frame, max_frames, as_status_words_device());
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// y[c][f*rep + r] = x[c][tok_of_frame[f]]     (T_en @ aln, and the decoder's nearest x2: models.py:367-368, :500)
__global__ void expand_kernel(const float* __restrict__ x, int ldx, int C, const int* __restrict__ tok_of_frame,
                              int n_frames, int rep, float* __restrict__ y, int ldy)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= n_frames * rep) return;
    y[(size_t)c * ldy + j] = x[(size_t)c * ldx + tok_of_frame[j / rep]];
}

extern "C" int as_expand_f32(const float* x, int ldx, int C, const int32_t* tok_of_frame, int n_frames, int repeat,
                             float* y, int ldy, as_stream_t stream)
{
    if (!x || !y || !tok_of_frame || C <= 0 || n_frames < 0 || repeat < 1 || ldy < n_frames * repeat) return AS_EINVAL;
    if (n_frames == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(expand_kernel, dim3(as_cdiv((long)n_frames * repeat, 256), C), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, C, tok_of_frame, n_frames, repeat, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// K11: reference-feature glue   models.py:431,447-449,655-660
// n[j] = (log || exp(4 mel - 4) ||_2 - e_mean) / e_std ; f0 = (f0 - p_mean)/p_std ; ema[c] = (ema[c]-m[c])/s[c]
// feat rows: 0 = n, 1 = f0, 2..11 = ema  ([12][N])
// ---------------------------------------------------------------------------------------------------
// block = 32 columns x 8 row slices: a thread sums the squares of every eighth mel row of its column (all its loads in flight at once),
// the slices meet in LDS; slice 0 finishes the column.  (One thread per column walked 80 dependent exp / load pairs: 37 us at the head of
// every step.)
__global__ void __launch_bounds__(256)
ref_features_kernel(const float* __restrict__ mel, int ldm, int n_mels, const float* __restrict__ f0_raw,
                    const float* __restrict__ ema_raw, int lde, int N, const float* __restrict__ stats,
                    float* __restrict__ feat, int ldf)
{
    __shared__ float part[8][32];
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + col;
    const int jc = j < N ? j : N - 1;
    constexpr int RMAX = 16;                                             // rows per slice held in registers: n_mels <= 128
    float v[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int m = sl + 8 * r;
        v[r] = mel[(size_t)(m < n_mels ? m : 0) * ldm + jc];
    }
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        if (sl + 8 * r < n_mels) {
            const float e = expf(v[r] * 4.0f + -4.0f);
            ss += e * e;
        }
    }
    for (int m = sl + 8 * RMAX; m < n_mels; m += 8) {                     // (more than 128 mel bins: the remaining rows one by one)
        const float e = expf(mel[(size_t)m * ldm + jc] * 4.0f + -4.0f);
        ss += e * e;
    }
    part[sl][col] = ss;
    __syncthreads();
    if (j >= N) return;
    // stats: [0]=energy_mean [1]=energy_std [2]=pitch_mean [3]=pitch_std [4..13]=EMA_mean [14..23]=EMA_std
    if (sl == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += part[q][col];
        feat[j] = (logf(sqrtf(tot)) - stats[0]) / stats[1];
    } else if (sl == 1) {
        feat[(size_t)ldf + j] = (f0_raw[j] - stats[2]) / stats[3];
    } else {
        for (int c = sl - 2; c < 10; c += 6) feat[(size_t)(2 + c) * ldf + j] = (ema_raw[(size_t)c * lde + j] - stats[4 + c]) / stats[14 + c];
    }
}

extern "C" int as_ref_features_f32(const float* mel, int ldm, int n_mels, const float* f0_raw, const float* ema_raw,
                                   int lde, int N, const float* stats24, float* feat, int ldf, as_stream_t stream)
{
    if (!mel || !f0_raw || !ema_raw || !stats24 || !feat || N < 0 || n_mels <= 0 || ldm < N || lde < N || ldf < N) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(ref_features_kernel, dim3(as_cdiv(N, 32)), dim3(256), 0, (hipStream_t)stream, mel, ldm, n_mels,
                       f0_raw, ema_raw, lde, N, stats24, feat, ldf);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// copy a [C][*] tensor's per-utterance column windows into another packed layout:
// dst[c][dst_off[b] + i] = src[c][src_off[b] + start + i], i < dst_len(b)  (the T-1 crop, models.py:459-471)
__global__ void crop_kernel(const float* __restrict__ src, int lds, const int* __restrict__ src_off, int start,
                            float* __restrict__ dst, int ldd, const int* __restrict__ dst_off, int C)
{
    const int b = blockIdx.y, c = blockIdx.z;
    const int L = dst_off[b + 1] - dst_off[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x)
        dst[(size_t)c * ldd + dst_off[b] + i] = src[(size_t)c * lds + src_off[b] + start + i];
}

extern "C" int as_crop_f32(const float* src, int lds, const int32_t* src_off, int start, float* dst, int ldd,
                           const int32_t* dst_off, int B, int C, int max_len, as_stream_t stream)
{
    if (!src || !dst || !src_off || !dst_off || B < 0 || C <= 0) return AS_EINVAL;
    if (B == 0 || max_len <= 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(crop_kernel, dim3(as_cdiv(max_len, 256), B, C), dim3(256), 0, (hipStream_t)stream, src, lds,
                       src_off, start, dst, ldd, dst_off, C);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// H channel rows of per-utterance column windows -> per-utterance H x W images, packed [1][sum_b H*W_b]:
// dst[img_off[b] + h*W_b + i] = src[h][src_off[b] + start + i]   (the mel / TV inputs of the 2-D towers, models.py:419-421)
__global__ void rows_to_images_kernel(const float* __restrict__ src, int lds, const int* __restrict__ src_off, int start,
                                      float* __restrict__ dst, const int* __restrict__ img_off, int H)
{
    const int b = blockIdx.y, h = blockIdx.z;
    const int Wb = (img_off[b + 1] - img_off[b]) / H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Wb; i += gridDim.x * blockDim.x)
        dst[(size_t)img_off[b] + (size_t)h * Wb + i] = src[(size_t)h * lds + src_off[b] + start + i];
}

extern "C" int as_rows_to_images_f32(const float* src, int lds, const int32_t* src_off, int start, int H, float* dst,
                                     const int32_t* img_off, int B, int max_w, as_stream_t stream)
{
    if (!src || !dst || !src_off || !img_off || B < 0 || H <= 0) return AS_EINVAL;
    if (B == 0 || max_w <= 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(rows_to_images_kernel, dim3(as_cdiv(max_w, 256), B, H), dim3(256), 0, (hipStream_t)stream, src, lds,
                       src_off, start, dst, img_off, H);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// JDCNet: BatchNorm2d(eval) -> LeakyReLU -> MaxPool2d((1, k)) over the mel axis
// (Utils/JDC/model.py:29-34 pool_block, :166-170 ResBlock.pre_conv).  Images [H = mel bins][W_b = frames].
// ---------------------------------------------------------------------------------------------------
__global__ void bn_lrelu_maxpool_rows_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ tok_off, int H, int k,
                                             const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                             float* __restrict__ y, int ldy, int to_channels)
{
    const int b = blockIdx.y, c = blockIdx.z;
    const int t0 = tok_off[b], Wb = tok_off[b + 1] - t0, Hout = H / k;
    const float* xr = x + (size_t)c * ldx + (size_t)H * t0;
    const float sc = scale[c], sh = shift[c];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hout * Wb; i += gridDim.x * blockDim.x) {
        const int ho = i / Wb, w = i - ho * Wb;
        float m = -INFINITY;
        for (int r = 0; r < k; ++r) {
            float v = xr[(size_t)(ho * k + r) * Wb + w] * sc + sh;
            v = v > 0.f ? v : slope * v;
            m = fmaxf(m, v);
        }
        if (to_channels) y[(size_t)(c * Hout + ho) * ldy + t0 + w] = m;
        else y[(size_t)c * ldy + (size_t)Hout * t0 + i] = m;
    }
}

extern "C" int as_bn_lrelu_maxpool_rows_f32(const float* x, int ldx, const int32_t* tok_off, int B, int C, int H, int k,
                                            const float* scale, const float* shift, float slope, float* y, int ldy,
                                            int to_channels, int total_frames, as_stream_t stream)
{
    if (!x || !y || !tok_off || !scale || !shift || B < 0 || C <= 0 || H <= 0 || k <= 0 || k > H) return AS_EINVAL;
    if (B == 0 || total_frames <= 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 4.0 * C * (double)total_frames * (H + H / k), (hipStream_t)stream);
    const int per_utt = as_cdiv((long)(H / k) * total_frames, B);          // average output pixels of an utterance
    int gx = as_cdiv(per_utt, 256);
    gx = gx < 1 ? 1 : gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(bn_lrelu_maxpool_rows_kernel, dim3(gx, B, C), dim3(256), 0, (hipStream_t)stream, x, ldx, tok_off, H, k, scale,
                       shift, slope, y, ldy, to_channels);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Style-tower helpers (K10): learned / average 2x down-sampling, im2col for the valid 5x5 convs,
// LeakyReLU + global average pool.  Images are [C][sum_b H*W_b] with per-utterance widths.
// ---------------------------------------------------------------------------------------------------
// LearnedDownSample (models.py:27-31): depthwise conv, 'half' = 3x3 s2 p1, 'channelpreserve' = 1x3 s(1,2) p(0,1);
// ResBlk1d.pool (models.py:116) is the H = 1 case of 'channelpreserve'.  Optional LeakyReLU on the result
// (the activation that follows it at models.py:94 / :148).
// A workgroup takes 8 consecutive channels of one utterance, a thread one output position of all eight: the values leave as
// fp32 rows (y) and / or as one 16-byte row per part of the consumer conv's operand image (yh: the output of
// models.py:27-31,116 feeds nothing but the block's second conv).
// Branch-free: every tap is loaded from a clamped position and multiplied by its weight or by zero (a tap outside the image), so the
// 8 x 3 KH loads of a thread are all in flight together -- with `continue` in the tap loops each load sat in its own divergent
// region and paid its own memory round trip (the 64-channel 509 440-column launch: 82 us for 200 MB).  Same summation order as before.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));      // a pair of floats at any 4-byte address (odd row starts)
template <int KH>
static __device__ __forceinline__ void dwconv_down_body(const AsDownArgs& a, int b, int g, int bx, int gxn)
{
    __shared__ float ws[8][KH * 3 + 1];                                 // the eight channels' taps and bias
    const float* __restrict__ x = a.x;
    const int ldx = a.ldx, Hin = a.Hin, ldy = a.ldy, Hout = a.Hout, C = a.C, Nout = a.n_out, act = a.lrelu;
    const int* __restrict__ in_off = a.in_off;
    const int* __restrict__ in_w = a.in_w;
    const int* __restrict__ out_off = a.out_off;
    const int* __restrict__ out_w = a.out_w;
    const float* __restrict__ w = a.w;
    const float* __restrict__ bias = a.bias;
    float* __restrict__ y = a.y;
    u32x4_t* __restrict__ yh = reinterpret_cast<u32x4_t*>(a.yh);
    const int sh = KH == 3 ? 2 : 1, ph = KH == 3 ? 1 : 0;
    const int c0 = g * 8;
    const int Wi = in_w[b], Wo = out_w[b];
    const size_t NX = (size_t)Nout + 1, plane = ((size_t)(g >> 1) * 4 + (g & 1)) * NX;
    if (yh && b == 0 && bx == 0 && threadIdx.x < 2) yh[plane + (size_t)threadIdx.x * 2 * NX + Nout] = u32x4_t{0u, 0u, 0u, 0u};
    if (threadIdx.x < 8 * (KH * 3 + 1)) {
        const int r = threadIdx.x / (KH * 3 + 1), k = threadIdx.x % (KH * 3 + 1), c = c0 + r;
        ws[r][k] = c < C ? (k < KH * 3 ? w[(size_t)c * KH * 3 + k] : bias[c]) : 0.f;
    }
    __syncthreads();
    const int ib = in_off[b], ob = out_off[b], lane = threadIdx.x & 63;
    // The window's columns 2 wo, 2 wo + 1 come as ONE 8-byte load per row and channel (a wave's loads are then whole cache lines; as three
    // 4-byte loads at a stride of two floats every instruction touched its lines half-used and the address path, not the memory, set
    // the pace: 2.5 TB/s); column 2 wo - 1 is the previous lane's second value -- the first lane of a wave fetches its own.  Rows of
    // one column (Wi = 1) and the last column of an odd row (no right neighbour) take the pair from one float earlier.
    for (int i0 = bx * 256; i0 < Hout * Wo; i0 += gxn * 256) {      // (whole waves stay in the loop: shuffles)
        const int i = i0 + threadIdx.x, ic = min(i, Hout * Wo - 1);
        const int ho = ic / Wo, wo = ic - ho * Wo;
        const bool has_r = 2 * wo + 1 < Wi;                              // the right neighbour exists
        const int pc = Wi >= 2 ? (has_r ? 2 * wo : 2 * wo - 1) : 0;      // first column of the pair that is loaded
        const bool own_left = lane == 0 && wo > 0;
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) t[r] = 0.f;
        // every load of the window is issued before anything waits for one (the pairs; then, in ONE divergent region, the first lane's
        // own left neighbours): a shuffle or a branch between the loads would serialise 24 memory round trips
        f32x2u p2[8][KH];
        float lo[8][KH];
        bool okh[KH];
        int base[KH];
#pragma unroll
        for (int a = 0; a < KH; ++a) {
            const int hi = ho * sh - ph + a, hc = min(max(hi, 0), Hin - 1);
            okh[a] = hi >= 0 && hi < Hin;
            base[a] = ib + hc * Wi;
        }
        if (Wi >= 2) {
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int r = 0; r < 8; ++r) p2[r][a] = *reinterpret_cast<const f32x2u*>(x + (size_t)min(c0 + r, C - 1) * ldx + base[a] + pc);
        } else {
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int r = 0; r < 8; ++r) { const float e = x[(size_t)min(c0 + r, C - 1) * ldx + base[a]]; p2[r][a] = f32x2u{e, e}; }
        }
        if (own_left) {
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int r = 0; r < 8; ++r) lo[r][a] = x[(size_t)min(c0 + r, C - 1) * ldx + base[a] + 2 * wo - 1];
        }
        float v[8][KH][3];
#pragma unroll
        for (int a = 0; a < KH; ++a)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float x0 = (has_r || Wi < 2) ? p2[r][a].x : p2[r][a].y, x1 = p2[r][a].y;   // (no right neighbour: x1 unused, its weight is skipped)
                const float xl = __shfl_up(x1, 1);                       // previous lane: (ho, wo - 1)'s right value = column 2 wo - 1
                v[r][a][0] = own_left ? lo[r][a] : xl; v[r][a][1] = x0; v[r][a][2] = x1;
            }
        // a previous lane whose pair was shifted (its has_r false) cannot be this lane's neighbour: that lane ends a row, this one starts the next (wo = 0)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < KH; ++a) {
                if (okh[a] && wo > 0) s += v[r][a][0] * ws[r][a * 3 + 0];
                if (okh[a]) s += v[r][a][1] * ws[r][a * 3 + 1];
                if (okh[a] && has_r) s += v[r][a][2] * ws[r][a * 3 + 2];
            }
            s += ws[r][KH * 3];
            if (act) s = lrelu02(s);
            if (c0 + r >= C) s = 0.f;
            else if (y && i < Hout * Wo) y[(size_t)(c0 + r) * ldy + ob + i] = s;
            t[r] = s;
        }
        if (yh && i < Hout * Wo) {
            u32x4_t h, l;
            split2(t, h, l);
            const size_t at = plane + ob + i;
            yh[at] = h;
            yh[at + 2 * NX] = l;
        }
    }
}

static int down_one(const AsDownArgs& a, hipStream_t stream);

static int dwconv_launch(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, float* y, int ldy, const int32_t* out_off,
                         const int32_t* out_w, int Hout, const float* w, const float* bias, int kh, int B, int C, int max_out, int lrelu,
                         uint16_t* yh, int Nout, hipStream_t stream)
{
    AsDownArgs a;
    memset(&a, 0, sizeof(a));
    a.kind = 0; a.x = x; a.ldx = ldx; a.in_off = in_off; a.in_w = in_w; a.Hin = Hin; a.y = y; a.ldy = ldy; a.out_off = out_off; a.out_w = out_w;
    a.Hout = Hout; a.w = w; a.bias = bias; a.kh = kh; a.B = B; a.C = C; a.max_out = max_out; a.lrelu = lrelu; a.yh = yh; a.n_out = Nout;
    return down_one(a, stream);
}

extern "C" int as_dwconv_down_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin,
                                  float* y, int ldy, const int32_t* out_off, const int32_t* out_w, int Hout,
                                  const float* w, const float* bias, int kh, int B, int C, int max_out, int lrelu,
                                  as_stream_t stream)
{
    return dwconv_launch(x, ldx, in_off, in_w, Hin, y, ldy, out_off, out_w, Hout, w, bias, kh, B, C, max_out, lrelu, nullptr, 0, (hipStream_t)stream);
}

extern "C" int as_dwconv_down_image_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, const int32_t* out_off,
                                        const int32_t* out_w, int Hout, const float* w, const float* bias, int kh, int B, int C, int max_out,
                                        int lrelu, uint16_t* yh, int n_out, as_stream_t stream)
{
    if (!yh) return AS_EINVAL;
    return dwconv_launch(x, ldx, in_off, in_w, Hin, nullptr, 0, out_off, out_w, Hout, w, bias, kh, B, C, max_out, lrelu, yh, n_out, (hipStream_t)stream);
}

// DownSample (models.py:43-57) / ResBlk1d.downsample (:127-130): replicate the last column when W is odd,
// then average pool (ph x 2); optionally  y = (pool(x) + res) / sqrt(2)  (the block's output, models.py:99-100).
template <int PH>
static __device__ __forceinline__ void avgpool_down_body(const AsDownArgs& a, int b, int g, int bx, int gxn)
{
    // 8 consecutive channels per workgroup, one output position of all eight per thread (as dwconv_down_body)
    const float* __restrict__ x = a.x;
    const int ldx = a.ldx, ldy = a.ldy, Hout = a.Hout, C = a.C, Nout = a.n_out, yh_lrelu = a.lrelu, ldr = a.ldr;
    const int* __restrict__ in_off = a.in_off;
    const int* __restrict__ in_w = a.in_w;
    const int* __restrict__ out_off = a.out_off;
    const int* __restrict__ out_w = a.out_w;
    const float* __restrict__ res = a.res;
    float* __restrict__ y = a.y;
    u32x4_t* __restrict__ yh = reinterpret_cast<u32x4_t*>(a.yh);
    const int c0 = g * 8;
    const int Wi = in_w[b], Wo = out_w[b], ib = in_off[b], ob = out_off[b];
    const size_t NX = (size_t)Nout + 1, plane = ((size_t)(g >> 1) * 4 + (g & 1)) * NX;
    if (yh && b == 0 && bx == 0 && threadIdx.x < 2) yh[plane + (size_t)threadIdx.x * 2 * NX + Nout] = u32x4_t{0u, 0u, 0u, 0u};
    for (int i = bx * 256 + threadIdx.x; i < Hout * Wo; i += gxn * 256) {
        const int ho = i / Wo, wo = i - ho * Wo;
        // the pair (2 wo, 2 wo + 1) as one 8-byte load (whole cache lines per wave); the last column of an odd row stands for its
        // missing neighbour: there the pair is loaded one float earlier and its second value is used twice.  Every load of the
        // thread (8 channels x PH rows, the residual's 8) is issued before the first sum.
        const bool has_r = 2 * wo + 1 < Wi;
        const int pc = Wi >= 2 ? (has_r ? 2 * wo : 2 * wo - 1) : 0;
        f32x2u p2[8][PH];
        float rv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float* xr = x + (size_t)min(c0 + r, C - 1) * ldx + ib;
#pragma unroll
            for (int a = 0; a < PH; ++a) {
                if (Wi >= 2) p2[r][a] = *reinterpret_cast<const f32x2u*>(xr + (size_t)(ho * PH + a) * Wi + pc);
                else { const float e = xr[(size_t)(ho * PH + a) * Wi]; p2[r][a] = f32x2u{e, e}; }
            }
            rv[r] = res ? res[(size_t)min(c0 + r, C - 1) * ldr + ob + i] : 0.f;
        }
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < PH; ++a) {
                s += (has_r || Wi < 2) ? p2[r][a].x : p2[r][a].y;
                s += p2[r][a].y;
            }
            s = s / (float)(2 * PH);
            if (res) s = (s + rv[r]) / 1.41421356237309504880f;
            if (c0 + r < C) {
                if (y) y[(size_t)(c0 + r) * ldy + ob + i] = s;
                if (yh_lrelu) s = lrelu02(s);
            } else {
                s = 0.f;
            }
            t[r] = s;
        }
        if (yh) {
            u32x4_t h, l;
            split2(t, h, l);
            const size_t at = plane + ob + i;
            yh[at] = h;
            yh[at + 2 * NX] = l;
        }
    }
}

static int avgpool_launch(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, float* y, int ldy, const int32_t* out_off,
                          const int32_t* out_w, int Hout, int pool_h, const float* res, int ldr, int B, int C, int max_out, uint16_t* yh, int Nout,
                          int yh_lrelu, hipStream_t stream)
{
    AsDownArgs a;
    memset(&a, 0, sizeof(a));
    a.kind = 1; a.x = x; a.ldx = ldx; a.in_off = in_off; a.in_w = in_w; a.Hin = Hin; a.y = y; a.ldy = ldy; a.out_off = out_off; a.out_w = out_w;
    a.Hout = Hout; a.pool_h = pool_h; a.res = res; a.ldr = ldr; a.B = B; a.C = C; a.max_out = max_out; a.yh = yh; a.n_out = Nout; a.lrelu = yh_lrelu;
    return down_one(a, stream);
}

extern "C" int as_avgpool_down_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin,
                                   float* y, int ldy, const int32_t* out_off, const int32_t* out_w, int Hout,
                                   int pool_h, const float* res, int ldr, int B, int C, int max_out, as_stream_t stream)
{
    if (!y) return AS_EINVAL;
    return avgpool_launch(x, ldx, in_off, in_w, Hin, y, ldy, out_off, out_w, Hout, pool_h, res, ldr, B, C, max_out, nullptr, 0, 0, (hipStream_t)stream);
}

extern "C" int as_avgpool_down_image_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, float* y, int ldy,
                                         const int32_t* out_off, const int32_t* out_w, int Hout, int pool_h, const float* res, int ldr, int B,
                                         int C, int max_out, uint16_t* yh, int n_out, int yh_lrelu, as_stream_t stream)
{
    if (!yh) return AS_EINVAL;
    return avgpool_launch(x, ldx, in_off, in_w, Hin, y, ldy, out_off, out_w, Hout, pool_h, res, ldr, B, C, max_out, yh, n_out, yh_lrelu,
                          (hipStream_t)stream);
}

// The learned shortcut's input of a tower's FIRST block, avgpool(stem(x)) (models.py:79-84 after the tower's Cin = 1 stem conv, :385,393),
// straight from the one-channel input: the stem's fp32 output (64 x 509 440 floats at the mel tower) is then neither written nor read
// back -- its only other consumer, the block's conv1, reads the LeakyReLU image the stem writes.  w = the stem's fp32 image [T][Kp][M]
// (k = 0 rows), taps in taps_2d(3, 3) / taps_1d(3) order, zero padding; then DownSample's (pool_h x 2) average with the last column
// replicated when W is odd, summed in avgpool_down_kernel's order.  Workgroup = 8 channels of one utterance, thread = one output position.
static __device__ __forceinline__ void stem_pool_image_body(const AsDownArgs& a, int b, int g, int bx, int gxn)
{
    __shared__ float ws[8][10];                                         // the eight channels' taps and bias
    const float* __restrict__ x = a.x;
    const int Hin = a.Hin, Hout = a.Hout, ph = a.pool_h, Kp = a.Kp, KH = a.kh, C = a.C, Nout = a.n_out;
    const int* __restrict__ in_off = a.in_off;
    const int* __restrict__ in_w = a.in_w;
    const int* __restrict__ out_off = a.out_off;
    const int* __restrict__ out_w = a.out_w;
    const float* __restrict__ w = a.w;
    const float* __restrict__ bias = a.bias;
    u32x4_t* __restrict__ yh = reinterpret_cast<u32x4_t*>(a.yh);
    const int c0 = g * 8;
    const int Wi = in_w[b], Wo = out_w[b];
    const size_t NX = (size_t)Nout + 1, plane = ((size_t)(g >> 1) * 4 + (g & 1)) * NX;
    if (b == 0 && bx == 0 && threadIdx.x < 2) yh[plane + (size_t)threadIdx.x * 2 * NX + Nout] = u32x4_t{0u, 0u, 0u, 0u};
    if (threadIdx.x < 80) {
        const int r = threadIdx.x / 10, k = threadIdx.x % 10, c = c0 + r;
        ws[r][k] = c < C ? (k < 9 ? (k < KH * 3 ? w[(size_t)k * Kp * C + c] : 0.f) : (bias ? bias[c] : 0.f)) : 0.f;
    }
    __syncthreads();
    const int ib = in_off[b], ob = out_off[b], pad = KH / 2;
    for (int i = bx * 256 + threadIdx.x; i < Hout * Wo; i += gxn * 256) {
        const int ho = i / Wo, wo = i - ho * Wo;
        float win[4][4];                                                // rows ho ph - pad .. + ph + KH - 2, columns 2 wo - 1 .. 2 wo + 2 (zero outside)
        // branch-free: the sixteen loads go to clamped positions, unconditionally, and the zero padding is a select
        // behind them -- with each load behind its own bounds test every one sat in its own divergent region and paid its own round trip
        // (round 5: this launch took 71 us for 50 MB of output)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hi = ho * ph - pad + r, hc = min(max(hi, 0), Hin - 1);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int wi = 2 * wo - 1 + cc, wc = min(max(wi, 0), Wi - 1);
                win[r][cc] = x[ib + hc * Wi + wc];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hi = ho * ph - pad + r;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int wi = 2 * wo - 1 + cc;
                const bool ok = r < ph + KH - 1 && hi >= 0 && hi < Hin && wi >= 0 && wi < Wi;
                win[r][cc] = ok ? win[r][cc] : 0.f;
            }
        }
        const bool dup = 2 * wo + 1 >= Wi;                              // odd width: the last column stands for its missing neighbour
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float wr[10];                                               // (registers: the products below would otherwise re-read LDS 36 times)
#pragma unroll
            for (int k = 0; k < 10; ++k) wr[k] = ws[r][k];
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (a >= ph) break;
                float v[2];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    float sum = 0.f;
#pragma unroll
                    for (int ta = 0; ta < 3; ++ta) {
                        if (ta >= KH) break;
#pragma unroll
                        for (int td = 0; td < 3; ++td) sum += wr[ta * 3 + td] * win[a + ta][cc + td];
                    }
                    v[cc] = sum + wr[9];
                }
                s += v[0];
                s += dup ? v[0] : v[1];
            }
            t[r] = c0 + r < C ? s / (float)(2 * ph) : 0.f;
        }
        u32x4_t h, l;
        split2(t, h, l);
        const size_t at = plane + ob + i;
        yh[at] = h;
        yh[at + 2 * NX] = l;
    }
}

extern "C" int as_stem_pool_image_f32(const float* x, const int32_t* in_off, const int32_t* in_w, int Hin, const int32_t* out_off,
                                      const int32_t* out_w, int Hout, int pool_h, const float* w, int Kp, const float* bias, int kh, int B, int C,
                                      int max_out, uint16_t* yh, int n_out, as_stream_t stream)
{
    AsDownArgs a;
    memset(&a, 0, sizeof(a));
    a.kind = 2; a.x = x; a.in_off = in_off; a.in_w = in_w; a.Hin = Hin; a.out_off = out_off; a.out_w = out_w; a.Hout = Hout; a.pool_h = pool_h;
    a.w = w; a.Kp = Kp; a.bias = bias; a.kh = kh; a.B = B; a.C = C; a.max_out = max_out; a.yh = yh; a.n_out = n_out;
    return down_one(a, (hipStream_t)stream);
}

// ONE launch for several of the tower down-sampling steps above (AsDownArgs: include/artspeech_hip.h): blockIdx.y walks the problems'
// workgroups back to back (exactly the workgroups every problem needs: a common 3-D grid sized for the widest problem was three quarters empty workgroups and slower than the launches it replaced).
// The four towers of the style path and dur_block (models.py:385-411, 530-535) are independent and march through their blocks in step,
// so their LearnedDownSample / DownSample launches -- 7-20 us each at C3 sizes, all latency -- come in sets of four.
#define DOWN_MAXP 6
struct DownMulti {
    int32_t n, pad_;
    int32_t blk0[DOWN_MAXP + 2];         // first workgroup of problem i; its workgroups: (x of gx, utterance of B, 8-channel group of gz), x fastest
    int32_t gx[DOWN_MAXP], gz[DOWN_MAXP];
    AsDownArgs a[DOWN_MAXP];
};
// HEAVY: the instantiation that also holds the 3-row depthwise conv and the stem bodies (200+ registers per lane); a set without them
// runs the light one (the 1-D / channel-preserving steps and the average pools: a quarter of the registers, four times the waves)
template <bool HEAVY>
__global__ void __launch_bounds__(256)
down_multi_kernel(const DownMulti dm)
{
    int pi = 0;
#pragma unroll
    for (int i = 1; i < DOWN_MAXP; ++i) pi += (i < dm.n && (int)blockIdx.x >= dm.blk0[i]) ? 1 : 0;
    const AsDownArgs& a = dm.a[pi];
    const int gxn = dm.gx[pi], local = (int)blockIdx.x - dm.blk0[pi];
    const int bx = local % gxn, rest = local / gxn, b = rest % a.B, g = rest / a.B;
    if (a.kind == 0) {
        if (a.kh == 3) { if constexpr (HEAVY) dwconv_down_body<3>(a, b, g, bx, gxn); }
        else dwconv_down_body<1>(a, b, g, bx, gxn);
    } else if (a.kind == 1) {
        if (a.pool_h == 2) avgpool_down_body<2>(a, b, g, bx, gxn);
        else avgpool_down_body<1>(a, b, g, bx, gxn);
    } else {
        if constexpr (HEAVY) stem_pool_image_body(a, b, g, bx, gxn);
    }
}

static int down_check(const AsDownArgs& a)
{
    if (!a.x || !a.in_off || !a.in_w || !a.out_off || !a.out_w || a.B < 0 || a.C <= 0) return AS_EINVAL;
    if (a.yh && ((reinterpret_cast<uintptr_t>(a.yh) & 15) != 0 || a.n_out < 0)) return AS_EINVAL;
    if (a.kind == 0) return ((!a.y && !a.yh) || !a.w || !a.bias || (a.kh != 1 && a.kh != 3)) ? AS_EINVAL : AS_OK;
    if (a.kind == 1) return ((!a.y && !a.yh) || (a.pool_h != 1 && a.pool_h != 2)) ? AS_EINVAL : AS_OK;
    if (a.kind == 2) return (!a.w || !a.yh || (a.pool_h != 1 && a.pool_h != 2) || (a.kh != 1 && a.kh != 3) || a.Kp <= 0) ? AS_EINVAL : AS_OK;
    return AS_EINVAL;
}

extern "C" int as_down_multi_f32(const AsDownArgs* list_host, int n, as_stream_t stream_)
{
    static_assert(DOWN_MAXP == AS_MAX_MULTI, "header and kernel disagree");
    if (!list_host || n < 1 || n > DOWN_MAXP) return AS_EINVAL;
    DownMulti dm;
    memset(&dm, 0, sizeof(dm));
    bool heavy = false;
    for (int i = 0; i < n; ++i) {
        const AsDownArgs& a = list_host[i];
        const int r = down_check(a);
        if (r != AS_OK) return r;
        if (a.B == 0 || a.max_out <= 0) continue;
        heavy = heavy || a.kind == 2 || (a.kind == 0 && a.kh == 3);
        dm.a[dm.n] = a;
        dm.gx[dm.n] = std::min(32, as_cdiv(a.max_out, 256));
        dm.gz[dm.n] = a.yh ? 2 * as_kbx(a.C) : as_cdiv(a.C, 8);
        dm.blk0[dm.n + 1] = dm.blk0[dm.n] + dm.gx[dm.n] * a.B * dm.gz[dm.n];
        ++dm.n;
    }
    if (dm.n == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream_);
    if (heavy) hipLaunchKernelGGL(down_multi_kernel<true>, dim3(dm.blk0[dm.n]), dim3(256), 0, (hipStream_t)stream_, dm);
    else hipLaunchKernelGGL(down_multi_kernel<false>, dim3(dm.blk0[dm.n]), dim3(256), 0, (hipStream_t)stream_, dm);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

static int down_one(const AsDownArgs& a, hipStream_t stream) { return as_down_multi_f32(&a, 1, stream); }

// im2col for the valid KxK convs that close the 2-D towers (models.py:391,399,535), with the LeakyReLU that
// precedes them (models.py:390,398,534) applied on the fly.  col[(c*K*K + a*K + d)][out_off[b] + ho*Wo + wo]
__global__ void __launch_bounds__(256)
im2col_valid_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ in_off, const int* __restrict__ in_w,
                    int Hin, float* __restrict__ col, int ldc, const int* __restrict__ out_off,
                    const int* __restrict__ out_w, int Hout, int K, int stride, int act, int B)
{
    // one workgroup per row ck = c*K*K + a*K + d of `col`, threads over its packed output columns (the outputs of a
    // tower's last conv are a handful of positions per utterance: a grid over (utterance, row) was 400 k tiny workgroups)
    const int ck = blockIdx.x;
    const int c = ck / (K * K), a = (ck / K) % K, d = ck % K;
    const int total = out_off[B];
    for (int j = threadIdx.x; j < total; j += blockDim.x) {
        int b = 0;
        while (b + 1 < B && out_off[b + 1] <= j) ++b;
        const int i = j - out_off[b], Wo = out_w[b], Wi = in_w[b];
        const int ho = i / Wo, wo = i - ho * Wo;
        float v = x[(size_t)c * ldx + in_off[b] + (size_t)(ho * stride + a) * Wi + wo * stride + d];
        if (act) v = lrelu02(v);
        col[(size_t)ck * ldc + j] = v;
    }
}

extern "C" int as_im2col_valid_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin,
                                   float* col, int ldc, const int32_t* out_off, const int32_t* out_w, int Hout, int K,
                                   int stride, int lrelu, int B, int C, int max_out, as_stream_t stream)
{
    if (!x || !col || !in_off || !in_w || !out_off || !out_w || K <= 0 || stride <= 0 || B < 0 || C <= 0) return AS_EINVAL;
    if (B == 0 || max_out <= 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(im2col_valid_kernel, dim3(C * K * K), dim3(256), 0, (hipStream_t)stream, x, ldx, in_off, in_w, Hin, col,
                       ldc, out_off, out_w, Hout, K, stride, lrelu, B);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// The same written directly as the operand image of the conv it feeds (nothing else reads it): a workgroup takes 8 consecutive
// rows ck of `col`, a thread one packed output column of all eight.
__global__ void __launch_bounds__(256)
im2col_valid_image_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ in_off, const int* __restrict__ in_w,
                          const int* __restrict__ out_off, const int* __restrict__ out_w, int K, int stride, int act, int B, int rows,
                          u32x4_t* __restrict__ yh)
{
    const int g = blockIdx.x;
    const int total = out_off[B];
    const size_t NX = (size_t)total + 1, plane = ((size_t)(g >> 1) * 4 + (g & 1)) * NX;
    if (threadIdx.x < 2) yh[plane + (size_t)threadIdx.x * 2 * NX + total] = u32x4_t{0u, 0u, 0u, 0u};
    for (int j = threadIdx.x; j < total; j += blockDim.x) {
        int b = 0;
        while (b + 1 < B && out_off[b + 1] <= j) ++b;
        const int i = j - out_off[b], Wo = out_w[b], Wi = in_w[b];
        const int ho = i / Wo, wo = i - ho * Wo;
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int ck = g * 8 + r;
            float v = 0.f;
            if (ck < rows) {
                const int c = ck / (K * K), a = (ck / K) % K, d = ck % K;
                v = x[(size_t)c * ldx + in_off[b] + (size_t)(ho * stride + a) * Wi + wo * stride + d];
                if (act) v = lrelu02(v);
            }
            t[r] = v;
        }
        u32x4_t h, l;
        split2(t, h, l);
        yh[plane + j] = h;
        yh[plane + j + 2 * NX] = l;
    }
}

extern "C" int as_im2col_valid_image_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, const int32_t* out_off,
                                         const int32_t* out_w, int K, int stride, int lrelu, int B, int C, uint16_t* yh, as_stream_t stream)
{
    if (!x || !yh || !in_off || !in_w || !out_off || !out_w || K <= 0 || stride <= 0 || B <= 0 || C <= 0 ||
        (reinterpret_cast<uintptr_t>(yh) & 15) != 0)
        return AS_EINVAL;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(im2col_valid_image_kernel, dim3(2 * as_kbx(C * K * K)), dim3(256), 0, (hipStream_t)stream, x, ldx, in_off, in_w, out_off,
                       out_w, K, stride, lrelu, B, C * K * K, reinterpret_cast<u32x4_t*>(yh));
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// y[b][c] = mean_j act(x[c][off[b] + j])     (LeakyReLU + AdaptiveAvgPool, models.py:392-393,400-401,405-406)
__global__ void __launch_bounds__(256)
mean_pool_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ col_off, int B, int C, int act,
                 float* __restrict__ y, int ldy)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= C * B) return;
    const int c = row / B, b = row - c * B;
    const int o0 = col_off[b], L = col_off[b + 1] - o0;
    float s = 0.f;
    for (int i = lane; i < L; i += 64) {
        float v = x[(size_t)c * ldx + o0 + i];
        if (act) v = lrelu02(v);
        s += v;
    }
    s = wave_sum(s);
    if (lane == 0) y[(size_t)b * ldy + c] = s / (float)L;
}

extern "C" int as_mean_pool_f32(const float* x, int ldx, const int32_t* col_off, int B, int C, int lrelu, float* y,
                                int ldy, as_stream_t stream)
{
    if (!x || !y || !col_off || B < 0 || C <= 0 || ldy < C) return AS_EINVAL;
    if (B == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(mean_pool_kernel, dim3(as_cdiv((long)C * B, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, col_off,
                       B, C, lrelu, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Capacity layouts (AsDynGeo, common.h): as_forward_test's second half when the frame counts exist on the device only
// (as_forward_io.frame_cap; models.py:361-368 sizes everything from the PREDICTED durations).  One launch derives every table of the
// half's four layouts from frame_off: blockIdx.z = 0 half rate, 1 mel rate, 2..4 the three groups of the tripled half-rate layout, 5..7
// those of the tripled mel-rate one, 8 = the odd jobs (the frame -> token map's tail, the segments' own offsets); blockIdx.y = utterance.
// Offsets are cut at the capacity (AS_STATUS_CAPACITY is raised): whatever the durations say, nothing is laid out past cap1 columns.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
dyn_geometry_kernel(const AsDynGeo g)
{
    const int z = blockIdx.z, b = blockIdx.y, B = g.B, cap1 = g.cap1;
    const int total_raw = g.frame_off[B];
    if (z == 8) {
        if (b == 0) {                                               // frames past the last one read token 0 (finite filler for the expansion)
            if (g.tof)
                for (int f = max(total_raw, 0) + (int)(blockIdx.x * blockDim.x + threadIdx.x); f < cap1; f += gridDim.x * blockDim.x) g.tof[f] = 0;
            if (blockIdx.x == 0 && threadIdx.x == 0 && total_raw > cap1) as_status_raise(g.status, AS_STATUS_CAPACITY);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {                  // the segment's own offsets (a merged call: every submission gets its own)
            int s = 0;
            while (s + 1 < g.n_seg && b >= g.seg_first[s + 1]) ++s;
            const int first = g.seg_first[s], o_first = min(g.frame_off[first], cap1);
            int32_t* fo = g.seg_frame_off[s];
            if (fo) fo[b - first] = min(g.frame_off[b], cap1) - o_first;
            if (b + 1 == g.seg_first[s + 1]) {
                const int len = min(g.frame_off[b + 1], cap1) - o_first;
                if (fo) fo[b + 1 - first] = len;
                if (len > g.seg_cap[s]) as_status_raise(g.status, AS_STATUS_CAPACITY);   // (its slot of the output holds seg_cap frames)
            }
        }
        return;
    }
    const int li = z == 0 ? 0 : (z == 1 ? 1 : (z < 5 ? 2 : 3));      // layout
    const int grp = z < 2 ? 0 : (z < 5 ? z - 2 : z - 5);
    const int sc = (li == 1 || li == 3) ? 2 : 1;
    const int o = min(g.frame_off[b], cap1), e = min(g.frame_off[b + 1], cap1);
    const int W = sc * (e - o), base = grp * sc * cap1 + sc * o, u = grp * B + b;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (g.w[li]) g.w[li][u] = W;
        if (g.off[li]) {
            g.off[li][u] = base;
            // the closing entry: the end of the last utterance of the last group (AdaIN launches on tripled layouts take widths from w)
            if (b == B - 1 && (li < 2 || grp == 2)) g.off[li][u + 1] = base + W;
        }
        if (li == 2 && g.src3) g.src3[u] = o;
        if (b == 0 && grp == 0 && g.nvalid[li]) *g.nvalid[li] = sc * min(max(total_raw, 0), cap1);
    }
    unsigned long long* meta = g.meta[li];
    if (!meta) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W; i += gridDim.x * blockDim.x)
        meta[base + i] = AS_META_PACK(0ull, (unsigned long long)i, 1ull, (unsigned long long)W);
}

int as_dyn_geometry_launch(const AsDynGeo& g, hipStream_t stream)
{
    if (!g.frame_off || g.B < 1 || g.B > 1024 || g.cap1 < 1 || g.n_seg < 1 || g.n_seg > 16) return AS_EINVAL;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, stream);
    const int gx = std::min(64, std::max(1, as_cdiv(as_cdiv(2L * g.cap1, g.B), 256)));
    hipLaunchKernelGGL(dyn_geometry_kernel, dim3(gx, g.B, 9), dim3(256), 0, stream, g);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// every segment's columns of the packed mel to the segment's own slot (a merged call with predicted durations: a submission finds its
// utterances from the first column of ITS output buffer on, whatever the submissions in front of it came to)
__global__ void __launch_bounds__(256)
seg_scatter_kernel(const AsSegScatter a)
{
    const int s = blockIdx.z, r = blockIdx.y;
    const int first = a.seg_first[s], last = a.seg_first[s + 1];
    const int o = 2 * a.frame_off[first], n = min(2 * (a.frame_off[last] - a.frame_off[first]), 2 * a.seg_cap[s]);
    const float* src = a.src + (size_t)r * a.ld_src + o;
    float* dst = a.dst[s] + (size_t)r * a.ld_dst[s];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

int as_seg_scatter_launch(const AsSegScatter& a, hipStream_t stream)
{
    if (!a.src || !a.frame_off || a.rows < 1 || a.n_seg < 1 || a.n_seg > 16) return AS_EINVAL;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, stream);
    int cap = 1;
    for (int s = 0; s < a.n_seg; ++s) cap = std::max(cap, 2 * a.seg_cap[s]);
    hipLaunchKernelGGL(seg_scatter_kernel, dim3(std::min(64, as_cdiv(cap, 256)), a.rows, a.n_seg), dim3(256), 0, stream, a);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
