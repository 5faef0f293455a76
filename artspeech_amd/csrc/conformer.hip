// Kernels of the EMA_Predictor (second half): the conformer blocks of
// Utils/EMA/conformer/conformer (attention.py, convolution.py) and the predictor's own glue (EMA_Predictor.py:65-82).
// The dense layers run through the conv GEMM; this file holds what is not a GEMM.
#include "common.h"
#include "artspeech_hip.h"
#define AS_FILE_CLS AS_CLS_ATTN

// ---------------------------------------------------------------------------------------------------
// Multi-head self-attention with Transformer-XL relative positions AS THE REFERENCE COMPUTES THEM
// (attention.py:77-109).  For an utterance of T frames, p = pos_proj(PE[0..T-1]) and
//   content[i][j] = (q_i + u) . k_j
//   P[i][c]       = (q_i + v) . p_c                       (c = 0 .. T-1)
//   pos           = _relative_shift(P): pad a zero column in front, view the [T][T+1] block as [T+1][T], drop row 0:
//                   flat index (i+1) T + j of the padded block.  With e = T-1-i+j that is
//                     e <  T : P[i][e]              = (q_i     + v) . p_e           (keys at or before the query)
//                     e == T : 0                                                     (j = i+1)
//                     e >  T : P[i+1][e-T-1]        = (q_{i+1} + v) . p_{e-T-1}     (the rows the shift wraps into)
//   score = (content + pos) / sqrt(d_model);  softmax over the utterance's own keys;  out = softmax . v
// Nothing of size T^2 is stored: scores are computed per 64-key tile with an online softmax (as K5 does).
// Workgroup = (utterance, head, 16-query tile), 4 waves x 4 queries, d_head = 64.
// ---------------------------------------------------------------------------------------------------
#define XQT 16
#define XKT 64
#define XPAD 65
#define XDK 64
#define XPW (XKT + XQT)            // window of p rows a (query tile, key tile) pair touches: 79 <= 80
#define XPP (XPW + 1)

static __device__ __forceinline__ float xwmax(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
static __device__ __forceinline__ float xwsum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256)
xl_attention_kernel(const float* __restrict__ qkv, int ld, int C, const float* __restrict__ pos, int ldp,
                    const float* __restrict__ u_bias, const float* __restrict__ v_bias, float inv_scale,
                    const int* __restrict__ col_off, float* __restrict__ out, int ldo)
{
    extern __shared__ __attribute__((aligned(16))) float xsm[];
    float* Ks = xsm;                              // [d][key]          (reused for the output transpose)
    float* Vs = Ks + XDK * XPAD;                  // [d][key]
    float* Pe = Vs + XDK * XPAD;                  // [d][x]: p row of e = e_min + x (zero where e == T or out of range)
    float* Qu = Pe + XDK * XPP;                   // q_i + u
    float* Qv = Qu + XQT * XDK;                   // q_i + v, rows i0 .. i0 + 16
    float* Ps = Qv + (XQT + 1) * XDK;             // [wave][key][4 queries]

    const int b = blockIdx.z, h = blockIdx.y;
    const int o0 = col_off[b], T = col_off[b + 1] - o0;
    const int q0 = blockIdx.x * XQT;
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const float* Qg = qkv + (size_t)(h * XDK) * ld + o0;
    const float* Kg = qkv + (size_t)(C + h * XDK) * ld + o0;
    const float* Vg = qkv + (size_t)(2 * C + h * XDK) * ld + o0;
    const float* Pg = pos + (size_t)(h * XDK) * ldp + o0;

    for (int i = tid; i < (XQT + 1) * XDK; i += 256) {
        const int q = i % (XQT + 1), d = i / (XQT + 1);
        const int qi = q0 + q;
        const float qv = qi < T ? Qg[(size_t)d * ld + qi] : 0.f;
        Qv[q * XDK + d] = qi < T ? qv + v_bias[h * XDK + d] : 0.f;
        if (q < XQT) Qu[q * XDK + d] = qi < T ? qv + u_bias[h * XDK + d] : 0.f;
    }

    float m[4], l[4], acc[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) { m[qq] = -INFINITY; l[qq] = 0.f; acc[qq] = 0.f; }
    float* Pw = Ps + wave * XKT * 4;

    for (int k0 = 0; k0 < T; k0 += XKT) {
        __syncthreads();                                    // previous tile fully consumed (and Qu/Qv visible)
        for (int i = tid; i < XDK * XKT; i += 256) {
            const int jj = i % XKT, d = i / XKT;
            const int kj = k0 + jj;
            const bool ok = kj < T;
            Ks[d * XPAD + jj] = ok ? Kg[(size_t)d * ld + kj] : 0.f;
            Vs[d * XPAD + jj] = ok ? Vg[(size_t)d * ld + kj] : 0.f;
        }
        const int e_min = T - 1 - (q0 + XQT - 1) + k0;      // e of (last query of the tile, first key of the tile)
        for (int i = tid; i < XDK * XPW; i += 256) {
            const int x = i % XPW, d = i / XPW;
            const int e = e_min + x;
            const int r = e < T ? e : e - T - 1;            // e == T -> -1: the zero entry
            Pe[d * XPP + x] = (e >= 0 && r >= 0 && r < T) ? Pg[(size_t)d * ldp + r] : 0.f;
        }
        __syncthreads();
        // ---- scores: lane = key
        const int kj = k0 + lane;
        // content and position sums; the position term of query q uses q's own vector for keys at or before it (e < T) and the
        // NEXT query's for the wrapped entries (e > T): both sums run (five broadcast rows serve the wave's four queries), the
        // lane picks one at the end.  Q rows are read four d at a time (one ds_read_b128 per row and four d).
        float sc[4], sa[4], sb[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { sc[qq] = 0.f; sa[qq] = 0.f; sb[qq] = 0.f; }
#pragma unroll 2
        for (int d = 0; d < XDK; d += 4) {
            float4 qu[4], qv[5];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) qu[qq] = *reinterpret_cast<const float4*>(Qu + (wave * 4 + qq) * XDK + d);
#pragma unroll
            for (int qq = 0; qq < 5; ++qq) qv[qq] = *reinterpret_cast<const float4*>(Qv + (wave * 4 + qq) * XDK + d);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const float kv = Ks[(d + dd) * XPAD + lane];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float pe = Pe[(d + dd) * XPP + (XQT - 1 - (wave * 4 + qq)) + lane];
                    const float u = dd == 0 ? qu[qq].x : dd == 1 ? qu[qq].y : dd == 2 ? qu[qq].z : qu[qq].w;
                    const float va = dd == 0 ? qv[qq].x : dd == 1 ? qv[qq].y : dd == 2 ? qv[qq].z : qv[qq].w;
                    const float vb = dd == 0 ? qv[qq + 1].x : dd == 1 ? qv[qq + 1].y : dd == 2 ? qv[qq + 1].z : qv[qq + 1].w;
                    sc[qq] += u * kv;
                    sa[qq] += va * pe;
                    sb[qq] += vb * pe;
                }
            }
        }
        float sp[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) sp[qq] = (T - 1 - (q0 + wave * 4 + qq) + kj) > T ? sb[qq] : sa[qq];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float s = (sc[qq] + sp[qq]) * inv_scale;
            if (kj >= T) s = -INFINITY;
            const float mn = fmaxf(m[qq], xwmax(s));
            const float p = (kj < T) ? expf(s - mn) : 0.f;
            const float corr = expf(m[qq] - mn);           // exp(-inf) = 0 on the first tile
            l[qq] = l[qq] * corr + xwsum(p);
            acc[qq] *= corr;
            m[qq] = mn;
            Pw[lane * 4 + qq] = p;
        }
        __syncthreads();
        // ---- PV: lane = channel
        const int jn = (T - k0) < XKT ? (T - k0) : XKT;
        for (int jj = 0; jj < jn; ++jj) {
            const float4 p4 = *reinterpret_cast<const float4*>(Pw + jj * 4);
            const float v0 = Vs[lane * XPAD + jj];
            acc[0] += p4.x * v0; acc[1] += p4.y * v0; acc[2] += p4.z * v0; acc[3] += p4.w * v0;
        }
    }
    __syncthreads();
    float* Os = Ks;                                         // [d][XQT]
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) Os[lane * XQT + wave * 4 + qq] = acc[qq] / l[qq];
    __syncthreads();
    for (int i = tid; i < XDK * XQT; i += 256) {
        const int q = i % XQT, d = i / XQT;
        if (q0 + q < T) out[(size_t)(h * XDK + d) * ldo + o0 + q0 + q] = Os[d * XQT + q];
    }
}

extern "C" int as_xl_attention_f32(const float* qkv, int ld, int C, int heads, const float* pos, int ldp, const float* u_bias,
                                   const float* v_bias, float inv_scale, const int32_t* col_off, int B, int max_len, float* out,
                                   int ldo, as_stream_t stream)
{
    if (!qkv || !pos || !u_bias || !v_bias || !col_off || !out || C <= 0 || heads <= 0 || C != heads * XDK || B < 0) return AS_EINVAL;
    if (B == 0 || max_len <= 0) return AS_OK;
    const size_t smem = sizeof(float) * ((size_t)2 * XDK * XPAD + XDK * XPP + XQT * XDK + (XQT + 1) * XDK + 4 * XKT * 4);
    AS_LDS_OPT_IN(xl_attention_kernel, 160 * 1024);       // > 64 KiB of dynamic LDS needs an explicit opt-in
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(xl_attention_kernel, dim3(as_cdiv(max_len, XQT), heads, B), dim3(256), smem, (hipStream_t)stream, qkv, ld, C,
                       pos, ldp, u_bias, v_bias, inv_scale, col_off, out, ldo);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Conformer convolution module, the part between its two pointwise convs (convolution.py:137-146):
//   g = a[0:C] * sigmoid(a[C:2C])                     GLU over channels
//   y = swish(BatchNorm(depthwise conv1d(g, k taps, 'same' zero padding inside the utterance)))
// a [2C][N] is the first pointwise conv's output; BatchNorm (eval) arrives as scale/shift.  Workgroup = (channel, utterance):
// the gated row is staged in LDS in chunks (each element's sigmoid is computed once, not k times).
// ---------------------------------------------------------------------------------------------------
#define DWK_MAX 63
#define DW_CHUNK 1024
__global__ void __launch_bounds__(256)
glu_dwconv_bn_swish_kernel(const float* __restrict__ a, int lda, int C, const float* __restrict__ w, int k,
                           const float* __restrict__ scale, const float* __restrict__ shift, const int* __restrict__ col_off,
                           float* __restrict__ y, int ldy)
{
    __shared__ float g[DW_CHUNK + DWK_MAX - 1];
    __shared__ float ws[DWK_MAX];
    const int c = blockIdx.x, b = blockIdx.y;
    const int o0 = col_off[b], T = col_off[b + 1] - o0;
    const int half = k / 2;
    const float* a0 = a + (size_t)c * lda + o0;
    const float* a1 = a + (size_t)(C + c) * lda + o0;
    if (threadIdx.x < k) ws[threadIdx.x] = w[c * k + threadIdx.x];
    const float sc = scale[c], sh = shift[c];
    for (int t0 = 0; t0 < T; t0 += DW_CHUNK) {
        __syncthreads();
        const int n = (T - t0) < DW_CHUNK ? (T - t0) : DW_CHUNK;
        for (int i = threadIdx.x; i < n + 2 * half; i += 256) {
            const int t = t0 + i - half;
            float v = 0.f;
            if (t >= 0 && t < T) v = a0[t] * (1.0f / (1.0f + expf(-a1[t])));
            g[i] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            float s = 0.f;
            for (int j = 0; j < k; ++j) s += ws[j] * g[i + j];
            s = s * sc + sh;
            y[(size_t)c * ldy + o0 + t0 + i] = s * (1.0f / (1.0f + expf(-s)));
        }
    }
}

extern "C" int as_glu_dwconv_bn_swish_f32(const float* a, int lda, int C, const float* w, int k, const float* scale,
                                          const float* shift, const int32_t* col_off, int B, float* y, int ldy, as_stream_t stream)
{
    if (!a || !w || !scale || !shift || !col_off || !y || C <= 0 || k <= 0 || k > DWK_MAX || (k & 1) == 0 || B < 0) return AS_EINVAL;
    if (B == 0) return AS_OK;
    AsProfScope prof__(AS_CLS_OTHER, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(glu_dwconv_bn_swish_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, a, lda, C, w, k, scale, shift,
                       col_off, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// EMA_Predictor.decoder2 (EMA_Predictor.py:43, :79): an nn.LSTM built WITHOUT batch_first and fed [1, T, 256], i.e. a
// "sequence" of length 1 with the T frames as its batch: every frame takes ONE step from the zero state,
//   c = sigmoid(i) * tanh(g),  h = sigmoid(o) * tanh(c)        (the forget gate multiplies c0 = 0; W_hh multiplies h0 = 0)
// in both directions.  gx [2 * 4H][N] = W_ih x + b_ih + b_hh (gate order i, f, g, o; forward rows first) -> h [2H][N].
// ---------------------------------------------------------------------------------------------------
__global__ void lstm_step0_kernel(const float* __restrict__ gx, int ldg, int H, int N, float* __restrict__ h, int ldh)
{
    const int row = blockIdx.y;                              // dir * H + unit
    const int dir = row / H, u = row - dir * H;
    const float* gi = gx + (size_t)(dir * 4 * H + u) * ldg;
    const float* gg = gx + (size_t)(dir * 4 * H + 2 * H + u) * ldg;
    const float* go = gx + (size_t)(dir * 4 * H + 3 * H + u) * ldg;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        const float c = (1.0f / (1.0f + expf(-gi[j]))) * tanhf(gg[j]);
        h[(size_t)row * ldh + j] = (1.0f / (1.0f + expf(-go[j]))) * tanhf(c);
    }
}

extern "C" int as_lstm_step0_f32(const float* gx, int ldg, int H, int N, float* h, int ldh, as_stream_t stream)
{
    if (!gx || !h || H <= 0 || N < 0 || ldg < N || ldh < N) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_CLS_LSTM, 0, 0, (hipStream_t)stream);
    int gx_blocks = as_cdiv(N, 256);
    gx_blocks = gx_blocks > 64 ? 64 : gx_blocks;
    hipLaunchKernelGGL(lstm_step0_kernel, dim3(gx_blocks, 2 * H), dim3(256), 0, (hipStream_t)stream, gx, ldg, H, N, h, ldh);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
