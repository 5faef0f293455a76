// Glue kernels of the HiFi-GAN generator (Vocoder/vocoder.py:75-125) -- SURVEY.md section 8(f) row N2; its
// convolutions are the conv GEMM of the acoustic path (conv_gemm*.hip).
#include "conv_gemm.h"
#define AS_FILE_CLS AS_CLS_OTHER

// ConvTranspose1d(k = 2u, stride u) runs as ONE 3-tap conv whose output rows are (phase r, channel m) (vocoder.py of
// this package builds the stacked weight); this kernel interleaves the phases into time order and adds the bias:
//   y[m][u*q + r] = z[r*C + m][q] + bias[m]
__global__ void interleave_phases_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ bias, int C, int u,
                                         int Nin, float* __restrict__ y, int ldy)
{
    const int m = blockIdx.y;
    const float b = bias ? bias[m] : 0.f;
    const long total = (long)Nin * u;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (long)gridDim.x * blockDim.x) {
        const int q = (int)(j / u), r = (int)(j - (long)q * u);
        y[(size_t)m * ldy + j] = z[(size_t)(r * C + m) * ldz + q] + b;
    }
}

extern "C" int as_interleave_phases_f32(const float* z, int ldz, const float* bias, int C, int u, int Nin, float* y, int ldy,
                                        as_stream_t stream)
{
    if (!z || !y || C <= 0 || u <= 0 || Nin < 0 || ldz < Nin || (long)ldy < (long)Nin * u) return AS_EINVAL;
    if (Nin == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    long blocks = ((long)Nin * u + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(interleave_phases_kernel, dim3((unsigned)blocks, C), dim3(256), 0, (hipStream_t)stream, z, ldz, bias, C, u,
                       Nin, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// y = (a + b + c) / 3: the average of the three residual stacks of a stage (vocoder.py:104-110)
__global__ void mean3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int ld, int N,
                             float* __restrict__ y, int ldy)
{
    const int m = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        const size_t i = (size_t)m * ld + j;
        y[(size_t)m * ldy + j] = ((a[i] + b[i]) + c[i]) / 3.0f;
    }
}

extern "C" int as_mean3_f32(const float* a, const float* b, const float* c, int ld, int C, int N, float* y, int ldy,
                            as_stream_t stream)
{
    if (!a || !b || !c || !y || C <= 0 || N < 0 || ld < N || ldy < N) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    int blocks = (N + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mean3_kernel, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a, b, c, ld, N, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// conv_post (vocoder.py:97, 111-113): wav = tanh(conv1d(LeakyReLU(x, 0.01), w [1][C][k]) + b), zero padding per utterance.  ONE output row:
// as a conv GEMM launch this was a 32-row matrix-core tile computing one useful row behind a split pass over the whole input (336 us at
// 32 channels x 1.92 M samples); as plain fp32 FMAs it is a read of x (245 MB).  A wave owns 256 consecutive columns, a lane the columns
// lane + 64 p: every load of a wave is one contiguous 256-byte run, and the k-fold re-read of a column comes from the CU's cache.
template <int K>
__global__ void __launch_bounds__(256) conv_post_kernel(const float* __restrict__ x, int ldx, int C, int N, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float slope, int tanh_out,
                                                        const unsigned long long* __restrict__ meta, float* __restrict__ y)
{
    constexpr int HALF = K / 2;
    const int j0 = (blockIdx.x * 256 + (threadIdx.x & ~63)) * 4 + (threadIdx.x & 63);
    if ((blockIdx.x * 256 + (threadIdx.x & ~63)) * 4 >= N) return;      // (the whole wave)
    // taps that stay inside the column's own utterance
    unsigned ok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int j = j0 + 64 * p;
        ok[p] = 0;
        if (j < N) {
            const unsigned long long md = meta[j];
            const int wj = AS_META_w(md), Wj = AS_META_W(md);
#pragma unroll
            for (int t = 0; t < K; ++t) ok[p] |= ((unsigned)(wj + t - HALF) < (unsigned)Wj ? 1u : 0u) << t;
        }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
        const float* xr = x + (size_t)c * ldx;
        float wt[K];
#pragma unroll
        for (int t = 0; t < K; ++t) wt[t] = w[c * K + t];               // (uniform: scalar loads)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int j = j0 + 64 * p;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const bool in = (ok[p] >> t) & 1u;
                float v = in ? xr[j + t - HALF] : 0.f;
                v = v > 0.f ? v : slope * v;
                acc[p] = __builtin_fmaf(wt[t], v, acc[p]);
            }
        }
    }
    const float b = bias ? bias[0] : 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int j = j0 + 64 * p;
        if (j < N) {
            const float v = acc[p] + b;
            y[j] = tanh_out ? tanhf(v) : v;
        }
    }
}

extern "C" int as_conv_post_f32(const float* x, int ldx, int C, int N, const float* w, const float* bias, int k, float in_slope,
                                int tanh_out, const uint64_t* meta, float* y, as_stream_t stream)
{
    if (!x || !w || !meta || !y || C <= 0 || N < 0 || ldx < N || (k != 3 && k != 5 && k != 7)) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_CLS_GEMM, 2.0 * C * k * (double)N, 4.0 * (C + 1.0) * N, (hipStream_t)stream, "conv_post");
    const dim3 grid(as_cdiv(N, 1024)), block(256);
    const unsigned long long* md = reinterpret_cast<const unsigned long long*>(meta);
    if (k == 3) hipLaunchKernelGGL(conv_post_kernel<3>, grid, block, 0, (hipStream_t)stream, x, ldx, C, N, w, bias, in_slope, tanh_out, md, y);
    else if (k == 5) hipLaunchKernelGGL(conv_post_kernel<5>, grid, block, 0, (hipStream_t)stream, x, ldx, C, N, w, bias, in_slope, tanh_out, md, y);
    else hipLaunchKernelGGL(conv_post_kernel<7>, grid, block, 0, (hipStream_t)stream, x, ldx, C, N, w, bias, in_slope, tanh_out, md, y);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// LeakyReLU((a + b + c) / 3) as the operand image of the conv that follows (the next stage's ConvTranspose1d, vocoder.py:101-110): the
// stage's mean is read by nothing else, so its fp32 copy and the split pass over it need not exist.  Thread geometry of split_f16x2_kernel.
__global__ void __launch_bounds__(256)
mean3_image_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int ld, int K, int N, float slope,
                   u32x4_t* __restrict__ xh)
{
    const int wcol = (blockIdx.x * 256 + (threadIdx.x & ~63)) * 4;      // the wave's first column
    const int col = wcol + (threadIdx.x & 63);
    const int g = blockIdx.y;                                           // 8-row group: kb = g / 2, kh = g % 2
    if (wcol > N) return;
    const int bytes = (int)(((unsigned)(K - 1) * ld + N) * 4u);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c), 0, bytes, 0x00020000);
    const size_t NX = (size_t)N + 1;
    const size_t base = ((size_t)(g >> 1) * 4 + (g & 1)) * NX + col;    // plane p*2 + kh of k-block kb
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        if (col + 64 * cc > N) break;                                   // (column N itself is written: the zero column)
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = g * 8 + r;
            const unsigned off = (k < K && col + 64 * cc < N) ? (unsigned)(k * ld + col + 64 * cc) * 4u : OOB;
            const float e = ((buf_load1(ra, off, 0) + buf_load1(rb, off, 0)) + buf_load1(rc, off, 0)) / 3.0f;
            t[r] = e > 0.f ? e : slope * e;
        }
        u32x4_t h, l;
        split2(t, h, l);
        xh[base + 64 * cc] = h;
        xh[base + 64 * cc + 2 * NX] = l;
    }
}

extern "C" int as_mean3_image_f32(const float* a, const float* b, const float* c, int ld, int C, int N, float slope, uint16_t* xh,
                                  as_stream_t stream)
{
    if (!a || !b || !c || !xh || C <= 0 || N < 0 || ld < N || (reinterpret_cast<uintptr_t>(xh) & 15) != 0) return AS_EINVAL;
    if ((double)C * ld * 4.0 >= 2147483648.0) return AS_EINVAL;         // 32-bit offsets in the buffer descriptors
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 16.0 * C * (double)N, (hipStream_t)stream);
    hipLaunchKernelGGL(mean3_image_kernel, dim3(as_cdiv(N + 1, 1024), 2 * as_kbx(C)), dim3(256), 0, (hipStream_t)stream, a, b, c, ld, C, N,
                       slope, reinterpret_cast<u32x4_t*>(xh));
    AS_CHECK_LAUNCH();
    return AS_OK;
}
