// Glue kernels of the HiFi-GAN generator (Vocoder/vocoder.py:75-125) -- SURVEY.md section 8(f) row N2; its
// convolutions are the conv GEMM of the acoustic path (conv_gemm*.hip).
#include "common.h"
#include "artspeech_hip.h"
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
