// wav -> log-mel front end of test.py:40-47: framing with centre / reflect padding, and the
// element-wise steps around the two GEMMs (windowed DFT basis; mel filterbank) that do the arithmetic.
#include "common.h"
#include "artspeech_hip.h"
#define AS_FILE_CLS AS_CLS_OTHER

// X[k][frame_off[b] + n] = wave_b[reflect(n * hop + k - n_fft / 2)]: the frames torch.stft(center=True, pad_mode="reflect") transforms
__global__ void frame_signal_kernel(const float* __restrict__ wave, const int* __restrict__ wav_off, const int* __restrict__ frame_off,
                                    int n_fft, int hop, float* __restrict__ X, int ldx)
{
    const int b = blockIdx.z, k = blockIdx.y;
    const int w0 = wav_off[b], L = wav_off[b + 1] - w0;
    const int f0 = frame_off[b], F = frame_off[b + 1] - f0;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < F; n += gridDim.x * blockDim.x) {
        int i = n * hop + k - n_fft / 2;
        i = i < 0 ? -i : i;
        i = i >= L ? 2 * (L - 1) - i : i;
        X[(size_t)k * ldx + f0 + n] = (i >= 0 && i < L) ? wave[w0 + i] : 0.f;
    }
}

extern "C" int as_frame_signal_f32(const float* wave, const int32_t* wav_off, const int32_t* frame_off, int B, int max_frames, int n_fft,
                                   int hop, float* X, int ldx, as_stream_t stream)
{
    if (!wave || !wav_off || !frame_off || !X || B < 0 || n_fft <= 0 || hop <= 0) return AS_EINVAL;
    if (B == 0 || max_frames <= 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    int gx = as_cdiv(max_frames, 256);
    gx = gx > 16 ? 16 : gx;
    hipLaunchKernelGGL(frame_signal_kernel, dim3(gx, n_fft, B), dim3(256), 0, (hipStream_t)stream, wave, wav_off, frame_off, n_fft, hop, X, ldx);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// P[f][n] = re[f][n]^2 + im[f][n]^2 from the stacked DFT output Y [2 F][N] (rows 0..F-1 real, F..2F-1 imaginary)
__global__ void spec_power_kernel(const float* __restrict__ Y, int ldy, int F, int N, float* __restrict__ P, int ldp)
{
    const int f = blockIdx.y;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const float re = Y[(size_t)f * ldy + n], im = Y[(size_t)(F + f) * ldy + n];
        P[(size_t)f * ldp + n] = re * re + im * im;
    }
}

extern "C" int as_spec_power_f32(const float* Y, int ldy, int F, int N, float* P, int ldp, as_stream_t stream)
{
    if (!Y || !P || F <= 0 || N < 0 || ldy < N || ldp < N) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 12.0 * F * (double)N, (hipStream_t)stream);
    int gx = as_cdiv(N, 256);
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(spec_power_kernel, dim3(gx, F), dim3(256), 0, (hipStream_t)stream, Y, ldy, F, N, P, ldp);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// y = (log(eps + x) - mean) / std          test.py:46
__global__ void log_norm_kernel(const float* __restrict__ x, int ldx, int N, float eps, float mean, float std, float* __restrict__ y, int ldy)
{
    const int c = blockIdx.y;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x)
        y[(size_t)c * ldy + n] = (logf(eps + x[(size_t)c * ldx + n]) - mean) / std;
}

extern "C" int as_log_norm_f32(const float* x, int ldx, int C, int N, float eps, float mean, float std, float* y, int ldy, as_stream_t stream)
{
    if (!x || !y || C <= 0 || N < 0 || ldx < N || ldy < N || std == 0.f) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_FILE_CLS, 0, 8.0 * C * (double)N, (hipStream_t)stream);
    int gx = as_cdiv(N, 256);
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(log_norm_kernel, dim3(gx, C), dim3(256), 0, (hipStream_t)stream, x, ldx, N, eps, mean, std, y, ldy);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
