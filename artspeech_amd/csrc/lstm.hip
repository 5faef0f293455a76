// K9 -- bidirectional single-layer LSTM recurrence (torch.nn.LSTM semantics: gate order i,f,g,o, both
// biases, reverse direction = forward over the time-flipped sequence), replacing
//   DurationPredictor.LSTM (models.py:526,555-561)  and  ArtsPredictor.{F0,N,EMA}_LSTM (models.py:589-591,606-618).
//
// The input projection W_ih x + b_ih + b_hh for ALL time steps is hoisted out of the recurrence into one
// MFMA GEMM (conv_gemm.hip, written time-major [N][8H] so a step reads contiguous gate rows); this kernel
// is the sequential part only.  Every utterance is its own recurrence (length-aware: the reverse pass
// starts at the utterance's own last frame, never in padding -- the reference's unpacked BiLSTMs get this
// wrong for padded batches), so a workgroup takes (direction, 8 utterances):
// thread = gate row, W_hh^T streams from L2 each step as coalesced rows, h lives in LDS, c in registers.
// Latency-bound by construction; reported in us/step.
#include "common.h"
#include "artspeech_hip.h"
#define AS_FILE_CLS AS_CLS_LSTM

#define NB 8

static __device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(1024)
bilstm_kernel(const float* __restrict__ gx, int ldg, const float* __restrict__ whh_t, const int* __restrict__ col_off,
              int B, int H, float* __restrict__ out, int ldo)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hs = sm;                         // [2][H][NB]
    float* gs = sm + 2 * H * NB;            // [4H][NB]
    __shared__ int s_off[NB], s_len[NB];

    const int dir = blockIdx.y;
    const int u0 = blockIdx.x * NB;
    const int r = threadIdx.x;              // gate row 0..4H-1
    const int G = 4 * H;
    if (r < NB) {
        const int u = u0 + r;
        s_off[r] = u < B ? col_off[u] : 0;
        s_len[r] = u < B ? col_off[u + 1] - col_off[u] : 0;
    }
    for (int i = r; i < 2 * H * NB; i += G) hs[i] = 0.f;
    __syncthreads();
    int Lmax = 0;
#pragma unroll
    for (int u = 0; u < NB; ++u) Lmax = s_len[u] > Lmax ? s_len[u] : Lmax;

    const float* w = whh_t + (size_t)dir * H * G + r;
    const int unit = r % H, q = r / H;      // cell-update role: (unit, utterances q and q+4)
    float c0 = 0.f, c1 = 0.f;

    for (int t = 0; t < Lmax; ++t) {
        const float* hc = hs + (t & 1) * H * NB;
        float* hn = hs + ((t + 1) & 1) * H * NB;
        float acc[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int L = s_len[u];
            const int pos = dir ? L - 1 - t : t;
            acc[u] = (t < L) ? gx[(size_t)(s_off[u] + pos) * ldg + dir * G + r] : 0.f;
        }
        for (int k = 0; k < H; ++k) {
            const float wk = w[(size_t)k * G];
            const float4 h0 = *reinterpret_cast<const float4*>(hc + k * NB);
            const float4 h1 = *reinterpret_cast<const float4*>(hc + k * NB + 4);
            acc[0] += wk * h0.x; acc[1] += wk * h0.y; acc[2] += wk * h0.z; acc[3] += wk * h0.w;
            acc[4] += wk * h1.x; acc[5] += wk * h1.y; acc[6] += wk * h1.z; acc[7] += wk * h1.w;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) gs[r * NB + u] = acc[u];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int u = q + 4 * s;
            const int L = s_len[u];
            if (t < L) {
                const float ig = sigmoidf_(gs[(unit)*NB + u]);
                const float fg = sigmoidf_(gs[(H + unit) * NB + u]);
                const float gg = tanhf(gs[(2 * H + unit) * NB + u]);
                const float og = sigmoidf_(gs[(3 * H + unit) * NB + u]);
                float& c = s ? c1 : c0;
                c = fg * c + ig * gg;
                const float hv = og * tanhf(c);
                hn[unit * NB + u] = hv;
                const int pos = dir ? L - 1 - t : t;
                out[(size_t)(dir * H + unit) * ldo + s_off[u] + pos] = hv;
            }
        }
        __syncthreads();
    }
}

extern "C" int as_bilstm_f32(const float* gx_tm, int ldg, const float* whh_t, const int32_t* col_off, int B, int H,
                             float* out, int ldo, as_stream_t stream)
{
    if (!gx_tm || !whh_t || !col_off || !out || B < 0 || H <= 0 || (4 * H) % 64 || 4 * H > 1024 || ldg < 8 * H) return AS_EINVAL;
    if (B == 0) return AS_OK;
    const size_t smem = sizeof(float) * ((size_t)2 * H * NB + (size_t)4 * H * NB);
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(bilstm_kernel, dim3(as_cdiv(B, NB), 2), dim3(4 * H), smem, (hipStream_t)stream, gx_tm, ldg, whh_t,
                       col_off, B, H, out, ldo);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
