// K9 -- bidirectional single-layer LSTM recurrence (torch.nn.LSTM semantics: gate order i,f,g,o, both
// biases, reverse direction = forward over the time-flipped sequence), replacing
//   DurationPredictor.LSTM (models.py:526,555-561)  and  ArtsPredictor.{F0,N,EMA}_LSTM (models.py:589-591,606-618).
//
// The input projection W_ih x + b_ih + b_hh for ALL time steps is hoisted out of the recurrence into one
// MFMA GEMM (conv_gemm.hip, written time-major [N][8H] so a step reads contiguous gate rows); this kernel
// is the sequential part only.  Every utterance is its own recurrence (length-aware: the reverse pass
// starts at the utterance's own last frame, never in padding -- the reference's unpacked BiLSTMs get this
// wrong for padded batches), so a workgroup takes (job, direction, NB = 2 utterances) and
// several independent LSTMs ("jobs": the F0 / energy / TV branches) share one launch so that their
// latency-bound recurrences overlap on different CUs.
// thread = gate row.  H <= 128: the thread's row of W_hh stays in REGISTERS for the whole sequence
// (persistent weights, 128 VGPRs); larger H streams W_hh^T rows from L2 with 8 loads in flight.
// h lives in LDS (read as one broadcast vector per k), c in registers, next step's input gates are
// prefetched during the current step.  Latency-bound by construction; reported in us/step.
#include "common.h"
#include "artspeech_hip.h"
#define AS_FILE_CLS AS_CLS_LSTM

#define NB 2

// exp and reciprocal via the hardware v_exp_f32 / v_rcp_f32 (1 ulp; ~1e-7 abs on the bounded gate outputs, the test bound is 2e-5).
// Not __frcp_rn: the correctly rounded reciprocal is a ten-instruction division sequence, five of them on every step's critical path.
static __device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
static __device__ __forceinline__ float tanhf_(float x)
{
    const float e = __expf(-2.0f * fabsf(x));              // in (0, 1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return x < 0.f ? -t : t;
}

// the value of another lane of the quad (DPP quad_perm: 0xB1 = lane ^ 1, 0x4E = lane ^ 2, 0x00 / 0x55 / 0xAA / 0xFF = lane 0 / 1 / 2 / 3)
template <int CTRL>
static __device__ __forceinline__ float quad_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

struct LstmJobs { BiLstmJob j[AS_MAX_LSTM_JOBS]; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait every step
// for the step's global store of h to be acknowledged and for the prefetch of the next step's input gates -- the
// recurrence needs neither (global data is never re-read by the workgroup).
static __device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// HREG = H when the weights are register resident (H in {16,32,64,128}), 0 for the streaming variant.
template <int HREG>
__global__ void __launch_bounds__(HREG ? 4 * HREG : 1024)
bilstm_kernel(const LstmJobs jobs, const int* __restrict__ col_off, int B, int Hrt)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = HREG ? HREG : Hrt;
    float* hs = sm;                         // [2][H][NB]
    float* gs = sm + 2 * H * NB;            // [4H][NB]
    __shared__ int s_off[NB], s_len[NB];

    const BiLstmJob job = jobs.j[blockIdx.z];
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

    const float* w = job.whh_t + (size_t)dir * H * G + r;
    float wreg[HREG ? HREG : 1];
    if (HREG) {
#pragma unroll
        for (int k = 0; k < HREG; ++k) wreg[k] = w[(size_t)k * G];
    }
    const int unit = r % H, q = r / H;      // cell-update role: (unit, utterance q)
    float c = 0.f;

    auto load_gx = [&](int t, float (&g)[NB]) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int L = s_len[u];
            const int pos = dir ? L - 1 - t : t;
            g[u] = (t < L) ? job.gx_tm[(size_t)(s_off[u] + pos) * job.ldg + dir * G + r] : 0.f;
        }
    };
    float gnext[NB];
    load_gx(0, gnext);

    for (int t = 0; t < Lmax; ++t) {
        const float* hc = hs + (t & 1) * H * NB;
        float* hn = hs + ((t + 1) & 1) * H * NB;
        float acc[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) acc[u] = gnext[u];
        if (t + 1 < Lmax) load_gx(t + 1, gnext);
        if (HREG) {
#pragma unroll
            for (int k = 0; k < HREG; ++k) {
                const float2 h2 = *reinterpret_cast<const float2*>(hc + k * NB);
                acc[0] += wreg[k] * h2.x; acc[1] += wreg[k] * h2.y;
            }
        } else {
            for (int k0 = 0; k0 < H; k0 += 8) {
                float wk[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wk[i] = w[(size_t)(k0 + i) * G];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float2 h2 = *reinterpret_cast<const float2*>(hc + (k0 + i) * NB);
                    acc[0] += wk[i] * h2.x; acc[1] += wk[i] * h2.y;
                }
            }
        }
        *reinterpret_cast<float2*>(gs + r * NB) = make_float2(acc[0], acc[1]);
        lds_barrier();
        if (q < NB) {                        // the first H*NB threads update one (unit, utterance) cell each
            const int u = q;
            const int L = s_len[u];
            if (t < L) {
                const float ig = sigmoidf_(gs[(unit)*NB + u]);
                const float fg = sigmoidf_(gs[(H + unit) * NB + u]);
                const float gg = tanhf_(gs[(2 * H + unit) * NB + u]);
                const float og = sigmoidf_(gs[(3 * H + unit) * NB + u]);
                c = fg * c + ig * gg;
                const float hv = og * tanhf_(c);
                hn[unit * NB + u] = hv;
                const int pos = dir ? L - 1 - t : t;
                job.out[(size_t)(dir * H + unit) * job.ldo + s_off[u] + pos] = hv;
            }
        }
        lds_barrier();
    }
}

// H = 256 (duration predictor, models.py:526; JDCNet's classifier, Utils/JDC/model.py:62-64): a direction's W_hh is 1 MB, twice
// the CU's register file, and streaming ALL of it from L2 every step (the kernel above: 11 us per step, latency-bound with 8
// loads in flight per thread) is the whole step time.  Here 512 threads own TWO gate rows each (rows t and t + 2H), keep the
// first KREG entries of both in registers, the next KLDS rows of W_hh^T live in LDS (144 KB), and the remaining rows are
// streamed per step, double-buffered.  Swept on MI355X (JDCNet, 200 steps): KREG 32 / 48 / 64 -> 1.98 / 2.36 / 3.1 ms against
// 2.33 for the kernel above -- hipcc spills beyond ~32 resident entries per row -- so this is a 15 % step, not the fix; the fix is
// W_hh split over several workgroups (DESIGN.md "next").  The accumulation order over k is unchanged (ascending).
template <int H, int KREG, int KLDS>
__global__ void __launch_bounds__(2 * H)
bilstm_big_kernel(const LstmJobs jobs, const int* __restrict__ col_off, int B)
{
    #ifndef LSTM_SB
#define LSTM_SB 4
#endif
    constexpr int G = 4 * H, NT = 2 * H, KS = H - KREG - KLDS, SB = LSTM_SB;   // SB: streamed rows per batch
    static_assert(KS >= 0 && KS % SB == 0, "split of the k range");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hs = sm;                         // [2][H][NB]
    float* gs = hs + 2 * H * NB;            // [4H][NB]
    float* wl = gs + G * NB;                // [KLDS][G]
    __shared__ int s_off[NB], s_len[NB];

    const BiLstmJob job = jobs.j[blockIdx.z];
    const int dir = blockIdx.y;
    const int u0 = blockIdx.x * NB;
    const int tid = threadIdx.x;            // gate rows tid and tid + NT
    if (tid < NB) {
        const int u = u0 + tid;
        s_off[tid] = u < B ? col_off[u] : 0;
        s_len[tid] = u < B ? col_off[u + 1] - col_off[u] : 0;
    }
    for (int i = tid; i < 2 * H * NB; i += NT) hs[i] = 0.f;
    const float* w = job.whh_t + (size_t)dir * H * G + tid;
    float wa[KREG], wb[KREG];
#pragma unroll
    for (int k = 0; k < KREG; ++k) {
        wa[k] = w[(size_t)k * G];
        wb[k] = w[(size_t)k * G + NT];
    }
#pragma unroll 4
    for (int k = 0; k < KLDS; ++k) {
        wl[k * G + tid] = w[(size_t)(KREG + k) * G];
        wl[k * G + NT + tid] = w[(size_t)(KREG + k) * G + NT];
    }
    const float* ws = w + (size_t)(KREG + KLDS) * G;
    __syncthreads();
    int Lmax = 0;
#pragma unroll
    for (int u = 0; u < NB; ++u) Lmax = s_len[u] > Lmax ? s_len[u] : Lmax;
    const int unit = tid % H, q = tid / H;  // cell-update role: (unit, utterance q); NT = H * NB threads
    float c = 0.f;

    auto load_gx = [&](int t, float (&ga)[NB], float (&gb)[NB]) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int L = s_len[u];
            const int pos = dir ? L - 1 - t : t;
            const float* p = job.gx_tm + (size_t)(s_off[u] + pos) * job.ldg + dir * G + tid;
            ga[u] = (t < L) ? p[0] : 0.f;
            gb[u] = (t < L) ? p[NT] : 0.f;
        }
    };
    float gna[NB], gnb[NB];
    load_gx(0, gna, gnb);

    for (int t = 0; t < Lmax; ++t) {
        asm volatile("" ::: "memory");      // the LDS-resident and streamed weights are loop invariant: without this hipcc hoists them into (spilled) registers
        const float* hc = hs + (t & 1) * H * NB;
        float* hn = hs + ((t + 1) & 1) * H * NB;
        float a0 = gna[0], a1 = gna[1], b0 = gnb[0], b1 = gnb[1];
        if (t + 1 < Lmax) load_gx(t + 1, gna, gnb);
        float ka[SB], kb[SB];                                  // the first streamed rows start moving before the resident part runs
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            ka[i] = ws[(size_t)i * G];
            kb[i] = ws[(size_t)i * G + NT];
        }
#pragma unroll
        for (int k = 0; k < KREG; ++k) {
            const float2 h2 = *reinterpret_cast<const float2*>(hc + k * NB);
            a0 += wa[k] * h2.x; a1 += wa[k] * h2.y;
            b0 += wb[k] * h2.x; b1 += wb[k] * h2.y;
            if ((k & 7) == 7) asm volatile("" ::: "memory");   // keeps hipcc from hoisting all 96 h reads up front (it spilled ~1000 registers)
        }
#pragma unroll 4
        for (int k = 0; k < KLDS; ++k) {
            const float va = wl[k * G + tid], vb = wl[k * G + NT + tid];
            const float2 h2 = *reinterpret_cast<const float2*>(hc + (KREG + k) * NB);
            a0 += va * h2.x; a1 += va * h2.y;
            b0 += vb * h2.x; b1 += vb * h2.y;
        }
        for (int k0 = 0; k0 < KS; k0 += SB) {
            float na[SB], nb[SB];
            const int kn = k0 + SB < KS ? k0 + SB : k0;       // (the last batch re-loads itself: branch-free)
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                na[i] = ws[(size_t)(kn + i) * G];
                nb[i] = ws[(size_t)(kn + i) * G + NT];
            }
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const float2 h2 = *reinterpret_cast<const float2*>(hc + (KREG + KLDS + k0 + i) * NB);
                a0 += ka[i] * h2.x; a1 += ka[i] * h2.y;
                b0 += kb[i] * h2.x; b1 += kb[i] * h2.y;
            }
#pragma unroll
            for (int i = 0; i < SB; ++i) { ka[i] = na[i]; kb[i] = nb[i]; }
        }
        *reinterpret_cast<float2*>(gs + tid * NB) = make_float2(a0, a1);
        *reinterpret_cast<float2*>(gs + (NT + tid) * NB) = make_float2(b0, b1);
        lds_barrier();
        {                                    // every thread updates one (unit, utterance) cell
            const int u = q;
            const int L = s_len[u];
            if (t < L) {
                const float ig = sigmoidf_(gs[(unit)*NB + u]);
                const float fg = sigmoidf_(gs[(H + unit) * NB + u]);
                const float gg = tanhf_(gs[(2 * H + unit) * NB + u]);
                const float og = sigmoidf_(gs[(3 * H + unit) * NB + u]);
                c = fg * c + ig * gg;
                const float hv = og * tanhf_(c);
                hn[unit * NB + u] = hv;
                const int pos = dir ? L - 1 - t : t;
                job.out[(size_t)(dir * H + unit) * job.ldo + s_off[u] + pos] = hv;
            }
        }
        lds_barrier();
    }
}

// H <= 128: thread = (hidden unit, K quarter).  A thread keeps the W_hh entries of ALL FOUR gates of its unit for its
// quarter of the hidden vector in registers (4 * H/4 = H VGPRs), accumulates 8 independent chains (4 gates x 2
// utterances) -- instruction-level parallelism instead of one 128-long dependent chain -- and the four quarters are
// summed inside the lane quad with two DPP shuffles, so the gate pre-activations of a unit meet in registers:
// no LDS round trip for them and ONE barrier per step (after the new h is published).
template <int H>
__global__ void __launch_bounds__(4 * H)
bilstm_quad_kernel(const LstmJobs jobs, const int* __restrict__ col_off, int B)
{
    constexpr int KQ = H / 4, G = 4 * H;
    __shared__ __attribute__((aligned(16))) float hs[2][H][NB];
    __shared__ int s_off[NB], s_len[NB];
    const BiLstmJob job = jobs.j[blockIdx.z];
    const int dir = blockIdx.y, u0 = blockIdx.x * NB;
    const int tid = threadIdx.x, unit = tid >> 2, kq = tid & 3;
    if (tid < NB) {
        const int u = u0 + tid;
        s_off[tid] = u < B ? col_off[u] : 0;
        s_len[tid] = u < B ? col_off[u + 1] - col_off[u] : 0;
    }
    for (int i = tid; i < 2 * H * NB; i += G) (&hs[0][0][0])[i] = 0.f;
    __syncthreads();
    const int L0 = s_len[0], L1 = s_len[1], o0 = s_off[0], o1 = s_off[1];
    const int Lmax = L0 > L1 ? L0 : L1;

    float w[4][KQ];
    const float* wp = job.whh_t + (size_t)dir * H * G + unit;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) w[g][kk] = wp[(size_t)(kq * KQ + kk) * G + g * H];

    // lane kq adds the input-projection term of gate kq (one coalesced-ish load per utterance per step)
    const float* gxp = job.gx_tm + dir * G + kq * H + unit;
    auto load_gx = [&](int t, float& g0, float& g1) {
        g0 = (t < L0) ? gxp[(size_t)(o0 + (dir ? L0 - 1 - t : t)) * job.ldg] : 0.f;
        g1 = (t < L1) ? gxp[(size_t)(o1 + (dir ? L1 - 1 - t : t)) * job.ldg] : 0.f;
    };
    // input-gate terms are prefetched PF steps ahead (a step is ~1 us, a miss beyond L2 is longer)
    constexpr int PF = 4;
    float gq0[PF], gq1[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_gx(j, gq0[j], gq1[j]);
    float c = 0.f;
    const int my_u = kq & 1, my_L = my_u ? L1 : L0, my_o = my_u ? o1 : o0;

    for (int t0 = 0; t0 < Lmax; t0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int t = t0 + j;
        if (t >= Lmax) break;
        const float(*hc)[NB] = hs[t & 1];
        float(*hn)[NB] = hs[(t + 1) & 1];
        float acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[g][0] = (g == kq) ? gq0[j] : 0.f;
            acc[g][1] = (g == kq) ? gq1[j] : 0.f;
        }
        load_gx(t + PF, gq0[j], gq1[j]);
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const float2 h2 = *reinterpret_cast<const float2*>(&hc[kq * KQ + kk][0]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g][0] += w[g][kk] * h2.x;
                acc[g][1] += w[g][kk] * h2.y;
            }
        }
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float a0 = acc[g][0], a1 = acc[g][1];
            a0 += __shfl_xor(a0, 1); a1 += __shfl_xor(a1, 1);
            a0 += __shfl_xor(a0, 2); a1 += __shfl_xor(a1, 2);
            pre[g] = my_u ? a1 : a0;
        }
        if (kq < 2 && t < my_L) {                        // lanes 0 / 1 of the quad own utterances 0 / 1
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
            c = fg * c + ig * gg;
            const float hv = og * tanhf_(c);
            hn[unit][my_u] = hv;
            job.out[(size_t)(dir * H + unit) * job.ldo + my_o + (dir ? my_L - 1 - t : t)] = hv;
        }
        lds_barrier();
      }
    }
}

// The same with ONE utterance per workgroup: half the multiply-adds and LDS reads per step, twice the workgroups -- used while
// they all fit the chip at once (the recurrence is latency-bound: a step is as long as one workgroup's dependent chain).
// The four gates of a unit are accumulated as two packed pairs (v_pk_fma_f32) in two independent chains (even / odd k).
typedef float lf2 __attribute__((ext_vector_type(2)));
template <int H>
__global__ void __launch_bounds__(4 * H)
bilstm_quad1_kernel(const LstmJobs jobs, const int* __restrict__ col_off, int B)
{
    constexpr int KQ = H / 4, G = 4 * H;
    static_assert(KQ % 4 == 0, "quarter of the hidden vector in float4 steps");
    __shared__ __attribute__((aligned(16))) float hs[2][H];
    const BiLstmJob job = jobs.j[blockIdx.z];
    const int dir = blockIdx.y, u = blockIdx.x;
    const int tid = threadIdx.x, unit = tid >> 2, kq = tid & 3;
    const int o0 = col_off[u], L = col_off[u + 1] - o0;
    if (L <= 0) return;
    for (int i = tid; i < 2 * H; i += G) (&hs[0][0])[i] = 0.f;
    __syncthreads();

    lf2 wif[KQ], wgo[KQ];                                                // (i, f) and (g, o) rows of this unit, this k quarter
    const float* wp = job.whh_t + (size_t)dir * H * G + unit;
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
        const float* r = wp + (size_t)(kq * KQ + kk) * G;
        wif[kk] = lf2{r[0], r[H]};
        wgo[kk] = lf2{r[2 * H], r[3 * H]};
    }
    // lane kq adds the input-projection term of gate kq
    const float* gxp = job.gx_tm + dir * G + kq * H + unit;
    auto load_gx = [&](int t) { return (t < L) ? gxp[(size_t)(o0 + (dir ? L - 1 - t : t)) * job.ldg] : 0.f; };
    constexpr int PF = 4;
    float gq[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) gq[j] = load_gx(j);
    float c = 0.f;
    float* outp = job.out + (size_t)(dir * H + unit) * job.ldo + o0;

    for (int t0 = 0; t0 < L; t0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int t = t0 + j;
            if (t >= L) break;
            const float* hc = hs[t & 1] + kq * KQ;
            float* hn = hs[(t + 1) & 1];
            const float gx = gq[j];
            gq[j] = load_gx(t + PF);
            lf2 aif0 = {0.f, 0.f}, ago0 = {0.f, 0.f}, aif1 = {0.f, 0.f}, ago1 = {0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < KQ; k4 += 4) {
                const float4 h4 = *reinterpret_cast<const float4*>(hc + k4);
                aif0 += wif[k4] * lf2{h4.x, h4.x};
                ago0 += wgo[k4] * lf2{h4.x, h4.x};
                aif1 += wif[k4 + 1] * lf2{h4.y, h4.y};
                ago1 += wgo[k4 + 1] * lf2{h4.y, h4.y};
                aif0 += wif[k4 + 2] * lf2{h4.z, h4.z};
                ago0 += wgo[k4 + 2] * lf2{h4.z, h4.z};
                aif1 += wif[k4 + 3] * lf2{h4.w, h4.w};
                ago1 += wgo[k4 + 3] * lf2{h4.w, h4.w};
            }
            float pre[4] = {aif0[0] + aif1[0], aif0[1] + aif1[1], ago0[0] + ago1[0], ago0[1] + ago1[1]};
            // the four quarters meet inside the lane quad through DPP (quad_perm, no LDS round trip: __shfl_xor is a ds_bpermute); then lane
            // kq activates gate kq -- one exp / rcp chain per lane instead of four on one lane of the quad with the other three masked off
            // (a step is issue-bound: eight waves of ~250 instructions on four SIMDs before this, ~150 after) -- and the activated gates
            // are handed round the quad again.  tanh(x) = 2 sigmoid(2x) - 1: the same code for all four gates.
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pre[g] += quad_dpp<0xB1>(pre[g]);          // lane ^ 1
                pre[g] += quad_dpp<0x4E>(pre[g]);          // lane ^ 2
            }
            const float sc = kq == 2 ? 2.f : 1.f;
            const float x = (kq == 0 ? pre[0] : kq == 1 ? pre[1] : kq == 2 ? pre[2] : pre[3]) + gx;   // lane kq holds gate kq's input term
            const float act = fmaf(__builtin_amdgcn_rcpf(1.0f + __expf(-sc * x)), sc, 1.0f - sc);
            const float ig = quad_dpp<0x00>(act), fg = quad_dpp<0x55>(act), gg = quad_dpp<0xAA>(act), og = quad_dpp<0xFF>(act);
            c = fg * c + ig * gg;                          // (every lane of the quad keeps the unit's cell state)
            const float hv = og * fmaf(__builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * c)), 2.0f, -1.0f);
            if (kq == 0) {
                hn[unit] = hv;
                outp[dir ? L - 1 - t : t] = hv;
            }
            lds_barrier();
        }
    }
}

extern "C" int as_bilstm_f32(const BiLstmJob* jobs_host, int n_jobs, const int32_t* col_off, int B, int H,
                             as_stream_t stream)
{
    if (!jobs_host || n_jobs <= 0 || n_jobs > AS_MAX_LSTM_JOBS || !col_off || B < 0 || H <= 0 || (4 * H) % 64 ||
        4 * H > 1024 || H % 8)
        return AS_EINVAL;
    LstmJobs jobs;
    for (int i = 0; i < AS_MAX_LSTM_JOBS; ++i) jobs.j[i] = jobs_host[i < n_jobs ? i : 0];
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs.j[i].gx_tm || !jobs.j[i].whh_t || !jobs.j[i].out || jobs.j[i].ldg < 8 * H) return AS_EINVAL;
    if (B == 0) return AS_OK;
    const size_t smem = sizeof(float) * ((size_t)2 * H * NB + (size_t)4 * H * NB);
    const dim3 grid(as_cdiv(B, NB), 2, n_jobs), block(4 * H);
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    // one utterance per workgroup while all the recurrences fit the chip together (AS_LSTM_NB=2 forces the pairs: experiments)
    const bool single = (H == 64 || H == 128) && (long)n_jobs * 2 * B <= 512 && !getenv("AS_LSTM_NB");
    if (single) {
        const dim3 g1(B, 2, n_jobs);
        if (H == 64) hipLaunchKernelGGL(bilstm_quad1_kernel<64>, g1, block, 0, (hipStream_t)stream, jobs, col_off, B);
        else hipLaunchKernelGGL(bilstm_quad1_kernel<128>, g1, block, 0, (hipStream_t)stream, jobs, col_off, B);
        AS_CHECK_LAUNCH();
        return AS_OK;
    }
    switch (H) {
    case 16: hipLaunchKernelGGL(bilstm_quad_kernel<16>, grid, block, 0, (hipStream_t)stream, jobs, col_off, B); break;
    case 32: hipLaunchKernelGGL(bilstm_quad_kernel<32>, grid, block, 0, (hipStream_t)stream, jobs, col_off, B); break;
    case 64: hipLaunchKernelGGL(bilstm_quad_kernel<64>, grid, block, 0, (hipStream_t)stream, jobs, col_off, B); break;
    case 128: hipLaunchKernelGGL(bilstm_quad_kernel<128>, grid, block, 0, (hipStream_t)stream, jobs, col_off, B); break;
    case 256: {
#ifndef LSTM_KREG
#define LSTM_KREG 32
#endif
        constexpr int KREG = LSTM_KREG, KLDS = 36;                    
        const size_t sm_big = smem + sizeof(float) * (size_t)KLDS * 1024;
        AS_LDS_OPT_IN((bilstm_big_kernel<256, KREG, KLDS>), (int)sm_big);   // (+ 16 bytes static: 160 KB would be refused)
        if (getenv("AS_LSTM_STREAM")) hipLaunchKernelGGL(bilstm_kernel<0>, grid, block, smem, (hipStream_t)stream, jobs, col_off, B, H);
        else hipLaunchKernelGGL((bilstm_big_kernel<256, KREG, KLDS>), grid, dim3(512), sm_big, (hipStream_t)stream, jobs, col_off, B);
        break;
    }
    default: hipLaunchKernelGGL(bilstm_kernel<0>, grid, block, smem, (hipStream_t)stream, jobs, col_off, B, H); break;
    }
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ---------------------------------------------------------------------------------------------------
// H = 256 across a CLUSTER of four workgroups (the duration predictor, models.py:526; 1024 sequential steps in the long-form
// configuration).  A direction's W_hh is 1 MB -- more than a CU's registers and LDS together -- so the single-workgroup kernel above
// re-streams ~750 KB of it from L2 every step (9.6 us per step, bound by the CU's L2 port).  Here workgroup `member` of a cluster owns
// 64 hidden units: all four gate rows of those units, 256 KB, register resident for the whole sequence (thread = (unit, k eighth),
// 128 weight registers as (i,f) / (g,o) pairs for v_pk_fma_f32).  Each step every member publishes its 64 new h values and reads the
// other 192 through a small exchange buffer in global memory: one 8-byte word per value, {h, tag}, tag = launch epoch and step, written
// and polled with agent-scope atomics, so a word validates itself (no separate flag, no fence) and a step costs ONE store -> load
// round trip through L2.  The four members sit on one XCD (linear workgroup ids 8 apart), i.e. behind one L2.
//   words: slot (t & 1) holds h after step t; a member overwrites slot (t+1) & 1 only after it has seen every member's step-t words,
//   which they publish after consuming step t-1's -- two slots suffice.
//   epoch: region[0] of the cluster, read by every member at start, bumped by member 0 at the end: words of earlier launches (the
//   buffer persists, hipGraph replays repeat the same arguments) never match.
// Forward progress: members spin, so a cluster needs its four workgroups resident together.  The launcher only uses this kernel
// when the whole grid fits the chip at once; beside other kernels (the path's side streams, a second plan's graph in flight) it rests
// on IN-ORDER DISPATCH of a grid's workgroups: a cluster's members have ids inside one window of 32 consecutive ids, so of the
// workgroups a launch has resident all but the last, partial cluster are complete, finish on their own and free their CUs for the
// next ids; two launches that share the chip cannot both be left with nothing but partial clusters.  HIP does not promise that
// order, so every poll is bounded (~1 s) and a poll that gives up RAISES AS_STATUS_LSTM_TIMEOUT (as_device_status): the launch still
// ends, its output is void, and the module entry points return AS_EDEVICE until the status is cleared.
// ---------------------------------------------------------------------------------------------------
#define CL_P 4
#define CL_REGION_WORDS 1032                       /* 1 header word + 2 slots x 256 units x 2 utterances, padded to 8 KB + 64 B */
static __device__ __forceinline__ float dpp_add8(float v)
{
    // sum over 8 consecutive lanes: two quad permutes, then the mirror inside each half row
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));
    return v;
}

template <int NBU>
__global__ void __launch_bounds__(512)
bilstm_cluster_kernel(const LstmJobs jobs, const int* __restrict__ col_off, int B, int n_clusters, unsigned long long* __restrict__ xchg,
                      unsigned* __restrict__ status, int spin_limit, int drop_member)
{
    constexpr int H = 256, G = 4 * H, UW = H / CL_P, KS = 8, KQ = H / KS, SL = KQ * NBU + 4, HB = KS * SL;
    __shared__ __attribute__((aligned(16))) float hs[2 * HB];
    const int lin = blockIdx.x;
    const int cl = (lin & 7) + 8 * ((lin >> 3) / CL_P), member = (lin >> 3) % CL_P;
    if (cl >= n_clusters) return;
    const int groups = (B + NBU - 1) / NBU;
    const int ug = cl % groups, dir = (cl / groups) & 1;
    const BiLstmJob job = jobs.j[cl / (2 * groups)];
    const int tid = threadIdx.x, ul = tid >> 3, kq = tid & 7, unit = member * UW + ul;
    int off[NBU], len[NBU], Lmax = 0;
#pragma unroll
    for (int i = 0; i < NBU; ++i) {
        const int u = ug * NBU + i;
        off[i] = u < B ? col_off[u] : 0;
        len[i] = u < B ? col_off[u + 1] - col_off[u] : 0;
        Lmax = len[i] > Lmax ? len[i] : Lmax;
    }
    if (Lmax <= 0) return;                                               // (every member of the cluster agrees)
    if (member == drop_member) return;                                   // test hook (as_bilstm_cluster_test_hooks): this member never shows up
    unsigned long long* region = xchg + (size_t)cl * CL_REGION_WORDS;
    const unsigned epoch = (unsigned)region[0] & 0x7FFFu;
    const unsigned tagbase = epoch << 17;                                 // tag = epoch : step + 1 (17 bits: the launcher bounds the lengths)
    unsigned long long* words = region + 1;                              // [2][H][NBU]
    for (int i = tid; i < 2 * HB; i += 512) hs[i] = 0.f;

    lf2 wif[KQ], wgo[KQ];
    const float* wp = job.whh_t + (size_t)dir * H * G + unit;
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
        const float* r = wp + (size_t)(kq * KQ + kk) * G;
        wif[kk] = lf2{r[0], r[H]};
        wgo[kk] = lf2{r[2 * H], r[3 * H]};
    }
    // lane kq carries the input-projection term of gate kq & 3 of utterance kq >> 2
    const int gu = kq >> 2, gL = gu < NBU ? len[gu < NBU ? gu : 0] : 0, gO = off[gu < NBU ? gu : 0];
    const float* gxp = job.gx_tm + dir * G + (kq & 3) * H + unit;
    auto load_gx = [&](int t) { return (t < gL) ? gxp[(size_t)(gO + (dir ? gL - 1 - t : t)) * job.ldg] : 0.f; };
    constexpr int PF = 4;
    float gq[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) gq[j] = load_gx(j);
    float c = 0.f;
    const int my_u = kq < NBU ? kq : 0, my_L = len[my_u], my_o = off[my_u];
    float* outp = job.out + (size_t)(dir * H + unit) * job.ldo + my_o;
    // the word this thread polls: value k = tid / NBU of utterance tid % NBU, stored where the (k / KQ) slice readers expect it
    const int pk = tid / NBU, pu = tid % NBU, ppos = (pk / KQ) * SL + (pk % KQ) * NBU + pu;
    __syncthreads();

    for (int t0 = 0; t0 < Lmax; t0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int t = t0 + j;
            if (t >= Lmax) break;
            float* hb = hs + (t & 1) * HB;
            if (t > 0 && tid < H * NBU) {
                const unsigned long long* w = words + (size_t)((t - 1) & 1) * H * NBU + tid;
                const unsigned want = tagbase | (unsigned)t;
                unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; (unsigned)(v >> 32) != want && spin < spin_limit; ++spin) {
                    __builtin_amdgcn_s_sleep(1);
                    v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // gave up: a peer never published this step (it got no CU, or died).  The word read last is NOT h(t-1): the recurrence
                // carries on so that the launch ends, and says that its output is void (as_device_status; the module entry points
                // return AS_EDEVICE from then on)
                if ((unsigned)(v >> 32) != want) as_status_raise(status, AS_STATUS_LSTM_TIMEOUT);
                hb[ppos] = __uint_as_float((unsigned)v);
            }
            lds_barrier();
            const float gx = gq[j];
            gq[j] = load_gx(t + PF);
            lf2 aif[NBU], ago[NBU];
#pragma unroll
            for (int u = 0; u < NBU; ++u) {
                aif[u] = lf2{kq == 4 * u ? gx : 0.f, kq == 4 * u + 1 ? gx : 0.f};
                ago[u] = lf2{kq == 4 * u + 2 ? gx : 0.f, kq == 4 * u + 3 ? gx : 0.f};
            }
            const float* hc = hb + kq * SL;
            if (NBU == 2) {
#pragma unroll
                for (int kk = 0; kk < KQ; kk += 2) {
                    const float4 h4 = *reinterpret_cast<const float4*>(hc + kk * 2);
                    aif[0] += wif[kk] * lf2{h4.x, h4.x};
                    ago[0] += wgo[kk] * lf2{h4.x, h4.x};
                    aif[NBU - 1] += wif[kk] * lf2{h4.y, h4.y};
                    ago[NBU - 1] += wgo[kk] * lf2{h4.y, h4.y};
                    aif[0] += wif[kk + 1] * lf2{h4.z, h4.z};
                    ago[0] += wgo[kk + 1] * lf2{h4.z, h4.z};
                    aif[NBU - 1] += wif[kk + 1] * lf2{h4.w, h4.w};
                    ago[NBU - 1] += wgo[kk + 1] * lf2{h4.w, h4.w};
                }
            } else {
                lf2 bif = {0.f, 0.f}, bgo = {0.f, 0.f};                  // a second chain: one utterance has only two accumulators
#pragma unroll
                for (int kk = 0; kk < KQ; kk += 4) {
                    const float4 h4 = *reinterpret_cast<const float4*>(hc + kk);
                    aif[0] += wif[kk] * lf2{h4.x, h4.x};
                    ago[0] += wgo[kk] * lf2{h4.x, h4.x};
                    bif += wif[kk + 1] * lf2{h4.y, h4.y};
                    bgo += wgo[kk + 1] * lf2{h4.y, h4.y};
                    aif[0] += wif[kk + 2] * lf2{h4.z, h4.z};
                    ago[0] += wgo[kk + 2] * lf2{h4.z, h4.z};
                    bif += wif[kk + 3] * lf2{h4.w, h4.w};
                    bgo += wgo[kk + 3] * lf2{h4.w, h4.w};
                }
                aif[0] += bif;
                ago[0] += bgo;
            }
            float pre[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NBU; ++u) {
                const float pi = dpp_add8(aif[u][0]), pf = dpp_add8(aif[u][1]), pg = dpp_add8(ago[u][0]), po = dpp_add8(ago[u][1]);
                if (u == my_u) { pre[0] = pi; pre[1] = pf; pre[2] = pg; pre[3] = po; }
            }
            if (kq < NBU) {                                              // lane u of the unit's eight owns utterance u
                float hv = 0.f;
                if (t < my_L) {
                    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
                    c = fg * c + ig * gg;
                    hv = og * tanhf_(c);
                    outp[dir ? my_L - 1 - t : t] = hv;
                }
                const unsigned long long word = ((unsigned long long)(tagbase | (unsigned)(t + 1)) << 32) | __float_as_uint(hv);
                __hip_atomic_store(words + (size_t)(t & 1) * H * NBU + unit * NBU + kq, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // The epoch is bumped by member 0 once it is done.  With two or more steps it cannot get here before every member has read the
    // epoch (it consumed their step Lmax - 2 words, which carry it).  With ONE step nothing was polled: wait for the others' step-0
    // words first -- a member that started late would otherwise read the bumped epoch and leave step-0 words that the NEXT launch's
    // step-1 poll takes for its own.
    if (member == 0) {
        if (Lmax == 1 && tid < H * NBU) {
            const unsigned long long* w = words + tid;
            const unsigned want = tagbase | 1u;
            unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int spin = 0; (unsigned)(v >> 32) != want && spin < spin_limit; ++spin) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if ((unsigned)(v >> 32) != want) as_status_raise(status, AS_STATUS_LSTM_TIMEOUT);
        }
        __syncthreads();
        if (tid == 0) region[0] = epoch + 1;
    }
}

// test hooks (tests/test_status_gpu.py): member `drop_member` of every cluster exits at once (-1 = none) and a poll gives up after
// `spin_limit` tries (0 = the default, ~1 s), so that the failure path runs in milliseconds
static int g_cl_drop_member = -1, g_cl_spin_limit = 1 << 21;
extern "C" int as_bilstm_cluster_test_hooks(int drop_member, int spin_limit)
{
    if (drop_member < -1 || drop_member >= CL_P || spin_limit < 0) return AS_EINVAL;
    g_cl_drop_member = drop_member;
    g_cl_spin_limit = spin_limit > 0 ? spin_limit : 1 << 21;
    return AS_OK;
}

extern "C" size_t as_bilstm_cluster_bytes(int n_jobs, int B)
{
    if (n_jobs <= 0 || B <= 0) return 0;
    return (size_t)n_jobs * 2 * B * CL_REGION_WORDS * sizeof(unsigned long long);
}

extern "C" int as_bilstm_cluster_f32(const BiLstmJob* jobs_host, int n_jobs, const int32_t* col_off, int B, int H, int max_len, void* xchg,
                                     size_t xchg_bytes, as_stream_t stream)
{
    if (!jobs_host || n_jobs <= 0 || n_jobs > AS_MAX_LSTM_JOBS || !col_off || B < 0 || H <= 0) return AS_EINVAL;
    if (B == 0) return AS_OK;
    // one utterance per cluster while every cluster fits the chip at once, else two; beyond that (or without an exchange buffer, or
    // another width, or lengths past the tag's step field) the single-workgroup kernels
    const long c1 = (long)n_jobs * 2 * B, c2 = (long)n_jobs * 2 * ((B + 1) / 2);
    const int nbu = c1 * CL_P <= 256 ? 1 : (c2 * CL_P <= 256 ? 2 : 0);
    const char* mode = getenv("AS_LSTM_CLUSTER");                        // experiments: "0" = never, "1" / "2" = utterances per cluster
    const int use = mode ? atoi(mode) : nbu;
    if (H != 256 || !xchg || use <= 0 || use > 2 || max_len <= 0 || max_len >= (1 << 17) || (use == 1 ? c1 : c2) * CL_P > 256 ||
        xchg_bytes < as_bilstm_cluster_bytes(n_jobs, B))
        return as_bilstm_f32(jobs_host, n_jobs, col_off, B, H, stream);
    LstmJobs jobs;
    for (int i = 0; i < AS_MAX_LSTM_JOBS; ++i) jobs.j[i] = jobs_host[i < n_jobs ? i : 0];
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs.j[i].gx_tm || !jobs.j[i].whh_t || !jobs.j[i].out || jobs.j[i].ldg < 8 * H) return AS_EINVAL;
    const int ncl = (int)(use == 1 ? c1 : c2);
    const dim3 grid(8 * CL_P * as_cdiv(ncl, 8)), block(512);
    unsigned* status = as_status_words_device();
    AsProfScope prof__(AS_FILE_CLS, 0, 0, (hipStream_t)stream);
    if (use == 1)
        hipLaunchKernelGGL(bilstm_cluster_kernel<1>, grid, block, 0, (hipStream_t)stream, jobs, col_off, B, ncl,
                           static_cast<unsigned long long*>(xchg), status, g_cl_spin_limit, g_cl_drop_member);
    else
        hipLaunchKernelGGL(bilstm_cluster_kernel<2>, grid, block, 0, (hipStream_t)stream, jobs, col_off, B, ncl,
                           static_cast<unsigned long long*>(xchg), status, g_cl_spin_limit, g_cl_drop_member);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

