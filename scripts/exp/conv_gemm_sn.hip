// K3/K4/K8/K10 at SMALL N -- the conv GEMM when the batch is one utterance (BASELINE config C2: N = 30 tokens / 150 frames), or any
// launch whose weight set sees at most 256 columns.  Replaces the same reference sites as conv_gemm_h3.hip (models.py:176-202,497-517,
// RelTransformerEnc.py:261-269, ...) with the same f16x3 arithmetic; what differs is what bounds it.
//
// With 150 columns a layer is a stream of WEIGHTS: 1024 x 1024 x 3 taps is 12.6 MB read once for 0.9 GFLOP, i.e. 2.5 us of HBM at full
// rate and 2 us of matrix-core time -- and a couple of microseconds of memory latency on top, which is what the launch really costs.
// The tiled kernel is built the other way round (tiles of output, both operands through a three-stage LDS ring: at most 64 KB in flight
// per CU, one workgroup per CU): 12 us per layer at batch 1 plus 6 us for its split-K reduction launch, 59 of those per utterance.
// Here:
//   * a workgroup = 128 output rows (4 waves x 32) x ALL columns of its weight set x a slice of (tap, k-block) iterations;
//   * the ACTIVATIONS of the slice (per iteration 4 planes x NgP columns x 16 bytes: small, L2 resident, shared by the four waves) are
//     staged by LDS-DMA up to twelve iterations at a time (a round: most slices are one round) -- a tap shifts a lane's source column,
//     an invalid tap reads the image's zero column, exactly as in the tiled kernel -- and stay for the round: no ring, no barrier
//     inside it;
//   * the WEIGHTS never touch LDS: a lane's fragment is 16 contiguous bytes of the weight image (its layout IS the fragment's), so a
//     wave loads its (h, l) fragments straight into a ring of SN_D register sets, SN_D iterations ahead: every byte of the slice's
//     weights is requested within the first microsecond of the launch;
//   * K is split over workgroups (slices) and the slices are summed INSIDE the launch: fp32 partial slabs, an arrival counter per
//     (row group, weight set), then every slice of the group reduces its own share of the columns in the fixed order s = 0 .. S-1
//     (deterministic) through the very epilogue code the split-K reduction launch runs (as_reduce_epilogue).  The hand-off is the
//     placement-independent form: write-through (sc1) slab stores -> every wave's vmcnt(0) -> workgroup barrier -> one relaxed agent-scope
//     atomic add; one relaxed poll -> agent-scope acquire -> barrier -> plain loads.  The counter resets itself (arrivals count to
//     S, departures to 2 S, the last one stores 0), so a hipGraph replay finds it as the first launch did.
// The S slices of a group wait for each other: the launcher keeps the grid within what is resident at once, the ids of a group are
// consecutive (in-order dispatch leaves at most one group incomplete), every spin is bounded and a give-up raises AS_STATUS_* like the
// clustered LSTM's (common.h).
#include "conv_gemm.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

typedef __attribute__((address_space(3))) void sn_lds_void;

#define SN_D 6                  // weight fragments in flight per wave: iterations ahead
#define SN_MAX_IT 12            // iterations per slice: 4 LDS-DMAs each, all outstanding at once beside the first weight loads (vmcnt is 6 bits)
#define SN_LDS_MAX (144 * 1024)
#define SN_MAX_ROUNDS 4         // rounds of staged activations per slice (a round = up to SN_MAX_IT iterations; two barriers each)

template <int I, int N, typename F>
static __device__ __forceinline__ void sn_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sn_for<I + 1, N>(f);
    }
}

// f(0); if (1 < n) { f(1); if (2 < n) { ... } }: straight-line code with early exits
template <int I, int N, typename F>
static __device__ __forceinline__ void sn_chain(int n, F&& f)
{
    if constexpr (I < N) {
        if (I < n) {
            f(std::integral_constant<int, I>{});
            sn_chain<I + 1, N>(n, f);
        }
    }
}

template <int TN, int NP>
__global__ void __launch_bounds__(256)
conv_gemm_sn_kernel(const ConvGemmArgs a, const H3Taps tp, int S, int slots, unsigned* __restrict__ sync)
{
    constexpr int NgP = ((TN + 1) / 2) * 64;                             // staged columns: whole waves of lanes
    constexpr int SLOT = NgP * 64;                                       // bytes per iteration: 4 planes x NgP columns x 16
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lk = lane >> 5;
    const int RG = (a.M + 127) >> 7, G = a.n_groups > 1 ? a.n_groups : 1;
    const int s = blockIdx.x % S, rgg = blockIdx.x / S, rg = rgg % RG, grp = rgg / RG;
    (void)G;
    const int m0 = rg * 128;
    const int n0 = a.n_groups > 1 ? grp * a.group_cols : 0;
    const int n_end = a.n_groups > 1 ? min(a.N, n0 + a.group_cols) : a.N;
    const int KB = a.Kp >> 4, KBx = (KB + 3) & ~3, NX = a.N + 1;
    const int IT = a.T * KB;
    const int it_lo = (int)((long)IT * s / S), n_it = (int)((long)IT * (s + 1) / S) - it_lo;

    const unsigned w_bytes = (unsigned)a.T * KBx * 4u * a.M * 16u;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.Wh) + (size_t)grp * w_bytes), 0, (int)w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.Xh), 0, (int)((unsigned)KBx * 4u * NX * 16u), 0x00020000);
    (void)rsX;

    // ---- weights: fragment (part p) of iteration (t, kb) for this lane = 16 bytes at plane p*2 + lk, row m0 + 32 wave + l31
    const int row = m0 + 32 * wave + l31;
    const unsigned w_lane = row < a.M ? (unsigned)((lk * a.M + row) * 16) : OOB;      // + ((t KBx + kb) 4 + 2 p) M 16
    int w_t = it_lo / KB, w_kb = it_lo - w_t * KB;                       // cursor of the next fragment to request
    f16x8 wa[SN_D][NP == 1 ? 1 : 2];
    // (always issued -- past the slice's end with an out-of-range offset, which costs no memory traffic: a load behind a branch would
    // leave the compiler unable to count the loads in flight, and it would drain them all at every use)
    auto w_load = [&](auto u_, bool valid) {
        constexpr int u = decltype(u_)::value;
        const int base = ((w_t * KBx + w_kb) * 4) * a.M * 16;
        const unsigned vo = valid ? w_lane : OOB;
#ifdef SN_EXP_NOW
        const unsigned vo2 = OOB; (void)vo;
#define vo vo2
#endif
        wa[u][0] = __builtin_bit_cast(f16x8, buf_load4(rsW, vo, base));
        if constexpr (NP != 1) wa[u][1] = __builtin_bit_cast(f16x8, buf_load4(rsW, vo, base + 2 * a.M * 16));
#ifdef SN_EXP_NOW
#undef vo
#endif
        if (++w_kb == KB) { w_kb = 0; ++w_t; }
    };
    sn_for<0, SN_D>([&](auto u_) { w_load(u_, decltype(u_)::value < n_it); });

    // ---- activations -> LDS, `slots` iterations per round (thread = one staged column, all four planes of an iteration)
    const int j = n0 + tid;
    unsigned tapmask = 0;
    int Wj = 0;
    if (tid < NgP && j < n_end) {
        if (a.meta) {
            const unsigned long long md = a.meta[j];
            const int h = (int)(md & 0xffff), w = (int)((md >> 16) & 0xffff), H = (int)((md >> 32) & 0xffff);
            Wj = (int)(md >> 48);
            for (int t = 0; t < a.T; ++t) {
                const int byte = (int)(h3_tap_word(tp, t) >> ((t & 7) * 8)) & 0xff;
                const int dh = tp.wide ? 0 : (byte >> 4) - 8, dw = tp.wide ? byte - 128 : (byte & 15) - 8;
                if ((unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)Wj) tapmask |= 1u << t;
            }
        } else {
            tapmask = 0xffffffffu;
        }
    }
    const int tA = tp.wide ? 16 : Wj, tC = j + (tp.wide ? -128 : -8 * Wj - 8);
    int x_t = it_lo / KB, x_kb = it_lo - x_t * KB;

    f32x16 acc[1][TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][jn][e] = 0.f;
    // fragment reads: plane p*2 + lk, column jn*32 + l31
    const unsigned char* bl = smem + (lk * NgP + l31) * 16;
    for (int r0 = 0; r0 < n_it; r0 += slots) {
        const int nr = min(slots, n_it - r0);
        if (r0 > 0) __syncthreads();                                     // every wave is done with the previous round's fragments
        if (tid < NgP) {                                                 // (whole waves: NgP is a multiple of 64)
            for (int i = 0; i < nr; ++i) {
                const int byte = (int)(h3_tap_word(tp, x_t) >> ((x_t & 7) * 8)) & 0xff;
                const unsigned ok = 0u - ((tapmask >> x_t) & 1u);
                const unsigned src = ((unsigned)((byte >> 4) * tA + (byte & 15) + tC) & ok) | ((unsigned)a.N & ~ok);
                (void)src;
#if __HIP_DEVICE_COMPILE__ && !defined(SN_EXP_NOX)   // (device pass only: with this builtin in the body hipcc 7.2's HOST pass drops the kernel's launch stub)
#pragma unroll
                for (int q = 0; q < (NP == 1 ? 2 : 4); ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (sn_lds_void*)(smem + i * SLOT + (q * NgP + wave * 64) * 16), 16,
                                                             (unsigned)(((x_kb * 4 + q) * NX) + src) * 16u, 0, 0, 0);
#endif
                if (++x_kb == KB) { x_kb = 0; ++x_t; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's DMAs (and the weight loads in flight) have landed
        __syncthreads();
        // a round is straight-line code (SN_MAX_IT iterations, nested early exits): the compiler then counts the weight loads in
        // flight exactly (vmcnt(10): five iterations ahead stay in flight); in a LOOP over ring slots it drains them all at the back edge
        auto iter = [&](auto i_) {
            constexpr int i = decltype(i_)::value, u = i % SN_D;
            const unsigned char* st = bl + i * SLOT;
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(st + jn * 32 * 16);
                if constexpr (NP != 1) {
                    const f16x8 bll = *reinterpret_cast<const f16x8*>(st + (2 * NgP + jn * 32) * 16);
                    acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u][0], bll, acc[0][jn], 0, 0, 0);     // smallest terms first
                    acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u][1], bh, acc[0][jn], 0, 0, 0);
                }
                acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u][0], bh, acc[0][jn], 0, 0, 0);
            }
            w_load(std::integral_constant<int, u>{}, r0 + i + SN_D < n_it);
        };
        sn_chain<0, SN_MAX_IT>(nr, iter);
    }

    const int rbase = m0 + 32 * wave + 4 * lk;
    if (S == 1) {
        epilogue_dispatch<1, TN>(a, acc, rbase, n0, l31, lk, grp, n_end);
        return;
    }
    // ---- split K: this slice's partial sums -> its slab; arrive; when all S are there, reduce this slice's share of the columns
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.ws) + (size_t)s * a.M * a.N, 0,
                                                                            (int)((unsigned)a.M * a.N * 4u), 0x00020000);
        slab_store<1, TN, 16>(a, acc, rs, rbase, n0, l31, n_end);          // sc1: write-through, no release fence below
    }
#ifdef SN_EXP_NOSYNC
    return;
#endif
    unsigned* cnt = sync + rgg;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; v < (unsigned)S && spin < (1 << 22); ++spin) {
            __builtin_amdgcn_s_sleep(2);
            v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // gave up: a slice of this group never arrived (it got no CU, or died).  The sums below are then not the layer's output: say
        // so (as_device_status); the module entry points return AS_EDEVICE from then on
        if (v < (unsigned)S) as_status_raise(a.status, AS_STATUS_GEMM_TIMEOUT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    {
        const int Ng = n_end - n0, cw = (Ng + S - 1) / S;
        const int c_lo = n0 + s * cw, c_hi = min(n_end, c_lo + cw);
        // the image's zero column belongs to the last slice of the last weight set
        const bool zero_col = a.Yh && s == S - 1 && grp == (a.n_groups > 1 ? a.n_groups : 1) - 1;
        const int ncols = max(c_hi - c_lo, 0) + (zero_col ? 1 : 0);
        const int g_tot = a.Yh ? max(2 * as_kbx(a.M), (a.M + 7) >> 3) : (a.M + 7) >> 3;
        const int g0 = rg * 16, ng = max(min(16, g_tot - g0), 0);
        for (int idx = tid; idx < ng * ncols; idx += 256) {
            const int c = idx % ncols, g = g0 + idx / ncols;
            const int j = (zero_col && c == ncols - 1) ? a.N : c_lo + c;
            as_reduce_epilogue(a, S, j, g);
        }
    }
    __syncthreads();
    if (tid == 0) {                                                      // departures: the last one leaves the counter at zero
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 2u * (unsigned)S - 1u) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------
struct SnPlan {
    int TN, S, n_it, slots, lds;
};

// The shapes this kernel takes and how it cuts them: every weight set sees at most 256 columns; a slice stages at most SN_MAX_IT
// iterations (and SN_LDS_MAX bytes of activations) per round and runs at most SN_MAX_ROUNDS rounds; slices are added until ~192 workgroups stream weights (more slices = more
// partial-slab traffic: a slab is 10 / n_it times its slice's weights at 160 columns), and the grid of a launch with S > 1 must be
// resident at once (one workgroup per CU is assumed: <= 256).
static bool sn_plan(const ConvGemmArgs& a, SnPlan* p)
{
    const char* env = getenv("AS_GEMM_SN");                              // tuning / experiments: "0" = never
    if (env && atoi(env) == 0) return false;
    if (!a.Xh || a.K == 1 || a.N <= 0) return false;
    const int G = a.n_groups > 1 ? a.n_groups : 1;
    const int ng = a.n_groups > 1 ? a.group_cols : a.N;
    if (ng > 256 || (a.n_groups > 1 && (long)a.group_cols * (G - 1) >= a.N)) return false;
    const int TN = as_cdiv(ng, 32), NgP = ((TN + 1) / 2) * 64, slot = NgP * 64;
    const int KB = a.Kp >> 4, IT = a.T * KB;
    const int RGG = as_cdiv(a.M, 128) * G;
    // iterations staged per round: a multiple of the weight ring's depth, so that an iteration's ring slot is its index in its round
    const int slots = std::min(SN_MAX_IT, SN_LDS_MAX / slot) / SN_D * SN_D;
    const int cap = slots * SN_MAX_ROUNDS;                               // iterations per slice
    const char* ew = getenv("AS_SN_WGS");                                // tuning: workgroups a launch aims for
    const int want = ew && atoi(ew) > 0 ? atoi(ew) : 192;
    const char* em = getenv("AS_SN_MINIT");                              // tuning: iterations a slice keeps at least
    const int min_it = em && atoi(em) > 0 ? atoi(em) : 8;
    int S = std::max(as_cdiv(IT, cap), std::min(std::max(IT / min_it, 1), as_cdiv(want, RGG)));
    if ((long)RGG * S > 256) S = std::max(as_cdiv(IT, cap), 256 / RGG);   // the slices of a group wait for each other: all resident
    if (S > 1 && ((long)RGG * S > 256 || !a.sync || a.sync_words < RGG || !a.ws)) {
        // cannot (or may not) wait for each other: one slice per group if it can hold every iteration, else not this kernel's shape
        if (as_cdiv(IT, cap) > 1) return false;
        S = 1;
    }
    if (S < 1) S = 1;
    const char* ems = getenv("AS_SN_MAXS");                              // tuning: largest number of slices this kernel may use
    if (ems && S > atoi(ems)) return false;
    p->TN = TN;
    p->S = S;
    p->n_it = as_cdiv(IT, S);
    p->slots = std::min(slots, p->n_it);
    p->lds = p->slots * slot;
    return true;
}

int as_conv_gemm_sn_slices(const ConvGemmArgs& a)
{
    SnPlan p;
    ConvGemmArgs q = a;
    if (!q.ws) q.ws = reinterpret_cast<void*>(16);                       // (a size query: the slabs are what is being sized)
    return sn_plan(q, &p) ? p.S : 0;
}

template <int TN, int NP>
static int sn_launch(const ConvGemmArgs& a, const H3Taps& tp, const SnPlan& p, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        AS_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_sn_kernel<TN, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, SN_LDS_MAX));
        attr_set = true;
    }
    const int G = a.n_groups > 1 ? a.n_groups : 1;
    const dim3 grid(as_cdiv(a.M, 128) * G * p.S);
    hipLaunchKernelGGL((conv_gemm_sn_kernel<TN, NP>), grid, dim3(256), p.lds, stream, a, tp, p.S, p.slots, a.sync);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

template <int NP>
static int sn_dispatch(const ConvGemmArgs& a, const H3Taps& tp, const SnPlan& p, hipStream_t stream)
{
    switch (p.TN) {
    case 1: return sn_launch<1, NP>(a, tp, p, stream);
    case 2: return sn_launch<2, NP>(a, tp, p, stream);
    case 3: return sn_launch<3, NP>(a, tp, p, stream);
    case 4: return sn_launch<4, NP>(a, tp, p, stream);
    case 5: return sn_launch<5, NP>(a, tp, p, stream);
    case 6: return sn_launch<6, NP>(a, tp, p, stream);
    case 7: return sn_launch<7, NP>(a, tp, p, stream);
    case 8: return sn_launch<8, NP>(a, tp, p, stream);
    default: return AS_EINVAL;
    }
}

int as_conv_gemm_sn_launch(const ConvGemmArgs& a, hipStream_t stream, bool* handled)
{
    SnPlan p;
    *handled = false;
    if (!sn_plan(a, &p)) return AS_OK;
    if (p.S > 1 && a.ws_bytes < (size_t)p.S * a.M * a.N * sizeof(float)) return AS_OK;   // no room for the slabs: the tiled kernel
    H3Taps tp;
    if (h3_pack_taps(a, &tp) != AS_OK) return AS_EINVAL;
    *handled = true;
    return a.n_prod == 1 ? sn_dispatch<1>(a, tp, p, stream) : sn_dispatch<3>(a, tp, p, stream);
}
