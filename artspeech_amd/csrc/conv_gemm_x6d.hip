// bf16x6 conv GEMM with BOTH operands arriving by LDS-DMA ("x6d").
//
// conv_gemm_x6.hip loads the fp32 activations into registers, splits them into three bf16 parts and stores them to
// LDS inside the k loop -- 8 loads, ~35 VALU and 3 LDS stores per thread and k-tile, and every output-channel tile of
// a launch repeats that work on the same activations.  Here the split is done ONCE per convolution by
// split_bf16x3_kernel (below) into the image the GEMM stages,
//     Xs[kb = k/16][p*2 + kh][column 0..N][8]  bf16   (k-blocks zero padded to a multiple of 4, like the weights; column N = 0)
// and the k loop of the GEMM moves both operands global -> LDS with buffer_load_dwordx4 ... lds: per k-tile a thread
// issues 6 LDS-DMAs, 12 fragment reads and 24 MFMAs.  Measured with the same stamps as the other kernels: 1 080 cycles
// per k-tile against 1 480 (lone workgroup), 30 % off on shapes with two workgroups per CU.
//
// A tap shifts the SOURCE column of a lane's 16-byte activation row (per-lane source addresses are what LDS-DMA
// offers); a tap that is invalid for an output column (conv zero padding / utterance wall) reads column N of the image,
// which the split kernel fills with zeros -- the image has N + 1 columns -- so the staged tile is exact and the k loop
// holds no masking at all.  LeakyReLU on the input is applied by the split kernel.
#include "x6_common.h"

template <int ACH, int BCH>
struct X6dPlan {                                        // staging pieces of one k-tile -> slot (behind which of the 24 MFMAs)
    static constexpr int M_TAP0 = 0;                    // the 8-tap word holding the staged tile's tap
    static constexpr int M_TAP1 = 1;                    // its byte; scalar offsets of the weight / activation k-blocks
    static constexpr int M_TAP2 = 2;                    // source column of this thread's activation rows
    static constexpr int M_DMA_A = 3;                   // ACH weight pieces
    static constexpr int M_DMA_B = M_DMA_A + ACH;       // BCH activation pieces
    static constexpr int M_ADV = M_DMA_B + BCH;         // advance the staging cursor
    static constexpr int M_FR = M_ADV + 1;              // 12 fragment reads of the next tile
    static constexpr int NM = M_FR + 12;
    // a slot holds what one wave can issue in the 32 cycles of its MFMA (~7 instructions): the tap decode takes three
    static constexpr int slot_of(int m)
    {
        if (m <= M_TAP2) return m;
        if (m < M_ADV) return 3 + ((m - M_DMA_A) * 9) / (ACH + BCH);     // LDS-DMA first: slots 3..11
        if (m == M_ADV) return 13;
        return 8 + (m - M_FR);                                           // fragment reads: slots 8..19
    }
};

// NS LDS stages.  With two, the tile staged during iteration `it` is read during it + 1, so its LDS-DMA must land within
// the iteration that issued it (measured: a lone workgroup then waits ~270 of its ~1 200 cycles per k-tile at the
// barrier); with three the tile is read during it + 2 and the end-of-iteration wait is vmcnt(this iteration's DMAs),
// i.e. only for the PREVIOUS iteration's.  Three stages of the 128x128 tile are 74 KB: two workgroups per CU still fit.
//
// OCC ("occupancy") variant: no software pipelining of the fragments at all -- an iteration reads its 12 fragments from
// the landed stage, issues the LDS-DMA of the NEXT tile into the other stage and multiplies; one fragment set instead of two
// keeps a wave under 170 registers, so THREE workgroups share a CU (two stages of the 128x128 tile are 49 KB) and cover each
// other's fragment latency and barriers, where the pipelined variant covers them inside one workgroup with twice the registers.
// Measured on MI355X (AS_X6D_OCC=1, kernel alone): 5-8 % SLOWER than the three-stage pipelined variant on every shape tried
// (M1024 N6400 K1024 T3: 244 vs 233 us; M128 N128000 K128 T9, four workgroups per CU available: 205 vs 189 us), so it stays an
// experiment: hiding latency inside the workgroup beats hiding it with occupancy here.
template <int WM, int WN, int WK, int NS, bool OCC = false>
__global__ void __launch_bounds__(64 * WM * WN * WK, OCC ? 3 : 1)
conv_gemm_x6d_kernel(const ConvGemmArgs a, const X6Taps tp)
{
    using C = X6Cfg<WM, WN, WK>;
    static_assert(NS == 2 || NS == 3, "stages");
    static_assert(!OCC || NS == 2, "the occupancy variant has two stages");
    constexpr int BM = C::BM, BN = C::BN, ACH = C::ACH, NT = C::NT;
    constexpr int BCH = WK * 6 * BN / NT;                                // 16-byte activation chunks per thread per k-tile
    static_assert((NT == 256 || NT == 512) && ACH * NT == WK * 6 * BM && BCH * NT == WK * 6 * BN, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    X6_STAMP(t0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn % WM, wn = wmn / WM;
    const int l31 = lane & 31, lk = lane >> 5;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tile = logical_tile();
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
    const int KB = a.Kp >> 4, KBx = (KB + 3) & ~3;
    const int NX = a.N + 1;                                              // columns of the activation image (the last one is zero)

    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>((a.n_split > 0 && n0 >= a.n_split) ? a.Wx2 : a.Wx), 0, (int)((unsigned)a.T * KBx * 6u * a.M * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.Xs), 0, (int)((unsigned)KBx * 6u * NX * 16u), 0x00020000);

    // chunk c = tid + NT i of a k-tile image [kblk][p*2+kh][row] -> LDS offset 16 c (both operands)
    unsigned a_voff[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = tid + NT * i, kblk = c / (6 * BM), rem = c % (6 * BM), pk = rem / BM, row = rem % BM;
        a_voff[i] = (m0 + row) < a.M ? (unsigned)(((kblk * 6 + pk) * a.M + m0 + row) * 16) : OOB;
    }
    unsigned b_plane[BCH];                                               // (kblk*6 + pk) * N * 16: plane of chunk i
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int c = tid + NT * i, kblk = c / (6 * BN), pk = (c % (6 * BN)) / BN;
        b_plane[i] = (unsigned)((kblk * 6 + pk) * NX) * 16u;
    }
    const int j = n0 + tid % BN;                                         // the column this thread stages (same for every chunk)
    unsigned tapmask = 0;                                                // taps that are valid for column j
    int Wj = 0;
    if (j < a.N) {
        if (a.meta) {
            const unsigned long long md = a.meta[j];
            const int h = (int)(md & 0xffff), w = (int)((md >> 16) & 0xffff);
            const int H = (int)((md >> 32) & 0xffff);
            Wj = (int)(md >> 48);
            for (int t = 0; t < a.T; ++t) {
                const int byte = x6_tap_byte(tp, t);
                const int dh = tp.wide ? 0 : (byte >> 4) - 8, dw = tp.wide ? byte - 128 : (byte & 15) - 8;
                if ((unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)Wj) tapmask |= 1u << t;
            }
        } else {
            tapmask = 0xffffffffu;
        }
    }
    const X6TapCol tcol(tp, j, Wj);

    const int kt_per_tap = (KB + WK - 1) / WK;
    const int nkt_all = a.T * kt_per_tap;
    const int S = gridDim.y;
    const int kt_lo = (int)((long)nkt_all * blockIdx.y / S);
    const int n_it = (int)((long)nkt_all * (blockIdx.y + 1) / S) - kt_lo;

    // cursor of the next tile to stage; past the end it stays on the last tile (loading it again into a stage nobody
    // reads is harmless)
    int c_t = kt_lo / kt_per_tap, c_kb = (kt_lo - c_t * kt_per_tap) * WK;
    auto advance = [&](int& t, int& kb) {
        int nkb = kb + WK, nt = t;
        if (nkb >= KB) { nkb = 0; nt += 1; }
        const bool ok = nt < a.T;
        kb = ok ? nkb : kb;
        t = ok ? nt : t;
    };
    unsigned c_col = 0;                                                  // source column * 16 of this thread's rows, staged tap
    int c_asoff = 0, c_bsoff = 0, c_byte = 0;
    unsigned long long c_word = 0;
    auto tap_word = [&]() {                                              // masks, not a select chain: see x6_tap_byte
        const int sel = c_t >> 3;
        const unsigned long long m0 = 0ull - (unsigned long long)(sel == 0), m1 = 0ull - (unsigned long long)(sel == 1),
                                 m2 = 0ull - (unsigned long long)(sel == 2), m3 = 0ull - (unsigned long long)(sel == 3);
        c_word = (tp.w0 & m0) | (tp.w1 & m1) | (tp.w2 & m2) | (tp.w3 & m3);
        asm volatile("" : "+s"(c_word));                                 // computed HERE (hipcc otherwise sinks it into a branch)
    };
    auto tap_byte = [&]() {
        c_byte = (int)(c_word >> ((c_t & 7) * 8)) & 0xff;
        c_asoff = (c_t * KBx + c_kb) * 6 * a.M * 16;
        c_bsoff = c_kb * 6 * NX * 16;
        asm volatile("" : "+s"(c_byte), "+s"(c_asoff), "+s"(c_bsoff));
    };
    auto tap_col = [&]() {                                               // an invalid tap reads the zero column N
        const unsigned ok = 0u - ((tapmask >> c_t) & 1u);                // all ones / zero: arithmetic select, no exec branch
        const unsigned src = ((unsigned)tcol.src(c_byte) & ok) | ((unsigned)a.N & ~ok);
        c_col = src * 16u;
    };
    auto dma_a = [&](auto i_, int stage) {
#if __HIP_DEVICE_COMPILE__ && !defined(X6_EXP_NODMA)   // (device pass only: with this builtin in the body hipcc 7.2's HOST pass drops the kernel's launch stub)
        constexpr int i = decltype(i_)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(smem + stage * C::STAGE + wave * 1024 + i * (NT * 16)), 16, a_voff[i],
                                                 c_asoff, 0, 0);
#endif
    };
    auto dma_b = [&](auto i_, int stage) {
#if __HIP_DEVICE_COMPILE__ && !defined(X6_EXP_NODMA)
        constexpr int i = decltype(i_)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(smem + stage * C::STAGE + WK * C::A_BLK + wave * 1024 + i * (NT * 16)), 16,
                                                 b_plane[i] + c_col, c_bsoff, 0, 0);
#endif
    };
    X6Frags fr[2];
    const int a_frag = (wk * 6 + lk) * BM * 16 + (wm * 64 + l31) * 16;
    const int b_frag = WK * C::A_BLK + (wk * 6 + lk) * BN * 16 + (wn * 64 + l31) * 16;
    auto read_frag = [&](auto q_, auto set_, int stage) {
#ifdef X6_EXP_NOFRAG
        return;
#endif
        constexpr int q = decltype(q_)::value, set = decltype(set_)::value, ab = q / 6, p = (q % 6) / 2, i = q % 2;
        const unsigned char* st = smem + stage * C::STAGE;
        if constexpr (ab == 0) fr[set].a[i][p] = *reinterpret_cast<const bf16x8*>(st + a_frag + p * 2 * BM * 16 + i * 32 * 16);
        else fr[set].b[i][p] = *reinterpret_cast<const bf16x8*>(st + b_frag + p * 2 * BN * 16 + i * 32 * 16);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    using I0 = std::integral_constant<int, 0>;
    // prologue: tiles 0 .. NS-1 staged, fragments of tile 0 in set 0 (OCC: tile 0 staged, nothing read yet)
    for (int st = 0; st < (OCC ? 1 : NS); ++st) {
        tap_word();
        tap_byte();
        tap_col();
        x6_for<0, ACH>([&](auto i_) { dma_a(i_, st); });
        x6_for<0, BCH>([&](auto i_) { dma_b(i_, st); });
        advance(c_t, c_kb);
    }
    __syncthreads();
    if constexpr (!OCC) {
        x6_for<0, 12>([&](auto q_) { read_frag(q_, I0{}, 0); });
        __syncthreads();      // every wave holds its fragments of tile 0 before the first iteration restages stage 0
    }

    using PL = X6dPlan<ACH, BCH>;
    // iteration `it`: stage P = it % NS holds tile it, whose fragments are in register set F = it % 2; the fragments of
    // tile it + 1 are read from stage P + 1 into set F ^ 1, tile it + NS is staged into stage P
    auto micro = [&](auto m_, auto p_, auto f_) {
        constexpr int M = decltype(m_)::value, P = decltype(p_)::value, F = decltype(f_)::value, PN = (P + 1) % NS;
        if constexpr (M == PL::M_TAP0) tap_word();
        else if constexpr (M == PL::M_TAP1) tap_byte();
        else if constexpr (M == PL::M_TAP2) tap_col();
        else if constexpr (M < PL::M_DMA_B) dma_a(std::integral_constant<int, M - PL::M_DMA_A>{}, OCC ? PN : P);
        else if constexpr (M < PL::M_ADV) dma_b(std::integral_constant<int, M - PL::M_DMA_B>{}, OCC ? PN : P);
        else if constexpr (M == PL::M_ADV) advance(c_t, c_kb);
        else if constexpr (!OCC) read_frag(std::integral_constant<int, M - PL::M_FR>{}, std::integral_constant<int, F ^ 1>{}, PN);
    };
    // MFMA n of the iteration: six groups of four, smallest terms first: (A part, B part) = (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
    auto step = [&](auto n_, auto p_, auto f_) {
        constexpr int N = decltype(n_)::value, F = decltype(f_)::value;
        constexpr int grp = N / 4, i = (N % 4) / 2, jn = N % 2;
        constexpr int PA = grp == 0 ? 0 : grp == 1 ? 2 : grp == 2 ? 1 : grp == 3 ? 0 : grp == 4 ? 1 : 0;
        constexpr int PB = grp == 0 ? 2 : grp == 1 ? 0 : grp == 2 ? 1 : grp == 3 ? 1 : grp == 4 ? 0 : 0;
        acc[i][jn] = X6_MFMA(fr[F].a[i][PA], fr[F].b[jn][PB], acc[i][jn]);
        x6_for<0, PL::NM>([&](auto m_) {
            if constexpr (PL::slot_of(decltype(m_)::value) == N) micro(m_, p_, f_);
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    // end of an iteration: the tile the NEXT iteration reads fragments from has landed (this wave's share) and everybody
    // is done with the stage the next iteration restages.  Not __syncthreads(): its fence waits for vmcnt(0).
    constexpr int PEND = NS == 2 ? 0 : ACH + BCH;                      // LDS-DMAs that may stay in flight across the barrier
    // OCC: this iteration's fragments first, in the order the MFMAs want them (weights h, activations l, ...)
#define X6D_ITER(U)                                                                                               \
    {                                                                                                             \
        if constexpr (OCC) x6_for<0, 12>([&](auto q_) { read_frag(q_, I0{}, (U) % 2); });                         \
        x6_for<0, 24>([&](auto n_) { step(n_, std::integral_constant<int, (U) % NS>{}, std::integral_constant<int, OCC ? 0 : (U) % 2>{}); }); \
        X6_BAR_BEGIN                                                                                              \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PEND) : "memory");                       \
        X6_BAR_END                                                                                                \
    }
    X6_STAMP(t1)
#ifdef X6_EXP_STAMPS
    unsigned long long w_bar = 0;
#endif
    int it = 0;
    if constexpr (NS == 2) {
        for (; it + 1 < n_it; it += 2) {
            X6D_ITER(0)
            X6D_ITER(1)
        }
        if (it < n_it) X6D_ITER(0)
    } else {                                                            // stage = it % 3, register set = it % 2: period 6
        for (; it + 5 < n_it; it += 6) {
            X6D_ITER(0)
            X6D_ITER(1)
            X6D_ITER(2)
            X6D_ITER(3)
            X6D_ITER(4)
            X6D_ITER(5)
        }
        if (it + 0 < n_it) X6D_ITER(0)
        if (it + 1 < n_it) X6D_ITER(1)
        if (it + 2 < n_it) X6D_ITER(2)
        if (it + 3 < n_it) X6D_ITER(3)
        if (it + 4 < n_it) X6D_ITER(4)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // re-staged tail tiles have landed before LDS is reused
    }
#undef X6D_ITER
    X6_STAMP(t2)

    if (WK > 1) {                                                       // sum the K groups: wk = 0 += wk = 1, 2, 3 in order
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        f32x4 v = {acc[i][jn][4 * e4], acc[i][jn][4 * e4 + 1], acc[i][jn][4 * e4 + 2], acc[i][jn][4 * e4 + 3]};
                        red[((((wk - 1) * (WM * WN) + wmn) * 4 + i * 2 + jn) * 4 + e4) * 64 + lane] = v;
                    }
        }
        __syncthreads();
        if (wk == 0)
#pragma unroll
            for (int s = 1; s < WK; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const f32x4 v = red[((((s - 1) * (WM * WN) + wmn) * 4 + i * 2 + jn) * 4 + e4) * 64 + lane];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[i][jn][4 * e4 + c] += v[c];
                        }
    }
    epilogue<2, 2>(a, acc, m0, n0, wm, wn, l31, lk, S, wk == 0);
    X6_STAMP(t3)
    X6_STAMPS_OUT
}

template <int WM, int WN, int WK, int NS, bool OCC = false>
static int launch_x6d(const ConvGemmArgs& a, int S, hipStream_t stream)
{
    using C = X6Cfg<WM, WN, WK>;
    constexpr int LDS = NS * C::STAGE > C::RED ? NS * C::STAGE : C::RED;
    static bool attr_set = false;
    if (!attr_set) {
        AS_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_x6d_kernel<WM, WN, WK, NS, OCC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    X6Taps tp;
    if (x6_pack_taps(a, &tp) != AS_OK) return AS_EINVAL;
    const dim3 grid(as_cdiv(a.M, C::BM) * as_cdiv(a.N, C::BN), S);
    hipLaunchKernelGGL((conv_gemm_x6d_kernel<WM, WN, WK, NS, OCC>), grid, dim3(C::NT), LDS, stream, a, tp);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

#ifndef LSTAGE12
#define LSTAGE12 2          // (three 37 KB stages on the 64x128 tile leave one workgroup per CU: 249 -> 313 us on M64 N509440 K64 T9)
#endif
int as_conv_gemm_x6d_launch(const ConvGemmArgs& a, int choice, int S, hipStream_t stream)
{
    if ((double)(((a.Kp >> 4) + 3) & ~3) * 6.0 * (a.N + 1.0) * 16.0 >= 2147483648.0) return AS_EINVAL;   // 32-bit offsets in the descriptor
    switch (choice) {
    case 22:
        if (getenv("AS_X6D_OCC")) return launch_x6d<2, 2, 1, 2, true>(a, S, stream);   // experiment: three workgroups per CU, no fragment pipelining
        return launch_x6d<2, 2, 1, 3>(a, S, stream);    // 3 x 24.5 KB of LDS: two workgroups per CU
    case 21: return launch_x6d<2, 1, 2, 2>(a, S, stream);    // (a third 37 KB stage would leave one workgroup per CU)
    case 12: return launch_x6d<1, 2, 2, LSTAGE12>(a, S, stream);
    default: return AS_EINVAL;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// X fp32 [K][ldx] -> Xs[(K/16 up to a multiple of 4)][p*2 + kh][N + 1][8] bf16, x = h + m + l exactly; optional LeakyReLU
// first; column N and rows >= K are zero.
// A thread owns 4 consecutive columns x 8 consecutive k: eight 16-byte loads (range-checked buffer loads: the last
// quad of a row may reach past N, and past the allocation on the last row), twelve 16-byte stores (a wave writes 4 KB runs).
__global__ void __launch_bounds__(256)
split_bf16x3_kernel(const float* __restrict__ x, int ldx, int K, int N, int lrelu, float slope, u32x4* __restrict__ xs)
{
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int g = blockIdx.y;                                           // 8-row group: kb = g / 2, kh = g % 2
    if (col > N) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)(((unsigned)(K - 1) * ldx + N) * 4u), 0x00020000);
    f32x4 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = g * 8 + r;
        v[r] = buf_load4(rs, k < K ? (unsigned)(k * ldx + col) * 4u : OOB, 0);
    }
    const size_t NX = (size_t)N + 1;
    const size_t base = ((size_t)(g >> 1) * 6 + (g & 1)) * NX + col;    // plane p*2 + kh of k-block kb
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (col + c > N) break;
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float e = col + c < N ? v[r][c] : 0.f;                      // column N: the zero column
            if (lrelu) e = e > 0.f ? e : slope * e;
            t[r] = e;
        }
        u32x4 h, m, l;
        split3(t, h, m, l);
        xs[base + c] = h;
        xs[base + c + 2 * NX] = m;
        xs[base + c + 4 * NX] = l;
    }
}

extern "C" size_t as_split_bf16x3_bytes(int K, int N)
{
    if (K <= 0 || N <= 0) return 0;
    const size_t kbx = (size_t)((((K + 15) >> 4) + 3) & ~3);
    return kbx * 6 * ((size_t)N + 1) * 16;
}

int as_split_bf16x3_launch(const float* x, int ldx, int K, int N, int lrelu, float slope, uint16_t* xs, hipStream_t stream)
{
    const int KBx = (((K + 15) >> 4) + 3) & ~3;
    if ((double)K * ldx * 4.0 >= 2147483648.0) return AS_EINVAL;        // 32-bit offsets in the buffer descriptor
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(as_cdiv(N + 1, 1024), 2 * KBx), dim3(256), 0, stream, x, ldx, K, N, lrelu, slope,
                       reinterpret_cast<u32x4*>(xs));
    AS_CHECK_LAUNCH();
    return AS_OK;
}

extern "C" int as_split_bf16x3_f32(const float* x, int ldx, int K, int N, int in_act, float in_slope, uint16_t* xs, as_stream_t stream)
{
    if (!x || !xs || K <= 0 || N < 0 || ldx < N || (in_act != 0 && in_act != 2)) return AS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(xs) & 15) != 0) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_CLS_OTHER, 0, 10.0 * K * (double)N, (hipStream_t)stream);
    return as_split_bf16x3_launch(x, ldx, K, N, in_act == 2, in_slope == 0.f ? 0.2f : in_slope, xs, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------------------------------
// AdaIN + LeakyReLU written DIRECTLY as the split image (the output of models.py:189-197's norm -> actv feeds nothing but
// the following conv, so the fp32 activations never exist): per-(channel, utterance) statistics exactly as
// adain_kernel (elementwise.hip) computes them -- one wave per channel, the same summation order -- then a thread takes
// 8 channels of one column, normalises, applies LeakyReLU(0.2), splits and stores three 16-byte rows.
// Workgroup = (8-channel group = one (k-block, k-half) of the image, utterance).
// ----------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ float xd_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256)
adain_split_kernel(const float* __restrict__ x, int ldx, int C, const float* __restrict__ gb, int ldgb,
                   const int* __restrict__ col_off, int N, int act, u32x4* __restrict__ xs)
{
    __shared__ float st[8][4];                                          // mean, rstd, 1 + gamma, beta
    const int g = blockIdx.x, b = blockIdx.y;
    const int c0 = g * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t NX = (size_t)N + 1;
    const size_t plane = ((size_t)(g >> 1) * 6 + (g & 1)) * NX;         // part h of this (k-block, k-half); m at + 2 NX, l at + 4 NX
    if (b == 0 && threadIdx.x < 3) xs[plane + (size_t)threadIdx.x * 2 * NX + N] = u32x4{0u, 0u, 0u, 0u};   // the zero column
    const int o0 = col_off[b], L = col_off[b + 1] - o0;
    if (L <= 0) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = c0 + wave * 2 + q;
        if (c < C) {                                                    // wave-uniform
            const float* xr = x + (size_t)c * ldx + o0;
            float s = 0.f;
            for (int i = lane; i < L; i += 64) s += xr[i];
            const float mean = xd_wave_sum(s) / (float)L;
            float v = 0.f;
            for (int i = lane; i < L; i += 64) { const float d = xr[i] - mean; v += d * d; }
            const float var = xd_wave_sum(v) / (float)L;
            if (lane == 0) {
                st[wave * 2 + q][0] = mean;
                st[wave * 2 + q][1] = 1.0f / sqrtf(var + 1e-5f);
                st[wave * 2 + q][2] = 1.0f + gb[(size_t)b * ldgb + c];
                st[wave * 2 + q][3] = gb[(size_t)b * ldgb + C + c];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += 256) {
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float o = 0.f;
            if (c0 + r < C) {
                o = st[r][2] * ((x[(size_t)(c0 + r) * ldx + o0 + i] - st[r][0]) * st[r][1]) + st[r][3];
                if (act) o = o > 0.f ? o : 0.2f * o;
            }
            t[r] = o;
        }
        u32x4 h, m, l;
        split3(t, h, m, l);
        const size_t at = plane + o0 + i;
        xs[at] = h;
        xs[at + 2 * NX] = m;
        xs[at + 4 * NX] = l;
    }
}

extern "C" int as_adain_split_f32(const float* x, int ldx, int C, const float* gamma_beta, int ldgb, const int32_t* col_off, int B,
                                  int N, int lrelu, uint16_t* xs, as_stream_t stream)
{
    if (!x || !gamma_beta || !col_off || !xs || C <= 0 || B <= 0 || N < 0 || ldgb < 2 * C) return AS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(xs) & 15) != 0) return AS_EINVAL;
    const int KBx = (((C + 15) >> 4) + 3) & ~3;
    AsProfScope prof__(AS_CLS_ADAIN, 0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(adain_split_kernel, dim3(2 * KBx, B), dim3(256), 0, (hipStream_t)stream, x, ldx, C, gamma_beta, ldgb, col_off, N,
                       lrelu, reinterpret_cast<u32x4*>(xs));
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Channel LayerNorm (+ReLU) written DIRECTLY as the split image: in the encoders the LayerNorm output feeds nothing but the
// following conv (RelTransformerEnc.py:72-87: norm_layers_1 -> attention's q/k/v convs, norm_layers_2 -> the FFN's first
// conv; :321-323 the prenet).  Same arithmetic as channel_ln_kernel (elementwise.hip: two-pass statistics over the channel
// axis, per-part partial sums reduced in the same order), but a thread owns 8 CONSECUTIVE channels per register group --
// one 16-byte row of the image -- instead of channels strided by 32.  Columns >= n_split take the second affine pair.
// ----------------------------------------------------------------------------------------------------------------
#define LNS_COLS 32
#define LNS_PARTS 32
#define LNS_MAXG 4                       // 8-channel groups per thread held in registers: C <= 8 * 32 * 4 = 1024
__global__ void __launch_bounds__(1024)
channel_ln_split_kernel(const float* __restrict__ x, int ldx, int C, int N, const float* __restrict__ gamma1,
                        const float* __restrict__ beta1, const float* __restrict__ gamma2, const float* __restrict__ beta2, int n_split,
                        float eps, int relu, u32x4* __restrict__ xs, int KBx)
{
    __shared__ float red[LNS_PARTS][LNS_COLS + 1];
    const int col = threadIdx.x % LNS_COLS, part = threadIdx.x / LNS_COLS;
    const int j = blockIdx.x * LNS_COLS + col;
    const bool ok = j < N;
    const bool second = gamma2 && j >= n_split;
    const float* gamma = second ? gamma2 : gamma1;
    const float* beta = second ? beta2 : beta1;
    const int ngroups = 2 * KBx;                                         // 8-channel groups of the image (zero beyond C)
    float v[LNS_MAXG][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LNS_MAXG; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int c = 8 * (part + i * LNS_PARTS) + r;
            v[i][r] = (ok && c < C) ? x[(size_t)c * ldx + j] : 0.f;
            s += v[i][r];
        }
    red[part][col] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < LNS_PARTS; ++q) tot += red[q][col];
    const float mean = tot / (float)C;
    __syncthreads();
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < LNS_MAXG; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int c = 8 * (part + i * LNS_PARTS) + r;
            const float d = v[i][r] - mean;
            if (c < C) q2 += d * d;
        }
    red[part][col] = q2;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int q = 0; q < LNS_PARTS; ++q) tot += red[q][col];
    const float rs = 1.0f / sqrtf(tot / (float)C + eps);
    const size_t NX = (size_t)N + 1;
    if (blockIdx.x == 0 && col == 0) {                                   // the zero column N of every plane this thread's groups own
#pragma unroll
        for (int i = 0; i < LNS_MAXG; ++i) {
            const int g = part + i * LNS_PARTS;
            if (g < ngroups)
                for (int p = 0; p < 3; ++p) xs[((size_t)(g >> 1) * 6 + (g & 1) + 2 * p) * NX + N] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < LNS_MAXG; ++i) {
        const int g = part + i * LNS_PARTS;
        if (g >= ngroups) continue;
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int c = 8 * g + r;
            float o = 0.f;
            if (c < C) {
                o = (v[i][r] - mean) * rs * gamma[c] + beta[c];
                if (relu) o = o > 0.f ? o : 0.f;
            }
            t[r] = o;
        }
        u32x4 h, m, l;
        split3(t, h, m, l);
        const size_t at = ((size_t)(g >> 1) * 6 + (g & 1)) * NX + j;
        xs[at] = h;
        xs[at + 2 * NX] = m;
        xs[at + 4 * NX] = l;
    }
}

extern "C" int as_channel_layernorm_split_f32(const float* x, int ldx, int C, int N, const float* gamma, const float* beta,
                                              const float* gamma2, const float* beta2, int n_split, float eps, int relu, uint16_t* xs,
                                              as_stream_t stream)
{
    if (!x || !xs || !gamma || !beta || C <= 0 || C > 8 * LNS_PARTS * LNS_MAXG || N <= 0 || ldx < N ||
        ((gamma2 == nullptr) != (beta2 == nullptr)) || (reinterpret_cast<uintptr_t>(xs) & 15) != 0)
        return AS_EINVAL;
    const int KBx = (((C + 15) >> 4) + 3) & ~3;
    AsProfScope prof__(AS_CLS_LN, 8.0 * C * N, 10.0 * C * N, (hipStream_t)stream);
    hipLaunchKernelGGL(channel_ln_split_kernel, dim3(as_cdiv(N, LNS_COLS)), dim3(1024), 0, (hipStream_t)stream, x, ldx, C, N, gamma, beta,
                       gamma2, beta2, n_split, eps, relu, reinterpret_cast<u32x4*>(xs), KBx);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
