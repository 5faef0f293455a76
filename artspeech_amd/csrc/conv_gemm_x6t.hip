// bf16x6 conv GEMM, "tap-shared" form: the activation tile of a k-block is split and staged ONCE for the three taps of a
// group (same row offset dh, consecutive column offsets dw0, dw0+1, dw0+2 -- every k = 3 / 9 1-D conv and every 3x3
// conv is made of such groups), because tap g of output column c reads row c + g of the staged tile.  Per 72 MFMAs
// (3 taps x 24) the staging of conv_gemm_x6.hip (8 loads, 35 VALU of splitting, 3 LDS stores, cursors) happens once
// instead of three times: ~3.5 instead of ~7 non-MFMA instructions per MFMA, which is what bounds that kernel.
//
// Tile 128 x 128, 4 waves (2 x 2, 64 x 64 each), k-block 16.  LDS stage = 3 weight tiles (one per tap, LDS-DMA) + one
// activation tile of 128 + 2 halo rows; two stages.  Super-iteration `it` (stage P holds tile it):
//   T0: MFMAs of tap 0 | loads + LDS-DMA of tile it+1 (into stage Q)
//   T1: MFMAs of tap 1 | fragment reads of tap 2, split + store of tile it+1's activations into stage Q
//   vmcnt(0) lgkmcnt(0), barrier            (stage Q complete; stage P's last reads retired)
//   T2: MFMAs of tap 2 | fragment reads of taps 0 and 1 of tile it+1 from stage Q
// (one fragment register set per tap; the sections are balanced so that the staging instructions of each fit in
// the issue gaps of its 24 MFMAs: measured with s_memtime, an unbalanced first version spent 1 610 / 1 380 / 915 cycles
// in T0 / T1 / T2 against 768 of MFMA each)
// A tap that is invalid for an output column (conv zero padding / utterance wall: the staged row then belongs to
// another image row) is removed by zeroing that lane's B fragments before the MFMAs (select on a per-column tap mask).
#include "x6_common.h"

#ifdef X6_EXP_STAMPS
#define XT_NOW(t) unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
#else
#define XT_NOW(t)
#endif

#define T_BM 128
#define T_BN 128
#define T_ROWS 136                        // rows of the staged activation tile: T_BN + 2 halo, padded to 8
#define T_A_TILE (6 * T_BM * 16)          // bytes: one tap's weight tile of a k-block
#define T_B_OFF (3 * T_A_TILE)
#define T_B_BLK (6 * T_ROWS * 16)
#define T_STAGE (T_B_OFF + T_B_BLK)
#define T_LDS (2 * T_STAGE)

struct X6TGroups {
    int ng;
    int dh[9], dw0[9];                    // group g = taps 3g, 3g+1, 3g+2 = (dh[g], dw0[g] + 0 / 1 / 2)
};

template <bool LRELU>
__global__ void __launch_bounds__(256)
conv_gemm_x6t_kernel(const ConvGemmArgs a, const X6TGroups tg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, lk = lane >> 5;
    const int tiles_m = (a.M + T_BM - 1) / T_BM;
    const int tile = logical_tile();
    const int m0 = (tile % tiles_m) * T_BM, n0 = (tile / tiles_m) * T_BN;
    const int KB = a.Kp >> 4, KBx = (KB + 3) & ~3;

    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.Wx), 0, (int)((unsigned)a.T * KBx * 6u * a.M * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((unsigned)a.K * a.ldx * 4u), 0x00020000);

    // weights: chunk c = tid + 256 i of a tap tile [p*2+kh][row] -> LDS offset 16 c
    unsigned a_voff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid + 256 * i, pk = c / T_BM, row = c % T_BM;
        a_voff[i] = (m0 + row) < a.M ? (unsigned)((pk * a.M + m0 + row) * 16) : OOB;
    }
    // activations: this thread stages row r (k half kh) of the tile; threads 0..3 also the halo rows 128, 129.
    // Row r of group g holds X at position p = n0 + r + dw0[g] (+ dh[g] image rows).
    const int r = tid % T_BN;
    const int kh = __builtin_amdgcn_readfirstlane(tid / T_BN);
    const int rx = T_BN + (tid & 1), khx = (tid >> 1) & 1;
    int rh = 0, rH = 1, rW = 0, xh = 0, xH = 1, xW = 0;    // image geometry at the rows' dh = 0 positions (2-D only)
    if (a.meta && tg.dh[0] != tg.dh[tg.ng - 1]) {           // some dh != 0: all groups share dw0 (host-checked)
        int p = n0 + r + tg.dw0[0];
        p = p < 0 ? 0 : p >= a.N ? a.N - 1 : p;
        unsigned long long md = a.meta[p];
        rh = (int)(md & 0xffff), rH = (int)((md >> 32) & 0xffff), rW = (int)(md >> 48);
        p = n0 + rx + tg.dw0[0];
        p = p < 0 ? 0 : p >= a.N ? a.N - 1 : p;
        md = a.meta[p];
        xh = (int)(md & 0xffff), xH = (int)((md >> 32) & 0xffff), xW = (int)(md >> 48);
    }
    // per output column of this lane (two 32-column MFMA tiles): bit t = tap t valid
    unsigned tapmask[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int j = n0 + wn * 64 + jn * 32 + l31;
        unsigned m = 0;
        if (j < a.N) {
            if (a.meta) {
                const unsigned long long md = a.meta[j];
                const int h = (int)(md & 0xffff), w = (int)((md >> 16) & 0xffff);
                const int H = (int)((md >> 32) & 0xffff), Wj = (int)(md >> 48);
                for (int t = 0; t < a.T; ++t)
                    if ((unsigned)(h + a.dh[t]) < (unsigned)H && (unsigned)(w + a.dw[t]) < (unsigned)Wj) m |= 1u << t;
            } else {
                m = 0xffffffffu;
            }
        }
        tapmask[jn] = m;
    }

    // most waves sit inside an utterance: every tap valid for all their columns -> no fragment masking at all
    const unsigned all_taps = a.T >= 32 ? 0xffffffffu : (1u << a.T) - 1u;
    const bool need_mask = __builtin_amdgcn_readfirstlane(
        __ballot((tapmask[0] & tapmask[1] & all_taps) == all_taps) != ~0ull);

    const int nkt_all = tg.ng * KB;
    const int S = gridDim.y;
    const int kt_lo = (int)((long)nkt_all * blockIdx.y / S);
    const int n_it = (int)((long)nkt_all * (blockIdx.y + 1) / S) - kt_lo;

    // cursor of the next tile to stage (group, k-block); past the end it stays on the last tile
    int cg = kt_lo / KB, ckb = kt_lo - cg * KB;
    int ug = cg;                                           // group of the tile being multiplied
    int ukb = ckb;
    auto advance = [&](int& g, int& kb) {
        int nkb = kb + 1, ng = g;
        if (nkb >= KB) { nkb = 0; ng += 1; }
        const bool ok = ng < tg.ng;
        kb = ok ? nkb : kb;
        g = ok ? ng : g;
    };
    const int ldx4 = a.ldx * 4, last_row = (a.K - 1) * ldx4;
    float rb[8], rbx[8];
    unsigned c_voff = OOB, c_voffx = OOB;
    int c_base = 0, c_basex = 0;
    auto row_setup = [&]() {                                // offsets of this thread's rows for group cg
        const int dh = tg.dh[cg < 9 ? cg : 8], dw0 = tg.dw0[cg < 9 ? cg : 8];
        const int p = n0 + r + dw0, px = n0 + rx + dw0;
        const bool ok = (unsigned)p < (unsigned)a.N && (unsigned)(rh + dh) < (unsigned)rH;
        const bool okx = (unsigned)px < (unsigned)a.N && (unsigned)(xh + dh) < (unsigned)xH;
        c_voff = ok ? (unsigned)(p + dh * rW) * 4u : OOB;
        c_voffx = okx ? (unsigned)(px + dh * xW) * 4u : OOB;
        c_base = (ckb * 16 + kh * 8) * ldx4;
        c_basex = (ckb * 16 + khx * 8) * ldx4;
    };
    auto load_row = [&](auto q_) {                          // k rows past K read row K-1 again (zero weight rows)
        constexpr int q = decltype(q_)::value;
        const int soff = c_base + q * ldx4;
        rb[q] = buf_load1(rsX, c_voff, soff < last_row ? soff : last_row);
    };
    auto load_halo = [&]() {
        if (tid < 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int off = c_basex + q * ldx4;
                rbx[q] = buf_load1(rsX, c_voffx == OOB ? OOB : c_voffx + (unsigned)(off < last_row ? off : last_row), 0);
            }
        }
    };
    int c_asoff = 0;
    auto dma_piece = [&](auto i_, int stage) {              // piece i = 0..8: tap i / 3, chunk i % 3
#if __HIP_DEVICE_COMPILE__
        constexpr int i = decltype(i_)::value, t3 = i / 3, ch = i % 3;
        if (ch == 0) c_asoff = (((cg * 3 + t3) * KBx + ckb) * 6) * a.M * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(smem + stage * T_STAGE + t3 * T_A_TILE + wave * 1024 + ch * 4096), 16,
                                                 a_voff[ch], c_asoff, 0, 0);
#endif
    };
    u32x4 sh, sm, sl;
    float c_r[4][2];
    auto split_half = [&](auto q_) {                         // q = 0..7: pair q / 2, first (x = h + r) or second (r = m + l) half
        constexpr int q = decltype(q_)::value, e = q / 2;
        if constexpr (q % 2 == 0) {
            float x0 = rb[2 * e], x1 = rb[2 * e + 1];
            if (LRELU) {
                x0 = vmax(x0, a.in_slope * x0);
                x1 = vmax(x1, a.in_slope * x1);
            }
            const unsigned h = pk_bf16(x0, x1);
            c_r[e][0] = x0 - bf_lo(h);
            c_r[e][1] = x1 - bf_hi(h);
            sh[e] = h;
        } else {
            const unsigned m = pk_bf16(c_r[e][0], c_r[e][1]);
            sm[e] = m;
            sl[e] = pk_bf16(c_r[e][0] - bf_lo(m), c_r[e][1] - bf_hi(m));
        }
    };
    auto store_part = [&](auto p_, int stage) {
        constexpr int p = decltype(p_)::value;
        unsigned char* b = smem + stage * T_STAGE + T_B_OFF + ((p * 2 + kh) * T_ROWS + r) * 16;
        *reinterpret_cast<u32x4*>(b) = p == 0 ? sh : p == 1 ? sm : sl;
    };
    auto halo_split_store = [&](int stage) {
        if (tid < 4) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = LRELU ? vmax(rbx[q], a.in_slope * rbx[q]) : rbx[q];
            u32x4 h, m, l;
            split3(x, h, m, l);
            unsigned char* b = smem + stage * T_STAGE + T_B_OFF + (khx * T_ROWS + rx) * 16;
            *reinterpret_cast<u32x4*>(b) = h;
            *reinterpret_cast<u32x4*>(b + 2 * T_ROWS * 16) = m;
            *reinterpret_cast<u32x4*>(b + 4 * T_ROWS * 16) = l;
        }
    };
    // fragments: weights of tap `tap`, activations shifted by `tap` rows
    X6Frags fr[3];                                           // one register set per tap
    const int a_frag = lk * T_BM * 16 + (wm * 64 + l31) * 16;
    const int b_frag = T_B_OFF + lk * T_ROWS * 16 + (wn * 64 + l31) * 16;
    auto read_frag = [&](auto q_, auto set_, int stage, int tap) {   // q = 0..11
        constexpr int q = decltype(q_)::value, set = decltype(set_)::value, ab = q / 6, p = (q % 6) / 2, i = q % 2;
        const unsigned char* st = smem + stage * T_STAGE;
        if constexpr (ab == 0) fr[set].a[i][p] = *reinterpret_cast<const bf16x8*>(st + tap * T_A_TILE + a_frag + p * 2 * T_BM * 16 + i * 32 * 16);
        else fr[set].b[i][p] = *reinterpret_cast<const bf16x8*>(st + b_frag + p * 2 * T_ROWS * 16 + (i * 32 + tap) * 16);
    };
    auto mask_frag = [&](auto set_, auto jn_, auto p_, int t) {      // zero B part p of column tile jn where tap t is invalid
        constexpr int set = decltype(set_)::value, jn = decltype(jn_)::value, p = decltype(p_)::value;
        const bool ok = (tapmask[jn] >> t) & 1u;
        u32x4 v = __builtin_bit_cast(u32x4, fr[set].b[jn][p]);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = ok ? v[c] : 0u;
        fr[set].b[jn][p] = __builtin_bit_cast(bf16x8, v);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // prologue: tile 0 -> stage 0, fragments of its tap 0 -> set 0
    row_setup();
    x6_for<0, 8>([&](auto q_) { load_row(q_); });
    load_halo();
    x6_for<0, 9>([&](auto i_) { dma_piece(i_, 0); });
    advance(cg, ckb);
    x6_for<0, 8>([&](auto q_) { split_half(q_); });
    store_part(I0{}, 0);
    store_part(I1{}, 0);
    store_part(I2{}, 0);
    halo_split_store(0);
    __syncthreads();
    x6_for<0, 12>([&](auto q_) { read_frag(q_, I0{}, 0, 0); });
    x6_for<0, 12>([&](auto q_) { read_frag(q_, I1{}, 0, 1); });
    __syncthreads();                                        // (tile 1 is not staged before every wave holds these)

    // MFMA n of a tap section: six groups of four, smallest terms first: (A part, B part) = (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
    auto mfma = [&](auto n_, auto set_) {
        constexpr int N = decltype(n_)::value, set = decltype(set_)::value;
        constexpr int grp = N / 4, i = (N % 4) / 2, jn = N % 2;
        constexpr int PA = grp == 0 ? 0 : grp == 1 ? 2 : grp == 2 ? 1 : grp == 3 ? 0 : grp == 4 ? 1 : 0;
        constexpr int PB = grp == 0 ? 2 : grp == 1 ? 0 : grp == 2 ? 1 : grp == 3 ? 1 : grp == 4 ? 0 : 0;
        acc[i][jn] = X6_MFMA(fr[set].a[i][PA], fr[set].b[jn][PB], acc[i][jn]);
    };
    // masks of the fragments a tap section is about to use, just ahead of their first MFMA: part 2 (l) before MFMA 0,
    // part 0 (h) before MFMA 4, part 1 (m) before MFMA 8
    auto masks_before = [&](auto n_, auto set_, int t) {
        constexpr int N = decltype(n_)::value;
        if constexpr (N == 0 || N == 4 || N == 8) {
            if (need_mask) {
                if constexpr (N == 0) { mask_frag(set_, I0{}, I2{}, t); mask_frag(set_, I1{}, I2{}, t); }
                if constexpr (N == 4) { mask_frag(set_, I0{}, I0{}, t); mask_frag(set_, I1{}, I0{}, t); }
                if constexpr (N == 8) { mask_frag(set_, I0{}, I1{}, t); mask_frag(set_, I1{}, I1{}, t); }
            }
        }
    };
#ifdef X6_EXP_STAMPS
    unsigned long long w_t0 = 0, w_t1 = 0, w_bar = 0, w_t2 = 0;
#endif
    // one super-iteration; P = stage of the tile.  Fragment set g belongs to tap g: sets 0 and 1 were filled during the
    // previous T2, set 2 is filled during T1.
    auto iteration = [&](auto p_) {
        constexpr int P = decltype(p_)::value, Q = P ^ 1;
        const int t0 = ug * 3;
        XT_NOW(s0)
        // ---- T0: tap 0 | loads and LDS-DMA of tile it+1 (into stage Q)
        x6_for<0, 24>([&](auto n_) {
            constexpr int N = decltype(n_)::value;
            masks_before(n_, I0{}, t0);
            mfma(n_, I0{});
            if constexpr (N == 0) row_setup();
            if constexpr (N >= 1 && N <= 15 && N % 2 == 1) load_row(std::integral_constant<int, (N - 1) / 2>{});
            if constexpr (N >= 2 && N <= 18 && N % 2 == 0) dma_piece(std::integral_constant<int, (N - 2) / 2>{}, Q);
            if constexpr (N == 17) load_halo();
            if constexpr (N == 20) advance(cg, ckb);
            __builtin_amdgcn_sched_barrier(0);
        });
        XT_NOW(s1)
        // ---- T1: tap 1 | fragments of tap 2 -> set 2, split + store tile it+1's activations into stage Q
        x6_for<0, 24>([&](auto n_) {
            constexpr int N = decltype(n_)::value;
            masks_before(n_, I1{}, t0 + 1);
            mfma(n_, I1{});
            if constexpr (N >= 1 && N <= 23 && N % 2 == 1) read_frag(std::integral_constant<int, (N - 1) / 2>{}, I2{}, P, 2);
            if constexpr (N >= 2 && N <= 16 && N % 2 == 0) split_half(std::integral_constant<int, (N - 2) / 2>{});
            if constexpr (N == 18) halo_split_store(Q);
            if constexpr (N >= 20 && N <= 22) store_part(std::integral_constant<int, N - 20>{}, Q);
            __builtin_amdgcn_sched_barrier(0);
        });
        XT_NOW(s2)
        __syncthreads();                                    // vmcnt(0) lgkmcnt(0) + barrier
        XT_NOW(s3)
        advance(ug, ukb);
        // ---- T2: tap 2 | fragments of taps 0 and 1 of tile it+1 from stage Q -> sets 0, 1
        x6_for<0, 24>([&](auto n_) {
            constexpr int N = decltype(n_)::value;
            masks_before(n_, I2{}, t0 + 2);
            mfma(n_, I2{});
            if constexpr (N < 12) read_frag(n_, I0{}, Q, 0);
            else read_frag(std::integral_constant<int, N - 12>{}, I1{}, Q, 1);
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef X6_EXP_STAMPS
        XT_NOW(s4)
        w_t0 += s1 - s0; w_t1 += s2 - s1; w_bar += s3 - s2; w_t2 += s4 - s3;
#endif
    };
    int it = 0;
    for (; it + 1 < n_it; it += 2) {
        iteration(I0{});
        iteration(I1{});
    }
    if (it < n_it) iteration(I0{});
    __syncthreads();
    epilogue<2, 2>(a, acc, m0, n0, wm, wn, l31, lk, S);
#ifdef X6_EXP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {                                          // thread 0 owns Y[m0 + 0..3][n0]
        a.Y[(size_t)(m0 + 0) * a.ldy + n0] = (float)w_t0 / n_it;
        a.Y[(size_t)(m0 + 1) * a.ldy + n0] = (float)w_t1 / n_it;
        a.Y[(size_t)(m0 + 2) * a.ldy + n0] = (float)w_bar / n_it;
        a.Y[(size_t)(m0 + 3) * a.ldy + n0] = (float)w_t2 / n_it;
    }
#endif
}

template <bool LRELU>
static int launch_x6t(const ConvGemmArgs& a, const X6TGroups& tg, int S, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        AS_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_x6t_kernel<LRELU>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS));
        attr_set = true;
    }
    const dim3 grid(as_cdiv(a.M, T_BM) * as_cdiv(a.N, T_BN), S);
    hipLaunchKernelGGL((conv_gemm_x6t_kernel<LRELU>), grid, dim3(256), T_LDS, stream, a, tg);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// Taps in groups of three (same dh, dw0, dw0+1, dw0+2), T a multiple of 3; with image rows involved (some dh != 0) every
// group must start at the same dw0.  Returns the number of k-tiles of the tap-shared kernel (groups x Kp/16), 0 if the
// taps do not have this form.
int as_conv_gemm_x6t_ktiles(const ConvGemmArgs& a)
{
    if (a.T % 3 || a.T > 27 || a.in_act == 1) return 0;
    bool any_dh = false;
    for (int g = 0; g < a.T / 3; ++g) {
        const int t = 3 * g;
        if (a.dh[t + 1] != a.dh[t] || a.dh[t + 2] != a.dh[t] || a.dw[t + 1] != a.dw[t] + 1 || a.dw[t + 2] != a.dw[t] + 2) return 0;
        any_dh |= a.dh[t] != a.dh[0];
    }
    if (any_dh)
        for (int g = 1; g < a.T / 3; ++g)
            if (a.dw[3 * g] != a.dw[0]) return 0;
    return (a.T / 3) * (a.Kp / 16);
}

int as_conv_gemm_x6t_launch(const ConvGemmArgs& a, int S, hipStream_t stream)
{
    X6TGroups tg;
    tg.ng = a.T / 3;
    for (int g = 0; g < 9; ++g) {
        tg.dh[g] = g < tg.ng ? a.dh[3 * g] : 0;
        tg.dw0[g] = g < tg.ng ? a.dw[3 * g] : 0;
    }
    if (a.in_act != 0 && a.in_act != 2) return AS_EINVAL;
    return a.in_act == 2 ? launch_x6t<true>(a, tg, S, stream) : launch_x6t<false>(a, tg, S, stream);
}
