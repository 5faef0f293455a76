// K3/K4/K8/K10 -- every dense convolution / linear layer of the path as ONE implicit-GEMM kernel on the fp16 matrix cores,
// fp32-accurate through a two-way operand split ("f16x3"):
//
//   Y[m][j] = epilogue( sum_t sum_k  W[t][k][m] * X[k][ j + dh[t]*Wj + dw[t] ] )      m < M, j < N
//
//   x = h + l,  h = fp16(x), l = fp16(x - h)  (22 significand bits);   x*w ~= h_x*l_w + l_x*h_w + h_x*h_w
//   on v_mfma_f32_32x32x16_f16 (32 cycles, K = 16), fp32 accumulation, smallest terms first.  The dropped l*l term is below
//   2^-22 |x w|; weights are pre-scaled by a power of two so their l parts are normal numbers (ConvGemmArgs.acc_scale undoes
//   it).  scripts/exp/f16x3_numerics.py runs the whole path this way on the CPU: the mel lands 6e-6 from the reference's
//   golden vectors, where exact fp32 products land 5e-6 (bound 1e-4).  Three matrix-core products per fp32 product instead of
//   round 1's six bf16 ones (bf16x6), and 4 bytes of operand image per element instead of 6.
//
// Both operands arrive by LDS-DMA (buffer_load_dwordx4 ... lds) from images that are stored in staging order:
//   Wh[tap][kb = k/16][q = p*2 + kh][m][8]       fp16, p = 0 (h) / 1 (l), kh = k-half     (as_prep_weight_f16x2_host)
//   Xh[kb][q][column 0..N][8]                     fp16, column N = 0                        (split_f16x2_kernel and the producers)
// so the k loop holds no conversion, no masking and no activation registers: a tap shifts the SOURCE column of a lane's
// 16-byte row (per-lane source addresses are what LDS-DMA offers); a tap that is invalid for an output column (conv zero
// padding / utterance wall) reads the zero column N.  Per 16-deep k-block a wave issues 4 LDS-DMAs (128x128 tile), 8
// ds_read_b128 fragment reads and 12 MFMAs.
//
// Workgroup = 4 waves, each a 64x64 output block (2x2 MFMA tiles, 64 accumulator registers): WM x WN x WK waves along M, N, K;
// tiles 128x128 <2,2,1>, 128x64 <2,1,2>, 64x128 <1,2,2>, 64x64 <1,1,4>; with WK > 1 the waves split K and sum their
// accumulators through LDS at the end in a fixed order.  KT k-blocks per wave and iteration (one barrier per iteration), NS
// LDS stages: with three, the tile staged in iteration `it` is read in it + 2 and the end-of-iteration wait is
// vmcnt(this iteration's DMAs) -- only the PREVIOUS iteration's must have landed -- followed by a raw s_barrier
// (__syncthreads() would wait for vmcnt(0)).
#include "conv_gemm.h"
#include <cstring>
#include <type_traits>

// knock-out switches of experiment builds only (scripts/build_exp.sh NAME -DH3_EXP_NOMFMA / _NODMA / _NOFRAG): what bounds the k loop
#ifdef H3_EXP_NOMFMA
#define H3_MFMA(A, B, C) (C)
#else
#define H3_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#endif

typedef __attribute__((address_space(3))) void lds_void;

template <int I, int N, typename F>
static __device__ __forceinline__ void h3_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        h3_for<I + 1, N>(f);
    }
}

// TM: 32-row MFMA tiles per wave (2: a wave owns 64 x 64; 4: 128 x 64 -- a third less LDS traffic per matrix-core product)
template <int WM, int WN, int WK, int KT, int NS, int NP, int TM = 2>
struct H3Cfg {
    static constexpr int QP = NP == 1 ? 2 : 4;                           // planes staged per k-block (h only / h and l)
    static constexpr int BM = 32 * TM * WM, BN = 64 * WN;
    static constexpr int KBS = WK * KT;                                  // k-blocks per stage
    static constexpr int A_BLK = QP * BM * 16, B_BLK = QP * BN * 16;     // bytes per 16-deep k-block
    static constexpr int A_ST = KBS * A_BLK;                             // weight part of a stage
    static constexpr int STAGE = KBS * (A_BLK + B_BLK);
    static constexpr int RED = (WK - 1) * WM * WN * 32 * TM * 64 * 4;    // cross-wave K reduction scratch
    static constexpr int LDS = NS * STAGE > RED ? NS * STAGE : RED;
    static constexpr int NT = 64 * WM * WN * WK;                         // threads
    static constexpr int ACH = KBS * QP * BM / NT;                       // 16-byte weight chunks per thread per iteration
    static constexpr int BCH = KBS * QP * BN / NT;                       // 16-byte activation chunks per thread per iteration
    static constexpr int NMF = 2 * TM * NP;                              // MFMAs per k-block
    static constexpr int NFR = (TM + 2) * (NP == 1 ? 1 : 2);             // fragment reads per k-block
};

#ifndef H3_OCC
#define H3_OCC 1
#endif
// which problem a workgroup belongs to (wave-uniform: scalar compares on the kernel arguments)
static __device__ __forceinline__ int h3_problem_of(const H3Multi& mm)
{
    int pi = 0;
#pragma unroll
    for (int i = 1; i < H3_MAXP; ++i) pi += (i < mm.n && (int)blockIdx.x >= mm.p[i].wg0) ? 1 : 0;
    return pi;
}

// one workgroup's tile of problem `prob` (the whole kernel but for the problem look-up: conv_gemm_h3_kernel runs it for every
// workgroup, conv_gemm_h3_mix_kernel runs one of two instantiations, by the problem's tile class)
template <int WM, int WN, int WK, int KT, int NS, int NP, int TM = 2>
static __device__ __forceinline__ void h3_tile(const H3Prob& prob, unsigned char* smem)
{
    using C = H3Cfg<WM, WN, WK, KT, NS, NP, TM>;
    static_assert(NS >= 2 && NS <= 4, "stages");
    constexpr int BM = C::BM, BN = C::BN, ACH = C::ACH, BCH = C::BCH, NT = C::NT, QP = C::QP, KBS = C::KBS;
    static_assert(NT == 256 && ACH * NT == KBS * QP * BM && BCH * NT == KBS * QP * BN && ACH >= 1 && BCH >= 1, "tile shape");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn % WM, wn = wmn / WM;
    const int l31 = lane & 31, lk = lane >> 5;
    const ConvGemmArgs& a = prob.a;
    const H3Taps& tp = prob.tp;
    const int S = prob.S;
    const int tiles_m = (a.M + BM - 1) / BM;
    int lt, slice, tile, m0, n0, grp = 0, n_end = prob.col1;
    if (a.n_valid && S == 1) {
        // Capacity layout (ConvGemmArgs.n_valid): the launch is sized for room, the leading *n_valid columns of every group hold
        // utterances, the rest is filler (zero-column reads, no stores, no work).  The VALID tiles are dealt out like the tiles of a
        // launch that has no others: XCD x gets a contiguous run of them -- with the launch's own order the filler, which is the tail
        // of the column range, would be whole XCDs' shares and the other XCDs would carry 8 / 6 of the work (measured: + 6 % per step
        // at 25 % of room to spare).  Tiles are counted column tile by column tile ACROSS the groups (tn outer, group inner), so that
        // the valid ones are a prefix.
        const int nv = __builtin_amdgcn_readfirstlane(*a.n_valid);
        const int G = a.n_groups > 1 ? a.n_groups : 1;
        const int span = G > 1 ? a.group_cols : prob.col1 - prob.col0;
        const int ctv = (min(max(nv, 1), span) + BN - 1) / BN;              // valid column tiles of a group (>= 1: tile 0 owns the image's zero column)
        const int tiles_v = tiles_m * G * ctv, per = (tiles_v + 7) >> 3;
        const int local = (int)blockIdx.x - prob.wg0;
        if ((local >> 3) >= per) return;
        lt = (local & 7) * per + (local >> 3);
        if (lt >= tiles_v) return;
        slice = 0;
        tile = lt;
        m0 = (lt % tiles_m) * BM;
        const int q = lt / tiles_m, tn = q / G;
        grp = q - tn * G;
        const int base = G > 1 ? grp * a.group_cols : prob.col0;
        n0 = base + tn * BN;
        n_end = min(G > 1 ? min(a.N, (grp + 1) * a.group_cols) : prob.col1, base + nv);
    } else {
        lt = logical_of((int)blockIdx.x - prob.wg0, prob.wgs);
        if (lt >= prob.tiles * S) return;                                // (the padding up to a multiple of 8; the whole workgroup)
        slice = lt / prob.tiles;
        tile = lt - slice * prob.tiles;
        m0 = (tile % tiles_m) * BM;
        // column tiles: per weight set when the launch is grouped (a group's columns are [grp * group_cols, (grp + 1) * group_cols);
        // its last tile is cut at the group's end, so group_cols needs no alignment)
        n0 = prob.col0 + (tile / tiles_m) * BN;
        if (a.n_groups > 1) {
            const int tpg = (a.group_cols + BN - 1) / BN, tn = tile / tiles_m;
            grp = tn / tpg;
            n0 = grp * a.group_cols + (tn - grp * tpg) * BN;
            n_end = min(a.N, (grp + 1) * a.group_cols);
        }
        if (a.n_valid) {                                                 // (K-sliced launches under a capacity: small ones; the launch's own order)
            n_end = min(n_end, (a.n_groups > 1 ? grp * a.group_cols : 0) + __builtin_amdgcn_readfirstlane(*a.n_valid));
            if (n0 >= n_end && n0 > 0) return;
        }
    }
    const int KB = a.Kp >> 4, KBx = (KB + 3) & ~3;
    const int NX = (a.src_col ? a.N_in : a.N) + 1;                       // columns of the activation image (the last one is zero)

    const int KB2 = (a.K2 + 15) >> 4, KBx2 = (KB2 + 3) & ~3;             // the second operand's k-blocks (0: none)
    const unsigned w_bytes = ((unsigned)a.T * KBx + KBx2) * 4u * a.M * 16u;   // one weight set
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.Wh) + (size_t)grp * w_bytes), 0, (int)w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.Xh), 0, (int)((unsigned)KBx * 4u * NX * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(KB2 ? a.Xh2 : a.Xh), 0, (int)((unsigned)KBx2 * 4u * NX * 16u), 0x00020000);

    (void)rsW;
    (void)rsX;      // (the host pass does not see the LDS-DMA builtins that use them)
    (void)rsX2;
    // chunk c = tid + NT i of an iteration's image [kblk][plane][row] -> LDS offset 16 c (both operands).  The global image
    // always has four planes per k-block; with NP = 1 only planes 0, 1 (the h parts) are staged.
    unsigned a_voff[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = tid + NT * i, kblk = c / (QP * BM), rem = c % (QP * BM), pk = rem / BM, row = rem % BM;
        a_voff[i] = (m0 + row) < a.M ? (unsigned)(((kblk * 4 + pk) * a.M + m0 + row) * 16) : OOB;
    }
    unsigned b_plane[BCH];                                               // (kblk*4 + pk) * NX * 16: plane of chunk i
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int c = tid + NT * i, kblk = c / (QP * BN), pk = (c % (QP * BN)) / BN;
        b_plane[i] = (unsigned)((kblk * 4 + pk) * NX) * 16u;
    }
    const int j = n0 + tid % BN;                                         // the column this thread stages (same for every chunk)
    unsigned tapmask = 0;                                                // taps that are valid for column j
    int Wj = 0;
    if (j < n_end) {
        if (a.meta) {
            const unsigned long long md = a.meta[j];
            const int h = AS_META_h(md), w = AS_META_w(md);
            const int H = AS_META_H(md);
            Wj = AS_META_W(md);
            for (int t = 0; t < a.T; ++t) {
                const int byte = (int)(h3_tap_word(tp, t) >> ((t & 7) * 8)) & 0xff;
                const int dh = tp.wide ? 0 : (byte >> 4) - 8, dw = tp.wide ? byte - 128 : (byte & 15) - 8;
                if ((unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)Wj) tapmask |= 1u << t;
            }
        } else {
            tapmask = 0xffffffffu;
        }
    }
    // source column of a tap for this thread's column, one formula for both encodings:
    //   src = j + (byte >> 4) * tA + (byte & 15) + tC;  narrow: tA = Wj, tC = -8 Wj - 8;  wide: tA = 16, tC = -128
    // (strided / valid convs: the output column's tap (0, 0) reads input column src_col[j]; meta describes that input position)
    const int jsrc = a.src_col ? (j < n_end ? a.src_col[j] : 0) : j;
    const int tA = tp.wide ? 16 : Wj, tC = jsrc + (tp.wide ? -128 : -8 * Wj - 8);

    // iterations: an iteration covers KBS k-blocks of one tap (the tail of a tap re-reads zero blocks: KBx is a multiple of 4
    // and the weights' zero rows make them harmless as long as KBS divides 4 or the cursor clamps -- see advance)
    // The second operand (ConvGemmArgs.Xh2: a block's learned shortcut) is one more "tap" after the last: no column shift, its own
    // image and k-block count, its weights behind the taps' in the weight set.
    const int it_per_tap = (KB + KBS - 1) / KBS;
    const int nit_all = a.T * it_per_tap + (KB2 + KBS - 1) / KBS;
    const int it_lo = (int)((long)nit_all * slice / S);
    const int n_it = (int)((long)nit_all * (slice + 1) / S) - it_lo;

    // Cursor of the next tile to stage.  Inside a tap an iteration only adds constants to two scalar offsets; the tap's decode
    // (source column of every thread's rows; weights / activation offsets back to the tap's first k-block) runs when the tap
    // CHANGES, once per KB / KBS iterations, behind a wave-uniform branch.  (Round 2 PMC: with the decode in every iteration the
    // loop carried 43 scalar + 23 vector instructions per 12 MFMAs and the waves' issue slots, not the matrix cores, set the pace.)
    // Past the end the cursor stays on the last tile (loading it again into a stage nobody reads is harmless).
    int s_t = min(it_lo / it_per_tap, a.T), s_kb = (it_lo - s_t * it_per_tap) * KBS;
    int kb_lim = s_t < a.T ? KB : KB2;                                   // k-blocks of the staged tap
    __amdgpu_buffer_rsrc_t rsXc = s_t < a.T ? rsX : rsX2;                // the image the staged tap reads
    const int a_step = KBS * 4 * a.M * 16, b_step = KBS * 4 * NX * 16;
    int a_soff = (s_t * KBx + s_kb) * 4 * a.M * 16, b_soff = s_kb * 4 * NX * 16;
    unsigned bv[BCH];                                                    // voffset of this thread's activation chunks for the staged tap
    auto set_tap = [&](int t) {                                          // an invalid tap reads the zero column N
        int byte = (int)(h3_tap_word(tp, t) >> ((t & 7) * 8)) & 0xff;
        unsigned ok = 0u - ((tapmask >> (t & 31)) & 1u);                 // all ones / zero: arithmetic select, no exec branch
        if (t >= a.T) {                                                  // (wave-uniform) the second operand: the column itself
            byte = tp.wide ? 128 : 0x88;
            ok = 0u - (unsigned)(j < n_end);
        }
        const unsigned src = ((unsigned)((byte >> 4) * tA + (byte & 15) + tC) & ok) | ((unsigned)(NX - 1) & ~ok);
#pragma unroll
        for (int i = 0; i < BCH; ++i) bv[i] = b_plane[i] + src * 16u;
    };
    set_tap(s_t);
    auto advance = [&]() {
        s_kb += KBS;
        a_soff += a_step;
        b_soff += b_step;
        if (s_kb >= kb_lim) {                                            // wave-uniform, once per tap
            if (s_t + 1 < a.T || (s_t + 1 == a.T && KB2 > 0)) {
                s_t += 1;
                s_kb = 0;
                a_soff = s_t * KBx * 4 * a.M * 16;
                b_soff = 0;
                if (s_t == a.T) {
                    kb_lim = KB2;
                    rsXc = rsX2;
                }
                set_tap(s_t);
            } else {
                s_kb -= KBS;
                a_soff -= a_step;
                b_soff -= b_step;
            }
        }
    };
    auto dma_a = [&](auto i_, int stage) {
#if __HIP_DEVICE_COMPILE__ && !defined(H3_EXP_NODMA)   // (device pass only: with this builtin in the body hipcc 7.2's HOST pass drops the kernel's launch stub)
        constexpr int i = decltype(i_)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(smem + stage * C::STAGE + wave * 1024 + i * (NT * 16)), 16, a_voff[i],
                                                 a_soff, 0, 0);
#endif
    };
    auto dma_b = [&](auto i_, int stage) {
#if __HIP_DEVICE_COMPILE__ && !defined(H3_EXP_NODMA)
        constexpr int i = decltype(i_)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsXc, (lds_void*)(smem + stage * C::STAGE + C::A_ST + wave * 1024 + i * (NT * 16)), 16, bv[i],
                                                 b_soff, 0, 0);
#endif
    };
    (void)bv;
    (void)a_soff;
    (void)b_soff;
    (void)rsXc;
    // fragments: [set][32-row / 32-column tile][part]
    f16x8 fa[2][TM][2], fb[2][2][2];
    const int a_frag = (wk * KT * QP + lk) * BM * 16 + (wm * 32 * TM + l31) * 16;
    const int b_frag = C::A_ST + (wk * KT * QP + lk) * BN * 16 + (wn * 64 + l31) * 16;
    // fragment read q of k-block step s of stage `stage` into register set `set`: q = ab*NFR/2 + p*2 + i
    auto read_frag = [&](auto q_, auto set_, auto s_, int stage) {
#ifdef H3_EXP_NOFRAG
        return;
#endif
        constexpr int q = decltype(q_)::value, set = decltype(set_)::value, s = decltype(s_)::value;
        // in the order the MFMAs first need them: (A h, B l) for h*l, then (A l, B h) for l*h; h*h reuses them
        // reads: TM x A h, 2 x B l, TM x A l, 2 x B h  (NP = 1: TM x A h, 2 x B h)
        constexpr int sec = q < TM ? 0 : (q < TM + 2 ? 1 : (q < 2 * TM + 2 ? 2 : 3));
        constexpr int i = sec == 0 ? q : (sec == 1 ? q - TM : (sec == 2 ? q - TM - 2 : q - 2 * TM - 2));
        constexpr int ab = sec == 0 || sec == 2 ? 0 : 1;
        constexpr int p = NP == 1 ? 0 : (sec == 0 || sec == 3 ? 0 : 1);
        const unsigned char* st = smem + stage * C::STAGE;
        if constexpr (ab == 0) fa[set][i][p] = *reinterpret_cast<const f16x8*>(st + a_frag + (s * QP + p * 2) * BM * 16 + i * 32 * 16);
        else fb[set][i][p] = *reinterpret_cast<const f16x8*>(st + b_frag + (s * QP + p * 2) * BN * 16 + i * 32 * 16);
    };
    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    using I0 = std::integral_constant<int, 0>;
    // Look-ahead L: iteration `it` stages tile it + L into stage (it + L) % NS.  KT = 1: L = NS -- the iteration's own stage,
    // whose fragments were all read during iteration it - 1, before the barrier.  KT > 1: steps 1.. of an iteration still read
    // the iteration's own stage, so the target is the stage of tile it - 1: L = NS - 1.
    constexpr int L = KT == 1 ? NS : NS - 1;
    static_assert(L >= 2, "KT > 1 needs three stages");
    // prologue: tiles 0 .. L-1 staged, fragments of (tile 0, step 0) in set 0
    for (int st = 0; st < L; ++st) {
        h3_for<0, ACH>([&](auto i_) { dma_a(i_, st); });
        h3_for<0, BCH>([&](auto i_) { dma_b(i_, st); });
        advance();
    }
    __syncthreads();
    h3_for<0, C::NFR>([&](auto q_) { read_frag(q_, I0{}, I0{}, 0); });
    __syncthreads();      // every wave holds its fragments of tile 0 before the first iteration restages stage 0

    // Iteration `it` (stage P = it % NS holds its tile): step s multiplies k-block s from register set F = (it KT + s) % 2
    // while the fragments of the next step -- k-block s + 1 of the same stage, or k-block 0 of stage P + 1 -- are read into
    // the other set, and the staging of tile it + L is dealt out behind the MFMAs: a wave issues about one instruction per
    // 4 cycles and an MFMA holds the matrix core for 32.
    constexpr int NMI = KT * C::NMF;                                     // MFMAs per iteration
    // staging micro-operations of an iteration: ACH + BCH DMAs, then the cursor advance
    constexpr int M_ADV = ACH + BCH, NMS = M_ADV + 1;
    constexpr int SLOT0 = 0;
    auto stage_micro = [&](auto m_, auto p_) {
        constexpr int Mi = decltype(m_)::value, P = (decltype(p_)::value + L) % NS;
#if defined(H3_EXP_BSKIP)                                   // knock-out: activations staged for the first tap only (what sharing a tile across taps could save)
        if constexpr (Mi < ACH) dma_a(std::integral_constant<int, Mi>{}, P);
        else if constexpr (Mi < M_ADV) { if (s_t == 0) dma_b(std::integral_constant<int, Mi - ACH>{}, P); }
        else advance();
#elif defined(H3_EXP_ASKIP)                                 // knock-out: weights staged every other k-block
        if constexpr (Mi < ACH) { if ((s_kb & 1) == 0) dma_a(std::integral_constant<int, Mi>{}, P); }
        else if constexpr (Mi < M_ADV) dma_b(std::integral_constant<int, Mi - ACH>{}, P);
        else advance();
#else
        if constexpr (Mi < ACH) dma_a(std::integral_constant<int, Mi>{}, P);
        else if constexpr (Mi < M_ADV) dma_b(std::integral_constant<int, Mi - ACH>{}, P);
        else advance();
#endif
    };
#ifndef H3_PLAN
#define H3_PLAN 0
#endif
    // H3_PLAN (experiment builds): 0 = staging spread over the whole iteration, interleaved with the fragment reads; 1 = fragment
    // reads first, staging in the second half; 2 = staging first, fragment reads in the second half
    auto slot_of_stage = [](int m) constexpr {
        if (KT == 1 && NP == 3 && H3_PLAN == 1) return 6 + (m * 6) / NMS;
        if (KT == 1 && NP == 3 && H3_PLAN == 2) return (m * 6) / NMS;
        return SLOT0 + (m * (NMI - SLOT0)) / NMS;
    };
    // slot of fragment read q of step s (reads of the NEXT step): spread over the first NMF - 2 slots of the step
    auto slot_of_frag = [](int s, int q) constexpr {
        if (KT == 1 && NP == 3 && H3_PLAN == 1) return (q * 6) / C::NFR;
        if (KT == 1 && NP == 3 && H3_PLAN == 2) return 5 + (q * 6) / C::NFR;
        return s * C::NMF + (q * (C::NMF - 2)) / C::NFR + (C::NMF > 4 ? 1 : 0);
    };
    // MFMA n of a step: groups of four, smallest terms first: (A part, B part) = (h,l) (l,h) (h,h); NP = 1: (h,h)
    auto step = [&](auto n_, auto s_, auto p_, auto f_) {
        constexpr int Nn = decltype(n_)::value, Ss = decltype(s_)::value, P = decltype(p_)::value, F = decltype(f_)::value;
        constexpr int grp_ = Nn / (2 * TM), i = (Nn % (2 * TM)) / 2, jn = Nn % 2;
        constexpr int PA = NP == 1 ? 0 : (grp_ == 1 ? 1 : 0);
        constexpr int PB = NP == 1 ? 0 : (grp_ == 0 ? 1 : 0);
        acc[i][jn] = H3_MFMA(fa[F][i][PA], fb[F][jn][PB], acc[i][jn]);
        constexpr int slot = Ss * C::NMF + Nn;
        h3_for<0, NMS>([&](auto m_) {
            if constexpr (slot_of_stage(decltype(m_)::value) == slot) stage_micro(m_, p_);
        });
        h3_for<0, C::NFR>([&](auto q_) {
            if constexpr (slot_of_frag(Ss, decltype(q_)::value) == slot) {
                if constexpr (Ss + 1 < KT) read_frag(q_, std::integral_constant<int, F ^ 1>{}, std::integral_constant<int, Ss + 1>{}, P);
                else read_frag(q_, std::integral_constant<int, F ^ 1>{}, I0{}, (P + 1) % NS);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    // end of an iteration: the tile the NEXT iteration reads fragments from has landed (this wave's share) and everybody
    // is done with the stage the next iteration restages.  Not __syncthreads(): its fence waits for vmcnt(0).
    constexpr int PEND = (L - 2) * (ACH + BCH);                          // LDS-DMAs that may stay in flight across the barrier: tiles it + 3 .. it + L
#if defined(H3_EXP_NOBAR)
#define H3_END_OF_ITER asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PEND) : "memory");
#elif defined(H3_EXP_NOVMWAIT)
#define H3_END_OF_ITER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
#define H3_END_OF_ITER asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PEND) : "memory");
#endif
#define H3_ITER(U)                                                                                                \
    {                                                                                                             \
        h3_for<0, KT>([&](auto s_) {                                                                              \
            h3_for<0, C::NMF>([&](auto n_) {                                                                      \
                step(n_, s_, std::integral_constant<int, (U) % NS>{},                                             \
                     std::integral_constant<int, ((U) * KT + decltype(s_)::value) % 2>{});                        \
            });                                                                                                   \
        });                                                                                                       \
        H3_END_OF_ITER                                                                                            \
    }
    // stage = it % NS, first register set = (it KT) % 2: period lcm(NS, KT odd ? 2 : 1)
    constexpr int PERIOD = (KT % 2 == 0) ? NS : (NS % 2 == 0 ? NS : 2 * NS);
    int it = 0;
    for (; it + PERIOD <= n_it; it += PERIOD) {
        h3_for<0, PERIOD>([&](auto u_) { H3_ITER(decltype(u_)::value) });
    }
    h3_for<0, PERIOD - 1>([&](auto u_) {
        if (it + decltype(u_)::value < n_it) H3_ITER(decltype(u_)::value)
    });
#undef H3_ITER
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");       // re-staged tail tiles have landed before LDS is reused

    if (WK > 1) {                                                       // sum the K groups: wk = 0 += wk = 1, 2, 3 in order
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        f32x4 v = {acc[i][jn][4 * e4], acc[i][jn][4 * e4 + 1], acc[i][jn][4 * e4 + 2], acc[i][jn][4 * e4 + 3]};
                        red[((((wk - 1) * (WM * WN) + wmn) * 2 * TM + i * 2 + jn) * 4 + e4) * 64 + lane] = v;
                    }
        }
        __syncthreads();
        if (wk == 0)
#pragma unroll
            for (int s = 1; s < WK; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const f32x4 v = red[((((s - 1) * (WM * WN) + wmn) * 2 * TM + i * 2 + jn) * 4 + e4) * 64 + lane];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[i][jn][4 * e4 + c] += v[c];
                        }
    }
    epilogue<TM, 2>(a, acc, m0, n0, wm, wn, l31, lk, S, slice, wk == 0, grp, n_end);
}

template <int WM, int WN, int WK, int KT, int NS, int NP, int TM = 2>
__global__ void __launch_bounds__(64 * WM * WN * WK, H3_OCC)
conv_gemm_h3_kernel(const H3Multi mm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    h3_tile<WM, WN, WK, KT, NS, NP, TM>(mm.p[h3_problem_of(mm)], smem);
}

// ONE conv on two tile shapes (as_conv_gemm_h3_launch_mix): the entries listed first run the 128 x 128 tile, the ones flagged `small`
// the 128 x 64 tile (four waves as two K halves of two 64 x 64 blocks).  A workgroup runs one of the two bodies; registers and LDS are
// the larger of the two (the small tile stages two k-blocks per stage: 72 KB, two workgroups per CU as before).
__global__ void __launch_bounds__(256, H3_OCC)
conv_gemm_h3_mix_kernel(const H3Multi mm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (two entries, constant indices: a dynamic index into the argument struct here made the compiler copy it to scratch)
    if ((int)blockIdx.x >= mm.p[1].wg0) h3_tile<2, 1, 2, 1, 3, 3>(mm.p[1], smem);
    else h3_tile<2, 2, 1, 1, 3, 3>(mm.p[0], smem);
}

int as_conv_gemm_h3_launch_mix(const ConvGemmArgs* a, int col_split, hipStream_t stream)
{
    using CB = H3Cfg<2, 2, 1, 1, 3, 3>;
    using CS = H3Cfg<2, 1, 2, 1, 3, 3>;
    constexpr int LDS = CB::LDS > CS::LDS ? CB::LDS : CS::LDS;
    if (!a || a->n_prod != 3 || a->n_groups > 1 || col_split <= 0 || col_split >= a->N || col_split % CB::BN) return AS_EINVAL;
    if ((double)as_kbx(a->K) * 4.0 * ((a->src_col ? a->N_in : a->N) + 1.0) * 16.0 >= 2147483648.0) return AS_EINVAL;
    AS_LDS_OPT_IN((&conv_gemm_h3_mix_kernel), LDS);
    H3Multi mm;
    memset(&mm, 0, sizeof(mm));
    mm.n = 2;
    int wg = 0;
    for (int i = 0; i < 2; ++i) {
        H3Prob& p = mm.p[i];
        p.a = *a;
        if (h3_pack_taps(p.a, &p.tp) != AS_OK) return AS_EINVAL;
        p.col0 = i == 0 ? 0 : col_split;
        p.col1 = i == 0 ? col_split : a->N;
        p.small = i;
        p.tiles = as_cdiv(a->M, 128) * as_cdiv(p.col1 - p.col0, i == 0 ? CB::BN : CS::BN);
        p.S = 1;
        p.wg0 = wg;
        p.wgs = (p.tiles + 7) & ~7;
        wg += p.wgs;
    }
    hipLaunchKernelGGL(conv_gemm_h3_mix_kernel, dim3(wg), dim3(256), LDS, stream, mm);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

template <int WM, int WN, int WK, int KT, int NS, int NP, int TM = 2>
static int launch_h3(const ConvGemmArgs* const* a, const int* S, int n, hipStream_t stream)
{
    using C = H3Cfg<WM, WN, WK, KT, NS, NP, TM>;
    AS_LDS_OPT_IN((&conv_gemm_h3_kernel<WM, WN, WK, KT, NS, NP, TM>), C::LDS);
    H3Multi mm;
    memset(&mm, 0, sizeof(mm));
    mm.n = n;
    int wg = 0;
    for (int i = 0; i < n; ++i) {
        H3Prob& p = mm.p[i];
        p.a = *a[i];
        if (h3_pack_taps(p.a, &p.tp) != AS_OK) return AS_EINVAL;
        const int tiles_n = p.a.n_groups > 1 ? p.a.n_groups * as_cdiv(p.a.group_cols, C::BN) : as_cdiv(p.a.N, C::BN);
        p.tiles = as_cdiv(p.a.M, C::BM) * tiles_n;
        p.col0 = 0;
        p.col1 = p.a.N;
        p.S = S[i];
        p.wg0 = wg;
        p.wgs = (p.tiles * p.S + 7) & ~7;
        wg += p.wgs;
    }
    hipLaunchKernelGGL((conv_gemm_h3_kernel<WM, WN, WK, KT, NS, NP, TM>), dim3(wg), dim3(C::NT), C::LDS, stream, mm);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// Pipeline shape per tile (k-blocks per wave and iteration, LDS stages): AS_H3_KT / AS_H3_NS select the alternatives
// that are compiled in (tuning runs only)
#ifndef H3_KT
#define H3_KT 1
#endif
template <int WM, int WN, int WK>
static int launch_h3_tile(const ConvGemmArgs* const* a, const int* S, int n, hipStream_t stream)
{
    if (a[0]->n_prod == 1) return launch_h3<WM, WN, WK, 1, 3, 1>(a, S, n, stream);
#ifdef AS_EXPERIMENTS   // (-DAS_EXPERIMENTS builds only: the shipped library carries the one pipeline shape the path launches)
    const char *ekt = getenv("AS_H3_KT"), *ens = getenv("AS_H3_NS");
    const int kt = ekt ? atoi(ekt) : H3_KT, ns = ens ? atoi(ens) : 3;
    if constexpr (H3Cfg<WM, WN, WK, 2, 3, 3>::LDS <= 160 * 1024) {
        if (kt == 2) return launch_h3<WM, WN, WK, 2, 3, 3>(a, S, n, stream);
    }
    if (ns == 2) return launch_h3<WM, WN, WK, 1, 2, 3>(a, S, n, stream);
    if constexpr (H3Cfg<WM, WN, WK, 1, 4, 3>::LDS <= 160 * 1024) {
        if (ns == 4) return launch_h3<WM, WN, WK, 1, 4, 3>(a, S, n, stream);
    }
#endif
    return launch_h3<WM, WN, WK, 1, 3, 3>(a, S, n, stream);
}

int as_conv_gemm_h3_launch(const ConvGemmArgs* const* a, const int* S, int n, int choice, hipStream_t stream)
{
    if (n < 1 || n > H3_MAXP) return AS_EINVAL;
    for (int i = 0; i < n; ++i) {
        if (a[i]->n_prod != a[0]->n_prod || S[i] < 1) return AS_EINVAL;
        if ((double)as_kbx(a[i]->K) * 4.0 * ((a[i]->src_col ? a[i]->N_in : a[i]->N) + 1.0) * 16.0 >= 2147483648.0) return AS_EINVAL;   // 32-bit offsets in the descriptor
    }
    switch (choice) {
#ifdef AS_EXPERIMENTS
    case 42:                                                            // 256 x 128, a wave owns 128 x 64
        if (a[0]->n_prod == 1) return AS_EINVAL;
        return launch_h3<2, 2, 1, 1, 3, 3, 4>(a, S, n, stream);
#endif
    case 2:                                                             // 32 x 128 (a wave owns 32 x 64): the vocoder's 32-channel stage, M <= 32
        if (a[0]->n_prod == 1) return AS_EINVAL;
        {
            static const bool ns3 = getenv("AS_TILE2_NS3") != nullptr;   // (tuning: the three-stage form)
            if (!ns3) return launch_h3<1, 2, 2, 1, 2, 3, 1>(a, S, n, stream);
        }
        return launch_h3<1, 2, 2, 1, 3, 3, 1>(a, S, n, stream);
    case 22: return launch_h3_tile<2, 2, 1>(a, S, n, stream);
    case 21: return launch_h3_tile<2, 1, 2>(a, S, n, stream);
    case 12: return launch_h3_tile<1, 2, 2>(a, S, n, stream);
    case 14:                                                            // 64 x 256: four waves side by side, each 64 x 64 over the whole k
        if (a[0]->n_prod == 1) return launch_h3_tile<1, 2, 2>(a, S, n, stream);   // (h-only operands: half a weight chunk per thread -- the 64 x 128 tile)
        return launch_h3<1, 4, 1, 1, 3, 3>(a, S, n, stream);
    case 11: return launch_h3_tile<1, 1, 4>(a, S, n, stream);
    default: return AS_EINVAL;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// X fp32 [K][ldx] -> Xh[(K/16 up to a multiple of 4)][p*2 + kh][N + 1][8] fp16, x = h + l; optional LeakyReLU first;
// column N and rows >= K are zero.
// A thread owns 8 consecutive k of 4 columns -- lane, lane + 64, lane + 128, lane + 192 of its wave's 256: every load is 256
// contiguous bytes of a row and every store 1 KB of an image row (with four CONSECUTIVE columns per lane a store put 16 bytes every 64:
// 32 quarter-filled lines per instruction).  Range-checked buffer loads: offsets of rows >= K fall outside the descriptor (zero).
__global__ void __launch_bounds__(256)
split_f16x2_kernel(const float* __restrict__ x, int ldx, int K, int N, int lrelu, float slope, u32x4_t* __restrict__ xh)
{
    const int wcol = (blockIdx.x * 256 + (threadIdx.x & ~63)) * 4;      // the wave's first column
    const int col = wcol + (threadIdx.x & 63);
    const int g = blockIdx.y;                                           // 8-row group: kb = g / 2, kh = g % 2
    if (wcol > N) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)(((unsigned)(K - 1) * ldx + N) * 4u), 0x00020000);
    float v[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = g * 8 + r;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[r][c] = buf_load1(rs, (k < K && col + 64 * c < N) ? (unsigned)(k * ldx + col + 64 * c) * 4u : OOB, 0);
    }
    const size_t NX = (size_t)N + 1;
    const size_t base = ((size_t)(g >> 1) * 4 + (g & 1)) * NX + col;    // plane p*2 + kh of k-block kb
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (col + 64 * c > N) break;                                    // (column N itself is written: the zero column)
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float e = v[r][c];                                          // zero past N and past K
            if (lrelu) e = e > 0.f ? e : slope * e;
            t[r] = e;
        }
        u32x4_t h, l;
        split2(t, h, l);
        xh[base + 64 * c] = h;
        xh[base + 64 * c + 2 * NX] = l;
    }
}

extern "C" size_t as_split_f16x2_bytes(int K, int N)
{
    if (K <= 0 || N <= 0) return 0;
    return (size_t)as_kbx(K) * 4 * ((size_t)N + 1) * 16;
}

int as_split_f16x2_launch(const float* x, int ldx, int K, int N, int lrelu, float slope, uint16_t* xh, hipStream_t stream)
{
    if ((double)K * ldx * 4.0 >= 2147483648.0) return AS_EINVAL;        // 32-bit offsets in the buffer descriptor
    hipLaunchKernelGGL(split_f16x2_kernel, dim3(as_cdiv(N + 1, 1024), 2 * as_kbx(K)), dim3(256), 0, stream, x, ldx, K, N, lrelu, slope,
                       reinterpret_cast<u32x4_t*>(xh));
    AS_CHECK_LAUNCH();
    return AS_OK;
}

extern "C" int as_split_f16x2_f32(const float* x, int ldx, int K, int N, int in_act, float in_slope, uint16_t* xh, as_stream_t stream)
{
    if (!x || !xh || K <= 0 || N < 0 || ldx < N || (in_act != 0 && in_act != 2) || !(fabsf(in_slope) <= 3.0e38f)) return AS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(xh) & 15) != 0) return AS_EINVAL;
    if (N == 0) return AS_OK;
    AsProfScope prof__(AS_CLS_OTHER, 0, 8.0 * K * (double)N, (hipStream_t)stream);
    return as_split_f16x2_launch(x, ldx, K, N, in_act == 2, in_slope, xh, (hipStream_t)stream);
}

// host-side weight preparation (see the header)
extern "C" size_t as_prep_weight_f16x2_sc_bytes(int G, int Cout, int Cin, int T, int Cin2)
{
    if (G <= 0 || Cout <= 0 || Cin <= 0 || T <= 0 || Cin2 < 0) return 0;
    return (size_t)G * ((size_t)T * as_kbx(Cin) + (Cin2 ? as_kbx(Cin2) : 0)) * 4 * (size_t)Cout * 16;
}
extern "C" size_t as_prep_weight_f16x2_bytes(int G, int Cout, int Cin, int T) { return as_prep_weight_f16x2_sc_bytes(G, Cout, Cin, T, 0); }

extern "C" int as_prep_weight_f16x2_sc_host(const float* w, const float* w2, int G, int Cout, int Cin, int T, int Cin2, uint16_t* wh,
                                            float* scale_out)
{
    if (!w || !wh || !scale_out || G <= 0 || Cout <= 0 || Cin <= 0 || T <= 0 || Cin2 < 0 || ((Cin2 > 0) != (w2 != nullptr))) return AS_EINVAL;
    const size_t n = (size_t)G * Cout * Cin * T, n2 = (size_t)G * Cout * Cin2;
    float mx = 0.f;
    for (size_t i = 0; i < n + n2; ++i) {
        const float x = i < n ? w[i] : w2[i - n];
        const float v = x < 0 ? -x : x;
        if (!(v <= 3.0e38f)) return AS_EINVAL;                           // NaN / inf
        mx = v > mx ? v : mx;
    }
    int e = 0;
    float scale = 1.0f;
    if (mx > 0.f) {
        frexpf(mx, &e);                                                  // mx = f * 2^e, f in [0.5, 1): mx in [2^(e-1), 2^e)
        scale = ldexpf(1.0f, 14 - e);                                    // max |w| * scale in [2^13, 2^14)
    }
    const int KBx = as_kbx(Cin), KBx2 = Cin2 ? as_kbx(Cin2) : 0;
    const size_t blocks = (size_t)T * KBx + KBx2;                        // k-blocks of one weight set
    const size_t total = as_prep_weight_f16x2_sc_bytes(G, Cout, Cin, T, Cin2) / 2;
    for (size_t i = 0; i < total; ++i) wh[i] = 0;
    auto put = [&](size_t block, int k, int m, float x) {                // element k of k-block `block` (absolute), output row m
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        const int kh = (k >> 3) & 1, e8 = k & 7;
        const size_t base = block * 4 * (size_t)Cout * 8;
        uint16_t hb, lb;
        __builtin_memcpy(&hb, &h, 2);
        __builtin_memcpy(&lb, &l, 2);
        wh[base + ((size_t)(0 * 2 + kh) * Cout + m) * 8 + e8] = hb;
        wh[base + ((size_t)(1 * 2 + kh) * Cout + m) * 8 + e8] = lb;
    };
    for (int g = 0; g < G; ++g)
        for (int m = 0; m < Cout; ++m) {
            for (int k = 0; k < Cin; ++k)
                for (int t = 0; t < T; ++t) put((size_t)g * blocks + (size_t)t * KBx + (k >> 4), k, m, w[(((size_t)g * Cout + m) * Cin + k) * T + t] * scale);
            for (int k = 0; k < Cin2; ++k) put((size_t)g * blocks + (size_t)T * KBx + (k >> 4), k, m, w2[((size_t)g * Cout + m) * Cin2 + k] * scale);
        }
    *scale_out = scale;
    return AS_OK;
}
extern "C" int as_prep_weight_f16x2_host(const float* w, int G, int Cout, int Cin, int T, uint16_t* wh, float* scale_out)
{
    return as_prep_weight_f16x2_sc_host(w, nullptr, G, Cout, Cin, T, 0, wh, scale_out);
}

// ----------------------------------------------------------------------------------------------------------------
// AdaIN + LeakyReLU written DIRECTLY as the split image (the output of models.py:189-197's norm -> actv feeds nothing but
// the following conv, so the fp32 activations never exist): per-(channel, utterance) statistics exactly as
// adain_kernel (elementwise.hip) computes them -- one wave per channel, the same summation order -- then a thread takes
// 8 channels of one column, normalises, applies LeakyReLU(0.2), splits and stores three 16-byte rows.
// Workgroup = (8-channel group = one (k-block, k-half) of the image, utterance).
// ----------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ float h3_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// AsAdainArgs addressing (include/artspeech_hip.h): gamma(u, c) = gb[gb_off[u] + c * gb_sc] (gb_off NULL: u * ldgb), beta at channel
// C + c; utterance u reads its input at columns src_off[u].. (NULL: col_off[u]) and writes at col_off[u].. -- at 2 col_off[u]
// with the fused depthwise ConvTranspose1d(k3, s2, p1, op1) x2 up-sampler (models.py:172,195): out[2i] = a[i] w1 + b,
// out[2i+1] = a[i] w2 + a[i+1] w0 + b, and x_up (fp32) gets the nearest x2 copy of x (the block's shortcut, models.py:184).
//
// A WAVE owns one 16-byte row group of the image -- 8 consecutive channels (k-block kb, k-half kh) of one utterance -- from the first
// load to the last store: lane = column (i = lane + 64 j), the 8 x RV values stay in registers through the two statistics passes
// (summation order of adain_kernel: a lane adds its elements in ascending order, then the wave's butterfly), the normalisation and
// the split, so the input is read ONCE, nothing goes through LDS and there is no barrier (the round-1 / early round-2 kernel computed
// the statistics per wave of four channels, parked them in LDS and re-read x from L2 for the second pass: 15-25 us per launch).  The
// up-sampler's a[i+1] is the next lane's value (lane 63: lane 0 of the next register).  Utterances longer than 64 RV frames take
// the same mapping with loops over global memory (three passes).  Workgroup = 4 waves = two k-blocks of one utterance.
template <bool UP>
__global__ void __launch_bounds__(256)
adain_image_kernel(const AsAdainArgs a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kb = blockIdx.x * 2 + (wave >> 1), kh = wave & 1, u = blockIdx.y;
    const int C = a.C;
    if (kb >= as_kbx(C)) return;
    const size_t NX = (size_t)a.N + 1;
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(a.yh);
    const size_t plane = ((size_t)kb * 4 + kh) * NX;                    // h part; the l part two planes on
    if (u == 0 && lane == 0) {                                          // the zero column
        xs[plane + a.N] = u32x4_t{0u, 0u, 0u, 0u};
        xs[plane + 2 * NX + a.N] = u32x4_t{0u, 0u, 0u, 0u};
    }
    const int o0 = a.col_off[u], L = a.col_w ? a.col_w[u] : a.col_off[u + 1] - o0;
    if (L <= 0) return;
    const int s0 = a.src_off ? a.src_off[u] : o0;
    const size_t gbase = a.gb_off ? (size_t)a.gb_off[u] : (size_t)u * a.ldgb;
    const int c0 = kb * 16 + kh * 8;
    // per-channel constants (wave-uniform addresses); channels >= C (the image's padding rows) come out as zeros
    float g1[8], bt[8], w0[8], w1[8], w2[8], pb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int c = c0 + r < C ? c0 + r : C - 1;
        g1[r] = 1.0f + a.gb[gbase + (size_t)c * a.gb_sc];
        bt[r] = a.gb[gbase + (size_t)(C + c) * a.gb_sc];
        if (UP) { w0[r] = a.pool_w[c * 3 + 0]; w1[r] = a.pool_w[c * 3 + 1]; w2[r] = a.pool_w[c * 3 + 2]; pb[r] = a.pool_b[c]; }
    }
    const float* xr0 = a.x + s0;
    const size_t at0 = plane + (UP ? 2 * (size_t)o0 : (size_t)o0);
    float mean[8], rstd[8];
    // utterances of up to 64 RV columns live in registers (one read of x): RV = 4 for L <= 256, 8 up to 512 -- with one fixed RV = 8 the
    // 200-column utterances of the benchmark issued twice the loads and arithmetic they needed (13.6 -> 11.8 us per launch)
    auto resident = [&](auto rv_) {
        constexpr int RV = decltype(rv_)::value;
        float v[8][RV];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float* xr = xr0 + (size_t)(c0 + r < C ? c0 + r : C - 1) * a.ldx;
#pragma unroll
            for (int j = 0; j < RV; ++j) v[r][j] = xr[min(lane + 64 * j, L - 1)];        // loads only; masked below
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < RV; ++j)
                if (lane + 64 * j < L) sacc += v[r][j];
            mean[r] = h3_wave_sum(sacc) / (float)L;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float vacc = 0.f;
#pragma unroll
            for (int j = 0; j < RV; ++j)
                if (lane + 64 * j < L) { const float d = __fsub_rn(v[r][j], mean[r]); vacc = __fmaf_rn(d, d, vacc); }   // (explicit: the same bits in every kernel that computes these statistics)
            rstd[r] = 1.0f / sqrtf(h3_wave_sum(vacc) / (float)L + 1e-5f);
        }
        if (UP && a.x_up) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (c0 + r >= C) continue;
                float* ur = a.x_up + (size_t)(c0 + r) * a.ld_up + 2 * o0;
#pragma unroll
                for (int j = 0; j < RV; ++j) {
                    const int i = lane + 64 * j;
                    if (i < L) { ur[2 * i] = v[r][j]; ur[2 * i + 1] = v[r][j]; }
                }
            }
        }
        // normalise in place; everything outside (channel >= C, column >= L) becomes zero
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < RV; ++j)
                v[r][j] = (c0 + r < C && lane + 64 * j < L) ? as_adain_val(v[r][j], mean[r], rstd[r], g1[r], bt[r], a.lrelu) : 0.f;
#pragma unroll
        for (int j = 0; j < RV; ++j) {
            const int i = lane + 64 * j;
            if (64 * j >= L) break;                                      // (wave-uniform)
            if (!UP) {
                float t[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) t[r] = v[r][j];
                u32x4_t h, l;
                split2(t, h, l);
                if (i < L) {
                    xs[at0 + i] = h;
                    xs[at0 + i + 2 * NX] = l;
                }
            } else {
                float e0[8], e1[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float dn = __shfl_down(v[r][j], 1);
                    const float wr = j + 1 < RV ? __shfl(v[r][j + 1 < RV ? j + 1 : j], 0) : 0.f;
                    const float an = i + 1 < L ? (lane == 63 ? wr : dn) : 0.f;
                    float o0v, o1v;
                    as_convt_pair(v[r][j], an, w0[r], w1[r], w2[r], pb[r], &o0v, &o1v);
                    e0[r] = c0 + r < C ? o0v : 0.f;
                    e1[r] = c0 + r < C ? o1v : 0.f;
                }
                u32x4_t h, l, h2, l2;
                split2(e0, h, l);
                split2(e1, h2, l2);
                if (i < L) {
                    xs[at0 + 2 * i] = h;
                    xs[at0 + 2 * i + 2 * NX] = l;
                    xs[at0 + 2 * i + 1] = h2;
                    xs[at0 + 2 * i + 1 + 2 * NX] = l2;
                }
            }
        }
    };
    if (L <= 256) { resident(std::integral_constant<int, 4>{}); return; }
    if (L <= 512) { resident(std::integral_constant<int, 8>{}); return; }
    // long utterances: the same mapping, three passes over global memory
    {
        size_t rowoff[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) rowoff[r] = (size_t)(c0 + r < C ? c0 + r : C - 1) * a.ldx;
        float acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = 0.f;
        // the eight channels side by side, four columns per trip: 32 loads in flight (a channel's elements still add up in ascending order)
        int i = lane;
        for (; i + 192 < L; i += 256) {
            float t[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 8; ++r) t[q][r] = xr0[rowoff[r] + i + 64 * q];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] += t[q][r];
        }
        for (; i < L; i += 64)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] += xr0[rowoff[r] + i];
#pragma unroll
        for (int r = 0; r < 8; ++r) { mean[r] = h3_wave_sum(acc[r]) / (float)L; acc[r] = 0.f; }
        i = lane;
        for (; i + 192 < L; i += 256) {
            float t[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 8; ++r) t[q][r] = xr0[rowoff[r] + i + 64 * q];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 8; ++r) { const float d = __fsub_rn(t[q][r], mean[r]); acc[r] = __fmaf_rn(d, d, acc[r]); }
        }
        for (; i < L; i += 64)
#pragma unroll
            for (int r = 0; r < 8; ++r) { const float d = __fsub_rn(xr0[rowoff[r] + i], mean[r]); acc[r] = __fmaf_rn(d, d, acc[r]); }
#pragma unroll
        for (int r = 0; r < 8; ++r) rstd[r] = 1.0f / sqrtf(h3_wave_sum(acc[r]) / (float)L + 1e-5f);
    }
    // third pass: two columns per trip (i and i + 64), every load before the first use
    for (int i0 = lane; i0 < L; i0 += 128) {
        float xi[2][8], xn[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = min(i0 + 64 * q, L - 1);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float* xr = xr0 + (size_t)(c0 + r < C ? c0 + r : C - 1) * a.ldx;
                xi[q][r] = xr[i];
                if (UP) xn[q][r] = xr[min(i + 1, L - 1)];
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = i0 + 64 * q;
            if (i >= L) break;
            float a0[8], a1[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool ok = c0 + r < C;
                a0[r] = ok ? as_adain_val(xi[q][r], mean[r], rstd[r], g1[r], bt[r], a.lrelu) : 0.f;
                a1[r] = (UP && ok && i + 1 < L) ? as_adain_val(xn[q][r], mean[r], rstd[r], g1[r], bt[r], a.lrelu) : 0.f;
                if (UP && a.x_up && ok) {
                    float* ur = a.x_up + (size_t)(c0 + r) * a.ld_up + 2 * o0 + 2 * i;
                    ur[0] = xi[q][r];
                    ur[1] = xi[q][r];
                }
            }
            u32x4_t h, l;
            if (!UP) {
                split2(a0, h, l);
                xs[at0 + i] = h;
                xs[at0 + i + 2 * NX] = l;
            } else {
                float e0[8], e1[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    as_convt_pair(a0[r], a1[r], w0[r], w1[r], w2[r], pb[r], &e0[r], &e1[r]);
                    if (c0 + r >= C) { e0[r] = 0.f; e1[r] = 0.f; }
                }
                split2(e0, h, l);
                xs[at0 + 2 * i] = h;
                xs[at0 + 2 * i + 2 * NX] = l;
                split2(e1, h, l);
                xs[at0 + 2 * i + 1] = h;
                xs[at0 + 2 * i + 1 + 2 * NX] = l;
            }
        }
    }
}

extern "C" int as_adain_image_f32(const AsAdainArgs* args_host, as_stream_t stream)
{
    if (!args_host) return AS_EINVAL;
    const AsAdainArgs& a = *args_host;
    if (!a.x || !a.gb || !a.col_off || !a.yh || a.C <= 0 || a.U <= 0 || a.N < 0 || a.gb_sc <= 0 || (!a.gb_off && a.ldgb <= 0)) return AS_EINVAL;
    if ((a.pool_w == nullptr) != (a.pool_b == nullptr) || (a.x_up && !a.pool_w)) return AS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a.yh) & 15) != 0) return AS_EINVAL;
    if (a.N == 0) return AS_OK;
    AsProfScope prof__(AS_CLS_ADAIN, 0, 8.0 * a.C * (double)a.N, (hipStream_t)stream);
    if (a.pool_w) hipLaunchKernelGGL(adain_image_kernel<true>, dim3(as_kbx(a.C) / 2, a.U), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(adain_image_kernel<false>, dim3(as_kbx(a.C) / 2, a.U), dim3(256), 0, (hipStream_t)stream, a);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

extern "C" int as_adain_split_f32(const float* x, int ldx, int C, const float* gamma_beta, int ldgb, const int32_t* col_off, int B,
                                  int N, int lrelu, uint16_t* xs, as_stream_t stream)
{
    if (!x || !gamma_beta || !col_off || !xs || C <= 0 || B <= 0 || N < 0 || ldgb < 2 * C) return AS_EINVAL;
    AsAdainArgs a = {};
    a.x = x; a.ldx = ldx; a.C = C; a.gb = gamma_beta; a.ldgb = ldgb; a.gb_sc = 1; a.col_off = col_off; a.U = B; a.N = N; a.lrelu = lrelu; a.yh = xs;
    return as_adain_image_f32(&a, stream);
}

// x [B][ldx] fp32 (one K-vector per utterance: the style vectors) -> the split operand image of its transpose [K][B]: the
// AdaIN fc layers of the whole model then run as ONE conv GEMM with the utterances as columns (models.py:237)
__global__ void __launch_bounds__(256)
rows_image_kernel(const float* __restrict__ x, int ldx, int K, int B, u32x4_t* __restrict__ xh)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int groups = 2 * as_kbx(K);
    if (i >= groups * (B + 1)) return;
    const int g = i / (B + 1), b = i % (B + 1);
    float t[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] = (b < B && g * 8 + r < K) ? x[(size_t)b * ldx + g * 8 + r] : 0.f;
    u32x4_t h, l;
    split2(t, h, l);
    const size_t at = ((size_t)(g >> 1) * 4 + (g & 1)) * (B + 1) + b;
    xh[at] = h;
    xh[at + 2 * (size_t)(B + 1)] = l;
}

extern "C" int as_rows_image_f32(const float* x, int ldx, int K, int B, uint16_t* xh, as_stream_t stream)
{
    if (!x || !xh || K <= 0 || B <= 0 || ldx < K || (reinterpret_cast<uintptr_t>(xh) & 15) != 0) return AS_EINVAL;
    AsProfScope prof__(AS_CLS_OTHER, 0, 8.0 * K * (double)B, (hipStream_t)stream);
    hipLaunchKernelGGL(rows_image_kernel, dim3(as_cdiv((long)2 * as_kbx(K) * (B + 1), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, K, B,
                       reinterpret_cast<u32x4_t*>(xh));
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
                        float eps, int relu, u32x4_t* __restrict__ xs, int KBx)
{
    __shared__ float red[LNS_PARTS][LNS_COLS + 1];
    const int col = threadIdx.x % LNS_COLS, part = threadIdx.x / LNS_COLS;
    const int j = blockIdx.x * LNS_COLS + col;
    const bool ok = j < N;
    const int grp = gamma2 ? j / n_split : 0;
    const float* gamma = gamma1 + (ptrdiff_t)grp * (gamma2 - gamma1);
    const float* beta = beta1 + (ptrdiff_t)grp * (beta2 - beta1);
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
            const float d = __fsub_rn(v[i][r], mean);
            if (c < C) q2 = __fmaf_rn(d, d, q2);                        // (explicit: the same bits in the reduction kernel that computes this LayerNorm, conv_gemm.hip)
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
                for (int p = 0; p < 2; ++p) xs[((size_t)(g >> 1) * 4 + (g & 1) + 2 * p) * NX + N] = u32x4_t{0u, 0u, 0u, 0u};
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
                o = __fmaf_rn(__fmul_rn(__fsub_rn(v[i][r], mean), rs), gamma[c], beta[c]);
                if (relu) o = o < 0.f ? 0.f : o;
            }
            t[r] = o;
        }
        u32x4_t h, l;
        split2(t, h, l);
        const size_t at = ((size_t)(g >> 1) * 4 + (g & 1)) * NX + j;
        xs[at] = h;
        xs[at + 2 * NX] = l;
    }
}

extern "C" int as_channel_layernorm_split_f32(const float* x, int ldx, int C, int N, const float* gamma, const float* beta,
                                              const float* gamma2, const float* beta2, int n_split, float eps, int relu, uint16_t* xs,
                                              as_stream_t stream)
{
    if (!x || !xs || !gamma || !beta || C <= 0 || C > 8 * LNS_PARTS * LNS_MAXG || N <= 0 || ldx < N ||
        ((gamma2 == nullptr) != (beta2 == nullptr)) || (gamma2 && n_split <= 0) || (reinterpret_cast<uintptr_t>(xs) & 15) != 0)
        return AS_EINVAL;
    const int KBx = as_kbx(C);
    AsProfScope prof__(AS_CLS_LN, 8.0 * C * N, 8.0 * C * (double)N, (hipStream_t)stream);
    hipLaunchKernelGGL(channel_ln_split_kernel, dim3(as_cdiv(N, LNS_COLS)), dim3(1024), 0, (hipStream_t)stream, x, ldx, C, N, gamma, beta,
                       gamma2, beta2, n_split, eps, relu, reinterpret_cast<u32x4_t*>(xs), KBx);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
