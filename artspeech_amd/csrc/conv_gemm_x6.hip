// The conv GEMM on the bf16 matrix cores at fp32 accuracy ("bf16x6").
//
// Every fp32 operand is split EXACTLY into three bf16 numbers, x = h + m + l (round-to-nearest-even at each step;
// 8 + 8 + 8 significand bits), and a product x*w is accumulated in fp32 from the six partial products of weight
// >= 2^-16:  h*l + l*h + m*m + h*m + m*h + h*h.  The three dropped terms (m*l, l*m, l*l) are below 2^-24 |x*w|, the
// rounding error of an fp32 product, so the result is fp32-accurate (scripts/exp/bf16x6_numerics.py: the whole path
// run this way is as far from the reference's golden mels, 5e-6, as run in fp32; with three terms it is 8e-5).
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each):
// 192 vs 512 matrix-core cycles per 32x32x16 block.
//
// Weights are split once at load time (ops.prep_weight) and stored as the kernel stages them:
//     Wx[tap][kb = k/16][p*2 + kh][m][8]  bf16        p = part (h, m, l), kh = (k%16)/8, last index k%8
// (k-blocks per tap padded with zeros to a multiple of 4) so a (tile, k-block) is six contiguous runs of BM x 16
// bytes, and the LDS image of a k-tile is lane-linear: it is filled by LDS-DMA (buffer_load_dwordx4 ... lds),
// no registers and no ds_write for the weights.
// Activations are fp32 in the packed-frames layout: a thread loads 8 consecutive k of ONE column with eight dword
// buffer loads (a wave = 64 consecutive columns of one k row: 256 contiguous bytes), splits them in registers and
// writes three 16-byte LDS rows -- the transposition [k][n] -> [n][k] the MFMA operand needs costs nothing extra,
// a tap that falls outside the column's utterance is an out-of-range buffer offset (hardware returns 0), and no
// alignment promise is needed (one staging for every layout).
//
// LDS image per k-block: [p*2 + kh][row][8 bf16]: a 32x32x16 operand fragment (lane: row = lane&31, k-half =
// lane>>5, 8 bf16) is one ds_read_b128 and each 16-lane group of that read covers 256 contiguous bytes (no bank
// conflict); the staging writes are contiguous 16-byte rows per 8-lane group (conflict-free too).
//
// Workgroup = 4 waves, each owning a 64x64 output block (2x2 MFMA tiles, 64 accumulator registers) of one
// 16-deep k-block per k-tile: WM x WN x WK waves along M, N and K (WM*WN*WK = 4), tile = 64*WM x 64*WN, k-tile =
// 16*WK.  Small outputs use WK > 1 (the waves split K and sum their accumulators through LDS at the end, in a fixed
// order) so that a 64x64 tile still feeds four matrix cores with 2x2 register blocking.
#include "common.h"
#include "conv_gemm.h"
#include <type_traits>

// timing-only experiment builds (scripts/build_exp.sh): a zero-record descriptor drops that operand's memory traffic
#ifdef X6_EXP_NOW
#define X6_EXP_W(n) 0
#else
#define X6_EXP_W(n) (n)
#endif
#ifdef X6_EXP_NOX
#define X6_EXP_X(n) 0
#else
#define X6_EXP_X(n) (n)
#endif

#include "x6_common.h"

// The staging work of one k-tile as a list of micro-operations with approximate instruction counts, dealt out over
// the 24 MFMAs of the iteration in order, by cumulative weight (x6 kernel, "slots").
template <int ACH, int UB>
struct X6Plan {
    static constexpr int M_TAPA = 0;                      // tap byte of the activation cursor             (8 instr)
    static constexpr int M_TAPB = 1;                      // column offset of that tap, first-row offset   (9)
    static constexpr int M_LD = 2;                        // 8*UB activation loads                         (3 each)
    static constexpr int M_XADV = M_LD + 8 * UB;          // advance the activation cursor                 (8)
    static constexpr int M_ASOFF = M_XADV + 1;            // weight tile offset                            (3)
    static constexpr int M_DMA = M_ASOFF + 1;             // ACH LDS-DMA pieces                            (2 each)
    static constexpr int M_WADV = M_DMA + ACH;            // advance the weight cursor                     (8)
    static constexpr int M_FR = M_WADV + 1;               // 12 fragment reads of the next tile            (1 each)
    static constexpr int M_SP = M_FR + 12;                // 4*UB element pairs x 2 halves of the split    (5, 6)
    static constexpr int M_ST = M_SP + 8 * UB;            // 3*UB LDS stores                               (1 each)
    static constexpr int NM = M_ST + 3 * UB;
    // Slot (0..23 = behind which MFMA) of each piece.  The categories are interleaved rather than issued one after
    // the other: a slot then mixes a VALU chain (split), a scalar chain (addresses, cursors) and a memory
    // instruction, which one wave can issue back to back, where a slot holding a single dependent chain stalls on it
    // (PMC: a third of the wave's cycles were such issue stalls).  Order constraints: TAPA < TAPB < loads < XADV;
    // ASOFF < DMA < WADV (in slot order AND in micro-op order within a slot); both halves of a pair in order and before the stores; everything before the barrier.
    static constexpr int slot_of(int m)
    {
        if (m == M_TAPA) return 0;
        if (m == M_TAPB) return 1;
        if (m < M_XADV) return 8 + (m - M_LD) / UB;                      // loads: slots 8..15
        if (m == M_XADV) return 16;
        if (m == M_ASOFF) return 1;
        if (m < M_WADV) return 2 + ((m - M_DMA) * 6) / ACH;              // LDS-DMA pieces first (slots 2..7): the longest latency
        if (m == M_WADV) return 17;
        if (m < M_SP) return 6 + (m - M_FR);                             // fragment reads: slots 6..17
        if (m < M_ST) return ((m - M_SP) * 22) / (8 * UB);               // split halves: spread over slots 0..21
        return 22 + ((m - M_ST) * 2) / (3 * UB);                         // stores: slots 22, 23
    }
};

// Software pipeline, per workgroup and k-tile `it` (P = it & 1):
//   LDS stage P holds tile it (its fragments are already in registers, read during it-1), stage P^1 holds tile it+1.
//   top:    request the fragments of tile it+1 (register set P^1), the activation rows of tile it+3 (register set
//           P^1) and the LDS-DMA of the weights of tile it+2 into stage P (free: everyone read it before the barrier)
//   middle: 24 MFMAs on register set P; behind the first eight, split the activations of tile it+2 (register set P,
//           loaded one iteration ago) and write them to stage P
//   end:    vmcnt(0) lgkmcnt(0) + barrier
// so a lone workgroup on a CU never starts a k-tile by waiting for LDS, and every global access has a whole k-tile
// (~800 cycles of MFMA) to land.
template <int WM, int WN, int WK, bool LRELU>
__global__ void __launch_bounds__(64 * WM * WN * WK)
conv_gemm_x6_kernel(const ConvGemmArgs a, const X6Taps tp)
{
    using C = X6Cfg<WM, WN, WK>;
    constexpr int BM = C::BM, BN = C::BN, ACH = C::ACH, UB = C::UB, NT = C::NT;
    static_assert((NT == 256 || NT == 512) && ACH * NT == WK * 6 * BM && UB * NT == WK * 2 * BN, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    X6_STAMP(t0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn % WM, wn = wmn / WM;
    const int l31 = lane & 31, lk = lane >> 5;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tile = logical_tile();
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
    const int KB = a.Kp >> 4;                                            // k-blocks per tap that hold weights
    const int KBx = (KB + 3) & ~3;                                       // k-blocks per tap in the image (zero padded)

    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>((a.n_split > 0 && n0 >= a.n_split) ? a.Wx2 : a.Wx), 0, X6_EXP_W((int)((unsigned)a.T * KBx * 6u * a.M * 16u)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, X6_EXP_X((int)((unsigned)a.K * a.ldx * 4u)), 0x00020000);

    // weights: chunk c = tid + NT i of the k-tile image [kblk][p*2+kh][row] goes to LDS offset 16 c (lane-linear, so
    // one LDS-DMA wave-instruction moves 64 chunks); rows past M are out of range (nothing that is stored reads them)
    unsigned a_voff[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = tid + NT * i, kblk = c / (6 * BM), rem = c % (6 * BM), pk = rem / BM, row = rem % BM;
        a_voff[i] = (m0 + row) < a.M ? (unsigned)(((kblk * 6 + pk) * a.M + m0 + row) * 16) : OOB;
    }
    const int kt_per_tap = (KB + WK - 1) / WK;
    const int nkt_all = a.T * kt_per_tap;
    const int S = gridDim.y;                                             // split-K over (tap, k-tile)
    const int kt_lo = (int)((long)nkt_all * blockIdx.y / S);
    const int n_it = (int)((long)nkt_all * (blockIdx.y + 1) / S) - kt_lo;

    // Two cursors over the (tap, k-block) sequence; the main loop is branch-free: past the end of the sequence a
    // cursor stays on the last tile (loading it again into a stage nobody reads is harmless).
    int wa_t = kt_lo / kt_per_tap, wa_kb = (kt_lo - wa_t * kt_per_tap) * WK;
    int xb_t = wa_t, xb_kb = wa_kb;
    auto advance = [&](int& t, int& kb) {
        int nkb = kb + WK, nt = t;
        if (nkb >= KB) { nkb = 0; nt += 1; }
        const bool ok = nt < a.T;
        kb = ok ? nkb : kb;
        t = ok ? nt : t;
    };
    auto dma_a = [&](int buf) {                                          // weights of a tile: global -> LDS, no registers
        const int a_soff = (wa_t * KBx + wa_kb) * 6 * a.M * 16;
        unsigned char* st = smem + buf * C::STAGE + wave * 1024;
        // (device pass only: with this builtin in the body hipcc 7.2's HOST pass silently drops the kernel's launch stub)
#if __HIP_DEVICE_COMPILE__
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
#ifdef X6_EXP_REGSTAGE
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, a_voff[i], a_soff, 0);
            *reinterpret_cast<u32x4*>(smem + buf * C::STAGE + (tid + NT * i) * 16) = v;
#else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(st + i * (NT * 16)), 16, a_voff[i], a_soff, 0, 0);
#endif
        }
#endif
        advance(wa_t, wa_kb);
    };
    // activations: unit u = tid + NT i -> column u % BN (the same for every i), k rows 8 g .. 8 g + 7, g = u / BN
    const int j = n0 + tid % BN;
    const int g0 = __builtin_amdgcn_readfirstlane(tid / BN);             // wave-uniform (BN >= 64)
    unsigned long long md = 0;
    if (a.meta && j < a.N) md = a.meta[j];
    // the weights of tiles 0 and 1 start moving now, behind the descriptor load (loads return in order: waiting for the
    // descriptor must not mean waiting for them): their latency overlaps the tap masks below
    dma_a(0);
    dma_a(1);
    unsigned tapmask = 0;
    int Wj = 0;
    if (j < a.N) {
        if (a.meta) {
            const int h = (int)(md & 0xffff), w = (int)((md >> 16) & 0xffff);
            const int H = (int)((md >> 32) & 0xffff);
            Wj = (int)(md >> 48);
            for (int t = 0; t < a.T; ++t) {                               // offsets from the packed bytes: no loads in this loop
                const int byte = x6_tap_byte(tp, t);
                const int dh = tp.wide ? 0 : (byte >> 4) - 8, dw = tp.wide ? byte - 128 : (byte & 15) - 8;
                if ((unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)Wj) tapmask |= 1u << t;
            }
        } else {
            tapmask = 0xffffffffu;
        }
    }

    const X6TapCol tcol(tp, j, Wj);

    float rb[2][UB][8];
    const int ldx4 = a.ldx * 4, last_row = (a.K - 1) * ldx4;
    auto gload_b = [&](float (&r8)[UB][8]) {
        const int src = tcol.src(x6_tap_byte(tp, xb_t));
        const unsigned b_voff = (((tapmask >> xb_t) & 1u) && src >= 0) ? (unsigned)src * 4u : OOB;
        // rows past K (last k-tile of a tap when K % 16 != 0, or zero-padded k-blocks): read row K-1 again -- their
        // weights are zero rows, and a clamp is two scalar instructions where a validity select is six
        const int base = (xb_kb * 16 + g0 * 8) * ldx4;
#pragma unroll
        for (int i = 0; i < UB; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int soff = base + (i * (NT / BN) * 8 + r) * ldx4;
                r8[i][r] = buf_load1(rsX, b_voff, soff < last_row ? soff : last_row);
            }
        advance(xb_t, xb_kb);
    };
    // split + LDS store of one register set, in two halves (elements 0..3, then 4..7 and the three 16-byte stores) so
    // that each half fits behind one group of four MFMAs
    u32x4 sh[UB], sm[UB], sl[UB];
    auto split_pair = [&](float x0, float x1, u32x4& hv, u32x4& mv, u32x4& lv, int e) {
        unsigned h, m, l;
        if (LRELU) {                                                    // LeakyReLU fused on the operand (models.py:89,142)
            x0 = vmax(x0, a.in_slope * x0);
            x1 = vmax(x1, a.in_slope * x1);
        }
#ifdef X6_EXP_NOSPLIT
        h = __builtin_bit_cast(unsigned, x0);
        m = __builtin_bit_cast(unsigned, x1);
        l = 0;
#else
        h = pk_bf16(x0, x1);                                            // x = h + m + l exactly (both subtractions exact)
        const float r0 = x0 - bf_lo(h), r1 = x1 - bf_hi(h);
        m = pk_bf16(r0, r1);
        l = pk_bf16(r0 - bf_lo(m), r1 - bf_hi(m));
#endif
        hv[e] = h;
        mv[e] = m;
        lv[e] = l;
    };
    auto lstore_b = [&](float (&r8)[UB][8], int buf, int half) {
        unsigned char* st = smem + buf * C::STAGE + WK * C::A_BLK + (tid % BN) * 16;
#pragma unroll
        for (int i = 0; i < UB; ++i) {
#pragma unroll
            for (int e = 2 * half; e < 2 * half + 2; ++e) {
                split_pair(r8[i][2 * e], r8[i][2 * e + 1], sh[i], sm[i], sl[i], e);
            }
            if (half == 1) {
                const int g = g0 + i * (NT / BN), kblk = g >> 1, kh = g & 1;
                unsigned char* b = st + (kblk * 6 + kh) * BN * 16;
                *reinterpret_cast<u32x4*>(b) = sh[i];
                *reinterpret_cast<u32x4*>(b + 2 * BN * 16) = sm[i];
                *reinterpret_cast<u32x4*>(b + 4 * BN * 16) = sl[i];
            }
        }
    };
    // fragment addresses of this wave's k-block: part p at + p * 2 * rows * 16
    const int a_frag = (wk * 6 + lk) * BM * 16 + (wm * 64 + l31) * 16;
    const int b_frag = WK * C::A_BLK + (wk * 6 + lk) * BN * 16 + (wn * 64 + l31) * 16;
    auto read_frags = [&](X6Frags& f, int buf) {
#ifdef X6_EXP_NOFRAG
        return;
#endif
        const unsigned char* st = smem + buf * C::STAGE;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f.a[i][p] = *reinterpret_cast<const bf16x8*>(st + a_frag + p * 2 * BM * 16 + i * 32 * 16);
                f.b[i][p] = *reinterpret_cast<const bf16x8*>(st + b_frag + p * 2 * BN * 16 + i * 32 * 16);
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    X6Frags fr[2];
    // prologue: tiles 0 and 1 staged, activations of tile 2 in flight, fragments of tile 0 requested
    gload_b(rb[0]);
    gload_b(rb[1]);
    lstore_b(rb[0], 0, 0);
    lstore_b(rb[0], 0, 1);
    gload_b(rb[0]);
    lstore_b(rb[1], 1, 0);
    lstore_b(rb[1], 1, 1);
    __syncthreads();
    read_frags(fr[0], 0);
    // every wave must HOLD its fragments of tile 0 before any wave lets tile 2 into stage 0 (the first iteration's
    // LDS-DMA and stores): in the loop the end-of-iteration barrier orders this, here nothing else does -- without it
    // a wave delayed by another kernel sharing the CU read tile 2's weights as tile 0's (seen only under concurrency)
    __syncthreads();

    // A wave issues one instruction per ~4 cycles and an MFMA occupies the matrix core for 32, so the ~150 staging
    // instructions of an iteration are dealt out five or six behind EACH of the 24 MFMAs (measured: issued as four
    // MFMAs then a block of staging, the two simply add up -- 1900 cycles per k-tile instead of ~800; and hipcc's
    // sched_group_barrier pipelines left most slots empty).  micro(m, p) is staging piece m for buffer parity p.
    using PL = X6Plan<ACH, UB>;
    unsigned c_voff = OOB;
    int c_base = 0, c_asoff = 0, c_byte = 0;
    float c_r[UB][4][2];
    auto micro = [&](auto m_, auto p_) {
        constexpr int M = decltype(m_)::value, P = decltype(p_)::value, Q = P ^ 1;
        if constexpr (M == PL::M_TAPA) {
            c_byte = x6_tap_byte(tp, xb_t);
        } else if constexpr (M == PL::M_TAPB) {
            const int src = tcol.src(c_byte);
            c_voff = (((tapmask >> xb_t) & 1u) && src >= 0) ? (unsigned)src * 4u : OOB;
            c_base = (xb_kb * 16 + g0 * 8) * ldx4;
        } else if constexpr (M < PL::M_XADV) {
            constexpr int q = M - PL::M_LD, i = q / 8, r = q % 8;
            const int soff = c_base + (i * (NT / BN) * 8 + r) * ldx4;
            rb[Q][i][r] = buf_load1(rsX, c_voff, soff < last_row ? soff : last_row);
        } else if constexpr (M == PL::M_XADV) {
            advance(xb_t, xb_kb);
        } else if constexpr (M == PL::M_ASOFF) {
            c_asoff = (wa_t * KBx + wa_kb) * 6 * a.M * 16;
        } else if constexpr (M < PL::M_WADV) {
#if __HIP_DEVICE_COMPILE__
            constexpr int i = M - PL::M_DMA;
#ifdef X6_EXP_REGSTAGE
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, a_voff[i], c_asoff, 0);
            *reinterpret_cast<u32x4*>(smem + P * C::STAGE + (tid + NT * i) * 16) = v;
#else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(smem + P * C::STAGE + wave * 1024 + i * (NT * 16)), 16,
                                                     a_voff[i], c_asoff, 0, 0);
#endif
#endif
        } else if constexpr (M == PL::M_WADV) {
            advance(wa_t, wa_kb);
        } else if constexpr (M < PL::M_SP) {
#ifndef X6_EXP_NOFRAG
            constexpr int q = M - PL::M_FR, ab = q / 6, p = (q % 6) / 2, i = q % 2;
            const unsigned char* st = smem + Q * C::STAGE;
            if constexpr (ab == 0) fr[Q].a[i][p] = *reinterpret_cast<const bf16x8*>(st + a_frag + p * 2 * BM * 16 + i * 32 * 16);
            else fr[Q].b[i][p] = *reinterpret_cast<const bf16x8*>(st + b_frag + p * 2 * BN * 16 + i * 32 * 16);
#endif
        } else if constexpr (M < PL::M_ST) {
            constexpr int q = M - PL::M_SP, i = q / 8, e = (q % 8) / 2, half = q % 2;
            if constexpr (half == 0) {                                  // x = h + r
                float x0 = rb[P][i][2 * e], x1 = rb[P][i][2 * e + 1];
                if (LRELU) {
                    x0 = vmax(x0, a.in_slope * x0);
                    x1 = vmax(x1, a.in_slope * x1);
                }
                const unsigned h = pk_bf16(x0, x1);
                c_r[i][e][0] = x0 - bf_lo(h);
                c_r[i][e][1] = x1 - bf_hi(h);
                sh[i][e] = h;
            } else {                                                    // r = m + l
                const unsigned m = pk_bf16(c_r[i][e][0], c_r[i][e][1]);
                sm[i][e] = m;
                sl[i][e] = pk_bf16(c_r[i][e][0] - bf_lo(m), c_r[i][e][1] - bf_hi(m));
            }
        } else {
            constexpr int q = M - PL::M_ST, i = q / 3, p = q % 3;
            const int g = g0 + i * (NT / BN), kblk = g >> 1, kh = g & 1;
            unsigned char* b = smem + P * C::STAGE + WK * C::A_BLK + (tid % BN) * 16 + (kblk * 6 + kh + 2 * p) * BN * 16;
            *reinterpret_cast<u32x4*>(b) = p == 0 ? sh[i] : p == 1 ? sm[i] : sl[i];
        }
    };
    // MFMA number n of the iteration: six groups of four, smallest terms first: (A part, B part) =
    // (h,l) (l,h) (m,m) (h,m) (m,h) (h,h); then the staging pieces of slot n
    auto step = [&](auto n_, auto p_) {
        constexpr int N = decltype(n_)::value, P = decltype(p_)::value;
        constexpr int grp = N / 4, i = (N % 4) / 2, jn = N % 2;
        constexpr int PA = grp == 0 ? 0 : grp == 1 ? 2 : grp == 2 ? 1 : grp == 3 ? 0 : grp == 4 ? 1 : 0;
        constexpr int PB = grp == 0 ? 2 : grp == 1 ? 0 : grp == 2 ? 1 : grp == 3 ? 1 : grp == 4 ? 0 : 0;
        acc[i][jn] = X6_MFMA(fr[P].a[i][PA], fr[P].b[jn][PB], acc[i][jn]);
        x6_for<0, PL::NM>([&](auto m_) {
            if constexpr (PL::slot_of(decltype(m_)::value) == N) micro(m_, p_);
        });
        __builtin_amdgcn_sched_barrier(0);
    };
#define X6_ITER(P)                                                                                                \
    {                                                                                                             \
        x6_for<0, 24>([&](auto n_) { step(n_, std::integral_constant<int, P>{}); });                              \
        X6_BAR_BEGIN                                                                                              \
        __syncthreads();                                                                                          \
        X6_BAR_END                                                                                                \
    }
    X6_STAMP(t1)
#ifdef X6_EXP_STAMPS
    unsigned long long w_bar = 0;
#endif
    int it = 0;
    for (; it + 1 < n_it; it += 2) {
        X6_ITER(0)
        X6_ITER(1)
    }
    if (it < n_it) X6_ITER(0)
    X6_STAMP(t2)
#undef X6_ITER

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

template <int WM, int WN, int WK, bool LRELU>
static int launch_x6(const ConvGemmArgs& a, int S, hipStream_t stream)
{
    using C = X6Cfg<WM, WN, WK>;
    static bool attr_set = false;
    if (!attr_set) {
        AS_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_x6_kernel<WM, WN, WK, LRELU>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr_set = true;
    }
    X6Taps tp;
    if (x6_pack_taps(a, &tp) != AS_OK) return AS_EINVAL;
    const dim3 grid(as_cdiv(a.M, C::BM) * as_cdiv(a.N, C::BN), S);
    hipLaunchKernelGGL((conv_gemm_x6_kernel<WM, WN, WK, LRELU>), grid, dim3(C::NT), C::LDS, stream, a, tp);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

int as_conv_gemm_x6_launch(const ConvGemmArgs& a, int choice, int S, hipStream_t stream)
{
    const bool lr = a.in_act == 2;
    switch (choice) {
    case 228: return lr ? launch_x6<2, 2, 2, true>(a, S, stream) : launch_x6<2, 2, 2, false>(a, S, stream);   // 8 waves
    case 218: return lr ? launch_x6<2, 1, 4, true>(a, S, stream) : launch_x6<2, 1, 4, false>(a, S, stream);
    case 128: return lr ? launch_x6<1, 2, 4, true>(a, S, stream) : launch_x6<1, 2, 4, false>(a, S, stream);
    case 22: return lr ? launch_x6<2, 2, 1, true>(a, S, stream) : launch_x6<2, 2, 1, false>(a, S, stream);
    case 21: return lr ? launch_x6<2, 1, 2, true>(a, S, stream) : launch_x6<2, 1, 2, false>(a, S, stream);
    case 12: return lr ? launch_x6<1, 2, 2, true>(a, S, stream) : launch_x6<1, 2, 2, false>(a, S, stream);
    default: return lr ? launch_x6<1, 1, 4, true>(a, S, stream) : launch_x6<1, 1, 4, false>(a, S, stream);
    }
}
