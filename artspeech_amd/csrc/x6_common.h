// Pieces shared by the bf16x6 GEMM kernels (conv_gemm_x6.hip, conv_gemm_x6t.hip).
#pragma once
#include "common.h"
#include "conv_gemm.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ unsigned pk_bf16(float a, float b)      // v_cvt_pk_bf16_f32 (RNE): a -> low half
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
static __device__ __forceinline__ float vmax(float a, float b)          // bare v_max_f32 (no canonicalising pre-max)
{
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
static __device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// 8 fp32 (consecutive k of one column) -> three rows of 8 bf16; x = h + m + l exactly (both subtractions are exact)
static __device__ __forceinline__ void split3(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = x[2 * e], x1 = x[2 * e + 1];
        const unsigned hu = pk_bf16(x0, x1);
        const float r0 = x0 - bf_lo(hu), r1 = x1 - bf_hi(hu);
        const unsigned mu = pk_bf16(r0, r1);
        const float s0 = r0 - bf_lo(mu), s1 = r1 - bf_hi(mu);
        h[e] = hu;
        m[e] = mu;
        l[e] = pk_bf16(s0, s1);
    }
}
#ifdef X6_EXP_NOMFMA
#define X6_MFMA(A, B, C) (C)
#else
#define X6_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#endif


// timing-only experiment build (scripts/build_exp.sh NAME -DX6_EXP_STAMPS): s_memtime segments written over Y[m0 + 0..3][n0]
#ifdef X6_EXP_STAMPS
#define X6_STAMP(t) unsigned long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
#define X6_BAR_BEGIN { unsigned long long b0__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b0__) :: "memory");
#define X6_BAR_END unsigned long long b1__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b1__) :: "memory"); w_bar += b1__ - b0__; }
#else
#define X6_STAMP(t)
#define X6_BAR_BEGIN
#define X6_BAR_END
#endif
#ifdef X6_EXP_STAMPS
#define X6_STAMPS_OUT                                                                                             \
    if (tid == 0) {                                                                                               \
        a.Y[(size_t)(m0 + 0) * a.ldy + n0] = (float)(t1 - t0);                                                    \
        a.Y[(size_t)(m0 + 1) * a.ldy + n0] = (float)(t2 - t1);                                                    \
        a.Y[(size_t)(m0 + 2) * a.ldy + n0] = (float)(t3 - t2);                                                    \
        a.Y[(size_t)(m0 + 3) * a.ldy + n0] = (float)w_bar;                                                        \
    }
#else
#define X6_STAMPS_OUT
#endif

struct X6Frags {
    bf16x8 a[2][3], b[2][3];                                             // [32-row / 32-column tile][part]
};

typedef __attribute__((address_space(3))) void lds_void;

template <int I, int N, typename F>
static __device__ __forceinline__ void x6_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        x6_for<I + 1, N>(f);
    }
}


template <int WM, int WN, int WK>
struct X6Cfg {
    static constexpr int BM = 64 * WM, BN = 64 * WN;
    static constexpr int A_BLK = 6 * BM * 16, B_BLK = 6 * BN * 16;      // bytes per 16-deep k-block
    static constexpr int STAGE = WK * (A_BLK + B_BLK);
    static constexpr int RED = (WK - 1) * WM * WN * 64 * 64 * 4;        // cross-wave K reduction scratch
    static constexpr int LDS = 2 * STAGE > RED ? 2 * STAGE : RED;
    static constexpr int NT = 64 * WM * WN * WK;                         // threads: one wave per (M, N, K) block, 4 or 8 waves
    static constexpr int ACH = WK * 6 * BM / NT;                         // 16-byte weight chunks per thread per k-tile
    static constexpr int UB = WK * 2 * BN / NT;                          // (column, 8 k) activation units per thread
};


// tap offsets packed one byte per tap, (dh+8) << 4 | (dw+8) (|dh|, |dw| <= 7, checked by the host), eight taps per
// word: the k loop then selects a tap with scalar ALU only (a scalar or scratch load inside it would stall the wave)
struct X6Taps {
    unsigned long long w0, w1, w2, w3;
    int wide;                             // 1: every dh = 0 and the byte is dw + 128 (dilated 1-D convs, |dw| <= 127)
};


#ifdef __HIPCC__
// Byte of tap t.  Written with masks: as a select chain hipcc turns it into scalar BRANCHES inside the k loop.
static __device__ __forceinline__ int x6_tap_byte(const X6Taps& tp, int t)
{
    const int s = t >> 3;
    const unsigned long long m0 = 0ull - (unsigned long long)(s == 0), m1 = 0ull - (unsigned long long)(s == 1),
                             m2 = 0ull - (unsigned long long)(s == 2), m3 = 0ull - (unsigned long long)(s == 3);
    const unsigned long long w = (tp.w0 & m0) | (tp.w1 & m1) | (tp.w2 & m2) | (tp.w3 & m3);
    return (int)(w >> ((t & 7) * 8)) & 0xff;
}
// Source column of a tap for a thread staging column j of an image Wj wide, one formula for both encodings:
//   src = j + (byte >> 4) * A + (byte & 15) + C;  narrow: A = Wj, C = -8 Wj - 8;  wide: A = 16, C = -128
struct X6TapCol {
    int A, C;
    __device__ __forceinline__ X6TapCol(const X6Taps& tp, int j, int Wj) : A(tp.wide ? 16 : Wj), C(j + (tp.wide ? -128 : -8 * Wj - 8)) {}
    __device__ __forceinline__ int src(int byte) const { return (byte >> 4) * A + (byte & 15) + C; }
};
#endif

// host: pack the tap offsets of `a` for the kernels (AS_EINVAL if they do not fit a byte)
static inline int x6_pack_taps(const ConvGemmArgs& a, X6Taps* out)
{
    X6Taps tp = {0, 0, 0, 0, 0};
    unsigned long long* w = &tp.w0;
    for (int t = 0; t < a.T; ++t)
        if (a.dh[t] < -7 || a.dh[t] > 7 || a.dw[t] < -7 || a.dw[t] > 7) tp.wide = 1;
    for (int t = 0; t < a.T; ++t) {
        if (tp.wide && (a.dh[t] != 0 || a.dw[t] < -127 || a.dw[t] > 127)) return AS_EINVAL;
        const int byte = tp.wide ? a.dw[t] + 128 : ((a.dh[t] + 8) << 4) | (a.dw[t] + 8);
        w[t >> 3] |= (unsigned long long)byte << ((t & 7) * 8);
    }
    *out = tp;
    return AS_OK;
}
