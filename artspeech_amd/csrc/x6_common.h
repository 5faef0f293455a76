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

