// Shared helpers for the gfx950 kernels behind include/artspeech_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AS_OK 0
#define AS_EINVAL (-1)

// Launch-error check: returns the hipError_t (>0) from the calling extern "C" function.
#define AS_CHECK_LAUNCH()                        \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

#define AS_CHECK(call)                           \
    do {                                         \
        hipError_t e__ = (call);                 \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

static inline int as_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#define AS_WAVE 64

// kernel classes for the optional event profiler (prof.hip)
enum { AS_CLS_GEMM = 0, AS_CLS_ADAIN = 1, AS_CLS_LN = 2, AS_CLS_ATTN = 3, AS_CLS_LSTM = 4, AS_CLS_MAS = 5, AS_CLS_OTHER = 6, AS_N_CLS = 7 };
struct AsProfScope {
    int idx;
    hipStream_t stream;
    AsProfScope(int cls, double flops, double bytes, hipStream_t s, const char* tag = nullptr);
    ~AsProfScope();
};
