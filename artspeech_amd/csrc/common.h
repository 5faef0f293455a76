// Shared helpers for the gfx950 kernels behind include/artspeech_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AS_OK 0
#define AS_EINVAL (-1)
#ifndef AS_EDEVICE
#define AS_EDEVICE (-3)
#endif

// device-side status words (status.hip; kinds in include/artspeech_hip.h)
unsigned* as_status_words_device();     // device address of the current device's words, or NULL
int as_status_peek();                   // host view: bit k = kind k raised

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

// The opt-in for more than 64 KiB of dynamic LDS is a per-device function attribute: one of these per call site sets it once on every
// device the process drives (the library keeps per-device state elsewhere too: status.hip), from any host thread.
struct AsLdsOptIn {
    unsigned long long done = 0;        // bit d: set on device d (read / written with __atomic builtins)
    int ensure(const void* fn, int bytes)
    {
        int d = 0;
        hipError_t e = hipGetDevice(&d);
        if (e != hipSuccess) return (int)e;
        const unsigned long long bit = 1ull << (d & 63);
        if (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit) return AS_OK;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
        return AS_OK;
    }
};
#define AS_LDS_OPT_IN(fn, bytes)                                                   \
    do {                                                                           \
        static AsLdsOptIn opt__;                                                   \
        const int r__ = opt__.ensure(reinterpret_cast<const void*>(fn), (bytes));  \
        if (r__ != AS_OK) return r__;                                              \
    } while (0)

#define AS_WAVE 64

// Capacity layouts of as_forward_test's second half (as_forward_io.frame_cap; csrc/model.hip "dynamic layouts", elementwise.hip): the
// utterances' frame counts exist on the device only -- frame_off [B + 1], half rate, packed from column 0 -- and every table the kernels of
// the second half read is derived from them by ONE launch.  Layout i of {0: half rate, 1: mel rate, 2: the batch three times at half rate
// (group g from column g * cap1), 3: the same at mel rate}: widths w[i], first columns off[i] (one closing entry), column descriptors
// meta[i], *nvalid[i] = the valid leading columns (of a group).  Any pointer may be NULL.
struct AsDynGeo {
    const int32_t* frame_off;
    int32_t B, cap1, n_seg;
    int32_t seg_first[17];            // segments = the submissions of a merged call: utterances [seg_first[s], seg_first[s + 1])
    int32_t seg_cap[16];              // ... and the half-rate frames the segment's output slot holds
    int32_t* seg_frame_off[16];       // optional: the segment's own frame offsets [its utterances + 1], from 0
    int32_t* w[4];
    int32_t* off[4];
    unsigned long long* meta[4];
    int32_t* nvalid[4];
    int32_t* src3;                    // [3 B]: where utterance b of group g reads the shared half-rate input (= off[0][b])
    int32_t* tof;                     // frame -> token map [cap1]: entries past the last frame are set to token 0
    unsigned* status;
};
int as_dyn_geometry_launch(const AsDynGeo& g, hipStream_t stream);
// mel [rows][ld_src] packed from column 0 -> every segment's columns to its own slot of dst: dst_s[r][i] = src[r][2 off[first_s] + i]
struct AsSegScatter {
    const float* src; int32_t ld_src, rows, n_seg;
    const int32_t* frame_off;
    int32_t seg_first[17];
    int32_t seg_cap[16];
    float* dst[16]; int32_t ld_dst[16];
};
int as_seg_scatter_launch(const AsSegScatter& a, hipStream_t stream);

// kernel classes for the optional event profiler (prof.hip)
enum { AS_CLS_GEMM = 0, AS_CLS_ADAIN = 1, AS_CLS_LN = 2, AS_CLS_ATTN = 3, AS_CLS_LSTM = 4, AS_CLS_MAS = 5, AS_CLS_OTHER = 6, AS_N_CLS = 7 };
struct AsProfScope {
    int idx;
    hipStream_t stream;
    AsProfScope(int cls, double flops, double bytes, hipStream_t s, const char* tag = nullptr);
    ~AsProfScope();
};

#ifdef __HIPCC__
// raise status kind `kind` (a kernel's failure path only): one system-scope store into pinned host memory
static __device__ __forceinline__ void as_status_raise(unsigned* words, int kind)
{
    if (words) __hip_atomic_store(words + kind, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// AdaIN1d value and the fused depthwise ConvTranspose1d(k3, s2, p1, op1) pair, written with explicit roundings so that every
// kernel that evaluates them (adain_kernel: fp32 output; adain_image_kernel: operand image) produces the same bits whatever
// the compiler would contract.
static __device__ __forceinline__ float as_adain_val(float x, float mean, float rstd, float one_plus_gamma, float beta, int lrelu)
{
    const float o = __fmaf_rn(one_plus_gamma, __fmul_rn(__fsub_rn(x, mean), rstd), beta);
    return (lrelu && o < 0.f) ? __fmul_rn(0.2f, o) : o;
}
// out[2i] = a[i] w1 + b ; out[2i+1] = a[i] w2 + a[i+1] w0 + b
static __device__ __forceinline__ void as_convt_pair(float a0, float a1, float w0, float w1, float w2, float pb, float* o0, float* o1)
{
    *o0 = __fmaf_rn(a0, w1, pb);
    *o1 = __fadd_rn(__fmaf_rn(a1, w0, __fmul_rn(a0, w2)), pb);
}
#endif
