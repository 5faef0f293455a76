// Optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; when on, every launcher brackets its kernel with an event pair.  as_prof_collect
// synchronises the recorded events (it is the only entry point of the library that blocks).
// If the environment variable AS_PROF_CSV names a file, as_prof_collect also appends one line per launch
// (class, tag, ms, algorithmic flop, algorithmic bytes) to it -- per-shape tuning data.
#include "common.h"
#include "artspeech_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; char tag[160]; };
static bool g_on = false;
static thread_local double g_hint_flops = 0, g_hint_bytes = 0;      // what the caller knows about the next launch (as_prof_hint)
static std::vector<ProfRec> g_recs;
static std::mutex g_mu;

AsProfScope::AsProfScope(int cls, double flops, double bytes, hipStream_t s, const char* tag) : idx(-1), stream(s)
{
    if (!g_on) return;
    ProfRec r;
    r.cls = cls; r.flops = flops > 0 ? flops : g_hint_flops; r.bytes = bytes > 0 ? bytes : g_hint_bytes;
    g_hint_flops = g_hint_bytes = 0;
    r.tag[0] = 0;
    if (tag) { strncpy(r.tag, tag, sizeof(r.tag) - 1); r.tag[sizeof(r.tag) - 1] = 0; }
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    std::lock_guard<std::mutex> g(g_mu);
    g_recs.push_back(r);
    idx = (int)g_recs.size() - 1;
}

AsProfScope::~AsProfScope()
{
    if (idx < 0) return;
    std::lock_guard<std::mutex> g(g_mu);
    (void)hipEventRecord(g_recs[idx].b, stream);
}

// Algorithmic flop / bytes of the NEXT launch on this host thread, for launchers whose arguments do not say (geometry tables live
// on the device): the module-level entry points know the layouts and tell the profiler.  No effect while profiling is off.
extern "C" int as_prof_hint(double flops, double bytes)
{
    if (g_on) { g_hint_flops = flops; g_hint_bytes = bytes; }
    return AS_OK;
}

extern "C" int as_prof_enable(int on)
{
    std::lock_guard<std::mutex> g(g_mu);
    for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_recs.clear();
    g_on = on != 0;
    return AS_OK;
}

extern "C" int as_prof_collect(double* ms, double* flops, double* bytes, int32_t* launches, int n_classes)
{
    if (!ms || !flops || !bytes || !launches || n_classes <= 0) return AS_EINVAL;
    std::lock_guard<std::mutex> g(g_mu);
    for (int i = 0; i < n_classes; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; launches[i] = 0; }
    const char* csv = getenv("AS_PROF_CSV");
    struct Csv {                                          // (closed on every path out of this function)
        FILE* f;
        ~Csv() { if (f) fclose(f); }
    } out{csv ? fopen(csv, "a") : nullptr};
    FILE* f = out.f;
    for (auto& r : g_recs) {
        AS_CHECK(hipEventSynchronize(r.b));
        float t = 0.f;
        AS_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        const int c = r.cls < n_classes ? r.cls : n_classes - 1;
        ms[c] += t; flops[c] += r.flops; bytes[c] += r.bytes; launches[c] += 1;
        if (f) fprintf(f, "%d,%s,%.6f,%.0f,%.0f\n", r.cls, r.tag, t, r.flops, r.bytes);
    }
    return AS_OK;
}

// What an event pair adds to the kernel it brackets.  An event is a marker in the queue: the time between two of them is the kernel's own
// duration PLUS the command processor's work around it (~4-5 us per bracket on MI355X), which a kernel trace (rocprofv3) does not count.
// Measured here so that bench.py can report durations that agree with the trace: brackets of 1, 2, 4 and 8 empty kernels; the intercept of
// the line through their mean times is the part that belongs to the bracket, not to a kernel.
__global__ void as_prof_null_kernel() {}

extern "C" int as_prof_bracket_overhead(as_stream_t stream_, double* overhead_ms)
{
    if (!overhead_ms) return AS_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int ks[4] = {1, 2, 4, 8}, REP = 48;
    double mean[4] = {0, 0, 0, 0};
    std::vector<hipEvent_t> ev(2 * REP);
    for (auto& e : ev) AS_CHECK(hipEventCreate(&e));
    for (int c = 0; c < 4; ++c) {
        for (int warm = 0; warm < 2; ++warm) {
            for (int r = 0; r < REP; ++r) {
                AS_CHECK(hipEventRecord(ev[2 * r], stream));
                for (int k = 0; k < ks[c]; ++k) hipLaunchKernelGGL(as_prof_null_kernel, dim3(1), dim3(64), 0, stream);
                AS_CHECK(hipEventRecord(ev[2 * r + 1], stream));
            }
            AS_CHECK(hipStreamSynchronize(stream));
        }
        double sum = 0;
        for (int r = 0; r < REP; ++r) {
            float t = 0.f;
            AS_CHECK(hipEventElapsedTime(&t, ev[2 * r], ev[2 * r + 1]));
            sum += t;
        }
        mean[c] = sum / REP;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    // least squares line mean = o + e k over k = 1, 2, 4, 8
    double sk = 0, sm = 0, skk = 0, skm = 0;
    for (int c = 0; c < 4; ++c) { sk += ks[c]; sm += mean[c]; skk += (double)ks[c] * ks[c]; skm += ks[c] * mean[c]; }
    const double slope = (4 * skm - sk * sm) / (4 * skk - sk * sk);
    const double o = (sm - slope * sk) / 4;
    *overhead_ms = o > 0 ? o : 0;
    return AS_OK;
}

// What the matrix cores of THIS device sustain on RANDOM fp16 operands (bench.py's roofline leg reports it beside the nominal peak).
// The chip lowers its clock under matrix-core load on non-trivial data (MI355X_MICROARCH.md, DVFS give-back): a bare loop of the conv
// GEMM's own MFMA (v_mfma_f32_32x32x16_f16, a wave's 2 x 2 accumulator tiles, the twelve products of one f16x3 k-block per trip) runs
// far below the 2516.6 TFLOP/s of the data sheet, with the operands in registers and -- closer to the kernel -- re-read from LDS at
// the kernel's ratio (8 ds_read_b128 per 12 MFMAs).  No global memory traffic, no barriers: an upper bound for any real kernel.
typedef float prof_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 prof_f16x8 __attribute__((ext_vector_type(8)));
template <int LDSF>
__global__ void __launch_bounds__(256) as_prof_mfma_kernel(const prof_f16x8* __restrict__ src, float* __restrict__ out, int iters)
{
    __shared__ __attribute__((aligned(16))) prof_f16x8 lds[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[(blockIdx.x * 2048 + i) & 65535];
    __syncthreads();
    prof_f16x8 f[8];                                                     // A h x2, A l x2, B h x2, B l x2
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = lds[(q * 64 + lane + wave * 17) & 2047];
    prof_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a >> 1][a & 1][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (LDSF) {
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = lds[((q + (it & 7) * 8) * 64 + lane) & 2047];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], f[6 + j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 + i], f[4 + j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], f[4 + j], acc[i][j], 0, 0, 0);
    }
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[a >> 1][a & 1][e];
    out[blockIdx.x * 256 + tid] = sum;
}

extern "C" int as_prof_mfma_sustained(as_stream_t stream_, double* tflops_registers, double* tflops_lds_fed)
{
    if (!tflops_registers || !tflops_lds_fed) return AS_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int blocks = 512, iters = 12000;                               // two workgroups per CU, ~6 ms per launch
    const size_t n = (size_t)65536 * 8;
    std::vector<_Float16> h(n);
    unsigned long long x = 88172645463325252ull;                         // xorshift: uniform in [-1, 1)
    for (size_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[i] = (_Float16)((float)((x >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f);
    }
    void *src = nullptr, *out = nullptr;
    AS_CHECK(hipMalloc(&src, n * 2));
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) { (void)hipFree(src); return (int)hipErrorOutOfMemory; }
    int rc = AS_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        rc = (int)hipErrorUnknown;
    for (int mode = 0; mode < 2 && rc == AS_OK; ++mode) {
        float ms = 0.f;
        for (int rep = 0; rep < 4; ++rep) {                              // the first launches let the clock settle; the last one is timed
            if (rep == 3) (void)hipEventRecord(e0, stream);
            if (mode == 0) hipLaunchKernelGGL(as_prof_mfma_kernel<0>, dim3(blocks), dim3(256), 0, stream, static_cast<const prof_f16x8*>(src), static_cast<float*>(out), iters);
            else hipLaunchKernelGGL(as_prof_mfma_kernel<1>, dim3(blocks), dim3(256), 0, stream, static_cast<const prof_f16x8*>(src), static_cast<float*>(out), iters);
        }
        (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) { rc = (int)hipErrorUnknown; break; }
        const double flop = (double)blocks * 4 * iters * 12 * 32768.0;
        (mode == 0 ? *tflops_registers : *tflops_lds_fed) = flop / ms / 1e9;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(src);
    (void)hipFree(out);
    return rc;
}
