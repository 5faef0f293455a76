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

struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; char tag[64]; };
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
    FILE* f = csv ? fopen(csv, "a") : nullptr;
    for (auto& r : g_recs) {
        AS_CHECK(hipEventSynchronize(r.b));
        float t = 0.f;
        AS_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        const int c = r.cls < n_classes ? r.cls : n_classes - 1;
        ms[c] += t; flops[c] += r.flops; bytes[c] += r.bytes; launches[c] += 1;
        if (f) fprintf(f, "%d,%s,%.6f,%.0f,%.0f\n", r.cls, r.tag, t, r.flops, r.bytes);
    }
    if (f) fclose(f);
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
