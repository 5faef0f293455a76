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
