// Optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; when on, every launcher brackets its kernel with an event pair.  as_prof_collect
// synchronises the recorded events (it is the only entry point of the library that blocks).
#include "common.h"
#include "artspeech_hip.h"
#include <vector>
#include <mutex>

struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; };
static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::mutex g_mu;

AsProfScope::AsProfScope(int cls, double flops, double bytes, hipStream_t s) : idx(-1), stream(s)
{
    if (!g_on) return;
    ProfRec r;
    r.cls = cls; r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    hipEventRecord(r.a, s);
    std::lock_guard<std::mutex> g(g_mu);
    g_recs.push_back(r);
    idx = (int)g_recs.size() - 1;
}

AsProfScope::~AsProfScope()
{
    if (idx < 0) return;
    std::lock_guard<std::mutex> g(g_mu);
    hipEventRecord(g_recs[idx].b, stream);
}

extern "C" int as_prof_enable(int on)
{
    std::lock_guard<std::mutex> g(g_mu);
    for (auto& r : g_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    g_recs.clear();
    g_on = on != 0;
    return AS_OK;
}

extern "C" int as_prof_collect(double* ms, double* flops, double* bytes, int32_t* launches, int n_classes)
{
    if (!ms || !flops || !bytes || !launches || n_classes <= 0) return AS_EINVAL;
    std::lock_guard<std::mutex> g(g_mu);
    for (int i = 0; i < n_classes; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; launches[i] = 0; }
    for (auto& r : g_recs) {
        AS_CHECK(hipEventSynchronize(r.b));
        float t = 0.f;
        AS_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        const int c = r.cls < n_classes ? r.cls : n_classes - 1;
        ms[c] += t; flops[c] += r.flops; bytes[c] += r.bytes; launches[c] += 1;
    }
    return AS_OK;
}
