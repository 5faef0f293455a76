// The throughput arrangement of the path as a piece of the library: N batches in flight on ONE model, every batch
// one chain on its own stream.  Built on the module-level C ABI only (as_plan_*, as_module_workspace_bytes, as_forward_test): a lane = a
// serial plan (as_plan_set_serial: the step's independent branches back to back) + a HIP stream + its two workspaces + the hipGraphs of
// the batch geometries it has replayed.  The reference has no counterpart (models.py:361-362 processes one utterance at a time; a
// server calling it would keep several requests in flight exactly like this).
//
//   first submit of a geometry on a lane   as_forward_test eagerly (a plan's first call with a geometry uploads its tables: not capturable)
//   second submit                          the same call captured into a hipGraph (needs batch->frames: with predicted durations the
//                                          call reads the frame counts back in the middle and stays eager), instantiated, launched
//   later submits                          one hipGraphLaunch
// A graph bakes the pointers in: it is reused only for the same geometry AND the same as_forward_io (a serving loop keeps one set of
// device buffers per lane and copies requests into them).
#include "common.h"
#include "artspeech_hip.h"
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Lane {
    as_plan* plan = nullptr;
    hipStream_t stream = nullptr;
    void *wa = nullptr, *wb = nullptr;
    size_t na = 0, nb = 0;
    std::unordered_map<std::string, hipGraphExec_t> graphs;      // key -> instantiated graph
    std::unordered_map<std::string, int> seen;                    // key -> eager calls so far
};

void drop_graphs(Lane& L)
{
    for (auto& kv : L.graphs) hipGraphExecDestroy(kv.second);
    L.graphs.clear();
}

bool grow(void** p, size_t* have, size_t need)
{
    if (need <= *have) return true;
    if (*p) hipFree(*p);
    *p = nullptr;
    *have = 0;
    const size_t n = need + need / 8;                             // some slack: lengths vary from batch to batch
    if (hipMalloc(p, n) != hipSuccess) return false;
    *have = n;
    return true;
}

std::string key_of(const as_batch* b, const as_forward_io* io)
{
    std::string k(reinterpret_cast<const char*>(&b->B), sizeof(b->B));
    k.append(reinterpret_cast<const char*>(b->tok_lens), sizeof(int32_t) * b->B);
    k.append(reinterpret_cast<const char*>(b->ref_lens), sizeof(int32_t) * b->B);
    if (b->frames) k.append(reinterpret_cast<const char*>(b->frames), sizeof(int32_t) * b->B);
    k.append(reinterpret_cast<const char*>(io), sizeof(*io));
    return k;
}

}  // namespace

struct as_lanes {
    const as_model* m = nullptr;
    std::vector<Lane> lanes;
    int next = 0;
};

extern "C" int as_lanes_destroy(as_lanes* q)
{
    if (!q) return AS_OK;
    for (Lane& L : q->lanes) {
        if (L.stream) hipStreamSynchronize(L.stream);
        drop_graphs(L);
        if (L.wa) hipFree(L.wa);
        if (L.wb) hipFree(L.wb);
        if (L.plan) as_plan_destroy(L.plan);
        if (L.stream) hipStreamDestroy(L.stream);
    }
    delete q;
    return AS_OK;
}

// (nothing may leave through the C boundary: the bodies below allocate host memory)
static int lanes_create(const as_model* m, int n_lanes, as_lanes** out)
{
    if (!m || !out || n_lanes < 1 || n_lanes > 64) return AS_EINVAL;
    *out = nullptr;
    as_lanes* q = new (std::nothrow) as_lanes;
    if (!q) return (int)hipErrorOutOfMemory;
    q->m = m;
    q->lanes.resize(n_lanes);
    for (Lane& L : q->lanes) {
        int rc = as_plan_create(m, &L.plan);
        if (rc == AS_OK) rc = as_plan_set_serial(L.plan, 1);
        if (rc == AS_OK && hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) rc = (int)hipErrorOutOfMemory;
        if (rc != AS_OK) {
            as_lanes_destroy(q);
            return rc;
        }
    }
    *out = q;
    return AS_OK;
}

extern "C" int as_lanes_create(const as_model* m, int n_lanes, as_lanes** out)
{
    try {
        return lanes_create(m, n_lanes, out);
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}

extern "C" int as_lanes_count(const as_lanes* q) { return q ? (int)q->lanes.size() : 0; }
extern "C" int as_lanes_next(const as_lanes* q) { return q ? q->next : -1; }
extern "C" as_stream_t as_lanes_stream(const as_lanes* q, int lane)
{
    return (q && lane >= 0 && lane < (int)q->lanes.size()) ? static_cast<as_stream_t>(q->lanes[lane].stream) : nullptr;
}

extern "C" int as_lanes_wait(as_lanes* q, int lane)
{
    if (!q || lane >= (int)q->lanes.size()) return AS_EINVAL;
    for (int i = 0; i < (int)q->lanes.size(); ++i)
        if (lane < 0 || lane == i)
            if (hipStreamSynchronize(q->lanes[i].stream) != hipSuccess) return (int)hipErrorUnknown;
    return as_device_status(0) ? AS_EDEVICE : AS_OK;
}

static int lanes_submit(as_lanes* q, const as_batch* batch, const as_forward_io* io, int32_t* frames_host_out, int32_t* lane_out)
{
    if (!q || !batch || !io || batch->B <= 0 || !batch->tok_lens || !batch->ref_lens) return AS_EINVAL;
    const int lane = q->next;
    Lane& L = q->lanes[lane];
    if (lane_out) *lane_out = lane;
    // the lane's previous batch has left its workspaces (and the caller's buffers of that lane)
    if (hipStreamSynchronize(L.stream) != hipSuccess) return (int)hipErrorUnknown;
    const size_t na = as_module_workspace_bytes(q->m, L.plan, AS_MOD_FORWARD_A, batch);
    if (!na) return AS_EINVAL;
    // workspace B depends on the frame counts: known (forced durations / a second pass), or sized for what the output buffer can hold
    std::vector<int32_t> cap;
    as_batch bb = *batch;
    if (!batch->frames) {
        cap.assign(batch->B, std::max(1, io->ld_out / 2 / batch->B));
        bb.frames = cap.data();
    }
    size_t nb = as_module_workspace_bytes(q->m, L.plan, AS_MOD_FORWARD_B, &bb);
    if (!nb) return AS_EINVAL;
    if (na > L.na || nb > L.nb) drop_graphs(L);                   // the graphs hold the old workspaces' addresses
    if (!grow(&L.wa, &L.na, na) || !grow(&L.wb, &L.nb, nb)) return (int)hipErrorOutOfMemory;
    q->next = (lane + 1) % (int)q->lanes.size();

    if (!batch->frames) {
        // predicted durations: the call synchronises once to read the frame counts; AS_ENOSPC = workspace B (sized for a capacity) is too
        // small for what came out -- frames_host_out says what is needed: size B for it and run again (the header's contract)
        std::vector<int32_t> fr(batch->B, 0);
        int32_t* fh = frames_host_out ? frames_host_out : fr.data();
        int rc = as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, fh, L.stream);
        if (rc == AS_ENOSPC) {
            as_batch fit = *batch;
            fit.frames = fh;
            long total = 0;
            for (int b = 0; b < batch->B; ++b) total += fh[b];
            if (2 * total > io->ld_out) return AS_ENOSPC;           // the caller's output buffer itself is too small
            nb = as_module_workspace_bytes(q->m, L.plan, AS_MOD_FORWARD_B, &fit);
            if (!nb) return AS_EINVAL;
            drop_graphs(L);
            if (!grow(&L.wb, &L.nb, nb)) return (int)hipErrorOutOfMemory;
            rc = as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, fh, L.stream);
        }
        return rc;
    }
    const std::string key = key_of(batch, io);
    auto g = L.graphs.find(key);
    if (g != L.graphs.end()) return hipGraphLaunch(g->second, L.stream) == hipSuccess ? AS_OK : (int)hipErrorUnknown;
    int& n_seen = L.seen[key];
    if (n_seen == 0) {                                              // first call with this geometry: tables are uploaded, the stream synchronised
        n_seen = 1;
        if (L.seen.size() > 256) { L.seen.clear(); drop_graphs(L); }
        return as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, frames_host_out, L.stream);
    }
    if (hipStreamBeginCapture(L.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return (int)hipErrorUnknown;
    const int rc = as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, nullptr, L.stream);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(L.stream, &graph);
    if (rc != AS_OK || ec != hipSuccess || !graph) {
        if (graph) hipGraphDestroy(graph);
        if (rc != AS_OK) return rc;
        // not capturable after all: run it eagerly
        return as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, frames_host_out, L.stream);
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) return as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, frames_host_out, L.stream);
    L.graphs[key] = exec;
    if (frames_host_out) memcpy(frames_host_out, batch->frames, sizeof(int32_t) * batch->B);
    return hipGraphLaunch(exec, L.stream) == hipSuccess ? AS_OK : (int)hipErrorUnknown;
}

extern "C" int as_lanes_submit(as_lanes* q, const as_batch* batch, const as_forward_io* io, int32_t* frames_host_out, int32_t* lane_out)
{
    try {
        return lanes_submit(q, batch, io, frames_host_out, lane_out);
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}
