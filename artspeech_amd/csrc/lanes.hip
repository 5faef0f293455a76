// The throughput arrangement of the path as a piece of the library: N batches in flight on ONE model, every batch
// one chain on its own stream.  Built on the module-level C ABI only (as_plan_*, as_module_workspace_bytes, as_forward_test): a lane = a
// HIP stream + its two workspaces + two serial plans (as_plan_set_serial: the step's independent branches back to back) + the hipGraphs of
// the batch geometries it has replayed.  The reference has no counterpart (models.py:361-362 processes one utterance at a time; a
// server calling it would keep several requests in flight exactly like this).
//
//   first submit of a (geometry, io) pair   as_forward_test eagerly on the lane's EAGER plan
//   second submit                           eagerly on the lane's GRAPH plan (a plan's first call with a geometry uploads its tables: not
//                                           capturable) -- only geometries that come back reach that plan
//   third submit                            the same call captured from the graph plan into a hipGraph (needs batch->frames: with predicted
//                                           durations the call reads the frame counts back in the middle and stays eager), launched
//   later submits                           one hipGraphLaunch
// A graph bakes in the pointers of the caller's buffers, of the workspaces and of the GRAPH plan's geometry tables:
//   * it is reused only for the same geometry AND the same as_forward_io (a serving loop keeps one set of device buffers per lane);
//   * the eager plan's layout cache may be flushed at any entry point (ever new ragged batches: as_plan_set_layout_cap) -- no graph
//     refers to it.  The graph plan never flushes by itself (cap = INT_MAX); its tables die together with the graphs: when a lane would
//     hold more than `graph_cap` graphs, or its workspaces move, the lane's stream is drained, every graph destroyed and the graph plan
//     reset (as_plan_reset_layouts: the table memory is reused, nothing is freed);
//   * nothing here frees device memory while other lanes may be busy (a hipFree synchronises the device): an outgrown workspace is
//     parked until as_lanes_wait(q, -1) / as_lanes_destroy.
#include "common.h"
#include "artspeech_hip.h"
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Lane {
    as_plan* plan = nullptr;                                      // eager calls, workspace queries: flushed whenever its cap says so
    as_plan* gplan = nullptr;                                     // geometries that are (about to be) replayed from graphs; reset with them
    hipStream_t stream = nullptr;
    void *wa = nullptr, *wb = nullptr;
    size_t na = 0, nb = 0;
    std::unordered_map<std::string, hipGraphExec_t> graphs;      // key -> instantiated graph
    std::unordered_set<std::string> once;                         // pairs that ran once, on the eager plan (a bounded memory of candidates)
    std::unordered_set<std::string> met;                          // pairs the graph plan has met eagerly: captured at their next submit
    int64_t n_drops = 0, n_launch = 0, n_eager = 0, n_capture = 0, n_merged = 0;
    // submissions waiting for their group (as_lanes_set_coalesce): host arrays copied, device pointers as given
    struct Pending {
        std::vector<int32_t> tok_lens, ref_lens, frames;
        as_forward_io io;
        float* out_host = nullptr;                                // as_lanes_submit_host: where the submission's mel goes once its group is out
        int32_t ld_out_host = 0;
        int32_t* foff_host = nullptr;                             // ... and, under a frame capacity, its frame offsets
        unsigned long long sum = 0;                               // debug mode: checksum of the device inputs as they were at submit
    };
    std::vector<Pending> pend;
    // as_lanes_submit_host: the lane's own device block -- the inputs of a group's submissions side by side (adjacent column ranges: one
    // batch as they lie) and the group's output
    struct Block {
        void* dev = nullptr;
        size_t bytes = 0;
        int cap_tok = 0, cap_ref = 0, cap_out = 0, cap_utt = 0;  // tokens / reference frames / mel frames (columns) / utterances the block holds
        int used_tok = 0, used_ref = 0, used_out = 0, used_utt = 0;   // ... of which the group that is being filled has taken
        int32_t *tokens = nullptr, *forced = nullptr, *foff = nullptr;   // (foff: frame offsets of submissions under a frame capacity, [utterances + 1] each)
        float *f0 = nullptr, *ema = nullptr, *mel = nullptr, *out = nullptr;
    } blk[2];                                                     // two, alternating from group to group: the next group's copies run under this group's kernels
    int cur = 0;                                                  // the block the group that is being filled lives in
    bool blk_used[2] = {false, false};                            // a group has gone out from it (ev_d2h[i] has been recorded)
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
    unsigned long long* dbg = nullptr;                            // debug mode: one device word for the checksum kernels
};

// the lane's stream is idle (lanes_submit synchronises it first): no graph is running, nothing reads the graph plan's tables
void drop_graphs(Lane& L)
{
    if (L.graphs.empty() && L.met.empty()) return;
    for (auto& kv : L.graphs) (void)hipGraphExecDestroy(kv.second);
    L.graphs.clear();
    for (const std::string& k : L.met) L.once.insert(k);          // (they have been seen; the graph plan has to meet them again)
    L.met.clear();
    (void)as_plan_reset_layouts(L.gplan);
    ++L.n_drops;
}

template <typename T>
void put(std::string& k, const T& v)
{
    k.append(reinterpret_cast<const char*>(&v), sizeof(T));
}

// Named fields only: the struct's padding bytes are whatever the caller's memory held
std::string key_of(const as_batch* b, const as_forward_io* io)
{
    std::string k;
    put(k, b->B);
    k.append(reinterpret_cast<const char*>(b->tok_lens), sizeof(int32_t) * b->B);
    k.append(reinterpret_cast<const char*>(b->ref_lens), sizeof(int32_t) * b->B);
    if (b->frames) k.append(reinterpret_cast<const char*>(b->frames), sizeof(int32_t) * b->B);
    put(k, io->tokens); put(k, io->mel); put(k, io->ld_mel); put(k, io->f0_raw); put(k, io->ema_raw); put(k, io->ld_ema);
    put(k, io->forced_dur); put(k, io->mel_out); put(k, io->ld_out); put(k, io->duration); put(k, io->dur_i); put(k, io->frame_off);
    put(k, io->style); put(k, io->feat12); put(k, io->ld_feat); put(k, io->t_en); put(k, io->a_en); put(k, io->ld_en);
    put(k, io->F0); put(k, io->N); put(k, io->EMA); put(k, io->ld_pred);
    put(k, io->frame_cap);
    if (io->segs) {
        const as_segments& g = *io->segs;
        put(k, g.n);
        for (int i = 0; i < g.n && i < AS_MAX_SEGMENTS; ++i) { put(k, g.first[i]); put(k, g.cap[i]); put(k, g.mel_out[i]); put(k, g.ld_out[i]); put(k, g.frame_off[i]); }
    }
    return k;
}

}  // namespace

struct as_lanes {
    const as_model* m = nullptr;
    std::vector<Lane> lanes;
    std::vector<void*> retired;                                   // outgrown workspaces: freed when every lane is idle
    int next = 0;
    size_t graph_cap = 256;
    int coalesce = 1;                                             // submissions of adjacent buffers launched as ONE as_forward_test call
    bool debug = false;                                           // as_lanes_set_debug / AS_DEBUG=1: a held-back submission's inputs are checksummed at submit and at its group's launch
    as_model_cfg cfg;
    // copy streams of the host submissions, ONE pair for all lanes (created with the first host submission): HIP maps streams onto four
    // hardware queues, and a copy stream per lane put lanes' copies behind other lanes' kernels (measured: + 7-10 % per step where the
    // copies on the lane's own stream cost + 4.7 %)
    hipStream_t h2d = nullptr, d2h = nullptr;
    bool copy_on_lane = false;                                    // AS_LANES_COPY_ON_LANE=1 (measurements): a lane's copies on its own stream
    hipStream_t up(const Lane& L) const { return copy_on_lane ? L.stream : h2d; }
    hipStream_t down(const Lane& L) const { return copy_on_lane ? L.stream : d2h; }
};

namespace {

void free_retired(as_lanes* q)
{
    for (void* p : q->retired) (void)hipFree(p);
    q->retired.clear();
}

bool grow(as_lanes* q, void** p, size_t* have, size_t need)
{
    if (need <= *have) return true;
    if (*p) q->retired.push_back(*p);                             // (not hipFree: it would stall the other lanes' batches)
    *p = nullptr;
    *have = 0;
    const size_t n = need + need / 8;                             // some slack: lengths vary from batch to batch
    if (hipMalloc(p, n) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    *have = n;
    return true;
}

}  // namespace

static int flush_lane(as_lanes* q, int lane);
extern "C" int as_lanes_destroy(as_lanes* q)
{
    if (!q) return AS_OK;
    // submissions still waiting for neighbours were accepted: they go out before the lanes do (their status has nobody left to go to)
    try {
        for (int i = 0; i < (int)q->lanes.size(); ++i)
            if (q->lanes[i].plan && q->lanes[i].gplan && q->lanes[i].stream) (void)flush_lane(q, i);
    } catch (...) {
    }
    for (Lane& L : q->lanes) {
        if (L.stream) (void)hipStreamSynchronize(L.stream);
    }
    if (q->h2d) (void)hipStreamSynchronize(q->h2d);
    if (q->d2h) (void)hipStreamSynchronize(q->d2h);
    for (Lane& L : q->lanes) {
        for (auto& kv : L.graphs) (void)hipGraphExecDestroy(kv.second);
        if (L.wa) (void)hipFree(L.wa);
        if (L.wb) (void)hipFree(L.wb);
        for (int i = 0; i < 2; ++i) {
            if (L.blk[i].dev) (void)hipFree(L.blk[i].dev);
            if (L.ev_h2d[i]) (void)hipEventDestroy(L.ev_h2d[i]);
            if (L.ev_comp[i]) (void)hipEventDestroy(L.ev_comp[i]);
            if (L.ev_d2h[i]) (void)hipEventDestroy(L.ev_d2h[i]);
        }
        if (L.dbg) (void)hipFree(L.dbg);
        if (L.plan) as_plan_destroy(L.plan);
        if (L.gplan) as_plan_destroy(L.gplan);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (q->h2d) (void)hipStreamDestroy(q->h2d);
    if (q->d2h) (void)hipStreamDestroy(q->d2h);
    free_retired(q);
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
    if (as_model_get_cfg(m, &q->cfg) != AS_OK) { delete q; return AS_EINVAL; }
    const char* dbg = getenv("AS_DEBUG");
    q->debug = dbg && *dbg && *dbg != '0';
    q->copy_on_lane = getenv("AS_LANES_COPY_ON_LANE") != nullptr;
    q->lanes.resize(n_lanes);
    for (Lane& L : q->lanes) {
        int rc = as_plan_create(m, &L.plan);
        if (rc == AS_OK) rc = as_plan_set_serial(L.plan, 1);
        if (rc == AS_OK) rc = as_plan_create(m, &L.gplan);
        if (rc == AS_OK) rc = as_plan_set_serial(L.gplan, 1);
        if (rc == AS_OK) rc = as_plan_set_layout_cap(L.gplan, INT_MAX);     // never flushes on its own: reset with the graphs
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

extern "C" int as_lanes_set_graph_cap(as_lanes* q, int max_graphs)
{
    if (!q || max_graphs < 1) return AS_EINVAL;
    q->graph_cap = (size_t)max_graphs;
    return AS_OK;
}

extern "C" int as_lanes_set_layout_cap(as_lanes* q, int max_layouts)
{
    if (!q) return AS_EINVAL;
    for (Lane& L : q->lanes) {
        const int rc = as_plan_set_layout_cap(L.plan, max_layouts);
        if (rc != AS_OK) return rc;
    }
    return AS_OK;
}

extern "C" int as_lanes_reserve(as_lanes* q, size_t bytes_a, size_t bytes_b)
{
    if (!q) return AS_EINVAL;
    for (Lane& L : q->lanes) {
        if (bytes_a <= L.na && bytes_b <= L.nb) continue;
        if (hipStreamSynchronize(L.stream) != hipSuccess) return (int)hipErrorUnknown;
        drop_graphs(L);                                           // the graphs hold the old workspaces' addresses
        if (!grow(q, &L.wa, &L.na, bytes_a) || !grow(q, &L.wb, &L.nb, bytes_b)) return (int)hipErrorOutOfMemory;
    }
    return AS_OK;
}

extern "C" int as_lanes_stats(const as_lanes* q, int lane, int64_t* out6)
{
    if (!q || !out6 || lane < 0 || lane >= (int)q->lanes.size()) return AS_EINVAL;
    const Lane& L = q->lanes[lane];
    out6[0] = (int64_t)L.graphs.size();
    out6[1] = L.n_drops;
    out6[2] = as_plan_layout_flushes(L.plan);
    out6[3] = L.n_launch;
    out6[4] = L.n_eager;
    out6[5] = L.n_capture;
    return AS_OK;
}

static int flush_lane(as_lanes* q, int lane);
extern "C" int as_lanes_wait(as_lanes* q, int lane)
{
    if (!q || lane >= (int)q->lanes.size()) return AS_EINVAL;
    try {                                                          // submissions still waiting for neighbours go out first
        for (int i = 0; i < (int)q->lanes.size(); ++i)
            if (lane < 0 || lane == i) {
                const int rc = flush_lane(q, i);
                if (rc != AS_OK) return rc;
            }
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
    for (int i = 0; i < (int)q->lanes.size(); ++i)
        if (lane < 0 || lane == i) {
            if (hipStreamSynchronize(q->lanes[i].stream) != hipSuccess) return (int)hipErrorUnknown;
            // (host submissions: their mel is at home when the lane's device -> host stream has drained)
            const Lane& Lw = q->lanes[i];
            for (int k = 0; k < 2; ++k)
                if (Lw.blk_used[k] && hipEventSynchronize(Lw.ev_d2h[k]) != hipSuccess) return (int)hipErrorUnknown;
        }
    if (lane < 0) free_retired(q);                                // every lane is idle: a free's device synchronisation costs nothing now
    return as_device_status(0) ? AS_EDEVICE : AS_OK;
}

// one as_forward_test call on lane q->next (then the next lane's turn)
static int lane_run(as_lanes* q, const as_batch* batch, const as_forward_io* io, int32_t* frames_host_out)
{
    const int lane = q->next;
    Lane& L = q->lanes[lane];
    // a status bit is up: as_forward_test would refuse anyway -- but it must do so before the pair's state moves on (a refused call under
    // capture would look up its layouts there, and a plan's first look-up of a geometry is an upload: not capturable)
    if (as_device_status(0)) return AS_EDEVICE;
    // the lane's previous batch has left its workspaces (and the caller's buffers of that lane)
    if (hipStreamSynchronize(L.stream) != hipSuccess) return (int)hipErrorUnknown;
    // (sizes come from the eager plan: a count pass adds host-side layout entries, which that plan may flush; the graph plan stays small)
    const size_t na = as_module_workspace_bytes(q->m, L.plan, AS_MOD_FORWARD_A, batch);
    if (!na) return AS_EINVAL;
    // workspace B depends on the frame counts: known (forced durations / a second pass), a capacity the caller named (frame_cap: the call
    // then runs like one with known counts -- eager, captured, replayed), or sized for what the output buffer can hold
    const bool cap_mode = !batch->frames && io->frame_cap > 0;
    std::vector<int32_t> cap;
    as_batch bb = *batch;
    if (cap_mode) {
        cap.assign(batch->B, 0);
        cap[0] = io->frame_cap;                                     // (AS_MOD_FORWARD_B_CAP: only the sum counts)
        bb.frames = cap.data();
    } else if (!batch->frames) {
        cap.assign(batch->B, std::max(1, io->ld_out / 2 / batch->B));
        bb.frames = cap.data();
    }
    size_t nb = as_module_workspace_bytes(q->m, L.plan, cap_mode ? AS_MOD_FORWARD_B_CAP : AS_MOD_FORWARD_B, &bb);
    if (!nb) return AS_EINVAL;
    if (na > L.na || nb > L.nb) drop_graphs(L);                   // the graphs hold the old workspaces' addresses
    if (!grow(q, &L.wa, &L.na, na) || !grow(q, &L.wb, &L.nb, nb)) return (int)hipErrorOutOfMemory;
    q->next = (lane + 1) % (int)q->lanes.size();

    if (!batch->frames && !cap_mode) {
        // predicted durations: the call synchronises once to read the frame counts; AS_ENOSPC = workspace B (sized for a capacity) is too
        // small for what came out -- frames_host_out says what is needed: size B for it and run again (the header's contract)
        std::vector<int32_t> fr(batch->B, 0);
        int32_t* fh = frames_host_out ? frames_host_out : fr.data();
        ++L.n_eager;
        int rc = as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, fh, L.stream);
        if (rc == AS_ENOSPC) {
            as_batch fit = *batch;
            fit.frames = fh;
            long total = 0;
            for (int b = 0; b < batch->B; ++b) total += fh[b];
            if (2 * total > io->ld_out) return AS_ENOSPC;           // the caller's output buffer itself is too small
            nb = as_module_workspace_bytes(q->m, L.plan, AS_MOD_FORWARD_B, &fit);
            if (!nb) return AS_EINVAL;
            if (hipStreamSynchronize(L.stream) != hipSuccess) return (int)hipErrorUnknown;
            if (nb > L.nb) drop_graphs(L);
            if (!grow(q, &L.wb, &L.nb, nb)) return (int)hipErrorOutOfMemory;
            rc = as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, fh, L.stream);
        }
        return rc;
    }
    if (frames_host_out && batch->frames) memcpy(frames_host_out, batch->frames, sizeof(int32_t) * batch->B);
    const std::string key = key_of(batch, io);
    auto g = L.graphs.find(key);
    if (g != L.graphs.end()) {
        ++L.n_launch;
        return hipGraphLaunch(g->second, L.stream) == hipSuccess ? AS_OK : (int)hipErrorUnknown;
    }
    int state = L.met.count(key) ? 2 : (L.once.count(key) ? 1 : 0);
    // graphs + pairs the graph plan has met (each may become a graph, and its tables already live there) never exceed the cap: a pair
    // that comes back when there is no room sends every graph and the graph plan's tables away first
    if (state == 1 && L.graphs.size() + L.met.size() + 1 > q->graph_cap) drop_graphs(L);
    if (state == 0) {                                               // first call with this pair: the eager plan (flushable); only the key is kept
        if (L.once.size() >= 4096) L.once.clear();                  // (ever new geometries: forget the candidates, never the graphs)
        L.once.insert(key);
        ++L.n_eager;
        return as_forward_test(q->m, L.plan, batch, io, L.wa, L.na, L.wb, L.nb, nullptr, L.stream);
    }
    if (state == 1) {                                               // it came back: the graph plan meets it eagerly (uploads its tables)
        L.once.erase(key);
        L.met.insert(key);
        ++L.n_eager;
        return as_forward_test(q->m, L.gplan, batch, io, L.wa, L.na, L.wb, L.nb, nullptr, L.stream);
    }
    L.met.erase(key);                                               // (whatever happens below, the pair leaves the waiting set)
    // third call: capture from the graph plan (its tables for this geometry exist; its cap is never reached, so no flush can fall into
    // the capture -- and as_plan's flush skips a capturing stream anyway)
    if (hipStreamBeginCapture(L.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return (int)hipErrorUnknown;
    const int rc = as_forward_test(q->m, L.gplan, batch, io, L.wa, L.na, L.wb, L.nb, nullptr, L.stream);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(L.stream, &graph);
    if (rc != AS_OK || ec != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (rc != AS_OK) return rc;
        // not capturable after all: run it eagerly
        ++L.n_eager;
        return as_forward_test(q->m, L.gplan, batch, io, L.wa, L.na, L.wb, L.nb, nullptr, L.stream);
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) {
        (void)hipGetLastError();
        ++L.n_eager;
        return as_forward_test(q->m, L.gplan, batch, io, L.wa, L.na, L.wb, L.nb, nullptr, L.stream);
    }
    L.graphs[key] = exec;
    ++L.n_capture;
    ++L.n_launch;
    return hipGraphLaunch(exec, L.stream) == hipSuccess ? AS_OK : (int)hipErrorUnknown;
}

// ---- coalescing (as_lanes_set_coalesce) ---------------------------------------------------------------------------------------------
// Utterances are concatenated along the column axis of every tensor of the path, so two submissions whose buffers are ADJACENT views of
// one block -- submission 2's tokens / reference features / output columns begin where submission 1's end, same leading dimensions --
// ARE one batch: the lane holds a submission back until k such neighbours have arrived (or something else is submitted, or the lane is
// flushed / waited for) and launches them as ONE as_forward_test call, with no copy.  What it buys: wider conv GEMM launches (fuller
// rounds of the chip, the weights fetched once for k batches).  The reference processes one utterance at a time (models.py:361-362):
// any grouping is legal, and every utterance still gets its batch-1 result (packed frames: no padding, no cross-utterance term).
static bool plain_io(const as_forward_io* io)                     // only the mel is wanted: the optional outputs have no per-submission home in a merged call
{
    return !io->duration && !io->dur_i && !io->frame_off && !io->style && !io->feat12 && !io->t_en && !io->a_en && !io->F0 && !io->N && !io->EMA;
}
static bool adjacent(const Lane::Pending& p, const as_forward_io* io, bool cap_mode)
{
    long nt = 0, nr = 0, nf = 0;
    for (int32_t v : p.tok_lens) nt += v;
    for (int32_t v : p.ref_lens) nr += v;
    for (int32_t v : p.frames) nf += v;
    const as_forward_io& a = p.io;
    if (cap_mode != (p.frames.empty() && a.frame_cap > 0)) return false;   // (a group is of one kind)
    const bool in = io->tokens == a.tokens + nt && io->mel == a.mel + nr && io->ld_mel == a.ld_mel && io->f0_raw == a.f0_raw + nr &&
                    io->ema_raw == a.ema_raw + nr && io->ld_ema == a.ld_ema;
    // under a frame capacity every submission keeps its own output buffer (as_segments: the merged call's mel is dealt out to them)
    if (cap_mode) return in && !io->forced_dur && !a.forced_dur;
    return in && ((!io->forced_dur && !a.forced_dur) || (io->forced_dur && a.forced_dur && io->forced_dur == a.forced_dur + nt)) &&
           io->mel_out == a.mel_out + 2 * nf && io->ld_out == a.ld_out;
}
// (frame capacity: frame_off is the one optional output a submission of a merged call can have -- it is how the caller finds its utterances)
static bool plain_cap_io(const as_forward_io* io)
{
    return !io->duration && !io->dur_i && !io->style && !io->feat12 && !io->t_en && !io->a_en && !io->F0 && !io->N && !io->EMA && !io->segs;
}

// ---- debug mode (as_lanes_set_debug, AS_DEBUG=1) -------------------------------------------------------------------------------------
// Under coalescing a submission's device buffers are read when its GROUP is launched, not when it is submitted: a caller that refills them
// in between corrupts a batch without any sign.  Debug mode makes that loud: the inputs of a held-back submission (tokens, forced
// durations, f0, the EMA and mel rows) are checksummed on the lane's stream when it is submitted -- the call then WAITS for that stream,
// so the sum is of the data the caller handed over -- and again when the group goes out; a difference raises AS_STATUS_BAD_LAYOUT (the
// group's launch returns AS_EDEVICE, as every entry point does while a bit is set).  Costs a stream synchronisation per submission.
__global__ void __launch_bounds__(256)
lanes_sum_kernel(const uint32_t* __restrict__ p, long n, long ld, unsigned long long salt, unsigned long long* __restrict__ out)
{
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    const long r = blockIdx.y;
    unsigned long long v = 0;
    if (j < n) v = (unsigned long long)p[r * ld + j] * (2ull * (unsigned long long)(r * n + j) + 1ull + salt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);          // (a sum: the order of the additions does not matter)
}

void as_status_raise_host(int kind);                               // status.hip

// checksum of a submission's device inputs as they are when the lane's stream gets here; blocks until it is known
static int inputs_sum(as_lanes* q, Lane& L, const std::vector<int32_t>& tok_lens, const std::vector<int32_t>& ref_lens, const as_forward_io& io,
                      unsigned long long* sum)
{
    if (!L.dbg && hipMalloc(reinterpret_cast<void**>(&L.dbg), sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();
        return (int)hipErrorOutOfMemory;
    }
    long nt = 0, nr = 0;
    for (int32_t v : tok_lens) nt += v;
    for (int32_t v : ref_lens) nr += v;
    AS_CHECK(hipMemsetAsync(L.dbg, 0, sizeof(unsigned long long), L.stream));
    auto add = [&](const void* ptr, long n, long rows, long ld, unsigned long long salt) {
        if (!ptr || n <= 0) return;
        hipLaunchKernelGGL(lanes_sum_kernel, dim3(as_cdiv(n, 256), (unsigned)rows), dim3(256), 0, L.stream, static_cast<const uint32_t*>(ptr), n, ld, salt, L.dbg);
    };
    add(io.tokens, nt, 1, nt, 0x100000000ull);
    add(io.forced_dur, nt, 1, nt, 0x200000000ull);
    add(io.f0_raw, nr, 1, nr, 0x300000000ull);
    add(io.ema_raw, nr, 10, io.ld_ema, 0x400000000ull);
    add(io.mel, nr, q->cfg.n_mels, io.ld_mel, 0x500000000ull);
    AS_CHECK_LAUNCH();
    AS_CHECK(hipMemcpyAsync(sum, L.dbg, sizeof(unsigned long long), hipMemcpyDeviceToHost, L.stream));
    AS_CHECK(hipStreamSynchronize(L.stream));
    return AS_OK;
}

// launch the pending group of lane `lane` (if any) as one call
static int flush_lane(as_lanes* q, int lane)
{
    Lane& L = q->lanes[lane];
    if (L.pend.empty()) return AS_OK;
    if (q->debug) {
        for (const Lane::Pending& p : L.pend) {
            unsigned long long now = 0;
            const int rc = inputs_sum(q, L, p.tok_lens, p.ref_lens, p.io, &now);
            if (rc != AS_OK) return rc;
            if (now != p.sum) {
                fprintf(stderr, "artspeech_hip: as_lanes (debug): the device buffers of a submission that was waiting for its group on lane %d "
                                "changed between as_lanes_submit and the group's launch\n", lane);
                as_status_raise_host(AS_STATUS_BAD_LAYOUT);
            }
        }
    }
    std::vector<int32_t> tl, rl, fr;
    for (const Lane::Pending& p : L.pend) {
        tl.insert(tl.end(), p.tok_lens.begin(), p.tok_lens.end());
        rl.insert(rl.end(), p.ref_lens.begin(), p.ref_lens.end());
        fr.insert(fr.end(), p.frames.begin(), p.frames.end());
    }
    as_batch b;
    b.B = (int32_t)tl.size();
    b.tok_lens = tl.data(); b.ref_lens = rl.data(); b.frames = fr.data();
    as_forward_io io = L.pend.front().io;
    as_segments segs;
    if (fr.empty()) {                                             // submissions under a frame capacity: the merged call deals its mel out to them
        b.frames = nullptr;
        if (L.pend.size() > 1) {
            memset(&segs, 0, sizeof(segs));
            segs.n = (int32_t)L.pend.size();
            long cap = 0;
            int32_t first = 0;
            for (size_t i = 0; i < L.pend.size(); ++i) {
                const Lane::Pending& p = L.pend[i];
                segs.first[i] = first;
                segs.cap[i] = p.io.frame_cap;
                segs.mel_out[i] = p.io.mel_out;
                segs.ld_out[i] = p.io.ld_out;
                segs.frame_off[i] = p.io.frame_off;
                first += (int32_t)p.tok_lens.size();
                cap += p.io.frame_cap;
            }
            segs.first[segs.n] = first;
            io.frame_cap = (int32_t)cap;
            io.frame_off = nullptr;
            io.segs = &segs;
        }
    }
    if (L.pend.size() > 1) ++L.n_merged;
    // (host submissions: where each one's mel goes once the group's kernels are enqueued)
    struct Out { float* host; int32_t ld; const float* dev; long cols; int32_t ld_dev; int32_t* foff_host; const int32_t* foff_dev; int n_foff; };
    std::vector<Out> outs;
    for (const Lane::Pending& p : L.pend)
        if (p.out_host) {
            long nf = 0;
            for (int32_t v : p.frames) nf += v;
            // (under a frame capacity the whole slot goes back: how much of it holds frames is known on the device only)
            outs.push_back({p.out_host, p.ld_out_host, p.io.mel_out, p.frames.empty() ? 2L * p.io.frame_cap : 2 * nf, p.io.ld_out, p.foff_host,
                            p.io.frame_off, (int)p.tok_lens.size() + 1});
        }
    L.pend.clear();
    const bool host_group = !outs.empty();
    const int bi = L.cur;
    if (host_group) {                                             // the group's kernels start behind its host -> device copies
        AS_CHECK(hipEventRecord(L.ev_h2d[bi], q->up(L)));
        AS_CHECK(hipStreamWaitEvent(L.stream, L.ev_h2d[bi], 0));
        L.cur ^= 1;                                               // (the next group of this lane fills the other block)
        L.blk_used[bi] = true;
    }
    const int keep = q->next;
    q->next = lane;
    const int rc = lane_run(q, &b, &io, nullptr);
    if (keep != lane) q->next = keep;                             // (a flush from as_lanes_wait does not change whose turn it is)
    if (host_group) {
        // ... and every submission's mel goes home behind them, on the lane's device -> host stream: the lane's next group computes meanwhile.
        // ev_d2h[bi] is what the block's NEXT group's copies wait for (recorded whatever happened: a block is never left without it)
        hipError_t e = hipEventRecord(L.ev_comp[bi], L.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(q->down(L), L.ev_comp[bi], 0);
        for (const Out& o : outs) {
            if (rc == AS_OK && e == hipSuccess && o.cols > 0)
                e = hipMemcpy2DAsync(o.host, (size_t)o.ld * 4, o.dev, (size_t)o.ld_dev * 4, (size_t)o.cols * 4, (size_t)q->cfg.n_mels,
                                     hipMemcpyDeviceToHost, q->down(L));
            if (rc == AS_OK && e == hipSuccess && o.foff_host)
                e = hipMemcpyAsync(o.foff_host, o.foff_dev, (size_t)o.n_foff * 4, hipMemcpyDeviceToHost, q->down(L));
        }
        const hipError_t e2 = hipEventRecord(L.ev_d2h[bi], q->down(L));
        if (rc == AS_OK && (e != hipSuccess || e2 != hipSuccess)) return (int)(e != hipSuccess ? e : e2);
    }
    return rc;
}

static int lanes_submit(as_lanes* q, const as_batch* batch, const as_forward_io* io, int32_t* frames_host_out, int32_t* lane_out,
                        float* out_host = nullptr, int32_t ld_out_host = 0, int32_t* foff_host = nullptr)
{
    if (!q || !batch || !io || batch->B <= 0 || !batch->tok_lens || !batch->ref_lens) return AS_EINVAL;
    Lane& L = q->lanes[q->next];
    // (a host submission always joins the group of its lane's block -- a group of one when coalescing is off)
    const bool cap_mode = !batch->frames && io->frame_cap > 0;
    const bool can_wait = (q->coalesce > 1 || out_host) && ((batch->frames && plain_io(io)) || (cap_mode && plain_cap_io(io)));
    size_t waiting = 0;                                           // utterances of the group that waits here
    for (const Lane::Pending& p : L.pend) waiting += p.tok_lens.size();
    // (a call takes at most 1024 utterances: as_durations_f32's one-workgroup scan; at most AS_MAX_SEGMENTS submissions under a capacity)
    if (!L.pend.empty() && !(can_wait && adjacent(L.pend.back(), io, cap_mode) && waiting + (size_t)batch->B <= 1024 &&
                             (!cap_mode || L.pend.size() < (size_t)AS_MAX_SEGMENTS))) {
        const int rc = flush_lane(q, q->next);                    // not a neighbour of what waits here: that group goes out first (and the turn passes on)
        if (rc != AS_OK) return rc;
    }
    if (lane_out) *lane_out = q->next;
    if (!can_wait) return lane_run(q, batch, io, frames_host_out);
    Lane& L2 = q->lanes[q->next];
    Lane::Pending p;
    p.tok_lens.assign(batch->tok_lens, batch->tok_lens + batch->B);
    p.ref_lens.assign(batch->ref_lens, batch->ref_lens + batch->B);
    if (batch->frames) p.frames.assign(batch->frames, batch->frames + batch->B);
    p.io = *io;
    p.out_host = out_host;
    p.ld_out_host = ld_out_host;
    p.foff_host = foff_host;
    if (q->debug && !out_host) {                                  // (a host submission's device buffers are the library's own)
        const int rc = inputs_sum(q, L2, p.tok_lens, p.ref_lens, p.io, &p.sum);
        if (rc != AS_OK) return rc;
    }
    L2.pend.push_back(std::move(p));
    if (frames_host_out && batch->frames) memcpy(frames_host_out, batch->frames, sizeof(int32_t) * batch->B);
    if ((int)L2.pend.size() >= q->coalesce) return flush_lane(q, q->next);
    return AS_OK;
}

extern "C" int as_lanes_submit(as_lanes* q, const as_batch* batch, const as_forward_io* io, int32_t* frames_host_out, int32_t* lane_out)
{
    try {
        return lanes_submit(q, batch, io, frames_host_out, lane_out);
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}

// ---- host submissions (as_lanes_submit_host) -----------------------------------------------------------------------------------------
// The reference's boundary hands over HOST arrays (test.py:96-113 moves tokens / mel to the device inside `synthesis`).  Here the lane owns
// the device side: one block per lane that holds the inputs of a group's submissions as ADJACENT column ranges (so that the group is one
// batch as it lies) and the group's output.  A submission's inputs are copied into the block's next free columns when it is submitted, the
// group's launch follows the last of them, and every submission's mel goes back to its own host array behind the launch.  TWO blocks per
// lane, alternating from group to group, and ONE copy stream for either direction shared by the lanes: host -> device copies of a lane's
// group g + 1 (stream h2d) run under the kernels of its group g (the lane's stream), whose results leave on stream d2h under the kernels of
// group g + 1.  Edges (events): kernels(g) wait for h2d(g); d2h(g) waits for kernels(g); h2d(g + 2) -- the same block -- waits for d2h(g).
// (With copies and kernels on the one stream the copies of a lane cost it 0.17 ms per step of 3.69: measured first.)  This is the
// buffer rule of as_lanes_set_coalesce kept by the library instead of the caller.
static size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }

// the block holds a group of `k` submissions like this one (with some slack); a block that has to grow is replaced while the lane is idle
static int block_fit(as_lanes* q, Lane& L, long nt, long nr, long nf2, int n_utt, int k)
{
    if (!q->h2d && (hipStreamCreateWithFlags(&q->h2d, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&q->d2h, hipStreamNonBlocking) != hipSuccess))
        return (int)hipErrorOutOfMemory;
    if (!L.ev_h2d[0]) {                                           // first host submission of this lane: its events
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&L.ev_h2d[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&L.ev_comp[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&L.ev_d2h[i], hipEventDisableTiming) != hipSuccess)
                return (int)hipErrorOutOfMemory;
    }
    Lane::Block& b = L.blk[L.cur];
    if (L.pend.empty()) b.used_tok = b.used_ref = b.used_out = b.used_utt = 0;   // a new group starts at the block's first column
    if (b.dev && b.used_tok + nt <= b.cap_tok && b.used_ref + nr <= b.cap_ref && b.used_out + nf2 <= b.cap_out &&
        b.used_utt + n_utt + 1 <= b.cap_utt)
        return AS_OK;
    if (!L.pend.empty()) return AS_ENOSPC;                        // (the caller sends the waiting group out first, then asks again)
    const auto grow_to = [](long need, int k_) { return (int)std::min<long>((long)INT_MAX / 64, (need * k_ * 5 + 3) / 4 + 64); };
    const int ct = std::max(b.cap_tok, grow_to(nt, k)), cr = std::max(b.cap_ref, grow_to(nr, k)), co = std::max(b.cap_out, grow_to(nf2, k));
    const int cu = std::max(b.cap_utt, grow_to(n_utt + 1, k));
    const int n_mels = q->cfg.n_mels;
    const size_t bytes = 2 * up256((size_t)ct * 4) + up256((size_t)cu * 4) + up256((size_t)cr * 4) + up256((size_t)10 * cr * 4) +
                         up256((size_t)n_mels * cr * 4) + up256((size_t)n_mels * co * 4);
    AS_CHECK(hipStreamSynchronize(L.stream));                     // the groups that used the old block have left it: kernels ...
    AS_CHECK(hipStreamSynchronize(q->d2h));                       // ... and the copies of their results
    drop_graphs(L);                                               // (its graphs hold the old block's addresses)
    if (b.dev) q->retired.push_back(b.dev);                       // (not hipFree: it would stall the other lanes)
    b = Lane::Block();
    L.blk_used[L.cur] = false;
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return (int)hipErrorOutOfMemory;
    }
    char* c = static_cast<char*>(d);
    b.dev = d; b.bytes = bytes; b.cap_tok = ct; b.cap_ref = cr; b.cap_out = co; b.cap_utt = cu;
    b.tokens = reinterpret_cast<int32_t*>(c); c += up256((size_t)ct * 4);
    b.forced = reinterpret_cast<int32_t*>(c); c += up256((size_t)ct * 4);
    b.foff = reinterpret_cast<int32_t*>(c); c += up256((size_t)cu * 4);
    b.f0 = reinterpret_cast<float*>(c); c += up256((size_t)cr * 4);
    b.ema = reinterpret_cast<float*>(c); c += up256((size_t)10 * cr * 4);
    b.mel = reinterpret_cast<float*>(c); c += up256((size_t)n_mels * cr * 4);
    b.out = reinterpret_cast<float*>(c);
    return AS_OK;
}

static int lanes_submit_host(as_lanes* q, const as_batch* batch, const as_host_io* h, int32_t* lane_out)
{
    if (!q || !batch || !h || batch->B <= 0 || !batch->tok_lens || !batch->ref_lens) return AS_EINVAL;
    if (!h->tokens || !h->mel || !h->f0_raw || !h->ema_raw || !h->mel_out) return AS_EINVAL;
    // frame counts from the caller, or a capacity (durations predicted on the device: the frame offsets come back with the mel)
    const bool cap_mode = !batch->frames;
    if (cap_mode && (h->frame_cap < 1 || !h->frame_off || h->forced_dur)) return AS_EINVAL;
    long nt = 0, nr = 0, nf = 0;
    for (int b = 0; b < batch->B; ++b) {
        if (batch->tok_lens[b] < 0 || batch->ref_lens[b] < 0 || (!cap_mode && batch->frames[b] < 0)) return AS_EINVAL;
        nt += batch->tok_lens[b]; nr += batch->ref_lens[b]; nf += cap_mode ? 0 : batch->frames[b];
    }
    if (cap_mode) nf = h->frame_cap;
    if (h->ld_mel < nr || h->ld_ema < nr || h->ld_out < 2 * nf) return AS_EINVAL;
    const int n_mels = q->cfg.n_mels;
    for (int attempt = 0;; ++attempt) {
        Lane& L = q->lanes[q->next];
        // what waits on this lane came with device buffers of the caller's, or the block is full: that group goes out first
        size_t waiting = 0;
        for (const Lane::Pending& p : L.pend) waiting += p.tok_lens.size();
        const bool foreign = !L.pend.empty() && (!L.pend.back().out_host || (L.pend.back().io.forced_dur != nullptr) != (h->forced_dur != nullptr) ||
                                                 waiting + (size_t)batch->B > 1024 || L.pend.back().frames.empty() != cap_mode ||
                                                 (cap_mode && L.pend.size() >= (size_t)AS_MAX_SEGMENTS));
        int rc = foreign ? AS_ENOSPC : block_fit(q, L, nt, nr, 2 * nf, batch->B, std::max(q->coalesce, 1));
        if (rc == AS_ENOSPC && attempt <= (int)q->lanes.size()) {
            rc = flush_lane(q, q->next);                          // (the turn passes on: the submission opens the next lane's group)
            if (rc != AS_OK) return rc;
            continue;
        }
        if (rc != AS_OK) return rc;
        Lane::Block& b = L.blk[L.cur];
        hipStream_t s = q->up(L);
        // the block's previous group (two groups back on this lane) has left it -- its kernels and the copies of its results -- before
        // the first copy of this one lands
        if (L.pend.empty() && L.blk_used[L.cur]) AS_CHECK(hipStreamWaitEvent(s, L.ev_d2h[L.cur], 0));
        AS_CHECK(hipMemcpyAsync(b.tokens + b.used_tok, h->tokens, (size_t)nt * 4, hipMemcpyHostToDevice, s));
        if (h->forced_dur) AS_CHECK(hipMemcpyAsync(b.forced + b.used_tok, h->forced_dur, (size_t)nt * 4, hipMemcpyHostToDevice, s));
        AS_CHECK(hipMemcpyAsync(b.f0 + b.used_ref, h->f0_raw, (size_t)nr * 4, hipMemcpyHostToDevice, s));
        if (nr > 0) {
            AS_CHECK(hipMemcpy2DAsync(b.ema + b.used_ref, (size_t)b.cap_ref * 4, h->ema_raw, (size_t)h->ld_ema * 4, (size_t)nr * 4, 10, hipMemcpyHostToDevice, s));
            AS_CHECK(hipMemcpy2DAsync(b.mel + b.used_ref, (size_t)b.cap_ref * 4, h->mel, (size_t)h->ld_mel * 4, (size_t)nr * 4, (size_t)n_mels,
                                      hipMemcpyHostToDevice, s));
        }
        as_forward_io io;
        memset(&io, 0, sizeof(io));
        io.tokens = b.tokens + b.used_tok;
        io.mel = b.mel + b.used_ref; io.ld_mel = b.cap_ref;
        io.f0_raw = b.f0 + b.used_ref;
        io.ema_raw = b.ema + b.used_ref; io.ld_ema = b.cap_ref;
        io.forced_dur = h->forced_dur ? b.forced + b.used_tok : nullptr;
        io.mel_out = b.out + b.used_out; io.ld_out = b.cap_out;
        if (cap_mode) {
            io.frame_cap = h->frame_cap;
            io.frame_off = b.foff + b.used_utt;
        }
        // (a group mixes forced and predicted-from-known-frames submissions only if all or none bring forced durations: `adjacent` says no otherwise)
        b.used_tok += (int)nt; b.used_ref += (int)nr; b.used_out += (int)(2 * nf); b.used_utt += batch->B + 1;
        return lanes_submit(q, batch, &io, nullptr, lane_out, h->mel_out, h->ld_out, cap_mode ? h->frame_off : nullptr);
    }
}

extern "C" int as_lanes_submit_host(as_lanes* q, const as_batch* batch, const as_host_io* io, int32_t* lane_out)
{
    try {
        return lanes_submit_host(q, batch, io, lane_out);
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}

extern "C" int64_t as_lanes_merged_calls(const as_lanes* q, int lane)
{
    if (!q || lane < 0 || lane >= (int)q->lanes.size()) return -1;
    return q->lanes[lane].n_merged;
}

extern "C" int as_lanes_flush(as_lanes* q)
{
    if (!q) return AS_EINVAL;
    try {
        for (int i = 0; i < (int)q->lanes.size(); ++i) {
            const int rc = flush_lane(q, i);
            if (rc != AS_OK) return rc;
        }
        return AS_OK;
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}

extern "C" int as_lanes_set_coalesce(as_lanes* q, int k)
{
    if (!q || k < 1 || k > 16) return AS_EINVAL;
    const int rc = as_lanes_flush(q);
    if (rc != AS_OK) return rc;
    q->coalesce = k;
    return AS_OK;
}

extern "C" int as_lanes_set_debug(as_lanes* q, int on)
{
    if (!q) return AS_EINVAL;
    const int rc = as_lanes_flush(q);                             // (what waits was submitted without a checksum)
    if (rc != AS_OK) return rc;
    q->debug = on != 0;
    return AS_OK;
}
