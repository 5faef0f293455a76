// Module-level C ABI (include/artspeech_hip.h, "Module-level entry points"): the acoustic model behind an opaque handle.
//
//   as_model_create   reads the checkpoint blob, folds weight_norm / spectral_norm (what the reference's modules do implicitly
//                     on every forward, models.py:685-701), lays every weight out for the kernels and uploads it
//   as_*_forward      the launch sequences of RelTransformerEncoder / StyleEncoder / DurationPredictor / ArtsPredictor / Decoder
//   as_forward_test   ArtsSpeech.forward(step="test"), models.py:356-371, batched on packed frames
//
// This file holds no kernels: it is the host side that orders the launches of the other files of this library, owns the
// batch geometry tables and hands out workspace memory.  One code path serves three passes over the same function:
//   prepare (as_model_create: every weight the sequence touches is built and uploaded; nothing launched),
//   count   (as_module_workspace_bytes: the bump allocator only adds up), and
//   run     (kernels are enqueued; nothing is allocated, nothing synchronises once the geometry's tables exist).
#include "common.h"
#include "conv_gemm.h"
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

constexpr int N_HEADS = 4;      // RelTransformerEnc.py:333
constexpr int WINDOW = 4;       // RelTransformerEnc.py:335
constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2;

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------------------------------------------
// device memory that lives as long as its owner (weights of a model, geometry tables of a plan)
// ------------------------------------------------------------------------------------------------------------------
struct DevPool {
    std::vector<void*> chunks;
    std::vector<size_t> sizes;
    size_t idx = 0;               // chunk being filled (chunks behind it are full, chunks after it are free: rewind() keeps them)
    char* cur = nullptr;
    size_t left = 0, chunk_bytes;
    explicit DevPool(size_t chunk) : chunk_bytes(chunk) {}
    void* alloc(size_t n)
    {
        n = align256(n ? n : 1);
        while (n > left) {
            if (cur && idx + 1 < chunks.size()) {              // a chunk kept by rewind()
                ++idx;
            } else if (!cur && !chunks.empty()) {
                idx = 0;
            } else {
                const size_t c = n > chunk_bytes ? n : chunk_bytes;
                void* p = nullptr;
                if (hipMalloc(&p, c) != hipSuccess) return nullptr;
                chunks.push_back(p);
                sizes.push_back(c);
                idx = chunks.size() - 1;
            }
            cur = static_cast<char*>(chunks[idx]);
            left = sizes[idx];
        }
        void* r = cur;
        cur += n;
        left -= n;
        return r;
    }
    // everything handed out so far is dead: start over in the memory already held (no hipFree: a free synchronises the whole device)
    void rewind()
    {
        idx = 0;
        cur = nullptr;
        left = 0;
    }
    void release()
    {
        for (void* p : chunks) (void)hipFree(p);
        chunks.clear();
        sizes.clear();
        idx = 0;
        cur = nullptr;
        left = 0;
    }
};

struct HostT {
    std::vector<int> dims;
    std::vector<float> v;
    size_t numel() const { return v.size(); }
    int dim(int i) const { return i < (int)dims.size() ? dims[i] : 1; }
};

struct GemmW {                  // a conv / linear weight prepared for as_conv_gemm_f32
    uint16_t* wh = nullptr;     // [G][T][KBx][4][M][8] fp16 split image
    float* w32 = nullptr;       // [T][Kp][M] fp32 (Cin = 1: the direct kernel)
    float scale = 1.f;
    int T = 0, Kp = 0, M = 0, K = 0, G = 1;
    int K2 = 0;                 // channels of the second operand whose 1x1 weights follow the taps (ConvGemmArgs.Xh2: a folded shortcut)
};
struct Vec {
    float* p = nullptr;
    size_t n = 0;
};
struct LstmW {
    const GemmW* wih = nullptr; // [8H][I] both directions
    float* bias = nullptr;      // [8H] b_ih + b_hh
    float* whh_t = nullptr;     // [2][H][4H]
    int H = 0;
};

}  // namespace

struct as_model {
    as_model_cfg cfg;
    int device = 0;
    std::unordered_map<std::string, HostT> raw;             // folded fp32 host tensors, reference names with plain ".weight"
    mutable std::unordered_map<std::string, GemmW> gemm;    // filled while !frozen (as_model_create), read-only afterwards
    mutable std::unordered_map<std::string, Vec> vecs;
    mutable std::unordered_map<std::string, LstmW> lstms;
    mutable DevPool pool{(size_t)256 << 20};
    mutable bool frozen = false;
    mutable int err = 0;

    bool has(const std::string& n) const { return raw.find(n) != raw.end(); }
    const HostT* host(const std::string& n) const
    {
        auto it = raw.find(n);
        if (it == raw.end()) { if (!err) { err = AS_EINVAL; fprintf(stderr, "artspeech_hip: checkpoint has no tensor '%s'\n", n.c_str()); } return nullptr; }
        return &it->second;
    }
    float* upload(const float* h, size_t n) const
    {
        float* d = static_cast<float*>(pool.alloc(n * sizeof(float)));
        if (!d || hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { if (!err) err = (int)hipErrorOutOfMemory; return nullptr; }
        return d;
    }
    // a weight given as host data [G][Cout][Cin][T]
    const GemmW* gemm_from(const std::string& key, const float* w, int G, int Cout, int Cin, int T, const float* w2 = nullptr, int Cin2 = 0) const
    {
        GemmW g;
        g.T = T; g.K = Cin; g.Kp = (Cin + 15) / 16 * 16; g.M = Cout; g.G = G; g.K2 = Cin2;
        const size_t bytes = as_prep_weight_f16x2_sc_bytes(G, Cout, Cin, T, Cin2);
        std::vector<uint16_t> img(bytes / 2);
        if (as_prep_weight_f16x2_sc_host(w, w2, G, Cout, Cin, T, Cin2, img.data(), &g.scale) != AS_OK) { if (!err) err = AS_EINVAL; return nullptr; }
        g.wh = static_cast<uint16_t*>(pool.alloc(bytes));
        if (!g.wh || hipMemcpy(g.wh, img.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { if (!err) err = (int)hipErrorOutOfMemory; return nullptr; }
        if (Cin == 1 && G == 1) {                                          // the direct kernel's fp32 image [T][Kp][M]
            std::vector<float> w32((size_t)T * g.Kp * Cout, 0.f);
            for (int m = 0; m < Cout; ++m)
                for (int t = 0; t < T; ++t) w32[((size_t)t * g.Kp) * Cout + m] = w[(size_t)m * T + t];
            g.w32 = upload(w32.data(), w32.size());
        }
        return &(gemm[key] = g);
    }
    // conv weights `names` ([Cout][Cin][k...] each, same shape) stacked as the weight sets of one grouped launch
    const GemmW* conv_stack(const std::vector<std::string>& names) const
    {
        std::string key = names[0];
        for (size_t i = 1; i < names.size(); ++i) key += "|" + names[i];
        auto it = gemm.find(key);
        if (it != gemm.end()) return &it->second;
        if (frozen) { if (!err) { err = AS_EINVAL; fprintf(stderr, "artspeech_hip: '%s' was not prepared by as_model_create\n", key.c_str()); } return nullptr; }
        const HostT* a = host(names[0] + ".weight");
        if (!a) return nullptr;
        const int Cout = a->dim(0), Cin = a->dim(1), T = (int)(a->numel() / ((size_t)Cout * Cin));
        if (names.size() == 1) return gemm_from(key, a->v.data(), 1, Cout, Cin, T);
        std::vector<float> w(a->v);
        for (size_t i = 1; i < names.size(); ++i) {
            const HostT* b = host(names[i] + ".weight");
            if (!b || b->dims != a->dims) { if (!err) err = AS_EINVAL; return nullptr; }
            w.insert(w.end(), b->v.begin(), b->v.end());
        }
        return gemm_from(key, w.data(), (int)names.size(), Cout, Cin, T);
    }
    // the same with the blocks' learned shortcuts `sc_names` ([Cout][Cin2][1], no bias: models.py:77,123,178) behind the taps of every
    // weight set: the shortcut is evaluated by the launch of the block's last conv (ConvGemmArgs.Xh2 / K2)
    const GemmW* conv_fold(const std::vector<std::string>& names, const std::vector<std::string>& sc_names) const
    {
        std::string key = "FOLD:";
        for (size_t i = 0; i < names.size(); ++i) key += names[i] + "+" + sc_names[i] + "|";
        auto it = gemm.find(key);
        if (it != gemm.end()) return &it->second;
        if (frozen) { if (!err) { err = AS_EINVAL; fprintf(stderr, "artspeech_hip: '%s' was not prepared by as_model_create\n", key.c_str()); } return nullptr; }
        const HostT *a = host(names[0] + ".weight"), *s0 = host(sc_names[0] + ".weight");
        if (!a || !s0 || names.size() != sc_names.size()) { if (!err) err = AS_EINVAL; return nullptr; }
        const int Cout = a->dim(0), Cin = a->dim(1), T = (int)(a->numel() / ((size_t)Cout * Cin)), Cin2 = s0->dim(1);
        std::vector<float> w, w2;
        for (size_t i = 0; i < names.size(); ++i) {
            const HostT *b = host(names[i] + ".weight"), *sc = host(sc_names[i] + ".weight");
            if (!b || !sc || b->dims != a->dims || sc->dims != s0->dims || sc->dim(0) != Cout || sc->numel() != (size_t)Cout * Cin2) {
                if (!err) err = AS_EINVAL;
                return nullptr;
            }
            w.insert(w.end(), b->v.begin(), b->v.end());
            w2.insert(w2.end(), sc->v.begin(), sc->v.end());
        }
        return gemm_from(key, w.data(), (int)names.size(), Cout, Cin, T, w2.data(), Cin2);
    }
    const GemmW* conv(const std::string& name, const std::string& name2 = std::string()) const
    {
        return name2.empty() ? conv_stack({name}) : conv_stack({name, name2});
    }
    // raw tensors (bias, gamma, table ...) on the device, stacked one behind the other
    const float* vec_stack(const std::vector<std::string>& names) const
    {
        std::string key = names[0];
        for (size_t i = 1; i < names.size(); ++i) key += "|" + names[i];
        auto it = vecs.find(key);
        if (it != vecs.end()) return it->second.p;
        if (frozen) { if (!err) { err = AS_EINVAL; fprintf(stderr, "artspeech_hip: '%s' was not prepared by as_model_create\n", key.c_str()); } return nullptr; }
        std::vector<float> v;
        size_t n0 = 0;
        for (size_t i = 0; i < names.size(); ++i) {
            const HostT* a = host(names[i]);
            if (!a || (i && a->numel() != n0)) { if (!err) err = AS_EINVAL; return nullptr; }
            n0 = a->numel();
            v.insert(v.end(), a->v.begin(), a->v.end());
        }
        Vec d;
        d.n = v.size();
        d.p = upload(v.data(), v.size());
        vecs[key] = d;
        return d.p;
    }
    const float* vec(const std::string& name, const std::string& name2 = std::string()) const
    {
        return name2.empty() ? vec_stack({name}) : vec_stack({name, name2});
    }
    const float* bias(const std::string& name, const std::string& name2 = std::string()) const
    {
        if (!has(name + ".bias")) return nullptr;
        return vec(name + ".bias", name2.empty() ? name2 : name2 + ".bias");
    }
    const float* bias_stack(const std::vector<std::string>& names) const
    {
        if (!has(names[0] + ".bias")) return nullptr;
        std::vector<std::string> n(names);
        for (auto& x : n) x += ".bias";
        return vec_stack(n);
    }
    // q / k / v projections of one attention layer as one [3C] GEMM (RelTransformerEnc.py:128-133)
    const GemmW* qkv(const std::vector<std::string>& ps, const float** bias_out) const
    {
        std::string key = "QKV:";
        for (const std::string& q : ps) key += q + "|";
        auto it = gemm.find(key);
        if (it == gemm.end()) {
            if (frozen) { if (!err) err = AS_EINVAL; return nullptr; }
            std::vector<float> w, b;
            int C = 0, G = 0;
            for (const std::string& q : ps) {
                ++G;
                for (const char* n : {"q", "k", "v"}) {
                    const HostT* wt = host(q + ".conv_" + n + ".weight");
                    const HostT* bt = host(q + ".conv_" + n + ".bias");
                    if (!wt || !bt) return nullptr;
                    C = wt->dim(1);
                    w.insert(w.end(), wt->v.begin(), wt->v.end());
                    b.insert(b.end(), bt->v.begin(), bt->v.end());
                }
            }
            if (!gemm_from(key, w.data(), G, 3 * C, C, 1)) return nullptr;
            Vec d;
            d.n = b.size();
            d.p = upload(b.data(), b.size());
            vecs[key] = d;
            it = gemm.find(key);
        }
        *bias_out = vecs[key].p;
        return &it->second;
    }
    // nn.LSTM(bidirectional): input projection of both directions as one GEMM, biases summed, W_hh transposed
    const LstmW* lstm(const std::string& p) const
    {
        auto it = lstms.find(p);
        if (it != lstms.end()) return &it->second;
        if (frozen) { if (!err) err = AS_EINVAL; return nullptr; }
        const HostT *wi = host(p + ".weight_ih_l0"), *wir = host(p + ".weight_ih_l0_reverse"), *wh = host(p + ".weight_hh_l0"),
                    *whr = host(p + ".weight_hh_l0_reverse"), *bi = host(p + ".bias_ih_l0"), *bh = host(p + ".bias_hh_l0"),
                    *bir = host(p + ".bias_ih_l0_reverse"), *bhr = host(p + ".bias_hh_l0_reverse");
        if (!wi || !wir || !wh || !whr || !bi || !bh || !bir || !bhr) return nullptr;
        LstmW L;
        L.H = wh->dim(1);
        const int H = L.H, I = wi->dim(1);
        std::vector<float> w(wi->v);
        w.insert(w.end(), wir->v.begin(), wir->v.end());
        L.wih = gemm_from("LSTM:" + p, w.data(), 1, 8 * H, I, 1);
        std::vector<float> b(8 * H);
        for (int i = 0; i < 4 * H; ++i) { b[i] = bi->v[i] + bh->v[i]; b[4 * H + i] = bir->v[i] + bhr->v[i]; }
        L.bias = upload(b.data(), b.size());
        std::vector<float> t((size_t)2 * H * 4 * H);
        for (int d = 0; d < 2; ++d) {
            const std::vector<float>& src = d ? whr->v : wh->v;                   // [4H][H] -> [H][4H]
            for (int r = 0; r < 4 * H; ++r)
                for (int k = 0; k < H; ++k) t[((size_t)d * H + k) * 4 * H + r] = src[(size_t)r * H + k];
        }
        L.whh_t = upload(t.data(), t.size());
        return &(lstms[p] = L);
    }
    // input projections of several LSTMs of the same shape as the weight sets of one grouped launch
    const GemmW* lstm_wih_stack(const std::vector<std::string>& names, const float** bias_out) const
    {
        std::string key = "LSTMS:";
        for (const auto& n : names) key += n + "|";
        auto it = gemm.find(key);
        if (it == gemm.end()) {
            if (frozen) { if (!err) err = AS_EINVAL; return nullptr; }
            std::vector<float> w, b;
            int H = 0, I = 0;
            for (const auto& p : names) {
                const HostT *wi = host(p + ".weight_ih_l0"), *wir = host(p + ".weight_ih_l0_reverse"), *bi = host(p + ".bias_ih_l0"),
                            *bh = host(p + ".bias_hh_l0"), *bir = host(p + ".bias_ih_l0_reverse"), *bhr = host(p + ".bias_hh_l0_reverse");
                if (!wi || !wir || !bi || !bh || !bir || !bhr) return nullptr;
                H = wi->dim(0) / 4;
                I = wi->dim(1);
                w.insert(w.end(), wi->v.begin(), wi->v.end());
                w.insert(w.end(), wir->v.begin(), wir->v.end());
                for (int i = 0; i < 4 * H; ++i) b.push_back(bi->v[i] + bh->v[i]);
                for (int i = 0; i < 4 * H; ++i) b.push_back(bir->v[i] + bhr->v[i]);
            }
            if (!gemm_from(key, w.data(), (int)names.size(), 8 * H, I, 1)) return nullptr;
            Vec d;
            d.n = b.size();
            d.p = upload(b.data(), b.size());
            vecs[key] = d;
            it = gemm.find(key);
        }
        *bias_out = vecs[key].p;
        return &it->second;
    }
    // Every AdaIN1d fc layer fed by one style vector as ONE weight matrix [Mtot][K] (models.py:237: h = fc(s); gamma, beta =
    // chunk(h)): `norms` = (layer name, first style entry it reads, entries it reads) -- a layer that reads a slice of the style
    // (models.py:499,597-599) gets zero columns elsewhere.  row0[name] = its first output row (gamma rows, then beta rows).
    struct FcAll {
        const GemmW* w = nullptr;
        const float* bias = nullptr;
        std::unordered_map<std::string, int> row0;
        int Mtot = 0, K = 0;
    };
    mutable std::unordered_map<std::string, FcAll> fcalls;
    struct NormSpec { std::string name; int off, S; };
    const FcAll* fc_all(const std::string& key, const std::vector<NormSpec>& norms, int K) const
    {
        auto it = fcalls.find(key);
        if (it != fcalls.end()) return &it->second;
        if (frozen) { if (!err) err = AS_EINVAL; return nullptr; }
        FcAll f;
        f.K = K;
        std::vector<float> w, b;
        for (const auto& n : norms) {
            const HostT *wt = host(n.name + ".fc.weight"), *bt = host(n.name + ".fc.bias");
            if (!wt || !bt || wt->dim(1) != n.S || n.off + n.S > K) { if (!err) err = AS_EINVAL; return nullptr; }
            const int M = wt->dim(0);
            f.row0[n.name] = f.Mtot;
            f.Mtot += M;
            const size_t base = w.size();
            w.resize(base + (size_t)M * K, 0.f);
            for (int m = 0; m < M; ++m)
                for (int k = 0; k < n.S; ++k) w[base + (size_t)m * K + n.off + k] = wt->v[(size_t)m * n.S + k];
            b.insert(b.end(), bt->v.begin(), bt->v.end());
        }
        f.w = gemm_from("FCALL:" + key, w.data(), 1, f.Mtot, K, 1);
        f.bias = upload(b.data(), b.size());
        if (!f.w || !f.bias) return nullptr;
        return &(fcalls[key] = f);
    }
    // decoder.F0_conv (1 -> 32), N_conv (1 -> 32), EMA_conv (10 -> 64) (models.py:480-482, 1x1, weight-norm) as ONE block-diagonal
    // 1x1 conv from the stacked [F0; N; EMA] rows (12) to the 128 channels the decoder concatenates (models.py:503-505)
    const GemmW* fne(const float** bias_out, const float** w32_out = nullptr) const
    {
        const std::string key = "FNE:decoder";
        auto it = gemm.find(key);
        if (it == gemm.end()) {
            if (frozen) { if (!err) err = AS_EINVAL; return nullptr; }
            const HostT *f = host("decoder.F0_conv.weight"), *n = host("decoder.N_conv.weight"), *e = host("decoder.EMA_conv.weight");
            const HostT *fb = host("decoder.F0_conv.bias"), *nb = host("decoder.N_conv.bias"), *eb = host("decoder.EMA_conv.bias");
            if (!f || !n || !e || !fb || !nb || !eb) return nullptr;
            const int mf = f->dim(0), mn = n->dim(0), me = e->dim(0), ke = e->dim(1), K = 2 + ke, M = mf + mn + me;
            std::vector<float> w((size_t)M * K, 0.f), b;
            for (int m = 0; m < mf; ++m) w[(size_t)m * K + 0] = f->v[m];
            for (int m = 0; m < mn; ++m) w[(size_t)(mf + m) * K + 1] = n->v[m];
            for (int m = 0; m < me; ++m)
                for (int k = 0; k < ke; ++k) w[(size_t)(mf + mn + m) * K + 2 + k] = e->v[(size_t)m * ke + k];
            b.insert(b.end(), fb->v.begin(), fb->v.end());
            b.insert(b.end(), nb->v.begin(), nb->v.end());
            b.insert(b.end(), eb->v.begin(), eb->v.end());
            if (!gemm_from(key, w.data(), 1, M, K, 1)) return nullptr;
            Vec d;
            d.n = b.size();
            d.p = upload(b.data(), b.size());
            vecs[key] = d;
            Vec d32;                                                     // the same matrix in fp32 [M][K] (as_pointwise_small_f32)
            d32.n = w.size();
            d32.p = upload(w.data(), w.size());
            vecs[key + ":w32"] = d32;
            it = gemm.find(key);
        }
        *bias_out = vecs[key].p;
        if (w32_out) *w32_out = vecs[key + ":w32"].p;
        return &it->second;
    }
};

namespace {

// ------------------------------------------------------------------------------------------------------------------
// packed-frames geometry: B utterances, utterance b is an H x w[b] image (H = 1: a sequence)
// ------------------------------------------------------------------------------------------------------------------
struct Lay {
    int B = 0, H = 1, N = 0, max_w = 0;
    std::vector<int> w, off;
    int32_t *d_w = nullptr, *d_off = nullptr;
    uint64_t* d_meta = nullptr;
    // a CAPACITY layout (as_forward_io.frame_cap): N columns are room, the utterances' widths exist on the device only -- w / off stay
    // empty, d_w / d_off / d_meta / d_nvalid are rewritten by every call's as_dyn_geometry_launch (kind: AsDynGeo's layout index;
    // dyn_B utterances per group, cap1 half-rate columns of room)
    bool dyn = false;
    int dyn_kind = 0, dyn_B = 0, cap1 = 0;
    int32_t* d_nvalid = nullptr;
    std::map<std::string, int32_t*> tabs;      // further per-utterance device tables of launches on this layout
    int max_cols() const { return H * max_w; }
};

}  // namespace

struct as_plan {
    const as_model* model = nullptr;
    std::map<std::pair<std::vector<int>, int>, std::unique_ptr<Lay>> lays;
    DevPool pool{(size_t)8 << 20};
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> events;
    size_t next_event = 0;
    bool serial = false;                  // run the independent branches back to back on the calling stream (one chain per batch)
    bool merge = true;                    // (serial plans) conv GEMMs of independent branches share launches: as_plan_set_merge
    bool timing = false;                  // record phase marks on the calling stream (as_plan_phase_ms)
    int n_prod = 3;                       // matrix-core products per fp32 product (as_plan_set_operand_mode)
    hipEvent_t marks[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void mark(int i, hipStream_t s)
    {
        if (!timing) return;
        if (!marks[i] && hipEventCreate(&marks[i]) != hipSuccess) { marks[i] = nullptr; return; }
        (void)hipEventRecord(marks[i], s);
    }
    // Layout cache.  A key is a whole length vector, so a server that sees ever new ragged batches adds ~10-20 entries per batch.
    // trim() runs at the START of an entry point, when no `const Lay*` of an earlier call is alive: above the cap it waits for the
    // streams this plan has launched on (kernels of earlier calls may still read the tables) -- not for the device: other plans' work
    // goes on --, drops every layout and rewinds the table pool (the memory is kept: a hipFree would synchronise the device), so neither
    // the host map nor device memory grows without bound.  Never while `s` is being captured (a synchronisation is illegal there): the
    // trim then waits for the next entry point.  A hipGraph captured from this plan holds table addresses: captured geometries get a plan
    // of their own that is reset only together with its graphs (as_plan_reset_layouts; csrc/lanes.hip does exactly that), or the owner
    // watches layout_flushes.
    size_t lay_cap = 4096;
    int layout_flushes = 0;
    std::vector<hipStream_t> used;        // calling streams of the run entry points since the last flush
    void note_stream(hipStream_t s)
    {
        if (std::find(used.begin(), used.end(), s) == used.end()) used.push_back(s);
    }
    int drop_layouts()
    {
        lays.clear();
        pool.rewind();
        lstm_xchg = nullptr;
        lstm_xchg_bytes = 0;
        used.clear();
        ++layout_flushes;
        return AS_OK;
    }
    int trim(hipStream_t s)
    {
        if (lays.size() <= lay_cap) return AS_OK;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return AS_OK; }
        if (cs != hipStreamCaptureStatusNone) return AS_OK;
        note_stream(s);
        for (hipStream_t u : used) {
            const hipError_t e = hipStreamSynchronize(u);
            if (e != hipSuccess) return (int)e;
        }
        for (hipStream_t u : side) {
            const hipError_t e = hipStreamSynchronize(u);
            if (e != hipSuccess) return (int)e;
        }
        return drop_layouts();
    }
    std::vector<int> frames_host;         // as_forward_test with unknown frame counts reads them here
    void* lstm_xchg = nullptr;            // as_bilstm_cluster_f32's exchange buffer (zero-filled once, then the library's)
    size_t lstm_xchg_bytes = 0;

    hipEvent_t event()
    {
        if (next_event == events.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            events.push_back(e);
        }
        return events[next_event++];
    }
    hipStream_t stream(int i)
    {
        while ((int)side.size() <= i) {
            hipStream_t s;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
            side.push_back(s);
        }
        return side[i];
    }
};

namespace {

// Deferred launches (serial plans with `merge`): inside a Fork every launch of a branch is RECORDED into the branch's queue instead of being
// enqueued; the root Fork's join() then plays the queues out on the calling stream in an order that keeps every queue's own order and the
// fork / join edges, and hands conv GEMMs that are ready at the same time to ONE launch (as_conv_gemm_multi_f32): on one stream a step costs
// the sum of its kernels' durations, and a 40-tile conv beside a 2 000-tile one costs next to nothing.
struct Op {
    int kind = 0;                                  // 0: a recorded launch, 1: a conv GEMM (as_conv_gemm_f32 arguments), 2: wait for other queues,
                                                   // 3: a tower down-sampling step (as_down_multi_f32 arguments)
    std::function<int()> fn;
    ConvGemmArgs g;
    AsAdainArgs post;                              // kind 1: the AdaIN that reads the conv's result (post.yh NULL: none), as_conv_gemm_multi_post_f32
    int post_max_w = 0;
    AsLnArgs post_ln;                              // kind 1: ... or the channel LayerNorm that does (post_ln.yh NULL: none)
    AsDownArgs d;
    hipStream_t s = nullptr;
    double hint_f = 0, hint_b = 0;                 // as_prof_hint that goes with the launch
    std::vector<std::pair<int, size_t>> deps;      // kind 2: queue q has played >= n ops
    const char* what = nullptr;
    int line = 0;
    bool side = false;                             // kind 0: a long launch on a handful of workgroups (the duration predictor's recurrence over
                                                   // hundreds of tokens): play() puts it on the plan's side stream and parks its queue, so that
                                                   // the other queues' launches run beside it
};
struct Sched {
    std::vector<std::vector<Op>> q;
    int new_queue() { q.emplace_back(); return (int)q.size() - 1; }
};

struct Ctx {
    const as_model& m;
    as_plan& p;
    hipStream_t s;                // the stream launches go to (a side stream inside a Fork)
    char* base;                   // workspace (nullptr when counting)
    size_t cap, off = 0;
    bool launch;                  // false: allocate only (count pass; prepare pass; the first half of as_forward_test_finish)
    bool count;                   // true: nothing behind the arena, geometry tables stay on the host
    int rc = 0;
    std::shared_ptr<Sched> sched; // launches are being recorded (inside a Fork of a serial, merging plan)
    int cur_q = -1;
    double hint_f = 0, hint_b = 0;
    bool deferring() const { return sched != nullptr && cur_q >= 0; }
    Op& push(int kind)
    {
        sched->q[cur_q].emplace_back();
        Op& o = sched->q[cur_q].back();
        o.kind = kind;
        o.s = s;
        o.hint_f = hint_f; o.hint_b = hint_b;
        hint_f = hint_b = 0;
        return o;
    }
    void hint(double f, double b)                  // as_prof_hint for the NEXT launch: it has to travel with a recorded one
    {
        if (deferring()) { hint_f = f; hint_b = b; }
        else as_prof_hint(f, b);
    }

    Ctx(const as_model& m_, as_plan& p_, hipStream_t s_, void* ws, size_t ws_bytes, bool launch_, bool count_)
        : m(m_), p(p_), s(s_), base(static_cast<char*>(ws)), cap(ws_bytes), launch(launch_), count(count_) {}
    void fail(int r, const char* what = nullptr, int line = 0)
    {
        if (!rc) {
            rc = r;
            if (getenv("AS_DEBUG")) fprintf(stderr, "artspeech_hip: model.hip:%d: rc %d %s\n", line, r, what ? what : "");
        }
    }
    void* raw_alloc(size_t bytes)
    {
        const size_t o = off;
        off += align256(bytes ? bytes : 1);
        if (launch && !count && getenv("AS_DEBUG_ALLOC")) fprintf(stderr, "artspeech_hip: arena %p + %zu : %zu bytes\n", (void*)base, o, bytes);
        // counting: a non-null placeholder (never dereferenced: nothing launches), so that code which branches on "is there an operand
        // image" takes the branch the run takes (their workspace needs differ)
        if (count) return reinterpret_cast<void*>((size_t)1 << 20) ;
        if (off > cap) { fail(AS_ENOSPC); return nullptr; }
        return base + o;
    }
    float* f32(size_t n) { return static_cast<float*>(raw_alloc(n * sizeof(float))); }
    int32_t* i32(size_t n) { return static_cast<int32_t*>(raw_alloc(n * sizeof(int32_t))); }
    uint16_t* image(int K, int N) { return static_cast<uint16_t*>(raw_alloc(as_split_f16x2_bytes(K, N > 0 ? N : 1))); }
    bool go() const { return launch && rc == 0 && m.err == 0; }

    // geometry (cached in the plan; device tables created on first real use: a blocking upload)
    const Lay* lay(const std::vector<int>& widths, int H = 1)
    {
        auto key = std::make_pair(widths, H);
        auto it = p.lays.find(key);
        Lay* L;
        if (it == p.lays.end()) {
            auto u = std::make_unique<Lay>();
            L = u.get();
            L->B = (int)widths.size();
            L->H = H;
            L->w = widths;
            L->off.resize(L->B + 1);
            L->off[0] = 0;
            for (int b = 0; b < L->B; ++b) {
                if (widths[b] < 0 || widths[b] > AS_META_MAX_W || H > AS_META_MAX_H) { fail(AS_EINVAL); return nullptr; }
                L->off[b + 1] = L->off[b] + H * widths[b];
                L->max_w = std::max(L->max_w, widths[b]);
            }
            L->N = L->off[L->B];
            p.lays[key] = std::move(u);                        // (never evicted inside a call: as_plan::trim runs between calls)
        } else {
            L = it->second.get();
        }
        if (!count && launch && !L->d_off) {
            L->d_w = static_cast<int32_t*>(p.pool.alloc((L->B + 1) * sizeof(int32_t)));
            L->d_off = static_cast<int32_t*>(p.pool.alloc((L->B + 1) * sizeof(int32_t)));
            if (!L->d_w || !L->d_off || hipMemcpy(L->d_w, L->w.data(), L->B * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(L->d_off, L->off.data(), (L->B + 1) * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
                fail((int)hipErrorOutOfMemory);
                return nullptr;
            }
        }
        return L;
    }
    // capacity layout `kind` (AsDynGeo: 0 half rate, 1 mel rate, 2 / 3 the batch three times) of B utterances in cap1 half-rate columns
    const Lay* dyn_lay(int kind, int B, int cap1)
    {
        if (kind < 0 || kind > 3 || B < 1 || cap1 < 1 || (double)cap1 * 6.0 > (double)AS_META_MAX_W) { fail(AS_EINVAL); return nullptr; }
        auto key = std::make_pair(std::vector<int>{-1 - kind, B, cap1}, 1);
        auto it = p.lays.find(key);
        Lay* L;
        if (it == p.lays.end()) {
            auto u = std::make_unique<Lay>();
            L = u.get();
            L->dyn = true; L->dyn_kind = kind; L->dyn_B = B; L->cap1 = cap1;
            L->B = kind < 2 ? B : 3 * B;
            L->H = 1;
            L->max_w = cap1 * ((kind & 1) ? 2 : 1);
            L->N = L->max_w * (kind < 2 ? 1 : 3);
            p.lays[key] = std::move(u);
        } else {
            L = it->second.get();
        }
        if (!count && launch && !L->d_off) {
            L->d_w = static_cast<int32_t*>(p.pool.alloc((L->B + 1) * sizeof(int32_t)));
            L->d_off = static_cast<int32_t*>(p.pool.alloc((L->B + 1) * sizeof(int32_t)));
            L->d_meta = static_cast<uint64_t*>(p.pool.alloc((size_t)L->N * sizeof(uint64_t)));
            L->d_nvalid = static_cast<int32_t*>(p.pool.alloc(sizeof(int32_t)));
            int32_t* src3 = kind == 2 ? static_cast<int32_t*>(p.pool.alloc((size_t)L->B * sizeof(int32_t))) : nullptr;
            if (!L->d_w || !L->d_off || !L->d_meta || !L->d_nvalid || (kind == 2 && !src3)) {
                L->d_off = nullptr;
                fail((int)hipErrorOutOfMemory);
                return nullptr;
            }
            if (src3) L->tabs["src3"] = src3;
        }
        return L;
    }
    const uint64_t* meta(const Lay* L)
    {
        if (!L || count || !launch) return nullptr;
        if (L->dyn) return L->d_meta;                              // (written by the call's as_dyn_geometry_launch)
        Lay* M = const_cast<Lay*>(L);
        if (!M->d_meta) {
            M->d_meta = static_cast<uint64_t*>(p.pool.alloc((size_t)std::max(L->N, 1) * sizeof(uint64_t)));
            if (!M->d_meta) { fail((int)hipErrorOutOfMemory); return nullptr; }
            const int r = as_make_meta(L->d_w, L->d_off, L->B, L->H, L->N, M->d_meta, s);
            if (r != AS_OK || hipStreamSynchronize(s) != hipSuccess) { fail(r ? r : (int)hipErrorUnknown); return nullptr; }
        }
        return M->d_meta;
    }
    // a per-utterance int32 table that belongs to layout L (built and uploaded on first real use, like L's own tables)
    template <typename F>
    const int32_t* itable(const Lay* L, const std::string& key, F&& build)
    {
        if (!L || count || !launch) return nullptr;
        Lay* M = const_cast<Lay*>(L);
        auto it = M->tabs.find(key);
        if (it != M->tabs.end()) return it->second;
        const std::vector<int32_t> h = build();
        int32_t* d = static_cast<int32_t*>(p.pool.alloc(h.size() * sizeof(int32_t)));
        if (!d || hipMemcpy(d, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { fail((int)hipErrorOutOfMemory); return nullptr; }
        return M->tabs[key] = d;
    }
    // the plan's exchange buffer for clustered recurrences (allocated and zero-filled on first real use, like the tables above)
    void* lstm_xchg(int n_jobs, int B, size_t* bytes)
    {
        *bytes = 0;
        if (count || !launch) return nullptr;
        const size_t need = as_bilstm_cluster_bytes(n_jobs, B);
        if (p.lstm_xchg_bytes < need) {
            void* d = p.pool.alloc(need);
            if (!d || hipMemset(d, 0, need) != hipSuccess) { fail((int)hipErrorOutOfMemory); return nullptr; }
            p.lstm_xchg = d;
            p.lstm_xchg_bytes = need;
        }
        *bytes = p.lstm_xchg_bytes;
        return p.lstm_xchg;
    }
    const Lay* scaled(const Lay* L, int k)
    {
        if (L->dyn) {
            if (k != 2 || (L->dyn_kind & 1)) { fail(AS_EINVAL); return nullptr; }
            return dyn_lay(L->dyn_kind + 1, L->dyn_B, L->cap1);
        }
        std::vector<int> w(L->w);
        for (int& v : w) v *= k;
        return lay(w, L->H);
    }
    const Lay* halved(const Lay* L, bool h_too)                 // W -> ceil(W/2); H -> H/2 when h_too
    {
        std::vector<int> w(L->w);
        for (int& v : w) v = (v + 1) / 2;
        return lay(w, h_too ? L->H / 2 : L->H);
    }
    const Lay* valid_conv(const Lay* L, int K, int stride)
    {
        std::vector<int> w(L->w);
        for (int& v : w) v = v >= K ? (v - K) / stride + 1 : 0;
        return lay(w, L->H >= K ? (L->H - K) / stride + 1 : 0);
    }
};

#ifdef AS_EXPERIMENTS
// timing experiments only (results are wrong): AS_EXP_SKIP="avgpool,dwconv" drops the RUN launches whose call text holds one of the words
static bool exp_skip(const char* call)
{
    static const char* e = getenv("AS_EXP_SKIP");
    if (!e || !*e) return false;
    std::string words(e);
    size_t a = 0;
    while (a <= words.size()) {
        size_t b = words.find(',', a);
        if (b == std::string::npos) b = words.size();
        if (b > a && strstr(call, words.substr(a, b - a).c_str())) return true;
        a = b + 1;
    }
    return false;
}
// AS_EXP_DUP="adain": those launches run TWICE (same arguments: results unchanged) -- what a class costs the step at the margin, without
// the knock-out's side effect on the data the later kernels see
static bool exp_dup(const char* call)
{
    static const char* e = getenv("AS_EXP_DUP");
    if (!e || !*e) return false;
    std::string words(e);
    size_t a = 0;
    while (a <= words.size()) {
        size_t b = words.find(',', a);
        if (b == std::string::npos) b = words.size();
        if (b > a && strstr(call, words.substr(a, b - a).c_str())) return true;
        a = b + 1;
    }
    return false;
}
#define EXP_SKIP(call) exp_skip(#call)
#define EXP_DUP(call) exp_dup(#call)
#else
#define EXP_SKIP(call) false
#define EXP_DUP(call) false
#endif
// (a recorded launch that play() moves to the plan's side stream: Op.side)
static thread_local hipStream_t tl_stream_override = nullptr;
// (recorded form: the closure copies what the call names -- argument structs and job arrays included -- and sees the stream as `c.s`)
#define RUN(c, call)                                   \
    do {                                               \
        if ((c).go() && !EXP_SKIP(call)) {             \
            if ((c).deferring()) {                     \
                const hipStream_t s__ = (c).s;         \
                Op& o__ = (c).push(0);                 \
                o__.what = #call; o__.line = __LINE__; \
                o__.fn = [=]() -> int {                \
                    struct { hipStream_t s; } c = {tl_stream_override ? tl_stream_override : s__}; \
                    (void)c;                           \
                    int r__ = (call);                  \
                    if (r__ == AS_OK && EXP_DUP(call)) r__ = (call); \
                    return r__;                        \
                };                                     \
            } else {                                   \
                int r__ = (call);                      \
                if (r__ == AS_OK && EXP_DUP(call)) r__ = (call); \
                if (r__ != AS_OK) (c).fail(r__, #call, __LINE__); \
            }                                          \
        }                                              \
    } while (0)

// Play the recorded queues out on stream `s` (see Op).  Greedy: every queue runs ahead through its plain launches; when every live head
// is a conv GEMM (or waits for another queue), the heads that can share a launch -- operand images in, the tiled kernel, one row class --
// go out together.
static bool gemm_mergeable(const ConvGemmArgs& g)
{
    if (!g.Xh || g.N <= 0) return false;
    int32_t kind = 0, tile = 0, slices = 0;
    return as_conv_gemm_plan(&g, &kind, &tile, &slices) == AS_OK && kind == 1;
}
static bool gemm_direct(const ConvGemmArgs& g)           // the Cin = 1 direct kernel's fast form
{
    int32_t kind = 1, tile = 0, slices = 0;
    return g.N > 0 && g.T <= 9 && as_conv_gemm_plan(&g, &kind, &tile, &slices) == AS_OK && kind == 0;
}
static bool gemm_tall(int M) { return M > 64 && (M % 128 == 0 || M % 128 > 64 || M >= 512); }   // (a 128-row tile is not half empty: conv_gemm.hip)

static void play(Ctx& c, Sched& S)
{
    const int nq = (int)S.q.size();
    std::vector<size_t> head(nq, 0);
    // a queue whose last launch went to the side stream (Op.side) is PARKED: its later ops wait until nothing else can go out, then the
    // calling stream waits for the side stream's event and the queue moves on.  Dependencies see a parked queue one op back.
    std::vector<hipEvent_t> parked(nq, nullptr);
    const bool no_side = getenv("AS_NO_SIDE_LSTM") != nullptr;          // experiments / tests: everything on the one stream
    auto unpark_all = [&]() -> bool {
        bool any = false;
        for (int qi = 0; qi < nq; ++qi)
            if (parked[qi]) {
                if (hipStreamWaitEvent(c.s, parked[qi], 0) != hipSuccess) c.fail((int)hipErrorUnknown, "unpark", __LINE__);
                parked[qi] = nullptr;
                any = true;
            }
        return any;
    };
    auto fail = [&](int r, const Op& o) { c.fail(r, o.what, o.line); };
    static const bool no_merge = getenv("AS_NO_MERGE") != nullptr;    // experiments: the recorded order, one launch per conv
    static const bool trace = getenv("AS_DEBUG_SCHED") != nullptr;     // print what goes out, in order
    if (trace) {
        fprintf(stderr, "artspeech_hip: playing %d recorded queues:", nq);
        for (int qi = 0; qi < nq; ++qi) fprintf(stderr, " %zu", S.q[qi].size());
        fprintf(stderr, " ops\n");
    }
    for (;;) {
        bool progress = false;
        for (int qi = 0; qi < nq && !c.rc; ++qi) {
            while (head[qi] < S.q[qi].size() && !parked[qi]) {
                Op& o = S.q[qi][head[qi]];
                if (o.kind == 2) {
                    bool ok = true;
                    for (auto& d : o.deps) ok = ok && head[d.first] - (parked[d.first] ? 1 : 0) >= d.second;
                    if (!ok) break;
                } else if (o.kind == 0) {
                    if (o.hint_f > 0 || o.hint_b > 0) as_prof_hint(o.hint_f, o.hint_b);
                    if (trace) fprintf(stderr, "  q%d  %.60s%s\n", qi, o.what ? o.what : "?", o.side && !no_side ? "  [side stream]" : "");
                    if (o.side && !no_side) {
                        // fork: the side stream continues from what the calling stream holds so far; join: when the queue is unparked
                        hipStream_t ss = c.p.stream(0);
                        hipEvent_t e1 = c.p.event(), e2 = c.p.event();
                        if (!ss || !e1 || !e2 || hipEventRecord(e1, c.s) != hipSuccess || hipStreamWaitEvent(ss, e1, 0) != hipSuccess) {
                            fail((int)hipErrorUnknown, o);
                            break;
                        }
                        tl_stream_override = ss;
                        const int r = o.fn();
                        tl_stream_override = nullptr;
                        if (r != AS_OK) { fail(r, o); break; }
                        if (hipEventRecord(e2, ss) != hipSuccess) { fail((int)hipErrorUnknown, o); break; }
                        parked[qi] = e2;
                        ++head[qi];
                        progress = true;
                        break;
                    }
                    const int r = o.fn();
                    if (r != AS_OK) { fail(r, o); break; }
                } else {
                    break;                                               // a conv GEMM / a down-sampling step: decided below, with the other queues' heads
                }
                ++head[qi];
                progress = true;
            }
        }
        if (c.rc) return;
        if (progress) continue;
        // every live head is a GEMM, a down-sampling step or a wait.  The towers' down-sampling steps that are ready together go out as one
        // launch first (they are what the towers' next convs wait for)
        {
            AsDownArgs dl[AS_MAX_MULTI];
            int dq[AS_MAX_MULTI], nd = 0;
            double hf = 0, hb = 0;
            for (int qi = 0; qi < nq && nd < AS_MAX_MULTI; ++qi)
                if (!parked[qi] && head[qi] < S.q[qi].size() && S.q[qi][head[qi]].kind == 3) {
                    const Op& o = S.q[qi][head[qi]];
                    dl[nd] = o.d;
                    dq[nd++] = qi;
                    hf += o.hint_f;
                    hb += o.hint_b;
                }
            if (nd > 0) {
                if (hf > 0 || hb > 0) as_prof_hint(hf, hb);
                if (trace) {
                    fprintf(stderr, "  DOWN x%d:", nd);
                    for (int i = 0; i < nd; ++i) fprintf(stderr, " q%d kind%d C%d |", dq[i], dl[i].kind, dl[i].C);
                    fprintf(stderr, "\n");
                }
                const Op& o0 = S.q[dq[0]][head[dq[0]]];
                const int r = no_merge ? AS_OK : as_down_multi_f32(dl, nd, o0.s);
                if (no_merge)
                    for (int i = 0; i < nd && !c.rc; ++i) {
                        const int r1 = as_down_multi_f32(&dl[i], 1, o0.s);
                        if (r1 != AS_OK) fail(r1, o0);
                    }
                if (r != AS_OK) { fail(r, o0); return; }
                if (c.rc) return;
                for (int i = 0; i < nd; ++i) ++head[dq[i]];
                continue;
            }
        }
        int heads[64], nh = 0;
        bool live = false;
        for (int qi = 0; qi < nq; ++qi)
            if (head[qi] < S.q[qi].size()) {
                live = true;
                if (!parked[qi] && S.q[qi][head[qi]].kind == 1 && nh < 64) heads[nh++] = qi;
            }
        if (nh == 0 && unpark_all()) continue;                               // nothing else can go out: the parked queues move on
        if (!live) return;
        if (nh == 0) { c.fail(AS_EINVAL, "recorded queues wait for each other", __LINE__); return; }
        // a queue with a side-stream launch still ahead of it goes FIRST and alone: its convs are what that launch waits for, and every
        // conv of another queue that goes out before it is one that could have run beside it (the duration predictor's blocks before its
        // recurrence; the encoders' last two layers then run while the recurrence does)
        if (!no_side) {
            int nu = 0, uh[64];
            for (int i = 0; i < nh; ++i) {
                const std::vector<Op>& Q = S.q[heads[i]];
                bool urgent = false;
                for (size_t k = head[heads[i]]; k < Q.size() && !urgent; ++k) urgent = Q[k].side;
                if (urgent) uh[nu++] = heads[i];
            }
            if (nu > 0 && nu < nh) {
                for (int i = 0; i < nu; ++i) heads[i] = uh[i];
                nh = nu;
            }
        }
        // a head that cannot share a launch (fp32 input still to be split, the direct Cin = 1 kernel) goes out first and alone: its queue
        // moves on to heads that can.  Otherwise the set = the mergeable heads of the row class that holds the most work.
        int pick[AS_MAX_MULTI], np = 0, lone = -1;
        for (int i = 0; i < nh && lone < 0; ++i)
            if (no_merge || !gemm_mergeable(S.q[heads[i]][head[heads[i]]].g)) lone = heads[i];
        if (lone >= 0 && !no_merge && gemm_direct(S.q[lone][head[lone]].g)) {        // the towers' Cin = 1 stems that are ready together: one direct launch
            for (int i = 0; i < nh && np < AS_MAX_MULTI; ++i)
                if (gemm_direct(S.q[heads[i]][head[heads[i]]].g)) pick[np++] = heads[i];
            if (np < 2) np = 0;
        }
        if (lone < 0) {
            double work[2] = {0, 0};
            for (int i = 0; i < nh; ++i) {
                const ConvGemmArgs& g = S.q[heads[i]][head[heads[i]]].g;
                work[gemm_tall(g.M) ? 1 : 0] += (double)g.M * g.N * ((double)g.K * g.T + g.K2);
            }
            const int cls = work[1] >= work[0] ? 1 : 0;
            // (a 64-channel conv may ride with a set of 128-row problems when it is a small part of the work: as_conv_gemm_multi_tile)
            const bool ride = cls == 1 && work[0] <= 0.1 * work[1];
            for (int i = 0; i < nh && np < AS_MAX_MULTI; ++i) {
                const ConvGemmArgs& g = S.q[heads[i]][head[heads[i]]].g;
                if (((gemm_tall(g.M) ? 1 : 0) == cls || ride) && (np == 0 || g.n_prod == S.q[pick[0]][head[pick[0]]].g.n_prod)) pick[np++] = heads[i];
            }
            if (np < 2) lone = pick[0];
        }
        if (np >= 2) {
            ConvGemmArgs list[AS_MAX_MULTI];
            AsAdainArgs posts[AS_MAX_MULTI];
            AsLnArgs lns[AS_MAX_MULTI];
            int32_t pmw[AS_MAX_MULTI];
            for (int i = 0; i < np; ++i) {
                const Op& oi = S.q[pick[i]][head[pick[i]]];
                list[i] = oi.g;
                posts[i] = oi.post;
                lns[i] = oi.post_ln;
                pmw[i] = oi.post_max_w;
            }
            if (trace) {
                fprintf(stderr, "  GEMM x%d:", np);
                for (int i = 0; i < np; ++i) fprintf(stderr, " q%d M%d N%d K%d T%d |", pick[i], list[i].M, list[i].N, list[i].K, list[i].T);
                fprintf(stderr, "\n");
            }
            const int r = as_conv_gemm_multi_post_f32(list, posts, pmw, lns, np, S.q[pick[0]][head[pick[0]]].s);
            if (r != AS_OK) { fail(r, S.q[pick[0]][head[pick[0]]]); return; }
            for (int i = 0; i < np; ++i) ++head[pick[i]];
        } else {
            Op& o = S.q[lone][head[lone]];
            if (trace) fprintf(stderr, "  GEMM alone: q%d M%d N%d K%d T%d (%d heads)\n", lone, o.g.M, o.g.N, o.g.K, o.g.T, nh);
            const int32_t mw1 = o.post_max_w;
            const int r = as_conv_gemm_multi_post_f32(&o.g, &o.post, &mw1, &o.post_ln, 1, o.s);
            if (r != AS_OK) { fail(r, o); return; }
            ++head[lone];
        }
    }
}

// Fork / join of independent branches over the plan's side streams: every branch first waits for the calling stream, the
// calling stream then waits for every branch (hipGraph capture records them as parallel nodes).
struct Fork {
    Ctx& c;
    hipStream_t main;
    int n, first;
    bool on_side;
    bool defer = false, root = false;      // recorded form (serial plans with `merge`): branch i = queue qs[i] of the Ctx's Sched
    int q0 = -1;
    std::vector<int> qs;
    Fork(Ctx& c_, int n_, int first_) : c(c_), main(c_.s), n(n_), first(first_), on_side(c_.go() && !c_.p.serial)
    {
        defer = c.go() && c.p.serial && c.p.merge;
        if (defer) {
            if (!c.sched) {
                c.sched = std::make_shared<Sched>();
                c.cur_q = c.sched->new_queue();                        // what the calling stream does from here on
                root = true;
            }
            q0 = c.cur_q;
            for (int i = 0; i < n; ++i) {
                const int qi = c.sched->new_queue();
                qs.push_back(qi);
                c.cur_q = qi;
                c.push(2).deps.push_back({q0, c.sched->q[q0].size()}); // a branch starts after what the calling stream has recorded so far
            }
            c.cur_q = q0;
            return;
        }
        if (!on_side) return;
        hipEvent_t e = c.p.event();
        if (!e || hipEventRecord(e, main) != hipSuccess) { c.fail((int)hipErrorUnknown); return; }
        for (int i = 0; i < n; ++i) {
            hipStream_t st = c.p.stream(first + i);
            if (!st || hipStreamWaitEvent(st, e, 0) != hipSuccess) { c.fail((int)hipErrorUnknown); return; }
        }
    }
    void branch(int i)
    {
        if (defer) { c.cur_q = qs[i]; return; }
        c.s = on_side ? c.p.stream(first + i) : main;
    }
    void back()                            // continue on the calling stream while the branches run; join() later
    {
        if (defer) { c.cur_q = q0; return; }
        c.s = main;
    }
    void wait_main(int i)                  // branch i continues only after what the calling stream has enqueued so far
    {
        if (defer) {
            const int keep = c.cur_q;
            c.cur_q = qs[i];
            c.push(2).deps.push_back({q0, c.sched->q[q0].size()});
            c.cur_q = keep;
            return;
        }
        if (!on_side) return;
        hipEvent_t e = c.p.event();
        if (!e || hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(c.p.stream(first + i), e, 0) != hipSuccess) c.fail((int)hipErrorUnknown);
    }
    void after(int i, std::initializer_list<int> js)   // (recorded form only) branch i goes on after everything branches js hold so far
    {
        if (!defer) return;
        const int keep = c.cur_q;
        c.cur_q = qs[i];
        Op& o = c.push(2);
        for (int j : js) o.deps.push_back({qs[j], c.sched->q[qs[j]].size()});
        c.cur_q = keep;
    }
    void join()
    {
        if (defer) {
            c.cur_q = q0;
            Op& o = c.push(2);
            for (int qi : qs) o.deps.push_back({qi, c.sched->q[qi].size()});
            if (root) {
                std::shared_ptr<Sched> S = c.sched;
                c.sched.reset();
                c.cur_q = -1;
                play(c, *S);
                S->q.clear();
            }
            return;
        }
        c.s = main;
        if (!on_side) return;
        for (int i = 0; i < n; ++i) {
            hipEvent_t e = c.p.event();
            if (!e || hipEventRecord(e, c.p.stream(first + i)) != hipSuccess || hipStreamWaitEvent(main, e, 0) != hipSuccess) {
                c.fail((int)hipErrorUnknown);
                return;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------------------------
struct Taps {
    int n = 0;
    int dh[AS_MAX_TAPS], dw[AS_MAX_TAPS];
};
Taps taps_1d(int k)
{
    Taps t;
    t.n = k;
    for (int i = 0; i < k; ++i) { t.dh[i] = 0; t.dw[i] = i - k / 2; }
    return t;
}
Taps taps_2d(int kh, int kw)
{
    Taps t;
    t.n = kh * kw;
    for (int a = 0; a < kh; ++a)
        for (int d = 0; d < kw; ++d) { t.dh[a * kw + d] = a - kh / 2; t.dw[a * kw + d] = d - kw / 2; }
    return t;
}

// gamma / beta of one AdaIN1d launch: AsAdainArgs addressing into the output of the fc GEMM ([rows][ldB], utterances as columns)
struct Norm {
    const float* gb = nullptr;
    const int32_t* gb_off = nullptr;   // per-utterance offsets (grouped launches); NULL: utterance u at gb + u
    int gb_sc = 0;                     // stride between channels = ldB
};

struct ConvOpt {
    const float* bias = nullptr;
    const float* res = nullptr;
    int ldr = 0;
    int act = ACT_NONE;
    bool div_sqrt2 = false;
    int in_act = 0;
    bool transpose_out = false;
    int group_cols = 0;
    bool want_yh = false;         // also / only write the output as the next conv's operand image `yh`
    uint16_t* yh = nullptr;
    bool yh_lrelu = false;
    bool in_image = false;        // set by conv(): the input is an operand image (xh), not fp32
    const uint16_t* x2h = nullptr; // second operand image (K2 channels) of a weight made by conv_fold
    int K2 = 0;
    const int32_t* src_col = nullptr;   // strided / valid conv: `lay` is the OUTPUT layout, the image has N_in columns, output column j
    const uint64_t* src_meta = nullptr; // reads input column src_col[j] (+ taps), src_meta[j] = that input position (ConvGemmArgs.src_col)
    int N_in = 0;
    bool probe = false;           // this launch tests its accumulators for inf / NaN (AS_PROBE_THIS): the path's last conv, always on
    // the AdaIN1d + LeakyReLU that reads this conv's result, written as the image post_yh by the same call (as_conv_gemm_multi_post_f32):
    // in the launch's reduction kernel when the conv is cut into K slices and no utterance is wider than 256 columns, else a launch behind it
    const Norm* post_n = nullptr;
    const Lay* post_lay = nullptr;
    uint16_t* post_yh = nullptr;
    // ... or the channel LayerNorm (+ ReLU) that reads it (the encoders: conv -> residual add -> LayerNorm -> conv): ln.yh = the image to write
    AsLnArgs ln = {nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0, nullptr};
};

// Y = epi(conv(W, X)); the input is fp32 X [K][ldx] (split by the library into the workspace) or the operand image xh
void conv_impl(Ctx& c, const GemmW* w, const float* X, int ldx, const uint16_t* xh, int K, const Lay* lay, const Taps& taps, float* Y,
               int ldy, const ConvOpt& o)
{
    if (!w || !lay) { c.fail(AS_EINVAL); return; }
    if (w->T != taps.n || K > w->Kp || K <= w->Kp - 16 || w->K2 != o.K2 || (o.K2 && !o.in_image)) { c.fail(AS_EINVAL); return; }
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.Wh = w->wh; a.W = w->w32; a.X = X; a.Xh = xh; a.Y = Y; a.Yh = o.yh;
    a.bias = o.bias; a.res = o.res;
    a.M = w->M; a.N = lay->N; a.K = K; a.T = w->T; a.Kp = w->Kp;
    a.ldx = X ? ldx : lay->N; a.ldy = ldy; a.ldr = o.ldr;
    a.act = o.act; a.div_sqrt2 = o.div_sqrt2; a.in_act = o.in_act; a.transpose_out = o.transpose_out; a.yh_lrelu = o.yh_lrelu;
    a.acc_scale = 1.0f / w->scale;
    a.in_slope = a.act_slope = AS_SLOPE_PATH;                           // LeakyReLU(0.2) everywhere on the path (models.py:163)
    a.n_prod = c.p.n_prod;
    a.n_groups = w->G; a.group_cols = o.group_cols;
    a.Xh2 = o.x2h; a.K2 = o.K2;
    a.N_in = o.N_in;
    a.range_probe = o.probe ? AS_PROBE_THIS : 0;
    for (int i = 0; i < taps.n; ++i) { a.dh[i] = taps.dh[i]; a.dw[i] = taps.dw[i]; }
    const bool pointwise = taps.n == 1 && taps.dh[0] == 0 && taps.dw[0] == 0;
    if (lay->N == 0) return;
    // A plan that records its branches (serial + merge): every conv should be able to share a launch, so fp32 activations are split into
    // their operand image by a launch of their own first (what as_conv_gemm_f32 would do inside the call), and the workspace holds the K
    // slices a merged launch may cut the problem into.  Decided by the PLAN's flags, the same in every pass.
    const bool rec = c.p.serial && c.p.merge;
    bool in_image = o.in_image;
    if (rec && !in_image && !(K == 1 && w->w32)) {
        uint16_t* img = c.image(K, lay->N);
        if (c.go()) RUN(c, as_split_f16x2_f32(X, ldx, K, a.N, a.in_act, a.in_slope, img, c.s));
        a.X = nullptr;
        a.Xh = img;
        a.in_act = 0;
        in_image = true;
    }
    // pointers that are null only because this pass does not run kernels must not change the plan the library makes
    ConvGemmArgs q = a;
    q.X = in_image ? nullptr : reinterpret_cast<const float*>(16);
    q.Xh = in_image ? reinterpret_cast<const uint16_t*>(16) : nullptr;
    q.Yh = o.want_yh ? reinterpret_cast<uint16_t*>(16) : nullptr;
    q.Xh2 = o.K2 ? reinterpret_cast<const uint16_t*>(16) : nullptr;
    const size_t wsb = rec ? as_conv_gemm_multi_workspace_bytes(&q) : as_conv_gemm_workspace_bytes(&q);
    if (getenv("AS_DEBUG_ALLOC"))
        fprintf(stderr, "artspeech_hip: conv %s M%d N%d K%d T%d G%d img%d -> ws %zu (arena at %zu)\n", c.count ? "count" : (c.launch ? "run" : "replay"), a.M,
                a.N, a.K, a.T, a.n_groups, (int)in_image, wsb, c.off);
    a.ws = wsb ? c.raw_alloc(wsb) : nullptr;
    a.ws_bytes = wsb;
    if (!c.go()) return;
    if (o.N_in && !o.src_col) { c.fail(AS_EINVAL); return; }
    a.meta = o.N_in ? o.src_meta : (pointwise ? nullptr : c.meta(lay));
    a.src_col = o.src_col;
    a.n_valid = lay->dyn ? lay->d_nvalid : nullptr;                     // (a capacity layout: the columns behind the utterances are filler)
    AsAdainArgs post;
    memset(&post, 0, sizeof(post));
    int32_t post_mw = 0;
    if (o.post_yh) {
        if (!o.post_n || !o.post_lay || o.post_lay->N != lay->N || !Y) { c.fail(AS_EINVAL); return; }
        post.gb = o.post_n->gb; post.gb_off = o.post_n->gb_off; post.ldgb = 1; post.gb_sc = o.post_n->gb_sc;
        post.col_off = o.post_lay->d_off; post.U = o.post_lay->B; post.lrelu = 1; post.yh = o.post_yh;
        post.col_w = o.post_lay->dyn ? o.post_lay->d_w : nullptr;
        post_mw = o.post_lay->max_w;                                      // (a capacity layout: no utterance is wider than the room)
    }
    if (o.ln.yh && (o.post_yh || !Y)) { c.fail(AS_EINVAL); return; }
    if (c.deferring() && !EXP_SKIP(as_conv_gemm_f32)) {                  // recorded: it may share its launch with other branches' convs
        Op& op = c.push(1);
        op.g = a;
        op.post = post;
        op.post_ln = o.ln;
        op.post_max_w = post_mw;
        op.what = "as_conv_gemm_f32";
        op.line = __LINE__;
        return;
    }
    const AsLnArgs ln1 = o.ln;
    RUN(c, as_conv_gemm_multi_post_f32(&a, &post, &post_mw, &ln1, 1, c.s));
}

// fp32 input X [K][ldx]
void conv_x(Ctx& c, const GemmW* w, const float* X, int ldx, int K, const Lay* lay, const Taps& taps, float* Y, int ldy, ConvOpt o)
{
    o.in_image = false;
    conv_impl(c, w, X, ldx, nullptr, K, lay, taps, Y, ldy, o);
}
// operand-image input xh (K channels)
void conv_h(Ctx& c, const GemmW* w, const uint16_t* xh, int K, const Lay* lay, const Taps& taps, float* Y, int ldy, ConvOpt o)
{
    o.in_image = true;
    conv_impl(c, w, nullptr, 0, xh, K, lay, taps, Y, ldy, o);
}
float* conv_x_new(Ctx& c, const GemmW* w, const float* X, int ldx, int K, const Lay* lay, const Taps& taps, const ConvOpt& o)
{
    if (!w || !lay) { c.fail(AS_EINVAL); return nullptr; }
    float* Y = c.f32((size_t)w->M * std::max(lay->N, 1));
    conv_x(c, w, X, ldx, K, lay, taps, Y, lay->N, o);
    return Y;
}
float* conv_h_new(Ctx& c, const GemmW* w, const uint16_t* xh, int K, const Lay* lay, const Taps& taps, const ConvOpt& o)
{
    if (!w || !lay) { c.fail(AS_EINVAL); return nullptr; }
    float* Y = c.f32((size_t)w->M * std::max(lay->N, 1));
    conv_h(c, w, xh, K, lay, taps, Y, lay->N, o);
    return Y;
}

// ------------------------------------------------------------------------------------------------------------------
// building blocks (the launch sequences of the reference's modules)
// ------------------------------------------------------------------------------------------------------------------

// Every AdaIN fc layer fed by `style` [B][lds] in one GEMM: gbT [Mtot][B] = W style^T + b (models.py:237).
// Returns the output; row0 of a layer through FcAll::row0.
struct FcOut {
    const as_model::FcAll* f = nullptr;
    float* gbT = nullptr;
    int B = 0;
    Norm norm(const std::string& name, int group_stride_rows = 0, const int32_t* gb_off = nullptr) const
    {
        Norm n;
        auto it = f ? f->row0.find(name) : decltype(f->row0.begin())();
        if (!f || it == f->row0.end()) return n;
        n.gb = gbT ? gbT + (size_t)it->second * B : nullptr;
        n.gb_off = gb_off;
        n.gb_sc = B;
        (void)group_stride_rows;
        return n;
    }
};

FcOut adain_fc_all(Ctx& c, const std::string& key, const std::vector<as_model::NormSpec>& norms, const float* style, int lds, int K, int B)
{
    FcOut o;
    o.B = B;
    o.f = c.m.fc_all(key, norms, K);
    if (!o.f) { c.fail(AS_EINVAL); return o; }
    uint16_t* sh = c.image(K, B);
    RUN(c, as_rows_image_f32(style, lds, K, B, sh, c.s));
    const Lay* lb = c.lay(std::vector<int>(B, 1));
    if (!lb) return o;
    o.gbT = c.f32((size_t)o.f->Mtot * B);
    ConvOpt q;
    q.bias = o.f->bias;
    conv_h(c, o.f->w, sh, K, lb, taps_1d(1), o.gbT, B, q);
    return o;
}

void adain_image(Ctx& c, const float* x, int ldx, int C, const Norm& n, const Lay* lay, const int32_t* src_off, int N_out, uint16_t* yh,
                 const float* pool_w = nullptr, const float* pool_b = nullptr, float* x_up = nullptr, int ld_up = 0)
{
    if (!lay) { c.fail(AS_EINVAL); return; }
    if (!c.go() || N_out == 0) return;
    AsAdainArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.C = C;
    a.gb = n.gb; a.gb_off = n.gb_off; a.ldgb = 1; a.gb_sc = n.gb_sc;
    a.col_off = lay->d_off; a.src_off = src_off; a.U = lay->B; a.N = N_out; a.lrelu = 1; a.yh = yh;
    a.col_w = lay->dyn ? lay->d_w : nullptr;
    a.pool_w = pool_w; a.pool_b = pool_b; a.x_up = x_up; a.ld_up = ld_up;
    RUN(c, as_adain_image_f32(&a, c.s));
}

struct Act {                     // fp32 activation [C][ld] on a layout (+ optionally its raw operand image)
    float* p = nullptr;
    int C = 0, ld = 0;
    const Lay* lay = nullptr;
    const uint16_t* h = nullptr;
    // the operand image of LeakyReLU(AdaIN(p)) under the NEXT block's norm1, written by the launch that produced p (BlkOpt.next_n1): that
    // block's first conv reads it and launches no AdaIN of its own
    const uint16_t* pre = nullptr;
    // a tower stem's output that exists only as its LeakyReLU image h (p null): the one-channel input and the stem's fp32 weights,
    // from which the first block's shortcut is computed directly (as_stem_pool_image_f32)
    const float* stem_x = nullptr;
    const GemmW* stem_w = nullptr;
    const float* stem_b = nullptr;
    int stem_kh = 0;
};

// AdainResBlk1d.forward (models.py:189-202), G blocks of the same shape side by side along the column axis when names.size() > 1
// (the F0 / energy / TV branches, models.py:606-618: weight set g serves the columns [g * group_cols, (g + 1) * group_cols)).
struct BlkOpt {
    std::vector<std::string> names;    // block prefixes (one per group)
    int group_cols = 0;                // columns per group of X's layout (ignored for one group)
    Norm n1, n2;
    bool upsample = false;
    const int32_t* src_off = nullptr;  // (upsample, grouped) where each utterance of the OUTPUT layout reads X
    const Lay* lay_out = nullptr;      // (upsample) the pre-doubling layout the outputs are laid out on (default X.lay)
    float* out = nullptr;              // destination (row stride ldo), or null: from the workspace
    int ldo = 0;
    bool want_yh = false;              // also write the output as an operand image
    uint16_t* yh = nullptr;            // (where; null with want_yh: from the workspace)
    const Norm* next_n1 = nullptr;     // norm1 of the block that reads this block's output (same layout, no up-sampling): conv2's call also
                                       // writes that block's conv1 operand (Act.pre) -- in its reduction kernel at batch-1 sizes
};

Act adain_resblk1d(Ctx& c, const Act& X, BlkOpt o)
{
    const as_model& m = c.m;
    Act Y;
    const int G = (int)o.names.size();
    auto sfx = [&](const char* s) { std::vector<std::string> v(o.names); for (auto& n : v) n += s; return v; };
    const bool has_sc = m.has(o.names[0] + ".conv1x1.weight");
    // A learned shortcut whose input exists as an operand image is summed by conv2's launch (K2 more channels of its reduction).
    // (The decoder's in-place blocks give that launch another image to write than the one it reads: decoder().)
    const bool fold = has_sc && X.h && !o.upsample;                     // (`yh`, if given, must not be the image X.h: that launch reads it)
    const GemmW *w1 = m.conv_stack(sfx(".conv1")), *w2 = fold ? m.conv_fold(sfx(".conv2"), sfx(".conv1x1")) : m.conv_stack(sfx(".conv2"));
    if (!w1 || !w2 || !X.lay) { c.fail(AS_EINVAL); return Y; }
    const int din = X.C, dout = w1->M;
    const Lay* lay_in = o.upsample && o.lay_out ? o.lay_out : X.lay;     // utterances of the block's (pre-doubling) output layout
    const Lay* lay2 = o.upsample ? c.scaled(lay_in, 2) : lay_in;
    if (!lay2) return Y;
    const int N2 = lay2->N, Nn2 = std::max(N2, 1);
    const int gc2 = G > 1 ? (o.upsample ? 2 * o.group_cols : o.group_cols) : 0;
    float* out = o.out;
    int ldo = o.ldo;
    if (!out) { out = c.f32((size_t)dout * Nn2); ldo = N2; }
    const Taps k3 = taps_1d(3), k1 = taps_1d(1);
    // norm1 -> LeakyReLU (-> depthwise ConvTranspose1d x2, models.py:172,195) exists only as conv1's operand image
    const bool have_pre = X.pre && !o.upsample;
    uint16_t* xs = have_pre ? const_cast<uint16_t*>(X.pre) : c.image(din, N2);
    const float* sc = X.p;
    int ldsc = X.ld;
    if (have_pre) {
        // (the producer's call wrote it)
    } else if (o.upsample) {
        float* up = c.f32((size_t)din * Nn2);                           // shortcut = nearest x2 (models.py:184,261-270)
        const float *pw = m.vec_stack(sfx(".pool.weight")), *pb = m.vec_stack(sfx(".pool.bias"));
        if (G > 1) { c.fail(AS_EINVAL); return Y; }                      // (grouped up-sampling blocks go through adain_image per group: see arts_predictor)
        adain_image(c, X.p, X.ld, din, o.n1, lay_in, o.src_off, N2, xs, pw, pb, up, N2);
        sc = up;
        ldsc = N2;
    } else {
        adain_image(c, X.p, X.ld, din, o.n1, lay_in, nullptr, N2, xs);
    }
    // norm2 -> LeakyReLU exists only as conv2's operand image, written by conv1's call (models.py:196-197)
    uint16_t* xs2 = c.image(dout, N2);
    ConvOpt q1;
    q1.bias = m.bias_stack(sfx(".conv1"));
    q1.group_cols = gc2;
    q1.post_n = &o.n2; q1.post_lay = lay2; q1.post_yh = xs2;
    conv_h_new(c, w1, xs, din, lay2, k3, q1);
    if (has_sc && !fold) {                                              // learned shortcut (models.py:185-186), no bias
        ConvOpt q;
        q.group_cols = gc2;
        float* dst = out;
        const int ldd = ldo;
        const GemmW* wsc = m.conv_stack(sfx(".conv1x1"));
        if (X.h && !o.upsample) conv_h(c, wsc, X.h, din, lay2, k1, dst, ldd, q);
        else conv_x(c, wsc, sc, ldsc, din, lay2, k1, dst, ldd, q);
        sc = dst;
        ldsc = ldd;
    }
    ConvOpt q2;
    q2.bias = m.bias_stack(sfx(".conv2"));
    if (fold) {
        q2.x2h = X.h;
        q2.K2 = din;
    } else {
        q2.res = sc;
        q2.ldr = ldsc;
    }
    q2.div_sqrt2 = true;                                                // (res + sc) / sqrt(2), models.py:201
    q2.group_cols = gc2;
    if (o.want_yh) {
        q2.want_yh = true;
        q2.yh = o.yh ? o.yh : c.image(dout, N2);
    }
    if (o.next_n1) {
        q2.post_n = o.next_n1; q2.post_lay = lay2;
        q2.post_yh = c.image(dout, N2);
    }
    conv_h(c, w2, xs2, dout, lay2, k3, out, ldo, q2);
    Y.p = out; Y.C = dout; Y.ld = ldo; Y.lay = lay2; Y.h = q2.yh; Y.pre = q2.post_yh;
    return Y;
}

// RelTransformerEncoder.forward (RelTransformerEnc.py:371-380) for SEVERAL encoders of the same shape on the same tokens
// (text_encoder, arts_encoder, durationPredictor.text_encoder: models.py:358-359,549) as ONE multi-width launch sequence: the tokens
// are laid out once per encoder, [utterances | filler up to a multiple of 128 columns | utterances | ...]; encoder g owns columns
// [g * gc, g * gc + N) and utterances [g * bg, g * bg + B); conv GEMMs pick weight set g per column tile (ConvGemmArgs.n_groups),
// embedding / LayerNorm / attention take parameter set g of a stack.  Encoders are listed deepest first: when a shallower one has run
// its last layer its result is finished (its own last LayerNorm) and the launches continue on the remaining, leading groups.
// Half the launches (a third with three encoders) and fuller tiles: M1024 N3840 is 240 tiles of 128 x 128, one round of the chip.
struct EncSpec {
    std::string prefix;
    int layers;
};
struct EncOut {
    float* y[4] = {nullptr, nullptr, nullptr, nullptr};   // [C][ld[g]] per encoder
    int ld[4] = {0, 0, 0, 0};
};
template <class Done>
bool rel_encoder_multi(Ctx& c, const std::vector<EncSpec>& enc, const int32_t* tokens, const Lay* lay1, EncOut* out, Done&& done)
{
    const as_model& m = c.m;
    const int G = (int)enc.size();
    if (G < 1 || G > 4 || !lay1) { c.fail(AS_EINVAL); return false; }
    for (int g = 1; g < G; ++g)
        if (enc[g].layers > enc[g - 1].layers) { c.fail(AS_EINVAL); return false; }
    const HostT* emb_h = m.host(enc[0].prefix + ".emb.weight");
    if (!emb_h) { c.fail(AS_EINVAL); return false; }
    const int V = emb_h->dim(0), C = emb_h->dim(1), N1 = lay1->N;
    const int pad = G > 1 ? (128 - N1 % 128) % 128 : 0, gc = N1 + pad, bg = lay1->B + (pad ? 1 : 0);
    // layouts of the first k groups, k = G .. 1
    std::vector<const Lay*> lays(G + 1, nullptr);
    {
        std::vector<int> w;
        for (int k = 1; k <= G; ++k) {
            if (k > 1 && pad) w.push_back(pad);
            w.insert(w.end(), lay1->w.begin(), lay1->w.end());
            lays[k] = c.lay(w);
            if (!lays[k]) return false;
        }
    }
    const int ld = std::max(lays[G]->N, 1);
    auto names = [&](const std::string& sfx, int k) {
        std::vector<std::string> n;
        for (int g = 0; g < k; ++g) n.push_back(enc[g].prefix + sfx);
        return n;
    };
    // every weight is looked up OUTSIDE the RUN(...) arguments: the prepare pass (as_model_create) runs this code with
    // launches disabled and must still see each name
    auto stack2 = [&](const std::string& sfx, int k, const float** second) {   // parameter stack of the first k encoders; *second = set 1
        const HostT* h0 = m.host(enc[0].prefix + sfx);
        const float* s0 = m.vec_stack(names(sfx, k));
        *second = (k > 1 && s0 && h0) ? s0 + h0->v.size() : nullptr;
        return s0;
    };
    float* x = c.f32((size_t)C * ld);
    {
        const float* e2 = nullptr;
        const float* e1 = stack2(".emb.weight", G, &e2);
        if (lays[G]->N > 0) RUN(c, as_embed_groups_f32(tokens, N1, e1, e2, gc, C, lays[G]->N, V, sqrtf((float)C), x, ld, c.s));
    }
    // conv(LayerNorm(xin)): the normalised activations exist only as the conv's operand image
    auto ln_image = [&](const float* xin, const std::string& ln, bool relu, int k) {
        const int N = lays[k]->N;
        uint16_t* xs = c.image(C, N);
        const float *g2 = nullptr, *b2 = nullptr;
        const float *g1 = stack2(ln + ".gamma", k, &g2), *b1 = stack2(ln + ".beta", k, &b2);
        if (N > 0) RUN(c, as_channel_layernorm_split_f32(xin, ld, C, N, g1, b1, g2, b2, gc, 1e-4f, relu, xs, c.s));
        return xs;
    };
    // ... written by the call of the conv that PRODUCES xin (ConvOpt.ln; round 5): the conv -> residual add -> LayerNorm -> conv of
    // RelTransformerEnc.py:72-87, 318-325 as conv(+ LayerNorm) -> conv.  At batch-1 sizes the producer's reduction kernel writes the image;
    // elsewhere the library launches the LayerNorm behind the conv (what ln_image does, issued one call earlier).  Only where producer and
    // consumer cover the same groups (the same columns).
    auto ln_post = [&](ConvOpt& q, const std::string& ln, bool relu, int k) {
        const int N = lays[k]->N;
        uint16_t* xs = c.image(C, N);
        const float *g2 = nullptr, *b2 = nullptr;
        const float *g1 = stack2(ln + ".gamma", k, &g2), *b1 = stack2(ln + ".beta", k, &b2);
        q.ln.gamma = g1; q.ln.beta = b1; q.ln.gamma2 = g2; q.ln.beta2 = b2; q.ln.n_split = gc; q.ln.eps = 1e-4f; q.ln.relu = relu ? 1 : 0;
        q.ln.yh = N > 0 ? xs : nullptr;
        return xs;
    };
    auto cw = [&](const std::string& name, int k) { return m.conv_stack(names(name, k)); };
    auto cb = [&](const std::string& name, int k) { return m.bias_stack(names(name, k)); };
    auto opt = [&](const std::string& bias_of, int k) {
        ConvOpt o;
        o.bias = cb(bias_of, k);
        o.group_cols = k > 1 ? gc : 0;
        return o;
    };
    const Taps k5 = taps_1d(5), k1 = taps_1d(1), k9 = taps_1d(9);
    float* h = c.f32((size_t)C * ld);
    const std::string e = ".encoder";
    uint16_t* xs_next = nullptr;                                               // the image of the NEXT LayerNorm, written by the conv call in front of it
    {                                                                          // ConvReluNorm, RelTransformerEnc.py:318-325
        ConvOpt q0 = opt(".pre.conv_layers.0", G);
        xs_next = ln_post(q0, ".pre.norm_layers.0", true, G);
        conv_x(c, cw(".pre.conv_layers.0", G), x, ld, C, lays[G], k5, h, ld, q0);
        for (int i = 0; i < 3; ++i) {
            uint16_t* xs = xs_next;
            const std::string nxt = i < 2 ? ".pre.conv_layers." + std::to_string(i + 1) : std::string(".pre.proj");
            ConvOpt q = opt(nxt, G);
            if (i == 2) { q.res = x; q.ldr = ld; }
            // (the proj conv's result is the encoder's input: layer 0's first LayerNorm reads it)
            xs_next = i < 2 ? ln_post(q, ".pre.norm_layers." + std::to_string(i + 1), true, G)
                            : (enc[0].layers > 0 ? ln_post(q, e + ".norm_layers_1.0", false, G) : nullptr);
            float* hn = c.f32((size_t)C * ld);
            conv_h(c, cw(nxt, G), xs, C, lays[G], i < 2 ? k5 : k1, hn, ld, q);
            h = hn;
        }
    }
    x = h;
    // one encoder's result: its own last LayerNorm on its columns
    auto finish = [&](int g0, int g1) {                                        // encoders g0 .. g1 - 1 (they share the remaining depth)
        const int Nk = (g1 - g0 - 1) * gc + N1;                                // columns from group g0's first to group g1 - 1's last
        float* y = c.f32((size_t)C * std::max(Nk, 1));
        std::vector<std::string> ng, nb;
        for (int g = g0; g < g1; ++g) { ng.push_back(enc[g].prefix + e + ".last_ln.gamma"); nb.push_back(enc[g].prefix + e + ".last_ln.beta"); }
        const float *lg = m.vec_stack(ng), *lb = m.vec_stack(nb);
        const float *lg2 = g1 - g0 > 1 && lg ? lg + C : nullptr, *lb2 = g1 - g0 > 1 && lb ? lb + C : nullptr;
        if (Nk > 0) RUN(c, as_channel_layernorm_groups_f32(x + (size_t)g0 * gc, ld, C, Nk, lg, lb, lg2, lb2, gc, 1e-4f, 0, y, Nk, c.s));
        for (int g = g0; g < g1; ++g) {
            out->y[g] = y + (size_t)(g - g0) * gc;
            out->ld[g] = std::max(Nk, 1);
        }
        for (int g = g0; g < g1; ++g) done(g);
    };
    int k = G;
    int xs_k = G;                                                              // the groups xs_next covers
    for (int i = 0; i < enc[0].layers; ++i) {                                  // Encoder.forward, RelTransformerEnc.py:66-90
        int kn = 0;
        while (kn < G && enc[kn].layers > i) ++kn;                             // encoders that have a layer i
        if (kn < k) { finish(kn, k); k = kn; }
        const Lay* lay = lays[k];
        const int N = lay->N;
        const std::string a = e + ".attn_layers." + std::to_string(i), f = e + ".ffn_layers." + std::to_string(i);
        const float* bqkv = nullptr;
        const GemmW* wqkv = m.qkv(names(a, k), &bqkv);
        ConvOpt o;
        o.bias = bqkv;
        o.group_cols = k > 1 ? gc : 0;
        // 128-channel heads (the shipped model): the q/k/v GEMM also writes its result as an operand image, the attention kernel takes
        // Q / K fragments straight from it and writes the o-projection's operand image -- no fp32 attention output, no split pass
        const bool img = C / N_HEADS == 128;
        if (img) {
            o.want_yh = true;
            o.yh = c.image(3 * C, N);
        }
        // norm_layers_1[i](x): written by the call that produced x (the proj conv / the previous layer's second FFN conv) when that call
        // covered the same groups; otherwise (a shallower encoder has just left) by a launch of its own
        uint16_t* xs1 = (xs_next && xs_k == k) ? xs_next : ln_image(x, e + ".norm_layers_1." + std::to_string(i), false, k);
        xs_next = nullptr;
        float* qkv = conv_h_new(c, wqkv, xs1, C, lay, k1, o);
        float* att = img ? nullptr : c.f32((size_t)C * std::max(N, 1));
        uint16_t* att_h = img ? c.image(C, N) : nullptr;
        const float *ek2 = nullptr, *ev2 = nullptr;
        const float *ek = stack2(a + ".emb_rel_k", k, &ek2), *ev = stack2(a + ".emb_rel_v", k, &ev2);
        if (N > 0 && img)
            RUN(c, as_relpos_attention_image_f32(qkv, N, o.yh, N, C, N_HEADS, WINDOW, ek, ev, ek2, ev2, bg, lay->d_off, lay->B, lay->max_w, nullptr,
                                                 0, att_h, c.s));
        else if (N > 0)
            RUN(c, as_relpos_attention_groups_f32(qkv, N, C, N_HEADS, WINDOW, ek, ev, ek2, ev2, bg, lay->d_off, lay->B, lay->max_w, att, N, c.s));
        ConvOpt oo = opt(a + ".conv_o", k);
        oo.res = x;
        oo.ldr = ld;
        uint16_t* xs2 = ln_post(oo, e + ".norm_layers_2." + std::to_string(i), false, k);   // norm_layers_2[i] of x + attention, by the o-projection's call
        float* xo = c.f32((size_t)C * ld);
        if (img) conv_h(c, cw(a + ".conv_o", k), att_h, C, lay, k1, xo, ld, oo);
        else conv_x(c, cw(a + ".conv_o", k), att, N, C, lay, k1, xo, ld, oo);
        // a shallower encoder's columns past the active groups keep their values in the OLD buffer: `finish` ran before this layer
        x = xo;
        // FFN (RelTransformerEnc.py:261-269): conv k9 -> ReLU exists only as the 1x1 conv's operand image
        const GemmW* w1 = cw(f + ".conv_1", k);
        if (!w1) { c.fail(AS_EINVAL); return false; }
        uint16_t* yh = c.image(w1->M, N);
        ConvOpt o1 = opt(f + ".conv_1", k);
        o1.act = ACT_RELU;
        o1.want_yh = true;
        o1.yh = yh;
        conv_h(c, w1, xs2, C, lay, k9, nullptr, N, o1);
        ConvOpt o2 = opt(f + ".conv_2", k);
        o2.res = x;
        o2.ldr = ld;
        // the next layer's norm_layers_1 by this call, if the next layer covers the same groups
        {
            int kn2 = 0;
            while (kn2 < G && enc[kn2].layers > i + 1) ++kn2;
            if (i + 1 < enc[0].layers && kn2 == k) { xs_next = ln_post(o2, e + ".norm_layers_1." + std::to_string(i + 1), false, k); xs_k = k; }
        }
        float* xf = c.f32((size_t)C * ld);
        conv_h(c, cw(f + ".conv_2", k), yh, w1->M, lay, k1, xf, ld, o2);
        x = xf;
    }
    finish(0, k);
    return c.rc == 0;
}

// the path's three encoders (models.py:358-359,549), deepest first
std::vector<EncSpec> path_encoders() { return {{"arts_encoder", 4}, {"text_encoder", 4}, {"durationPredictor.text_encoder", 2}}; }
enum { ENC_ARTS = 0, ENC_TEXT = 1, ENC_DUR = 2 };

// one down-sampling step of a tower (as_down_multi_f32): launched, or recorded so that the towers' steps share a launch
void down(Ctx& c, const AsDownArgs& a, int line)
{
    if (!c.go()) return;
    if (c.deferring()) {
        Op& o = c.push(3);
        o.d = a;
        o.what = "as_down_multi_f32";
        o.line = line;
        return;
    }
    const int r = as_down_multi_f32(&a, 1, c.s);
    if (r != AS_OK) c.fail(r, "as_down_multi_f32", line);
}
AsDownArgs down_args(int kind, const float* x, int ldx, const Lay* lay, const Lay* lay2, int C)
{
    AsDownArgs a;
    memset(&a, 0, sizeof(a));
    a.kind = kind; a.x = x; a.ldx = ldx; a.in_off = lay->d_off; a.in_w = lay->d_w; a.Hin = lay->H;
    a.out_off = lay2->d_off; a.out_w = lay2->d_w; a.Hout = lay2->H; a.B = lay->B; a.C = C; a.max_out = lay2->max_cols(); a.n_out = lay2->N;
    return a;
}

// ResBlk (models.py:79-100) / ResBlk1d(downsample=True) (models.py:127-156).  X.h, if set, is the operand image of
// LeakyReLU(X) (what conv1 reads); the result carries the same for the next block when want_image.
Act resblk_down(Ctx& c, const std::string& p, const Act& X, bool half, bool one_d, bool want_image)
{
    const as_model& m = c.m;
    const Lay* lay = X.lay;
    Act Y;
    const Lay* lay2 = c.halved(lay, half);
    if (!lay2) return Y;
    const int cin = X.C, N2 = std::max(lay2->N, 1);
    const Taps taps = one_d ? taps_1d(3) : taps_2d(3, 3);
    ConvOpt o;
    o.bias = m.bias(p + ".conv1");
    float* r;
    if (X.h) r = conv_h_new(c, m.conv(p + ".conv1"), X.h, cin, lay, taps, o);
    else {
        o.in_act = ACT_LRELU;
        r = conv_x_new(c, m.conv(p + ".conv1"), X.p, X.ld, cin, lay, taps, o);
    }
    // LearnedDownSample / pool (models.py:27-31,116) -> LeakyReLU: read by conv2 only, so it exists only as conv2's operand image
    const std::string dname = p + (one_d ? ".pool" : ".downsample_res.conv");
    uint16_t* r2h = c.image(cin, lay2->N);
    const float *dww = m.vec(dname + ".weight"), *dwb = m.vec(dname + ".bias");
    c.hint(0, 4.0 * cin * ((double)lay->N + lay2->N));
    {
        AsDownArgs a = down_args(0, r, lay->N, lay, lay2, cin);
        a.w = dww; a.bias = dwb; a.kh = half ? 3 : 1; a.lrelu = 1; a.yh = r2h;
        down(c, a, __LINE__);
    }
    const bool has_sc = m.has(p + ".conv1x1.weight");
    const GemmW* w2 = has_sc ? m.conv_fold({p + ".conv2"}, {p + ".conv1x1"}) : m.conv(p + ".conv2");
    if (!w2) { c.fail(AS_EINVAL); return Y; }
    ConvOpt o2;
    o2.bias = m.bias(p + ".conv2");
    float* out;
    uint16_t* outh = want_image ? c.image(w2->M, lay2->N) : nullptr;
    if (has_sc) {
        // shortcut = avgpool(conv1x1(x)) (models.py:79-84).  Both are linear and the 1x1 conv has no bias, so it is evaluated as
        // conv1x1(avgpool(x)): a quarter of the columns -- as `cin` more channels of conv2's reduction (conv_fold), the merge
        // (x + r)/sqrt(2) in that launch's epilogue: the residual branch never exists on its own.
        uint16_t* xsh = c.image(cin, lay2->N);
        c.hint(0, 4.0 * cin * ((double)lay->N + lay2->N));
        if (X.stem_x) {
            AsDownArgs a = down_args(2, X.stem_x, 0, lay, lay2, cin);
            a.pool_h = half ? 2 : 1; a.w = X.stem_w->w32; a.Kp = X.stem_w->Kp; a.bias = X.stem_b; a.kh = X.stem_kh; a.yh = xsh;
            down(c, a, __LINE__);
        } else {
            AsDownArgs a = down_args(1, X.p, X.ld, lay, lay2, cin);
            a.pool_h = half ? 2 : 1; a.yh = xsh;
            down(c, a, __LINE__);
        }
        o2.x2h = xsh;
        o2.K2 = cin;
        o2.div_sqrt2 = true;
        o2.want_yh = want_image;
        o2.yh = outh;
        o2.yh_lrelu = true;
        out = conv_h_new(c, w2, r2h, cin, lay2, taps, o2);
    } else {
        float* r3 = conv_h_new(c, w2, r2h, cin, lay2, taps, o2);
        out = c.f32((size_t)w2->M * N2);
        c.hint(0, 4.0 * cin * ((double)lay->N + (want_image ? 3.0 : 2.0) * lay2->N));
        {
            AsDownArgs a = down_args(1, X.p, X.ld, lay, lay2, cin);
            a.y = out; a.ldy = lay2->N; a.pool_h = half ? 2 : 1; a.res = r3; a.ldr = lay2->N;
            if (want_image) { a.yh = outh; a.lrelu = 1; }
            down(c, a, __LINE__);
        }
    }
    Y.p = out; Y.C = w2->M; Y.ld = lay2->N; Y.lay = lay2; Y.h = outh;
    return Y;
}

// first conv of a tower (Cin = 1: the direct kernel): fp32 output for the first block's shortcut + the LeakyReLU image its conv1 reads
// (image_only: a one-channel stem whose first block has a learned shortcut -- that shortcut is computed from X itself, resblk_down)
Act tower_stem(Ctx& c, const std::string& name, const float* X, int ldx, const Lay* lay, const Taps& taps, bool image_only)
{
    Act x;
    const GemmW* w0 = c.m.conv(name);
    if (!w0 || !lay) { c.fail(AS_EINVAL); return x; }
    ConvOpt o;
    o.bias = c.m.bias(name);
    o.want_yh = true;
    o.yh = c.image(w0->M, lay->N);
    o.yh_lrelu = true;
    if (image_only && w0->K == 1 && w0->w32 && (taps.n == 9 || taps.n == 3)) {
        conv_x(c, w0, X, ldx, w0->K, lay, taps, nullptr, lay->N, o);
        x.stem_x = X; x.stem_w = w0; x.stem_b = o.bias; x.stem_kh = taps.n == 9 ? 3 : 1;
    } else {
        x.p = conv_x_new(c, w0, X, ldx, w0->K, lay, taps, o);
    }
    x.C = w0->M; x.ld = lay->N; x.lay = lay; x.h = o.yh;
    return x;
}

// Mel_block / EMA_block / dur_block + their Linear (models.py:385-401,412-413,530-538) -> y [B][ldy] (M entries per row)
void tower2d(Ctx& c, const std::string& p, const float* X, const Lay* lay, const std::vector<bool>& halves, int last_idx, int last_stride,
             const std::string& linear, float* y, int ldy)
{
    const as_model& m = c.m;
    if (!lay) { c.fail(AS_EINVAL); return; }
    Act x = tower_stem(c, p + ".0", X, lay->N, lay, taps_2d(3, 3), m.has(p + ".1.conv1x1.weight"));
    if (!x.lay) return;
    for (size_t i = 0; i < halves.size(); ++i) {
        x = resblk_down(c, p + "." + std::to_string(i + 1), x, halves[i], false, true);   // (the last block's LeakyReLU image feeds the valid conv)
        if (!x.lay) return;
    }
    const int K = 5, C = x.C;
    const Lay* lout = c.valid_conv(x.lay, K, last_stride);
    if (!lout) return;
    if (lout->H < 1 || *std::min_element(lout->w.begin(), lout->w.end()) < 1) {
        // reference utterance too short for the 5x5 valid conv (T_ref >= 66)
        c.fail(AS_EINVAL);
        return;
    }
    const std::string ln = p + "." + std::to_string(last_idx);
    const GemmW* wl = m.conv(ln);
    if (!wl || wl->T != K * K || wl->K != C) { c.fail(AS_EINVAL); return; }
    // LeakyReLU -> K x K valid conv (models.py:390-391,398-399,534-535) straight from the last block's LeakyReLU image: the outputs are their
    // own layout, output (ho, wo) reads the input at (ho s + a, wo s + d), a, d = 0 .. K-1 (ConvGemmArgs.src_col).  No im2col matrix.
    const Lay* lin = x.lay;
    std::string key = "valid:" + std::to_string(K) + ":" + std::to_string(last_stride) + ":" + std::to_string(lin->H);
    for (int v : lin->w) key += ":" + std::to_string(v);
    const int32_t* tab = c.itable(lout, key, [&]() {                      // [N_out] source columns, then [N_out] descriptors (two words each)
        std::vector<int32_t> t((size_t)lout->N * 3 + 1, 0);
        const size_t mo = ((size_t)lout->N + 1) & ~(size_t)1;             // 8-byte aligned start of the descriptors
        for (int b = 0, j = 0; b < lout->B; ++b)
            for (int ho = 0; ho < lout->H; ++ho)
                for (int wo = 0; wo < lout->w[b]; ++wo, ++j) {
                    const int h = ho * last_stride, w = wo * last_stride;
                    t[j] = lin->off[b] + h * lin->w[b] + w;
                    const uint64_t md = AS_META_PACK(h, w, lin->H, lin->w[b]);
                    memcpy(&t[mo + 2 * (size_t)j], &md, 8);
                }
        return t;
    });
    ConvOpt o2;
    o2.bias = m.bias(ln);
    o2.act = ACT_LRELU;
    o2.src_col = tab;
    o2.src_meta = tab ? reinterpret_cast<const uint64_t*>(tab + (((size_t)lout->N + 1) & ~(size_t)1)) : nullptr;
    o2.N_in = lin->N;
    Taps tv;
    tv.n = K * K;
    for (int a = 0; a < K; ++a)
        for (int d = 0; d < K; ++d) { tv.dh[a * K + d] = a; tv.dw[a * K + d] = d; }
    float* z = conv_h_new(c, wl, x.h, C, lout, tv, o2);
    float* pooled = c.f32((size_t)lay->B * wl->M);
    RUN(c, as_mean_pool_f32(z, lout->N, lout->d_off, lay->B, wl->M, 0, pooled, wl->M, c.s));
    const HostT* lw = m.host(linear + ".weight");
    if (!lw) return;
    const float *lww = m.vec(linear + ".weight"), *lwb = m.vec(linear + ".bias");
    RUN(c, as_linear_rows_f32(pooled, wl->M, lww, lwb, lay->B, lw->dim(0), lw->dim(1), y, ldy, c.s));
}

// F0_block / energy_block + Linear (models.py:402-411,414-415)
void tower1d(Ctx& c, const std::string& p, const float* X, int ldx, const Lay* lay, const std::string& linear, float* y, int ldy)
{
    const as_model& m = c.m;
    if (!lay) { c.fail(AS_EINVAL); return; }
    Act x = tower_stem(c, p + ".0", X, ldx, lay, taps_1d(3), m.has(p + ".1.conv1x1.weight"));
    if (!x.lay) return;
    for (int i = 1; i <= 4; ++i) {
        x = resblk_down(c, p + "." + std::to_string(i), x, false, true, i < 4);
        if (!x.lay) return;
    }
    float* pooled = c.f32((size_t)lay->B * x.C);
    RUN(c, as_mean_pool_f32(x.p, x.ld, x.lay->d_off, lay->B, x.C, 1, pooled, x.C, c.s));
    const HostT* lw = m.host(linear + ".weight");
    if (!lw) return;
    const float *lww = m.vec(linear + ".weight"), *lwb = m.vec(linear + ".bias");
    RUN(c, as_linear_rows_f32(pooled, x.C, lww, lwb, lay->B, lw->dim(0), lw->dim(1), y, ldy, c.s));
}

// ------------------------------------------------------------------------------------------------------------------
// modules
// ------------------------------------------------------------------------------------------------------------------
struct StyleIn {                  // the T-1 crop of the reference features and the towers' images (models.py:459-471)
    float* crop = nullptr;        // [2][N1]: rows 0 n, 1 f0
    const Lay *l1 = nullptr, *lm = nullptr, *le = nullptr;
    float *mel_img = nullptr, *ema_img = nullptr;
};

StyleIn style_inputs(Ctx& c, const float* feat12, int ldf, const float* mel, int ldm, const Lay* ref)
{
    StyleIn s;
    const int n_mels = c.m.cfg.n_mels;
    std::vector<int> w(ref->w);
    for (int& v : w) v = v > 0 ? v - 1 : 0;                               // start = randint(0, 1) = 0, length T - 1
    s.l1 = c.lay(w);
    s.lm = c.lay(w, n_mels);
    s.le = c.lay(w, 10);
    if (!s.l1 || !s.lm || !s.le) return s;
    const int N1 = std::max(s.l1->N, 1);
    // the T - 1 window of the energy / F0 rows (the 1-D towers' input); the 2-D towers' images are cut straight out of the mel and the TV
    // rows (as_rows_to_images_f32 takes the utterances' source offsets and the images' widths: no cropped copy of 90 rows in between)
    s.crop = c.f32((size_t)2 * N1);
    RUN(c, as_crop_f32(feat12, ldf, ref->d_off, 0, s.crop, s.l1->N, s.l1->d_off, ref->B, 2, s.l1->max_cols(), c.s));
    s.mel_img = c.f32((size_t)std::max(s.lm->N, 1));
    s.ema_img = c.f32((size_t)std::max(s.le->N, 1));
    RUN(c, as_rows_to_images_f32(mel, ldm, ref->d_off, 0, n_mels, s.mel_img, s.lm->d_off, ref->B, s.l1->max_cols(), c.s));
    RUN(c, as_rows_to_images_f32(feat12 + (size_t)2 * ldf, ldf, ref->d_off, 0, 10, s.ema_img, s.le->d_off, ref->B, s.l1->max_cols(), c.s));
    return s;
}

// one of the four towers of StyleEncoder.style_extractor (models.py:417-424) -> its slice of Style [B][2 * style_dim]
void style_tower(Ctx& c, int which, const StyleIn& s, float* style)
{
    const std::string p = "style_encoder";
    const int sd = c.m.cfg.style_dim, lds = 2 * sd;
    if (which == 0) tower2d(c, p + ".Mel_block", s.mel_img, s.lm, {true, true, true, true}, 6, 1, p + ".Mellinear", style, lds);
    else if (which == 1) tower2d(c, p + ".EMA_block", s.ema_img, s.le, {false, false, true}, 5, 2, p + ".EMAlinear", style ? style + sd : nullptr, lds);
    // energy (crop row 0) and F0 (row 1) towers: twins merged at load (merge_twin_towers) into one tower on the two-row input; its Linear
    // writes the F0 slice and the energy slice of Style, which are adjacent
    else if (which == 2) tower1d(c, p + ".ENF0_block", s.crop, s.l1->N, s.l1, p + ".ENF0linear", style ? style + sd + sd / 2 : nullptr, lds);
}

// DurationPredictor (models.py:540-566) in three pieces
void duration_style(Ctx& c, const float* ema_ext, int lde, const Lay* ref, float* ds /* [B][style_dim / 4] */)
{
    // dur_block + dur_linear on the FULL-length TV track (models.py:543-546)
    const std::string p = "durationPredictor";
    const Lay* limg = c.lay(ref->w, 10);
    if (!limg) return;
    float* img = c.f32((size_t)std::max(limg->N, 1));
    RUN(c, as_rows_to_images_f32(ema_ext, lde, ref->d_off, 0, 10, img, limg->d_off, ref->B, ref->max_cols(), c.s));
    tower2d(c, p + ".dur_block", img, limg, {false, false, true}, 5, 2, p + ".dur_linear", ds, c.m.cfg.style_dim / 4);
}

// 3 x AdainResBlk1d -> BiLSTM -> duration_proj -> [1][N] (into `dst`, the caller's buffer, when given)
float* duration_tail(Ctx& c, float* d, const float* ds, const Lay* tok, float* dst = nullptr)
{
    const as_model& m = c.m;
    const std::string p = "durationPredictor";
    const int C = m.cfg.hidden_dim, S = m.cfg.style_dim / 4;
    std::vector<as_model::NormSpec> norms;
    for (int i = 0; i < 3; ++i)
        for (const char* n : {".norm1", ".norm2"}) norms.push_back({p + ".duration." + std::to_string(i) + n, 0, S});
    const FcOut fc = adain_fc_all(c, "duration", norms, ds, S, S, tok->B);
    Act x;
    x.p = d; x.C = C; x.ld = tok->N; x.lay = tok;
    for (int i = 0; i < 3; ++i) {
        BlkOpt o;
        o.names = {p + ".duration." + std::to_string(i)};
        o.n1 = fc.norm(o.names[0] + ".norm1");
        o.n2 = fc.norm(o.names[0] + ".norm2");
        o.want_yh = i == 2;                                             // the last block feeds the LSTM's input projection
        const Norm nn = i < 2 ? fc.norm(p + ".duration." + std::to_string(i + 1) + ".norm1") : Norm();
        if (i < 2) o.next_n1 = &nn;
        x = adain_resblk1d(c, x, o);
        if (!x.lay) return nullptr;
    }
    const LstmW* L = m.lstm(p + ".LSTM");
    if (!L || !L->wih) { c.fail(AS_EINVAL); return nullptr; }
    const int H = L->H, Nn = std::max(tok->N, 1);
    float* gx = c.f32((size_t)Nn * 8 * H);
    ConvOpt q;
    q.bias = L->bias;
    q.transpose_out = true;
    conv_h(c, L->wih, x.h, x.C, tok, taps_1d(1), gx, 8 * H, q);
    float* h = c.f32((size_t)2 * H * Nn);
    BiLstmJob job;
    job.gx_tm = gx; job.whh_t = L->whh_t; job.out = h; job.ldg = 8 * H; job.ldo = tok->N;
    if (tok->N > 0 && c.go()) {
        size_t xb = 0;
        void* xchg = c.lstm_xchg(1, tok->B, &xb);
        RUN(c, as_bilstm_cluster_f32(&job, 1, tok->d_off, tok->B, H, tok->max_w, xchg, xb, c.s));
        // hundreds of tokens: milliseconds on a few dozen workgroups (C5: 1.8 ms on 16) while the text / articulatory encoders' last two
        // layers -- which do not depend on it (models.py:356-360) -- have the chip's worth of GEMMs to run: a recording plan puts the
        // launch on its side stream (one fork / join pair of event edges: worth it only for a long launch)
        static const int side_min = getenv("AS_SIDE_LSTM_MIN") ? atoi(getenv("AS_SIDE_LSTM_MIN")) : 200;
        if (c.deferring() && c.go() && tok->max_w >= side_min && !c.sched->q[c.cur_q].empty()) c.sched->q[c.cur_q].back().side = true;
    }
    float* y = c.f32(Nn);
    if (dst) y = dst;
    const float *pw = m.vec(p + ".duration_proj.linear_layer.weight"), *pb = m.vec(p + ".duration_proj.linear_layer.bias");
    RUN(c, as_project_cols_f32(h, tok->N, 2 * H, tok->N, pw, pb, 1, y, tok->N, c.s));
    return y;
}

// The fc layers of every AdaIN1d that reads the Style vector (artsPredictor + decoder): layer name, first entry, entries read.
// The three branches' layers of one position are adjacent, so a grouped AdaIN launch addresses group g at + g * 2C rows.
std::vector<as_model::NormSpec> style_norms(const as_model& m)
{
    const int sd = m.cfg.style_dim;
    std::vector<as_model::NormSpec> v;
    const char* br[3] = {"F0", "N", "EMA"};
    const int off[3] = {sd + sd / 2, sd + sd / 2 + sd / 4, sd}, len[3] = {sd / 4, sd / 4, sd / 2};   // models.py:597-599
    for (const char* n : {".norm1", ".norm2"}) v.push_back({std::string("artsPredictor.shared") + n, 0, 2 * sd});
    for (int blk = 0; blk < 3; ++blk)
        for (const char* n : {".norm1", ".norm2"})
            for (int g = 0; g < 3; ++g)
                v.push_back({std::string("artsPredictor.") + br[g] + "." + std::to_string(blk) + n, blk == 0 ? 0 : off[g], blk == 0 ? 2 * sd : len[g]});
    for (const char* n : {".norm1", ".norm2"}) v.push_back({std::string("decoder.encode") + n, 0, 2 * sd});
    for (int i = 0; i < 6; ++i)
        for (const char* n : {".norm1", ".norm2"})
            v.push_back({"decoder.decode." + std::to_string(i) + n, 0, i < 3 ? 2 * sd : sd});          // decode.3..5: Mel_style = Style[:, :style_dim]
    return v;
}

// ArtsPredictor.forward (models.py:596-621): a [C][N1] on `lay` -> fne [12][ldp]: row 0 F0, 1 N, 2..11 EMA, on the x2 layout.
// The F0 / N / EMA branches (three AdainResBlk1d each, identical shapes) run as ONE grouped launch sequence on a layout that holds
// the batch three times; the three BiLSTMs share one recurrence launch.
void arts_predictor(Ctx& c, const float* a_en, int lda, const Lay* lay, const FcOut& fc, float* fne, int ldp)
{
    const as_model& m = c.m;
    const std::string p = "artsPredictor";
    const int C = m.cfg.hidden_dim, B = lay->B, N1 = lay->N;
    const char* br[3] = {"F0", "N", "EMA"};
    Act a;
    a.p = const_cast<float*>(a_en); a.C = C; a.ld = lda; a.lay = lay;
    {
        BlkOpt o;
        o.names = {p + ".shared"};
        o.n1 = fc.norm(p + ".shared.norm1");
        o.n2 = fc.norm(p + ".shared.norm2");
        a = adain_resblk1d(c, a, o);
        if (!a.lay) return;
    }
    // grouped layouts: the batch three times
    std::vector<int> w3;
    for (int g = 0; g < 3; ++g) w3.insert(w3.end(), lay->w.begin(), lay->w.end());
    const Lay* layG1 = lay->dyn ? c.dyn_lay(2, lay->dyn_B, lay->cap1) : c.lay(w3);
    const Lay* layG2 = layG1 ? c.scaled(layG1, 2) : nullptr;
    const Lay* lay2 = c.scaled(lay, 2);
    if (!layG1 || !layG2 || !lay2) return;
    const int N2 = lay2->N, NG2 = layG2->N, Nn = std::max(NG2, 1);
    // per-utterance tables of the grouped launches: where group g's utterance b reads the shared input, and its gamma / beta
    const int32_t* src_off = c.itable(layG1, "src3", [&]() {
        std::vector<int32_t> t(3 * B);
        for (int g = 0; g < 3; ++g)
            for (int b = 0; b < B; ++b) t[g * B + b] = lay->off[b];
        return t;
    });
    auto gb_table = [&](int rows_per_group) {                           // group g at + g * rows_per_group rows of gbT ([rows][B])
        return c.itable(layG1, "gb3:" + std::to_string(rows_per_group), [&]() {
            std::vector<int32_t> t(3 * B);
            for (int g = 0; g < 3; ++g)
                for (int b = 0; b < B; ++b) t[g * B + b] = g * rows_per_group * B + b;
            return t;
        });
    };
    auto gnorm = [&](int blk, const char* n, int Cn) {
        Norm x = fc.norm(p + "." + br[0] + "." + std::to_string(blk) + n);
        x.gb_off = gb_table(2 * Cn);
        return x;
    };
    auto names = [&](int blk) { std::vector<std::string> v; for (int g = 0; g < 3; ++g) v.push_back(p + "." + br[g] + "." + std::to_string(blk)); return v; };
    Act x;
    {   // block 0 (up-sampling; models.py:579,582,585): AdaIN on the shared input, three parameter sets
        const std::vector<std::string> nm = names(0);
        auto sfx = [&](const char* s) { std::vector<std::string> v(nm); for (auto& n : v) n += s; return v; };
        const GemmW *w1 = m.conv_stack(sfx(".conv1")), *w2 = m.conv_stack(sfx(".conv2"));
        if (!w1 || !w2) { c.fail(AS_EINVAL); return; }
        uint16_t* xs = c.image(C, NG2);
        float* up = c.f32((size_t)C * Nn);
        // the depthwise ConvTranspose1d weights differ per branch: one launch per group on its third of the grouped layout
        const Norm n1 = gnorm(0, ".norm1", C);
        for (int g = 0; g < 3; ++g) {
            const float *pw = m.vec(nm[g] + ".pool.weight"), *pb = m.vec(nm[g] + ".pool.bias");
            if (!c.go() || NG2 == 0) continue;
            AsAdainArgs q;
            memset(&q, 0, sizeof(q));
            q.x = a.p; q.ldx = a.ld; q.C = C;
            q.gb = n1.gb; q.gb_off = n1.gb_off + g * B; q.ldgb = 1; q.gb_sc = n1.gb_sc;
            q.col_off = layG1->d_off + g * B; q.src_off = src_off + g * B; q.U = B; q.N = NG2; q.lrelu = 1; q.yh = xs;
            q.col_w = layG1->dyn ? layG1->d_w + g * B : nullptr;
            q.pool_w = pw; q.pool_b = pb; q.x_up = up; q.ld_up = NG2;
            RUN(c, as_adain_image_f32(&q, c.s));
        }
        ConvOpt q1;
        q1.bias = m.bias_stack(sfx(".conv1"));
        q1.group_cols = 2 * N1;
        uint16_t* xs2 = c.image(w1->M, NG2);
        const Norm n2 = gnorm(0, ".norm2", w1->M);
        q1.post_n = &n2; q1.post_lay = layG2; q1.post_yh = xs2;
        conv_h_new(c, w1, xs, C, layG2, taps_1d(3), q1);
        ConvOpt q2;
        q2.bias = m.bias_stack(sfx(".conv2"));
        q2.res = up;
        q2.ldr = NG2;
        q2.div_sqrt2 = true;
        q2.group_cols = 2 * N1;
        q2.want_yh = true;                                              // block 1's learned shortcut reads the image
        q2.yh = c.image(w2->M, NG2);
        const Norm nn = gnorm(1, ".norm1", w2->M);                       // block 1's norm1, by the same call
        q2.post_n = &nn; q2.post_lay = layG2; q2.post_yh = c.image(w2->M, NG2);
        x.p = conv_h_new(c, w2, xs2, w1->M, layG2, taps_1d(3), q2);
        x.C = w2->M; x.ld = NG2; x.lay = layG2; x.h = q2.yh; x.pre = q2.post_yh;
    }
    for (int blk = 1; blk <= 2; ++blk) {
        BlkOpt o;
        o.names = names(blk);
        o.group_cols = 2 * N1;
        const GemmW* w1 = m.conv_stack({o.names[0] + ".conv1", o.names[1] + ".conv1", o.names[2] + ".conv1"});
        if (!w1) { c.fail(AS_EINVAL); return; }
        o.n1 = gnorm(blk, ".norm1", x.C);
        o.n2 = gnorm(blk, ".norm2", w1->M);
        o.want_yh = true;
        Norm nn;
        if (blk == 1) {                                                  // block 2's norm1 by block 1's last call (conv1: din -> dout, conv2: dout -> dout)
            nn = gnorm(2, ".norm1", w1->M);
            o.next_n1 = &nn;
        }
        x = adain_resblk1d(c, x, o);
        if (!x.lay) return;
    }
    // BiLSTMs: one grouped input projection, one recurrence launch for the three (each on its own third of the columns)
    std::vector<std::string> ln;
    for (int g = 0; g < 3; ++g) ln.push_back(p + "." + br[g] + "_LSTM");
    const float* lb = nullptr;
    const GemmW* wih = m.lstm_wih_stack(ln, &lb);
    const LstmW* L0 = m.lstm(ln[0]);
    if (!wih || !L0) { c.fail(AS_EINVAL); return; }
    const int H = L0->H;
    float* gx = c.f32((size_t)Nn * 8 * H);
    ConvOpt q;
    q.bias = lb;
    q.transpose_out = true;
    q.group_cols = 2 * N1;
    conv_h(c, wih, x.h, x.C, layG2, taps_1d(1), gx, 8 * H, q);
    float* hs = c.f32((size_t)2 * H * Nn);
    BiLstmJob jobs[3];
    for (int g = 0; g < 3; ++g) {
        const LstmW* L = m.lstm(ln[g]);
        if (!L) { c.fail(AS_EINVAL); return; }
        jobs[g].gx_tm = gx + (size_t)g * N2 * 8 * H; jobs[g].whh_t = L->whh_t; jobs[g].out = hs + (size_t)g * N2; jobs[g].ldg = 8 * H; jobs[g].ldo = NG2;
    }
    if (N2 > 0) RUN(c, as_bilstm_f32(jobs, 3, lay2->d_off, B, H, c.s));
    const int rows[3] = {0, 1, 2}, M[3] = {1, 1, 10};
    for (int g = 0; g < 3; ++g) {
        const float *pw = m.vec(p + "." + br[g] + "_proj.weight"), *pb = m.vec(p + "." + br[g] + "_proj.bias");
        RUN(c, as_project_cols_f32(hs + (size_t)g * N2, NG2, 2 * H, N2, pw, pb, M[g], fne + (size_t)rows[g] * ldp, ldp, c.s));
    }
}

void copy_rows(Ctx& c, float* dst, int ldd, const float* src, int lds, int rows, int cols)
{
    if (c.go() && rows > 0 && cols > 0 &&
        hipMemcpy2DAsync(dst, (size_t)ldd * 4, src, (size_t)lds * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToDevice, c.s) != hipSuccess)
        c.fail((int)hipErrorUnknown);
}

// Decoder.forward (models.py:497-517).  x0 [C + 128][N2]: rows 0..C-1 already hold the up-sampled text encoding (models.py:500);
// fne [12][ldp] = F0, N, EMA; mel [n_mels][ldo].
// One concat buffer: the decode blocks write their fp32 result over the rows they were computed from, so cat([x, asr_res, F0, N, EMA])
// (models.py:510) never copies anything; the operand image of the concatenation is the concatenation of the parts' images (k-blocks are
// the image's outer axis), each written by its producer -- kept twice, because a block's conv2 launch reads the block's input image
// (the folded shortcut) while it writes the result image.
struct DecPre {                   // the decoder's buffers and the part of it that needs nothing from the predictors
    uint16_t *x0h = nullptr, *cath = nullptr, *cath2 = nullptr;
    float* catb = nullptr;
    bool ok = false;
};
// x0 rows 0..C-1 = the up-sampled text encoding: its operand image and asr_res (models.py:507) -- independent of ArtsPredictor, so a
// recording plan runs them beside it (forward_b)
DecPre decoder_pre(Ctx& c, float* x0, const Lay* lay2)
{
    const as_model& m = c.m;
    const std::string p = "decoder";
    DecPre d;
    const int C = m.cfg.hidden_dim, N2 = lay2->N, Nn2 = std::max(N2, 1), bott = 2 * C, cat = bott + 64 + 128;
    const size_t blk_bytes = (size_t)4 * (N2 + 1) * 16;
    if (C % 64 || (C + 128) % 64) { c.fail(AS_EINVAL); return d; }      // parts must start on the image's 4-block granularity
    d.x0h = c.image(C + 128, N2);                                       // image of x0 = [text encoding | F0 N EMA convs]
    d.catb = c.f32((size_t)cat * Nn2);                                  // [x (2C) | asr_res (64) | F0 N EMA convs (128)]
    d.cath = c.image(cat, N2);
    d.cath2 = c.image(cat, N2);                                         // (see decoder(): the in-place blocks ping-pong between two images)
    RUN(c, as_split_f16x2_f32(x0, N2, C, N2, 0, 0.f, d.x0h, c.s));      // text encoding part of x0's image
    ConvOpt q;
    q.bias = m.bias(p + ".asr_res.0");
    q.want_yh = true;
    q.yh = d.cath ? reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d.cath) + (size_t)(bott / 16) * blk_bytes) : nullptr;
    conv_h(c, m.conv(p + ".asr_res.0"), d.x0h, C, lay2, taps_1d(1), d.catb ? d.catb + (size_t)bott * N2 : nullptr, N2, q);
    d.ok = c.rc == 0;
    return d;
}

void decoder(Ctx& c, const DecPre& dp, float* x0, const Lay* lay2, const float* fne, int ldp, const FcOut& fc, float* mel, int ldo)
{
    const as_model& m = c.m;
    const std::string p = "decoder";
    const int C = m.cfg.hidden_dim, N2 = lay2->N, bott = 2 * C, cat = bott + 64 + 128;
    const Taps k1 = taps_1d(1);
    const size_t blk_bytes = (size_t)4 * (N2 + 1) * 16;                 // one k-block of an image over N2 columns
    auto at_block = [&](uint16_t* img, int kb) { return img ? reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(img) + (size_t)kb * blk_bytes) : nullptr; };
    if (!dp.ok) { c.fail(AS_EINVAL); return; }
    uint16_t *x0h = dp.x0h, *cath = dp.cath, *cath2 = dp.cath2;
    float* catb = dp.catb;
    // (cath2: a second image of the concat buffer.  decode.0 / decode.1 write their result image over the channels they were computed from,
    // and their learned shortcut -- K2 = 1216 more channels of conv2's reduction -- reads the block's input image in that same launch: the
    // two blocks ping-pong between two images; the 192 persistent channels (asr_res, F0 / N / EMA convs) are copied across once)
    const float *fb = nullptr, *w32 = nullptr;
    const GemmW* wf = m.fne(&fb, &w32);
    if (!wf) { c.fail(AS_EINVAL); return; }
    {   // F0_conv / N_conv / EMA_conv (models.py:503-505) as one block-diagonal 1x1 conv of 12 input rows, written where both consumers
        // read it: rows C.. of x0 with their blocks of x0's image, rows bott + 64.. of the concat buffer with theirs (one launch)
        if (wf->K > 16) { c.fail(AS_EINVAL); return; }
        RUN(c, as_pointwise_small_f32(fne, ldp, wf->K, N2, w32, fb, wf->M, x0 + (size_t)C * N2, N2, at_block(x0h, C / 16),
                                      catb + (size_t)(bott + 64) * N2, N2, at_block(cath, (bott + 64) / 16), c.s));
    }
    Act x;
    x.p = x0; x.C = C + 128; x.ld = N2; x.lay = lay2; x.h = x0h;
    {
        BlkOpt o;
        o.names = {p + ".encode"};
        o.n1 = fc.norm(p + ".encode.norm1");
        o.n2 = fc.norm(p + ".encode.norm2");
        o.out = catb; o.ldo = N2;
        o.want_yh = true; o.yh = cath;
        adain_resblk1d(c, x, o);
    }
    if (c.go() && hipMemcpyAsync(at_block(cath2, bott / 16), at_block(cath, bott / 16), (size_t)((cat - bott) / 16) * blk_bytes,
                                 hipMemcpyDeviceToDevice, c.s) != hipSuccess)
        c.fail((int)hipErrorUnknown);
    Act xc;
    xc.p = catb; xc.C = cat; xc.ld = N2; xc.lay = lay2; xc.h = cath;
    for (int i = 0; i < 2; ++i) {                                       // decode.0, decode.1: 1216 -> 1024, fp32 in place on the concat buffer
        BlkOpt o;
        o.names = {p + ".decode." + std::to_string(i)};
        o.n1 = fc.norm(o.names[0] + ".norm1");
        o.n2 = fc.norm(o.names[0] + ".norm2");
        o.out = catb; o.ldo = N2;
        o.want_yh = true;
        o.yh = i == 0 ? cath2 : cath;                                   // (the image the block does NOT read)
        xc.h = i == 0 ? cath : cath2;
        adain_resblk1d(c, xc, o);
    }
    xc.h = cath;
    Act y;
    for (int i = 2; i <= 5; ++i) {
        BlkOpt o;
        o.names = {p + ".decode." + std::to_string(i)};
        o.n1 = fc.norm(o.names[0] + ".norm1");
        o.n2 = fc.norm(o.names[0] + ".norm2");
        o.want_yh = i == 5;                                             // to_out reads the image
        const Norm nn = i < 5 ? fc.norm(p + ".decode." + std::to_string(i + 1) + ".norm1") : Norm();
        if (i < 5) o.next_n1 = &nn;
        y = adain_resblk1d(c, i == 2 ? xc : y, o);
        if (!y.lay) return;
    }
    ConvOpt q;
    q.bias = m.bias(p + ".to_out.0");
    q.probe = true;                                                     // a non-finite mel raises AS_STATUS_F16_RANGE (16 compares per tile of an 80-row conv)
    conv_h(c, m.conv(p + ".to_out.0"), y.h, y.C, lay2, k1, mel, ldo, q);
}

// ------------------------------------------------------------------------------------------------------------------
// ArtsSpeech.forward(step="test"), models.py:356-371
// ------------------------------------------------------------------------------------------------------------------
struct PhaseA {                   // what the first half leaves in workspace A for the second
    FcOut fc;                     // (recording plans) the AdaIN fc GEMM of the predictors / decoder, run beside the encoders' last layers
    bool has_fc = false;
    float *feat12 = nullptr, *style = nullptr, *a_en = nullptr, *t_en = nullptr, *duration = nullptr;
    int ld_en = 0;
    int32_t *dur_i = nullptr, *frame_off = nullptr;
    const Lay *tok = nullptr, *ref = nullptr;
};

bool batch_ok(const as_batch* b, bool tok, bool ref, bool frames)
{
    return b && b->B > 0 && (!tok || b->tok_lens) && (!ref || b->ref_lens) && (!frames || b->frames);
}
std::vector<int> vec_of(const int32_t* p, int n) { return std::vector<int>(p, p + n); }

PhaseA forward_a(Ctx& c, const as_batch* batch, const as_forward_io* io)
{
    const as_model& m = c.m;
    PhaseA A;
    const int C = m.cfg.hidden_dim, B = batch->B, n_mels = m.cfg.n_mels, sd2 = 2 * m.cfg.style_dim;
    A.tok = c.lay(vec_of(batch->tok_lens, B));
    A.ref = c.lay(vec_of(batch->ref_lens, B));
    if (!A.tok || !A.ref) return A;
    const int Nt = std::max(A.tok->N, 1), Nr = std::max(A.ref->N, 1);
    A.feat12 = c.f32((size_t)12 * Nr);
    // results the caller asked for are written where the caller wants them (no copy nodes in the graph)
    A.style = c.f32((size_t)B * sd2);
    if (io->style) A.style = io->style;
    A.duration = nullptr;
    A.dur_i = c.i32(Nt);
    A.frame_off = c.i32(B + 1);
    if (!batch->frames) {
        if (io->dur_i) A.dur_i = io->dur_i;
        if (io->frame_off) A.frame_off = io->frame_off;
    }
    float* ds = c.f32((size_t)B * (m.cfg.style_dim / 4));
    const float* stats = m.vec("__stats24");
    if (c.go()) c.p.mark(0, c.s);
    c.hint(0, 4.0 * (n_mels + 11.0 + 12.0) * A.ref->N);
    RUN(c, as_ref_features_f32(io->mel, io->ld_mel, n_mels, io->f0_raw, io->ema_raw, io->ld_ema, A.ref->N, stats, A.feat12, A.ref->N, c.s));
    const StyleIn si = style_inputs(c, A.feat12, A.ref->N, io->mel, io->ld_mel, A.ref);
    if (!si.l1) return A;
    if (c.go()) c.p.mark(1, c.s);
    // The articulatory + text encoders (twins: one double-width encoder), the mel tower, the duration predictor and the three
    // small towers are mutually independent (models.py:358-360) and individually too small to fill 256 CUs: four concurrent
    // branches (measured in round 1: 3 branches 9.35 ms, these 4 8.89 ms, 5 branches 9.76 ms per step).
    // (AS_ONLY_BRANCH=k: timing experiments -- only branch k is launched, the others are allocated but skipped; results invalid)
    const char* only_env = getenv("AS_ONLY_BRANCH");
    const int only = only_env ? atoi(only_env) : -1;
    const bool launch0 = c.launch;
    auto gate = [&](int k) { c.launch = launch0 && (only < 0 || only == k); };
    // Side streams: the mel tower; the small towers; the duration predictor's dur_block.  The calling stream runs the three encoders as one
    // triple-width encoder; when the duration predictor's (2 layers) is finished the third side stream waits for it and runs the
    // predictor's tail while the text / articulatory pair runs its last two layers.  (Every edge is calling stream <-> side stream:
    // side-to-side edges break hipGraph instantiation.)
    const bool early = getenv("AS_DUR_EARLY") != nullptr;               // experiment: dur_block from the start on its own stream (slower in a graph)
    // Serial plans that merge (as_plan_set_serial + as_plan_set_merge, the default of a serial plan): the branches are RECORDED (Fork, Op)
    // and played out on the one stream with their ready conv GEMMs sharing launches -- queues cost nothing, so every tower is its own
    // branch and dur_block starts with the others; the predictor's tail joins its queue once the duration encoder is done.  The order of
    // the calls (hence of the workspace allocations) depends on the PLAN's flags only, never on the pass (count / prepare / run).
    const bool rec = c.p.serial && c.p.merge;
    Fork f(c, rec ? 5 : (early ? 3 : 2), 0);
    f.branch(0);
    gate(1);
    style_tower(c, 0, si, A.style);
    f.branch(1);
    gate(3);
    if (rec) {
        style_tower(c, 1, si, A.style);
        f.branch(2);
        style_tower(c, 2, si, A.style);
        f.branch(3);
        gate(2);
        duration_style(c, A.feat12 ? A.feat12 + (size_t)2 * A.ref->N : nullptr, A.ref->N, A.ref, ds);
        // every AdaIN fc layer of the predictors and the decoder (75 MB of weights streamed for 32 columns: bandwidth-bound) needs the
        // Style vector only -- here it rides along with the encoders' compute-bound convs instead of opening the second half alone
        f.after(4, {0, 1, 2});
        f.branch(4);
        gate(3);
        A.fc = adain_fc_all(c, "style", style_norms(m), A.style, sd2, sd2, B);
        A.has_fc = true;
    } else {
        for (int t = 1; t <= 3; ++t) style_tower(c, t, si, A.style);
        if (early) {
            f.branch(2);
            gate(2);
            duration_style(c, A.feat12 ? A.feat12 + (size_t)2 * A.ref->N : nullptr, A.ref->N, A.ref, ds);
        }
    }
    f.back();
    gate(0);
    EncOut eo;
    std::unique_ptr<Fork> f2;
    rel_encoder_multi(c, path_encoders(), io->tokens, A.tok, &eo, [&](int g) {
        if (g != ENC_DUR) return;
        const bool l0 = c.launch;
        if (rec) {
            f.wait_main(3);                                              // the duration encoder's result
            f.branch(3);
        } else if (early) {
            f.wait_main(2);
            f.branch(2);
        } else {
            f2.reset(new Fork(c, 1, 2));
            f2->branch(0);
        }
        gate(2);
        if (!early && !rec) duration_style(c, A.feat12 ? A.feat12 + (size_t)2 * A.ref->N : nullptr, A.ref->N, A.ref, ds);
        A.duration = duration_tail(c, eo.y[ENC_DUR], ds, A.tok, io->duration);
        f.back();
        c.launch = l0;
    });
    A.a_en = eo.y[ENC_ARTS];
    A.t_en = eo.y[ENC_TEXT];
    A.ld_en = eo.ld[ENC_ARTS];
    if (f2) f2->join();
    c.launch = launch0;
    f.join();
    if (c.go()) c.p.mark(2, c.s);
    // round half even -> clamp(min = 1) (or the forced durations), per-utterance frame offsets (models.py:361-366)
    // (with the frame counts given nobody reads this half's copy: the second half computes durations, offsets and the frame -> token map)
    // (... and so does a call under a frame capacity, as_forward_io.frame_cap)
    if (!batch->frames && io->frame_cap <= 0) RUN(c, as_durations_f32(A.duration, io->forced_dur, A.tok->d_off, B, A.dur_i, A.frame_off, nullptr, 0, c.s));
    (void)C;
    return A;
}

void forward_b(Ctx& c, const PhaseA& A, const as_batch* batch, const as_forward_io* io)
{
    const as_model& m = c.m;
    const int C = m.cfg.hidden_dim, B = batch->B, n_mels = m.cfg.n_mels;
    // frame counts from the host (forced durations / a second pass), or -- as_forward_io.frame_cap -- a capacity: the layouts are then
    // room, and what lies where is derived on the device from the durations this half computes (no read-back, capturable)
    const bool dyn = !batch->frames;
    const as_segments* segs = dyn ? io->segs : nullptr;
    const Lay* lay1 = dyn ? c.dyn_lay(0, B, io->frame_cap) : c.lay(vec_of(batch->frames, B));
    if (!lay1) return;
    const Lay* lay2 = c.scaled(lay1, 2);
    if (!lay2) return;
    const int N1 = lay1->N, N2 = lay2->N;
    if (segs) {
        long sum = 0;
        if (segs->n < 1 || segs->n > AS_MAX_SEGMENTS || segs->first[0] != 0 || segs->first[segs->n] != B) { c.fail(AS_EINVAL); return; }
        for (int i = 0; i < segs->n; ++i) {
            if (segs->first[i + 1] <= segs->first[i] || segs->cap[i] < 1 || !segs->mel_out[i]) { c.fail(AS_EINVAL); return; }
            if (segs->ld_out[i] < 2 * segs->cap[i]) { c.fail(AS_ENOSPC); return; }
            sum += segs->cap[i];
        }
        if (sum != N1 || io->F0) { c.fail(AS_EINVAL); return; }         // (the predictions have no per-submission home)
    } else if (io->ld_out < N2 || (io->F0 && io->ld_pred < N2)) { c.fail(AS_ENOSPC); return; }
    int32_t* tof = c.i32((size_t)std::max(N1, 1));
    int32_t* dur_i = c.i32((size_t)std::max(A.tok->N, 1));
    int32_t* frame_off = c.i32(B + 1);
    float* mel_packed = dyn ? c.f32((size_t)n_mels * std::max(N2, 1)) : nullptr;   // (a merged call's mel before it goes to the submissions' slots)
    if (io->dur_i) dur_i = io->dur_i;
    if (io->frame_off && !segs) frame_off = io->frame_off;
    RUN(c, as_durations_f32(A.duration, io->forced_dur, A.tok->d_off, B, dur_i, frame_off, tof, N1, c.s));
    if (dyn) {
        const Lay *lg1 = c.dyn_lay(2, B, io->frame_cap), *lg2 = lg1 ? c.scaled(lg1, 2) : nullptr;
        if (!lg1 || !lg2) return;
        if (c.go()) {
            AsDynGeo g;
            memset(&g, 0, sizeof(g));
            g.frame_off = frame_off; g.B = B; g.cap1 = N1;
            g.n_seg = segs ? segs->n : 1;
            for (int i = 0; i < g.n_seg; ++i) {
                g.seg_first[i] = segs ? segs->first[i] : 0;
                g.seg_cap[i] = segs ? segs->cap[i] : N1;
                g.seg_frame_off[i] = segs ? segs->frame_off[i] : nullptr;
            }
            g.seg_first[g.n_seg] = B;
            const Lay* ls[4] = {lay1, lay2, lg1, lg2};
            for (int i = 0; i < 4; ++i) {
                g.w[i] = ls[i]->d_w; g.off[i] = ls[i]->d_off; g.nvalid[i] = ls[i]->d_nvalid;
                g.meta[i] = reinterpret_cast<unsigned long long*>(ls[i]->d_meta);
            }
            auto t3 = lg1->tabs.find("src3");
            g.src3 = t3 != lg1->tabs.end() ? t3->second : nullptr;
            g.tof = tof;
            g.status = as_status_words_device();
            RUN(c, as_dyn_geometry_launch(g, c.s));
        }
    }
    float* a_ex = c.f32((size_t)C * std::max(N1, 1));
    float* fne = c.f32((size_t)12 * std::max(N2, 1));                    // rows: F0, N, EMA[10] (what the three branches predict)
    float* x0 = c.f32((size_t)(C + 128) * std::max(N2, 1));
    // every AdaIN fc layer of the predictors and the decoder (~75 MB of weights): one GEMM on the style vectors
    const FcOut fc = A.has_fc ? A.fc : adain_fc_all(c, "style", style_norms(m), A.style, 2 * m.cfg.style_dim, 2 * m.cfg.style_dim, B);
    // T_en @ pred_aln_trg is a column gather (models.py:367-368); the text encoding at the mel rate is nearest x2 of it (models.py:500).
    // What the decoder can do before the predictors are done -- x0's text part, its image, asr_res -- a recording plan runs BESIDE them
    // (one more queue: the 64-channel asr_res conv then rides in a predictor launch).
    const bool rec = c.p.serial && c.p.merge;
    DecPre dp;
    {
        std::unique_ptr<Fork> f;
        if (rec) {
            f.reset(new Fork(c, 1, 0));
            f->branch(0);
            c.hint(0, 4.0 * C * ((double)A.tok->N + N2));
            RUN(c, as_expand_f32(A.t_en, A.ld_en, C, tof, N1, 2, x0, N2, c.s));
            dp = decoder_pre(c, x0, lay2);
            f->back();
        }
        c.hint(0, 4.0 * C * ((double)A.tok->N + N1));
        RUN(c, as_expand_f32(A.a_en, A.ld_en, C, tof, N1, 1, a_ex, N1, c.s));
        arts_predictor(c, a_ex, N1, lay1, fc, fne, N2);
        if (f) f->join();
    }
    if (c.go()) c.p.mark(3, c.s);
    if (!rec) {
        c.hint(0, 4.0 * C * ((double)A.tok->N + N2));
        RUN(c, as_expand_f32(A.t_en, A.ld_en, C, tof, N1, 2, x0, N2, c.s));
        dp = decoder_pre(c, x0, lay2);
    }
    if (segs) {
        decoder(c, dp, x0, lay2, fne, N2, fc, mel_packed, N2);
        if (c.go()) {
            AsSegScatter sc;
            memset(&sc, 0, sizeof(sc));
            sc.src = mel_packed; sc.ld_src = N2; sc.rows = n_mels; sc.n_seg = segs->n; sc.frame_off = frame_off;
            for (int i = 0; i < segs->n; ++i) {
                sc.seg_first[i] = segs->first[i]; sc.seg_cap[i] = segs->cap[i];
                sc.dst[i] = segs->mel_out[i]; sc.ld_dst[i] = segs->ld_out[i];
            }
            sc.seg_first[segs->n] = B;
            RUN(c, as_seg_scatter_launch(sc, c.s));
        }
    } else {
        decoder(c, dp, x0, lay2, fne, N2, fc, io->mel_out, io->ld_out);
    }
    if (c.go()) c.p.mark(4, c.s);
    if (c.go()) {
        if (io->F0) copy_rows(c, io->F0, io->ld_pred, fne, N2, 1, N2);
        if (io->N) copy_rows(c, io->N, io->ld_pred, fne + (size_t)N2, N2, 1, N2);
        if (io->EMA) copy_rows(c, io->EMA, io->ld_pred, fne + (size_t)2 * N2, N2, 10, N2);
    }
    (void)n_mels;
}

void outputs_a(Ctx& c, const PhaseA& A, const as_batch* batch, const as_forward_io* io)
{
    if (!c.go()) return;
    const int C = c.m.cfg.hidden_dim, Nt = A.tok->N, Nr = A.ref->N;
    if (io->feat12) copy_rows(c, io->feat12, io->ld_feat, A.feat12, Nr, 12, Nr);
    if (io->t_en) copy_rows(c, io->t_en, io->ld_en, A.t_en, A.ld_en, C, Nt);
    if (io->a_en) copy_rows(c, io->a_en, io->ld_en, A.a_en, A.ld_en, C, Nt);
    (void)batch;
}

bool io_ok(const as_forward_io* io, bool need_out)
{
    return io && io->tokens && io->mel && io->f0_raw && io->ema_raw && (!need_out || io->mel_out);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
namespace {

bool read_blob(const void* blob, size_t bytes, std::unordered_map<std::string, HostT>* raw_out)
{
    // Every bound is checked in subtraction form against what is left (o <= bytes always holds): nothing here can wrap for any
    // header content, and nothing is allocated from a count the file merely claims.
    const unsigned char* b = static_cast<const unsigned char*>(blob);
    size_t o = 0;
    auto need = [&](size_t n) { return n <= bytes - o; };
    if (!need(12) || memcmp(b, "ASWBLOB1", 8) != 0) return false;
    o = 8;
    uint32_t n;
    memcpy(&n, b + o, 4);
    o += 4;
    if ((size_t)n > (bytes - o) / 11) return false;         // an entry is at least 2 + 0 + 1 + 0 + 8 bytes
    struct Ent { std::string name; std::vector<int> dims; uint64_t off; size_t numel; };
    std::vector<Ent> ents(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint16_t nl;
        if (!need(2)) return false;
        memcpy(&nl, b + o, 2);
        o += 2;
        if (!need((size_t)nl + 1)) return false;
        ents[i].name.assign(reinterpret_cast<const char*>(b + o), nl);
        o += nl;
        const int nd = b[o++];
        if (nd > 8 || !need((size_t)nd * 4 + 8)) return false;
        size_t numel = 1;
        for (int d = 0; d < nd; ++d) {
            uint32_t v;
            memcpy(&v, b + o, 4);
            o += 4;
            if (v > (uint32_t)INT32_MAX) return false;
            if (v != 0 && numel > (SIZE_MAX / 4) / v) return false;   // numel * 4 must not wrap either
            numel *= v;
            ents[i].dims.push_back((int)v);
        }
        ents[i].numel = numel;
        memcpy(&ents[i].off, b + o, 8);
        o += 8;
    }
    uint64_t data_bytes;
    if (!need(8)) return false;
    memcpy(&data_bytes, b + o, 8);
    o += 8;
    if (data_bytes > bytes - o) return false;
    for (auto& e : ents) {
        if (e.off > data_bytes || e.numel * 4 > data_bytes - e.off) return false;
        HostT t;
        t.dims = e.dims;
        t.v.resize(e.numel);
        if (e.numel) memcpy(t.v.data(), b + o + e.off, e.numel * 4);
        (*raw_out)[e.name] = std::move(t);
    }
    return true;
}

bool ends_with(const std::string& s, const char* suf)
{
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// weight_norm (torch old-style, dim 0): w = v * g / ||v|| over all dims but 0; spectral_norm in eval: w = W / (u . (W2d v))
// (artspeech_amd/weights.py is the Python statement of the same)
bool fold(const std::unordered_map<std::string, HostT>& in, std::unordered_map<std::string, HostT>* out)
{
    for (const auto& kv : in) {
        std::string k = kv.first;
        if (k.compare(0, 7, "module.") == 0) k = k.substr(7);
        if (k.compare(0, 30, "style_encoder.pitch_extractor.") == 0 || k.compare(0, 28, "style_encoder.ema_extractor.") == 0) continue;
        auto get = [&](const std::string& name) -> const HostT* {
            auto it = in.find(name);
            if (it == in.end()) it = in.find("module." + name);
            return it == in.end() ? nullptr : &it->second;
        };
        if (ends_with(k, ".weight_g")) {
            const std::string p = k.substr(0, k.size() - 9);
            const HostT* v = get(p + ".weight_v");
            if (!v || v->dims.empty() || kv.second.numel() != (size_t)v->dims[0]) return false;
            HostT w;
            w.dims = v->dims;
            w.v.resize(v->numel());
            const size_t rows = v->dims[0], per = v->numel() / rows;
            for (size_t r = 0; r < rows; ++r) {
                double s = 0;
                for (size_t i = 0; i < per; ++i) s += (double)v->v[r * per + i] * v->v[r * per + i];
                const float scale = kv.second.v[r] / (float)sqrt(s);
                for (size_t i = 0; i < per; ++i) w.v[r * per + i] = v->v[r * per + i] * scale;
            }
            (*out)[p + ".weight"] = std::move(w);
        } else if (ends_with(k, ".weight_orig")) {
            const std::string p = k.substr(0, k.size() - 12);
            const HostT *u = get(p + ".weight_u"), *vv = get(p + ".weight_v");
            const HostT& W = kv.second;
            if (!u || !vv || W.dims.empty() || u->numel() != (size_t)W.dims[0] || vv->numel() * W.dims[0] != W.numel()) return false;
            const size_t rows = W.dims[0], per = W.numel() / rows;
            double sigma = 0;
            for (size_t r = 0; r < rows; ++r) {
                double s = 0;
                for (size_t i = 0; i < per; ++i) s += (double)W.v[r * per + i] * vv->v[i];
                sigma += (double)u->v[r] * (double)(float)s;
            }
            HostT w;
            w.dims = W.dims;
            w.v.resize(W.numel());
            const float sg = (float)sigma;
            for (size_t i = 0; i < W.numel(); ++i) w.v[i] = W.v[i] / sg;
            (*out)[p + ".weight"] = std::move(w);
        } else if (ends_with(k, ".weight_v") || ends_with(k, ".weight_u")) {
            continue;
        } else {
            (*out)[k] = kv.second;
        }
    }
    return true;
}

size_t count_module(const as_model* m, as_plan* p, int module, const as_batch* batch, bool prepare);

}  // namespace

namespace {
// Two towers of the same architecture on different input channels (style_encoder.energy_block / F0_block + their Linear layers,
// models.py:402-411,414-415) as ONE tower of twice the width: every conv weight becomes block diagonal (the off-diagonal zeros multiply
// exactly to zero), depthwise weights and biases are concatenated.  Half the launches -- these 1-D towers are ~20 kernels of 8-10 us
// each, all at the fixed cost of a launch -- for twice the (negligible) flop.  Channel order: tower a first.
bool merge_twin_towers(std::unordered_map<std::string, HostT>* raw, const std::string& a, const std::string& b, const std::string& merged)
{
    std::vector<std::pair<std::string, HostT>> add;
    for (const auto& kv : *raw) {
        if (kv.first.compare(0, a.size() + 1, a + ".") != 0) continue;
        const std::string sfx = kv.first.substr(a.size());
        const auto itb = raw->find(b + sfx);
        if (itb == raw->end() || itb->second.dims != kv.second.dims) return false;
        const HostT &ta = kv.second, &tb = itb->second;
        HostT t;
        const bool is_w = sfx.size() >= 7 && sfx.compare(sfx.size() - 7, 7, ".weight") == 0;
        if (!is_w || sfx.find(".pool.") != std::string::npos) {           // bias / depthwise [C][1][k]: concatenate
            t.dims = ta.dims;
            t.dims[0] *= 2;
            t.v = ta.v;
            t.v.insert(t.v.end(), tb.v.begin(), tb.v.end());
        } else {                                                          // [Cout][Cin][k]: block diagonal
            const int co = ta.dim(0), ci = ta.dim(1), k = (int)(ta.numel() / ((size_t)co * ci));
            t.dims = ta.dims;
            t.dims[0] = 2 * co;
            t.dims[1] = 2 * ci;
            t.v.assign((size_t)4 * co * ci * k, 0.f);
            for (int m = 0; m < co; ++m)
                for (int c = 0; c < ci; ++c)
                    for (int j = 0; j < k; ++j) {
                        t.v[((size_t)m * 2 * ci + c) * k + j] = ta.v[((size_t)m * ci + c) * k + j];
                        t.v[((size_t)(co + m) * 2 * ci + ci + c) * k + j] = tb.v[((size_t)m * ci + c) * k + j];
                    }
        }
        add.emplace_back(merged + sfx, std::move(t));
    }
    if (add.empty()) return false;
    for (auto& kv : add) (*raw)[kv.first] = std::move(kv.second);
    return true;
}
// Linear layers of the twins -> one block Linear on [pooled a | pooled b] whose output rows are [rows of lb | rows of la] (the Style
// vector holds the F0 slice before the energy slice, models.py:423)
bool merge_twin_linears(std::unordered_map<std::string, HostT>* raw, const std::string& la, const std::string& lb, const std::string& merged)
{
    const auto wa = raw->find(la + ".weight"), wb = raw->find(lb + ".weight"), ba = raw->find(la + ".bias"), bb = raw->find(lb + ".bias");
    if (wa == raw->end() || wb == raw->end() || ba == raw->end() || bb == raw->end() || wa->second.dims != wb->second.dims) return false;
    const int o = wa->second.dim(0), i = wa->second.dim(1);
    HostT w, bias;
    w.dims = {2 * o, 2 * i};
    w.v.assign((size_t)4 * o * i, 0.f);
    for (int m = 0; m < o; ++m)
        for (int c = 0; c < i; ++c) {
            w.v[(size_t)m * 2 * i + i + c] = wb->second.v[(size_t)m * i + c];          // rows 0 .. o-1: lb on the second half of the input
            w.v[(size_t)(o + m) * 2 * i + c] = wa->second.v[(size_t)m * i + c];        // rows o .. : la on the first half
        }
    bias.dims = {2 * o};
    bias.v = bb->second.v;
    bias.v.insert(bias.v.end(), ba->second.v.begin(), ba->second.v.end());
    (*raw)[merged + ".weight"] = std::move(w);
    (*raw)[merged + ".bias"] = std::move(bias);
    return true;
}
}  // namespace

static int model_create(const void* blob_host, size_t blob_bytes, const as_model_cfg* cfg, as_model** out)
{
    std::unordered_map<std::string, HostT> blob;
    if (!read_blob(blob_host, blob_bytes, &blob)) return AS_EINVAL;
    struct Guard {                                                       // every error return below gives the device memory back
        std::unique_ptr<as_model> m;
        ~Guard() { if (m) m->pool.release(); }
    } g;
    g.m.reset(new as_model());
    as_model* m = g.m.get();
    m->cfg = *cfg;
    if (!fold(blob, &m->raw)) return AS_EINVAL;
    blob.clear();
    // energy tower (input: crop row 0) + F0 tower (row 1) as one double-width tower
    if (!merge_twin_towers(&m->raw, "style_encoder.energy_block", "style_encoder.F0_block", "style_encoder.ENF0_block") ||
        !merge_twin_linears(&m->raw, "style_encoder.Energylinear", "style_encoder.F0linear", "style_encoder.ENF0linear"))
        return AS_EINVAL;
    AS_CHECK(hipGetDevice(&m->device));
    {
        HostT st;
        st.dims = {24};
        st.v.assign(cfg->stats, cfg->stats + 24);
        m->raw["__stats24"] = st;
    }
    // prepare pass: walk the whole launch sequence once on a minimal geometry; every weight it touches is built and uploaded
    as_plan* plan = nullptr;
    int rc = as_plan_create(m, &plan);
    if (rc != AS_OK) return rc;
    const int32_t tl[1] = {8}, rl[1] = {96}, fr[1] = {12};
    as_batch b = {1, tl, rl, fr};
    for (int mod : {AS_MOD_FORWARD_A, AS_MOD_FORWARD_B}) count_module(m, plan, mod, &b, true);
    as_plan_destroy(plan);
    if (m->err) return m->err;
    m->frozen = true;
    AS_CHECK(hipDeviceSynchronize());
    *out = g.m.release();
    return AS_OK;
}

extern "C" int as_model_create(const void* blob_host, size_t blob_bytes, const as_model_cfg* cfg, as_model** out)
{
    if (!blob_host || !cfg || !out || cfg->hidden_dim <= 0 || cfg->hidden_dim % 16 || cfg->dim_in <= 0 || cfg->style_dim <= 0 || cfg->style_dim % 4 ||
        cfg->n_mels <= 0)
        return AS_EINVAL;
    try {                                                                // nothing may unwind through the C boundary
        return model_create(blob_host, blob_bytes, cfg, out);
    } catch (const std::bad_alloc&) {
        return (int)hipErrorOutOfMemory;
    } catch (...) {
        return AS_EINVAL;
    }
}

extern "C" int as_model_destroy(as_model* m)
{
    if (!m) return AS_EINVAL;
    m->pool.release();
    delete m;
    return AS_OK;
}

extern "C" int as_model_get_cfg(const as_model* m, as_model_cfg* out)
{
    if (!m || !out) return AS_EINVAL;
    *out = m->cfg;
    return AS_OK;
}

extern "C" int as_plan_create(const as_model* m, as_plan** out)
{
    if (!m || !out) return AS_EINVAL;
    as_plan* p = new as_plan();
    p->model = m;
    *out = p;
    return AS_OK;
}

extern "C" int as_plan_destroy(as_plan* p)
{
    if (!p) return AS_EINVAL;
    for (hipStream_t s : p->side) (void)hipStreamDestroy(s);
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->marks)
        if (e) (void)hipEventDestroy(e);
    p->pool.release();
    delete p;
    return AS_OK;
}

extern "C" int as_plan_set_serial(as_plan* p, int on)
{
    if (!p) return AS_EINVAL;
    p->serial = on != 0;
    return AS_OK;
}

extern "C" int as_plan_set_merge(as_plan* p, int on)
{
    if (!p) return AS_EINVAL;
    p->merge = on != 0;
    return AS_OK;
}

extern "C" int as_plan_set_operand_mode(as_plan* p, int n_prod)
{
    if (!p || (n_prod != 1 && n_prod != 3)) return AS_EINVAL;
    p->n_prod = n_prod;
    return AS_OK;
}

extern "C" int as_plan_set_timing(as_plan* p, int on)
{
    if (!p) return AS_EINVAL;
    p->timing = on != 0;
    return AS_OK;
}

// ms between the phase marks of the last as_forward_test on this plan: [0] reference features + tower inputs, [1] the four
// concurrent branches (encoders, towers, duration predictor), [2] durations + AdaIN fc GEMM + predictors, [3] decoder
extern "C" int as_plan_phase_ms(as_plan* p, float* ms, int n)
{
    if (!p || !ms || n < 4) return AS_EINVAL;
    for (int i = 0; i < 5; ++i)
        if (!p->marks[i]) return AS_EINVAL;
    AS_CHECK(hipEventSynchronize(p->marks[4]));
    for (int i = 0; i < 4; ++i) AS_CHECK(hipEventElapsedTime(&ms[i], p->marks[i], p->marks[i + 1]));
    return AS_OK;
}

namespace {

const as_forward_io* dummy_io()
{
    static as_forward_io io;
    static bool init = false;
    if (!init) {
        memset(&io, 0, sizeof(io));
        io.tokens = reinterpret_cast<const int32_t*>(16);
        io.mel = io.f0_raw = io.ema_raw = reinterpret_cast<const float*>(16);
        io.mel_out = reinterpret_cast<float*>(16);
        io.ld_mel = io.ld_ema = io.ld_out = io.ld_pred = 1 << 30;
        init = true;
    }
    return &io;
}

// the sequence of one module with nothing behind it: workspace bytes (and, for as_model_create, the weights it touches)
size_t count_module(const as_model* m, as_plan* p, int module, const as_batch* batch, bool prepare)
{
    Ctx c(*m, *p, nullptr, nullptr, 0, false, true);
    (void)prepare;
    const as_forward_io* io = dummy_io();
    const int B = batch->B, C = m->cfg.hidden_dim;
    switch (module) {
    case AS_MOD_FORWARD_A:
        if (!batch_ok(batch, true, true, false)) return 0;
        forward_a(c, batch, io);
        break;
    case AS_MOD_FORWARD_B: {
        if (!batch_ok(batch, true, true, true)) return 0;
        Ctx ca(*m, *p, nullptr, nullptr, 0, false, true);
        const PhaseA A = forward_a(ca, batch, io);
        if (ca.rc || !A.tok) return 0;
        forward_b(c, A, batch, io);
        break;
    }
    case AS_MOD_FORWARD_B_CAP: {                                         // batch->frames = capacities: their sum is the call's frame_cap
        if (!batch_ok(batch, true, true, true)) return 0;
        long cap = 0;
        for (int b = 0; b < B; ++b) cap += std::max(batch->frames[b], 0);
        if (cap < 1 || cap > (1 << 28)) return 0;
        as_forward_io ioc = *io;
        ioc.frame_cap = (int32_t)cap;
        as_batch b2 = *batch;
        b2.frames = nullptr;
        Ctx ca(*m, *p, nullptr, nullptr, 0, false, true);
        const PhaseA A = forward_a(ca, &b2, &ioc);
        if (ca.rc || !A.tok) return 0;
        forward_b(c, A, &b2, &ioc);
        break;
    }
    case AS_MOD_ENCODER: {
        if (!batch_ok(batch, true, false, false)) return 0;
        const Lay* lay = c.lay(vec_of(batch->tok_lens, B));
        EncOut eo;
        if (lay) rel_encoder_multi(c, path_encoders(), io->tokens, lay, &eo, [](int) {});
        break;
    }
    case AS_MOD_STYLE: {
        if (!batch_ok(batch, false, true, false)) return 0;
        const Lay* ref = c.lay(vec_of(batch->ref_lens, B));
        if (!ref) return 0;
        const StyleIn si = style_inputs(c, nullptr, ref->N, nullptr, ref->N, ref);
        if (si.l1) for (int t = 0; t < 4; ++t) style_tower(c, t, si, nullptr);
        break;
    }
    case AS_MOD_DURATION: {
        if (!batch_ok(batch, true, true, false)) return 0;
        const Lay *tok = c.lay(vec_of(batch->tok_lens, B)), *ref = c.lay(vec_of(batch->ref_lens, B));
        if (!tok || !ref) return 0;
        float* ds = c.f32((size_t)B * (m->cfg.style_dim / 4));
        duration_style(c, nullptr, ref->N, ref, ds);
        EncOut eo;
        rel_encoder_multi(c, path_encoders(), io->tokens, tok, &eo, [](int) {});
        duration_tail(c, eo.y[ENC_DUR], ds, tok);
        break;
    }
    case AS_MOD_ARTS: {
        if (!batch_ok(batch, false, false, true)) return 0;
        const Lay* lay = c.lay(vec_of(batch->frames, B));
        if (!lay) return 0;
        const FcOut fc = adain_fc_all(c, "style", style_norms(*m), nullptr, 2 * m->cfg.style_dim, 2 * m->cfg.style_dim, B);
        float* fne = c.f32((size_t)12 * std::max(2 * lay->N, 1));
        arts_predictor(c, nullptr, lay->N, lay, fc, fne, 2 * lay->N);
        break;
    }
    case AS_MOD_DECODER: {
        if (!batch_ok(batch, false, false, true)) return 0;
        const Lay* lay = c.lay(vec_of(batch->frames, B));
        if (!lay) return 0;
        const Lay* lay2 = c.scaled(lay, 2);
        c.i32((size_t)std::max(lay->N, 1));
        float* x0 = c.f32((size_t)(C + 128) * std::max(lay2->N, 1));
        float* fne = c.f32((size_t)12 * std::max(lay2->N, 1));
        const FcOut fc = adain_fc_all(c, "style", style_norms(*m), nullptr, 2 * m->cfg.style_dim, 2 * m->cfg.style_dim, B);
        decoder(c, decoder_pre(c, x0, lay2), x0, lay2, fne, lay2->N, fc, nullptr, lay2->N);
        break;
    }
    default: return 0;
    }
    return c.rc ? 0 : c.off + 256;
}

struct Call {                     // common prologue of the run entry points
    Ctx c;
    Call(const as_model* m, as_plan* p, void* ws, size_t bytes, as_stream_t stream, bool launch = true, bool first = true)
        : c(*m, *p, static_cast<hipStream_t>(stream), ws, bytes, launch, false)
    {
        if (launch) {
            p->next_event = 0;
            const int t = first ? p->trim(static_cast<hipStream_t>(stream)) : AS_OK;   // (as_forward_test_finish continues _begin's call: it needs _begin's layouts)
            if (t != AS_OK) c.fail(t);
            p->note_stream(static_cast<hipStream_t>(stream));
            if (as_status_peek()) c.fail(AS_EDEVICE);      // a kernel of earlier work reported a failure: sticky until as_device_status(1)
        }
        if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0) c.fail(AS_EINVAL);
    }
    int done() const { return c.rc ? c.rc : c.m.err; }
};

}  // namespace

extern "C" size_t as_module_workspace_bytes(const as_model* m, as_plan* p, int module, const as_batch* batch)
{
    if (!m || !p || !batch) return 0;
    // (no trim here: a caller asks for workspace B's size between as_forward_test_begin and _finish, which share layouts)
    return count_module(m, p, module, batch, false);
}

extern "C" int as_plan_set_layout_cap(as_plan* p, int max_layouts)
{
    if (!p || max_layouts < 64) return AS_EINVAL;          // (one forward touches ~20 layouts: the cap must hold a call's own)
    p->lay_cap = (size_t)max_layouts;
    return AS_OK;
}

extern "C" int as_plan_layout_flushes(const as_plan* p) { return p ? p->layout_flushes : AS_EINVAL; }
extern "C" int as_plan_layout_count(const as_plan* p) { return p ? (int)p->lays.size() : AS_EINVAL; }

extern "C" int as_plan_reset_layouts(as_plan* p)
{
    if (!p) return AS_EINVAL;
    return p->drop_layouts();
}

extern "C" int as_encoder_forward(const as_model* m, as_plan* p, int which, const as_batch* batch, const int32_t* tokens, float* out, int ldo,
                                  void* ws, size_t ws_bytes, as_stream_t stream)
try {                                                                    // nothing may unwind through the C boundary
    if (!m || !p || !batch_ok(batch, true, false, false) || !tokens || !out || which < 0 || which > 2) return AS_EINVAL;
    Call k(m, p, ws, ws_bytes, stream);
    Ctx& c = k.c;
    const Lay* lay = c.lay(vec_of(batch->tok_lens, batch->B));
    if (!lay || ldo < lay->N) return AS_EINVAL;
    // the three encoders exist as ONE stacked weight set (they always run together in the path): a single one is asked for by
    // running all of them and returning its columns
    EncOut eo;
    rel_encoder_multi(c, path_encoders(), tokens, lay, &eo, [](int) {});
    const int g = which == 0 ? ENC_TEXT : (which == 1 ? ENC_ARTS : ENC_DUR);
    copy_rows(c, out, ldo, eo.y[g], eo.ld[g], m->cfg.hidden_dim, lay->N);
    return k.done();
} catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
} catch (...) {
    return AS_EINVAL;
}

extern "C" int as_style_forward(const as_model* m, as_plan* p, const as_batch* batch, const float* mel, int ldm, const float* f0_raw,
                                const float* ema_raw, int lde, float* feat12, int ldf, float* style, void* ws, size_t ws_bytes, as_stream_t stream)
try {                                                                    // nothing may unwind through the C boundary
    if (!m || !p || !batch_ok(batch, false, true, false) || !mel || !f0_raw || !ema_raw || !feat12 || !style) return AS_EINVAL;
    Call k(m, p, ws, ws_bytes, stream);
    Ctx& c = k.c;
    const Lay* ref = c.lay(vec_of(batch->ref_lens, batch->B));
    if (!ref || ldm < ref->N || lde < ref->N || ldf < ref->N) return AS_EINVAL;
    const float* stats = m->vec("__stats24");
    RUN(c, as_ref_features_f32(mel, ldm, m->cfg.n_mels, f0_raw, ema_raw, lde, ref->N, stats, feat12, ldf, c.s));
    const StyleIn si = style_inputs(c, feat12, ldf, mel, ldm, ref);
    if (si.l1) for (int t = 0; t < 4; ++t) style_tower(c, t, si, style);
    return k.done();
} catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
} catch (...) {
    return AS_EINVAL;
}

extern "C" int as_duration_forward(const as_model* m, as_plan* p, const as_batch* batch, const int32_t* tokens, const float* ema_ext, int lde,
                                   float* duration, void* ws, size_t ws_bytes, as_stream_t stream)
try {                                                                    // nothing may unwind through the C boundary
    if (!m || !p || !batch_ok(batch, true, true, false) || !tokens || !ema_ext || !duration) return AS_EINVAL;
    Call k(m, p, ws, ws_bytes, stream);
    Ctx& c = k.c;
    const Lay *tok = c.lay(vec_of(batch->tok_lens, batch->B)), *ref = c.lay(vec_of(batch->ref_lens, batch->B));
    if (!tok || !ref || lde < ref->N) return AS_EINVAL;
    float* ds = c.f32((size_t)batch->B * (m->cfg.style_dim / 4));
    duration_style(c, ema_ext, lde, ref, ds);
    EncOut eo;
    rel_encoder_multi(c, path_encoders(), tokens, tok, &eo, [](int) {});
    duration_tail(c, eo.y[ENC_DUR], ds, tok, duration);
    return k.done();
} catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
} catch (...) {
    return AS_EINVAL;
}

extern "C" int as_arts_forward(const as_model* m, as_plan* p, const as_batch* batch, const float* a_ens, int lda, const float* style, float* F0,
                               float* N, float* EMA, int ldp, void* ws, size_t ws_bytes, as_stream_t stream)
try {                                                                    // nothing may unwind through the C boundary
    if (!m || !p || !batch_ok(batch, false, false, true) || !a_ens || !style || !F0 || !N || !EMA) return AS_EINVAL;
    Call k(m, p, ws, ws_bytes, stream);
    Ctx& c = k.c;
    const Lay* lay = c.lay(vec_of(batch->frames, batch->B));
    if (!lay || lda < lay->N || ldp < 2 * lay->N) return AS_EINVAL;
    const int N2 = 2 * lay->N, sd2 = 2 * m->cfg.style_dim;
    const FcOut fc = adain_fc_all(c, "style", style_norms(*m), style, sd2, sd2, batch->B);
    float* fne = c.f32((size_t)12 * std::max(N2, 1));
    arts_predictor(c, a_ens, lda, lay, fc, fne, N2);
    copy_rows(c, F0, ldp, fne, N2, 1, N2);
    copy_rows(c, N, ldp, fne + (size_t)N2, N2, 1, N2);
    copy_rows(c, EMA, ldp, fne + (size_t)2 * N2, N2, 10, N2);
    return k.done();
} catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
} catch (...) {
    return AS_EINVAL;
}

extern "C" int as_decoder_forward(const as_model* m, as_plan* p, const as_batch* batch, const float* asr, int lda, const float* style,
                                  const float* F0, const float* N, const float* EMA, int ldp, float* mel, int ldo, void* ws, size_t ws_bytes,
                                  as_stream_t stream)
try {                                                                    // nothing may unwind through the C boundary
    if (!m || !p || !batch_ok(batch, false, false, true) || !asr || !style || !F0 || !N || !EMA || !mel) return AS_EINVAL;
    Call k(m, p, ws, ws_bytes, stream);
    Ctx& c = k.c;
    const Lay* lay = c.lay(vec_of(batch->frames, batch->B));
    if (!lay) return AS_EINVAL;
    const Lay* lay2 = c.scaled(lay, 2);
    if (!lay2 || lda < lay->N || ldp < lay2->N || ldo < lay2->N) return AS_EINVAL;
    // nearest x2 of the text encoding (models.py:500) = a column gather with every frame as its own token
    int32_t* ident = c.i32((size_t)std::max(lay->N, 1));
    if (c.go() && lay->N > 0) {
        std::vector<int32_t> h(lay->N);
        for (int i = 0; i < lay->N; ++i) h[i] = i;
        if (hipMemcpyAsync(ident, h.data(), (size_t)lay->N * 4, hipMemcpyHostToDevice, c.s) != hipSuccess || hipStreamSynchronize(c.s) != hipSuccess)
            c.fail((int)hipErrorUnknown);
    }
    const int C = m->cfg.hidden_dim;
    float* x0 = c.f32((size_t)(C + 128) * std::max(lay2->N, 1));
    RUN(c, as_expand_f32(asr, lda, C, ident, lay->N, 2, x0, lay2->N, c.s));
    float* fne = c.f32((size_t)12 * std::max(lay2->N, 1));
    copy_rows(c, fne, lay2->N, F0, ldp, 1, lay2->N);
    copy_rows(c, fne + (size_t)lay2->N, lay2->N, N, ldp, 1, lay2->N);
    copy_rows(c, fne + (size_t)2 * lay2->N, lay2->N, EMA, ldp, 10, lay2->N);
    const int sd2 = 2 * m->cfg.style_dim;
    const FcOut fc = adain_fc_all(c, "style", style_norms(*m), style, sd2, sd2, batch->B);
    decoder(c, decoder_pre(c, x0, lay2), x0, lay2, fne, lay2->N, fc, mel, ldo);
    return k.done();
} catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
} catch (...) {
    return AS_EINVAL;
}

extern "C" int as_forward_test_begin(const as_model* m, as_plan* p, const as_batch* batch, const as_forward_io* io, void* ws_a, size_t ws_a_bytes,
                                     as_stream_t stream)
try {                                                                    // nothing may unwind through the C boundary
    if (!m || !p || !batch_ok(batch, true, true, false) || !io_ok(io, false)) return AS_EINVAL;
    Call k(m, p, ws_a, ws_a_bytes, stream);
    const PhaseA A = forward_a(k.c, batch, io);
    if (A.tok && A.ref) outputs_a(k.c, A, batch, io);
    return k.done();
} catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
} catch (...) {
    return AS_EINVAL;
}

extern "C" int as_forward_test_finish(const as_model* m, as_plan* p, const as_batch* batch, const as_forward_io* io, void* ws_a, size_t ws_a_bytes,
                                      void* ws_b, size_t ws_b_bytes, as_stream_t stream)
try {                                                                    // nothing may unwind through the C boundary
    if (!m || !p || !batch_ok(batch, true, true, true) || !io_ok(io, true)) return AS_EINVAL;
    // recover where the first half left its results: the same allocation sequence, nothing launched
    Call ka(m, p, ws_a, ws_a_bytes, stream, false);
    const PhaseA A = forward_a(ka.c, batch, io);
    if (ka.done() || !A.tok || !A.ref) return ka.done() ? ka.done() : AS_EINVAL;
    Call kb(m, p, ws_b, ws_b_bytes, stream, true, false);
    forward_b(kb.c, A, batch, io);
    return kb.done();
} catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
} catch (...) {
    return AS_EINVAL;
}

extern "C" int as_forward_test(const as_model* m, as_plan* p, const as_batch* batch, const as_forward_io* io, void* ws_a, size_t ws_a_bytes,
                               void* ws_b, size_t ws_b_bytes, int32_t* frames_host_out, as_stream_t stream)
try {                                                                    // nothing may unwind through the C boundary
    if (!m || !p || !batch_ok(batch, true, true, false) || !io_ok(io, true)) return AS_EINVAL;
    Call ka(m, p, ws_a, ws_a_bytes, stream);
    const PhaseA A = forward_a(ka.c, batch, io);
    if (ka.done() || !A.tok || !A.ref) return ka.done() ? ka.done() : AS_EINVAL;
    if (!batch->frames && io->frame_cap > 0) {
        // predicted durations under a frame capacity: the second half is sized by the capacity and finds the utterances' extents on the
        // device -- no read-back, nothing here waits for the stream, the whole call is capturable
        if (io->forced_dur) return AS_EINVAL;                               // (forced durations are host data: their sums are batch->frames)
        outputs_a(ka.c, A, batch, io);
        if (ka.done()) return ka.done();
        Ctx cb(*m, *p, static_cast<hipStream_t>(stream), ws_b, ws_b_bytes, true, false);
        if ((reinterpret_cast<uintptr_t>(ws_b) & 255) != 0) return AS_EINVAL;
        forward_b(cb, A, batch, io);
        return cb.rc ? cb.rc : m->err;
    }
    as_batch b2 = *batch;
    if (!batch->frames) {                                                   // the one device -> host read of the path
        std::vector<int32_t> off(batch->B + 1);
        AS_CHECK(hipMemcpyAsync(off.data(), A.frame_off, off.size() * 4, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
        AS_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        if (as_status_peek()) return AS_EDEVICE;                            // e.g. the duration predictor's recurrence timed out
        p->frames_host.resize(batch->B);
        for (int i = 0; i < batch->B; ++i) p->frames_host[i] = off[i + 1] - off[i];
        b2.frames = p->frames_host.data();
    }
    if (frames_host_out) memcpy(frames_host_out, b2.frames, (size_t)batch->B * 4);
    outputs_a(ka.c, A, &b2, io);
    if (ka.done()) return ka.done();
    Ctx cb(*m, *p, static_cast<hipStream_t>(stream), ws_b, ws_b_bytes, true, false);
    if ((reinterpret_cast<uintptr_t>(ws_b) & 255) != 0) return AS_EINVAL;
    forward_b(cb, A, &b2, io);
    return cb.rc ? cb.rc : m->err;
} catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
} catch (...) {
    return AS_EINVAL;
}
