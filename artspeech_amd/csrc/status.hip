// Device-side status: what a kernel reports when it cannot do what it was asked (include/artspeech_hip.h: as_device_status).
//
// The reference's operators cannot produce silently stale results: nn.Embedding raises on an id >= n_token
// (RelTransformerEnc.py:11-16), cuDNN's LSTM (models.py:555-561) and the CPU loops of S_monotonic_align.py have no
// cross-workgroup protocol that could time out.  Three kernels of this library can: the clustered H = 256 recurrence and the banded
// MAS poll words written by other workgroups with a bounded spin, and the embedding clamps ids it cannot look up.  Each of them
// raises a bit here instead of carrying on unnoticed.
//
// One word per event kind in pinned, device-mapped HOST memory, one set per HIP device of the process: a kernel raises a
// kind with a plain system-scope store (only on the failure path: nothing is written in a healthy run), the host reads it
// without a device synchronisation.  The word is sticky until cleared.  What the host sees is final once the stream that
// ran the kernel has been synchronised; before that a check is best effort (it reports failures of earlier, finished work).
#include "common.h"
#include "artspeech_hip.h"
#include <mutex>

namespace {
constexpr int MAX_DEV = 64;
struct Slot {
    unsigned* host = nullptr;   // [AS_STATUS_KINDS]
    unsigned* dev = nullptr;    // the same words as the device addresses them
};
Slot g_slot[MAX_DEV];
std::mutex g_mu;

Slot* slot_of_current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEV) return nullptr;
    Slot& s = g_slot[d];
    // fast path (every conv / embedding / LSTM launch asks): the slot exists -- `dev` is published last, with release order
    if (__atomic_load_n(&s.dev, __ATOMIC_ACQUIRE)) return &s;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!s.host) {
        void* h = nullptr;
        if (hipHostMalloc(&h, AS_STATUS_KINDS * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        for (int i = 0; i < AS_STATUS_KINDS; ++i) static_cast<volatile unsigned*>(h)[i] = 0u;
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(h);
            return nullptr;
        }
        s.host = static_cast<unsigned*>(h);
        __atomic_store_n(&s.dev, static_cast<unsigned*>(dp), __ATOMIC_RELEASE);
    }
    return &s;
}
}  // namespace

// device address of the current device's status words (kernels take it as an argument); NULL if pinned memory cannot be had --
// kernels then skip the report (a host without it still gets the bounded spins, as before)
unsigned* as_status_words_device()
{
    Slot* s = slot_of_current_device();
    return s ? s->dev : nullptr;
}

// host view: bit k set = kind k was raised since the last clear
int as_status_peek()
{
    Slot* s = slot_of_current_device();
    if (!s) return 0;
    int bits = 0;
    for (int k = 0; k < AS_STATUS_KINDS; ++k)
        if (static_cast<volatile unsigned*>(s->host)[k] != 0u) bits |= 1 << k;
    return bits;
}

extern "C" int as_device_status(int clear)
{
    Slot* s = slot_of_current_device();
    if (!s) return 0;
    const int bits = as_status_peek();
    if (clear)
        for (int k = 0; k < AS_STATUS_KINDS; ++k) static_cast<volatile unsigned*>(s->host)[k] = 0u;
    return bits;
}

// a failure the HOST side of the library found (as_lanes' debug checks): the same sticky word a kernel would have written
void as_status_raise_host(int kind)
{
    Slot* s = slot_of_current_device();
    if (s && kind >= 0 && kind < AS_STATUS_KINDS) static_cast<volatile unsigned*>(s->host)[kind] = 1u;
}

// test hook: raises `kind` from a kernel, exactly as a failing kernel would
__global__ void status_raise_kernel(unsigned* words, int kind) { as_status_raise(words, kind); }

extern "C" int as_device_status_raise_for_test(int kind, as_stream_t stream)
{
    if (kind < 0 || kind >= AS_STATUS_KINDS) return AS_EINVAL;
    unsigned* w = as_status_words_device();
    if (!w) return (int)hipErrorOutOfMemory;
    hipLaunchKernelGGL(status_raise_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, w, kind);
    AS_CHECK_LAUNCH();
    return AS_OK;
}
