// Round 6: what does an in-launch split-K seam cost on this chip, against a reduction launch?
// The batch-1 step (BASELINE C2) is 43 conv GEMM + 39 reduction launches; DESIGN.md section 3.1 measured an in-launch seam at 15-25 us
// (round 3, scripts/exp/conv_gemm_sn.hip), the microarchitecture guide prices the agent-scope acquire alone at ~1.7 us and a whole
// "splitk-seam" at 5-13 us.  This probe isolates the seam at the shape of the batch-1 decoder convs (M512 N150 K1536: 8 tiles of 128 x 128,
// S K slices = 8 S workgroups): every slice "computes" for a fixed time (an ALU spin standing in for the k loop), stores its 128 x 128
// partial tile into its slab, and then
//   form 0  nothing more (the slices alone: the baseline)
//   form 1  the two-launch form of the library: a reduction kernel behind it (8 rows x 64 columns per workgroup, four slabs in flight)
//   form 2  in-launch, plain stores: vmcnt(0), barrier, agent-scope release fence, ticket; the last arriver acquires and reduces its tile
//   form 3  in-launch, write-through (sc1) stores: vmcnt(0), barrier, ticket; the last arriver reads the slabs with sc1 loads
// as hipGraphs of 100 repetitions; us per repetition.  hipcc --offload-arch=gfx950 -O3 -o /tmp/seam_probe scripts/exp/seam_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the stand-in for a slice's k loop: busy until `iters` x 100 shader-clock cycles have passed (the same in every form: an ALU loop was
// compiled differently from form to form)
__device__ __forceinline__ float spin(float v, int iters)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)iters * 100ull) __builtin_amdgcn_s_sleep(1);
    return v;
}

template <int FORM>
__global__ void __launch_bounds__(256)
slices_kernel(float* __restrict__ slab, float* __restrict__ out, unsigned* __restrict__ ticket, int M, int N, int S, int iters, unsigned epoch)
{
    const int tiles_m = M / 128, tiles = tiles_m * ((N + 127) / 128);
    const int tile = blockIdx.x % tiles, s = blockIdx.x / tiles;
    const int m0 = (tile % tiles_m) * 128, n0 = (tile / tiles_m) * 128;
    const int tid = threadIdx.x;
    // a thread owns 16 rows x 4 consecutive columns of the tile (16-byte stores, a row of the tile = 512 bytes = 32 lanes)
    const int c4 = (tid & 31) * 4, r0 = (tid >> 5) * 16;
    const float seed = spin((float)(s + 1), iters);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab + (size_t)s * M * N, 0, (int)((unsigned)M * N * 4u), 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + r0 + r, col = n0 + c4;
        const f32x4 v = {seed + row, seed + row + 1, seed + row + 2, seed + row + 3};
        // (N is a multiple of 2 here; columns past N fall out of the descriptor through the offset)
        const unsigned off = col + 3 < N ? (unsigned)(row * N + col) * 4u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, FORM == 3 ? 16 : 0);   // aux 16 = sc1 (write-through)
    }
    if (FORM < 2) return;
    __shared__ unsigned last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (FORM == 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned t = __hip_atomic_fetch_add(ticket + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == epoch * (unsigned)S + (unsigned)S - 1u) ? 1u : 0u;
        if (last && FORM == 2) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last) return;
    // the last arriver sums the tile's S partials in slice order and writes the result
    f32x4 acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < S; ++q) {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(slab + (size_t)q * M * N, 0, (int)((unsigned)M * N * 4u), 0x00020000);
        f32x4 t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + r0 + r, col = n0 + c4;
            const unsigned off = col + 3 < N ? (unsigned)(row * N + col) * 4u : 0x80000000u;
            t[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, FORM == 3 ? 16 : 0));
        }
        __builtin_amdgcn_sched_barrier(0);                  // (all sixteen loads of a slab in flight before the first add: left alone, hipcc kept three)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += t[r];
        __builtin_amdgcn_sched_barrier(0);
    }
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((unsigned)M * N * 4u), 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + r0 + r, col = n0 + c4;
        const unsigned off = col + 3 < N ? (unsigned)(row * N + col) * 4u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[r]), ro, off, 0, 0);
    }
}

// the library's reduction launch in miniature: a thread = 8 rows of one column, four slabs' loads in flight (splitk_reduce_kernel)
__global__ void __launch_bounds__(64)
reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int M, int N, int S)
{
    const int j = blockIdx.x * 64 + threadIdx.x, g = blockIdx.y;
    if (j >= N) return;
    const size_t total = (size_t)M * N;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        float t[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 8; ++r) t[u][r] = slab[(size_t)(s + u) * total + (size_t)(8 * g + r) * N + j];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] += t[u][r];
    }
    for (; s < S; ++s)
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] += slab[(size_t)s * total + (size_t)(8 * g + r) * N + j];
#pragma unroll
    for (int r = 0; r < 8; ++r) out[(size_t)(8 * g + r) * N + j] = acc[r];
}

static float run(int form, float* slab, float* out, unsigned* ticket, int M, int N, int S, int iters, int reps, hipStream_t st, bool check)
{
    const int tiles = (M / 128) * ((N + 127) / 128);
    CK(hipMemsetAsync(ticket, 0, tiles * sizeof(unsigned), st));
    CK(hipMemsetAsync(out, 0, (size_t)M * N * 4, st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) {
        if (form == 0) hipLaunchKernelGGL(slices_kernel<0>, dim3(tiles * S), dim3(256), 0, st, slab, out, ticket, M, N, S, iters, (unsigned)i);
        if (form == 1) {
            hipLaunchKernelGGL(slices_kernel<1>, dim3(tiles * S), dim3(256), 0, st, slab, out, ticket, M, N, S, iters, (unsigned)i);
            hipLaunchKernelGGL(reduce_kernel, dim3((N + 63) / 64, M / 8), dim3(64), 0, st, slab, out, M, N, S);
        }
        if (form == 2) hipLaunchKernelGGL(slices_kernel<2>, dim3(tiles * S), dim3(256), 0, st, slab, out, ticket, M, N, S, iters, (unsigned)i);
        if (form == 3) hipLaunchKernelGGL(slices_kernel<3>, dim3(tiles * S), dim3(256), 0, st, slab, out, ticket, M, N, S, iters, (unsigned)i);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int k = 0; k < 4; ++k) {
        CK(hipMemsetAsync(ticket, 0, tiles * sizeof(unsigned), st));
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (k > 0 && ms < best) best = ms;
    }
    if (check && form > 0) {
        std::vector<float> h((size_t)M * N);
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        // every slice stored seed(s) + row + (col % 4): the sum over s is known up to the spin's value, which is the same per slice index
        double bad = 0;
        for (int row = 0; row < M; row += 37)
            for (int col = 0; col + 3 < N; col += 5) {
                const double d = (double)h[(size_t)row * N + col] - (double)h[(size_t)(row > 0 ? row - 1 : row) * N + col];
                if (row > 0 && !(d > 0.99 * S && d < 1.01 * S)) bad += 1;        // rows differ by exactly S (one per slice)
            }
        if (bad > 0) printf("   form %d: %g sampled elements are NOT the sum of the %d slices\n", form, bad, S);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best * 1e3f / reps;
}

int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    const int reps = 100;
    for (int shape = 0; shape < 3; ++shape) {
        const int M = shape == 1 ? 1024 : 512, N = shape == 2 ? 400 : 150;
        for (int S : {4, 8, 16}) {
            float *slab, *out; unsigned* ticket;
            CK(hipMalloc(&slab, (size_t)S * M * N * 4)); CK(hipMalloc(&out, (size_t)M * N * 4)); CK(hipMalloc(&ticket, 4096));
            for (int iters : {50, 200}) {
                float t[4];
                for (int f = 0; f < 4; ++f) t[f] = run(f, slab, out, ticket, M, N, S, iters, reps, st, iters == 50);
                printf("M%d N%d S%d (%d workgroups), slice busy for %3d x 100 cycles: slices alone %6.2f us | + reduction launch %6.2f (+%.2f) | in-launch, plain + release/acquire %6.2f (+%.2f) | "
                       "in-launch, sc1 stores and loads %6.2f (+%.2f)\n", M, N, S, (M / 128) * ((N + 127) / 128) * S, iters, t[0], t[1], t[1] - t[0], t[2], t[2] - t[0], t[3], t[3] - t[0]);
            }
            CK(hipFree(slab)); CK(hipFree(out)); CK(hipFree(ticket));
        }
    }
    return 0;
}
