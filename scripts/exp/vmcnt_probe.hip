// Does a wave with MORE than 63 vector-memory loads in flight read wrong data?  (vmcnt is a 6-bit counter.)  Round 2 met an AdaIN variant
// that kept > 60 loads in flight per thread and produced wrong values only while the conv GEMM ran beside it ("a hazard met on
// the way").  This probe: every thread issues NL independent 4-byte loads (strided far apart: every one a miss), then adds them up; a
// second kernel streams 256 MB beside it to stretch the latencies.  A wrong sum = the counter wrapped or the wait was too short.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NL>
__global__ void __launch_bounds__(256) probe(const unsigned* __restrict__ src, size_t stride, unsigned* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) v[k] = src[i + (size_t)k * stride];
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) s += v[k] * (unsigned)(k + 1);
    out[i] = s;
}
__global__ void stream_kernel(const uint4* __restrict__ p, size_t n, uint4* __restrict__ q)
{
    uint4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = p[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
    if (a.x == 0x1234567) q[0] = a;
}
template <int NL> int run(const unsigned* d, const std::vector<unsigned>& h, size_t stride, int blocks, const uint4* big, size_t nbig, uint4* sink, bool company)
{
    unsigned* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    int bad = 0;
    std::vector<unsigned> ho((size_t)blocks * 256);
    for (int rep = 0; rep < 20; ++rep) {
        if (company) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s2, big, nbig, sink);
        hipLaunchKernelGGL(probe<NL>, dim3(blocks), dim3(256), 0, s1, d, stride, out);
        hipDeviceSynchronize();
        hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < ho.size(); ++i) {
            unsigned s = 0;
            for (int k = 0; k < NL; ++k) s += h[i + (size_t)k * stride] * (unsigned)(k + 1);
            if (s != ho[i]) ++bad;
        }
    }
    printf("NL %3d %s: %d wrong sums of %zu\n", NL, company ? "beside a 256 MB stream" : "alone                 ", bad, ho.size() * 20);
    hipFree(out);
    return bad;
}
int main()
{
    const int blocks = 1024;
    const size_t stride = (size_t)blocks * 256 + 4099, n = stride * 128 + blocks * 256;
    std::vector<unsigned> h(n);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    unsigned* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    const size_t nbig = (size_t)(256 << 20) / 16;
    uint4 *big, *sink; hipMalloc(&big, nbig * 16); hipMemset(big, 1, nbig * 16); hipMalloc(&sink, 16);
    int bad = 0;
    for (int c = 0; c < 2; ++c) {
        bad += run<32>(d, h, stride, blocks, big, nbig, sink, c);
        bad += run<60>(d, h, stride, blocks, big, nbig, sink, c);
        bad += run<64>(d, h, stride, blocks, big, nbig, sink, c);
        bad += run<72>(d, h, stride, blocks, big, nbig, sink, c);
        bad += run<96>(d, h, stride, blocks, big, nbig, sink, c);
        bad += run<128>(d, h, stride, blocks, big, nbig, sink, c);
    }
    printf(bad ? "WRONG SUMS SEEN\n" : "all sums right\n");
    return 0;
}
