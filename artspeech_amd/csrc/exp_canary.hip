// Experiment builds only (scripts/build_exp.sh NAME -DAS_EXP_CANARY): a kernel that fills its own LDS with a pattern,
// waits, and counts the words that changed -- run beside another kernel to see whether that one writes foreign LDS.
#ifdef AS_EXP_CANARY
#include "common.h"
__global__ void __launch_bounds__(256) lds_canary_kernel(int words, int spins, unsigned* bad, unsigned* first_bad)
{
    extern __shared__ unsigned cs[];
    for (int i = threadIdx.x; i < words; i += 256) cs[i] = 0xC0DE0000u ^ (unsigned)i;
    __syncthreads();
    unsigned long long t0 = clock64();
    while (clock64() - t0 < (unsigned long long)spins) {}
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += 256)
        if (cs[i] != (0xC0DE0000u ^ (unsigned)i)) {
            atomicAdd(bad, 1u);
            atomicMin(first_bad, (unsigned)i);
        }
}
extern "C" int as_exp_lds_canary(int bytes, int blocks, int spins, unsigned* bad, unsigned* first_bad, void* stream)
{
    static bool set = false;
    if (!set) {
        AS_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        set = true;
    }
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, bytes / 4, spins, bad, first_bad);
    AS_CHECK_LAUNCH();
    return 0;
}
#endif
