// Positive control of tests/test_emulated_tsan.py: a kernel on the host model that hands data from lane to lane through LDS with
// (sync = 1) and without (sync = 0) the barrier in between.  ThreadSanitizer must report the second and only the second.
#include <hip/hip_runtime.h>
#include "msda_device.h"
__global__ void neighbour_kernel(int *out, int sync)
{
    MSDA_DYNAMIC_LDS(int, lds);
    const int t = threadIdx.x;
    lds[t] = t * 3;
    if (sync) __syncthreads();
    out[t] = lds[(t + 1) % blockDim.x];
}
extern "C" int run(int *out, int sync)
{
    hipLaunchKernelGGL(neighbour_kernel, dim3(1), dim3(128), 512, 0, out, sync);
    return 0;
}
