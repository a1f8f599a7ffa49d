// mfma_rate.hip -- issue rate of v_mfma_f32_32x32x16_bf16 and of ds_read_b64_tr_b16 on gfx950, in the pattern of
// csrc/token_gemm.hip (8 MFMAs on 4 accumulators per step, 16 transpose reads per step, one s_barrier per step).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_rate tools/ubench/mfma_rate.hip && tools/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OFF>
__device__ __forceinline__ s16x4 lds_tr_read(unsigned addr)
{
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// MODE bit 0: MFMAs, bit 1: LDS transpose reads (conflict-free lane-linear addresses), bit 2: barrier per step
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(int steps, float *out, unsigned long long *cycles)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned addr = lds0 + lane * 8 + (threadIdx.x >> 6) * 4096;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    union { s16x4 h[2]; bf16x8 v; } a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i].h[0] = a[i].h[1] = s16x4{1, 2, 3, 4}; b[i].h[0] = b[i].h[1] = s16x4{1, 2, 3, 4}; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        if (MODE & 4) __builtin_amdgcn_s_barrier();
        for (int half = 0; half < 2; ++half) {
            if (MODE & 2) {
                s16x4 r[8];
                r[0] = lds_tr_read<0>(addr); r[1] = lds_tr_read<512>(addr); r[2] = lds_tr_read<1024>(addr); r[3] = lds_tr_read<1536>(addr);
                r[4] = lds_tr_read<2048>(addr); r[5] = lds_tr_read<2560>(addr); r[6] = lds_tr_read<3072>(addr); r[7] = lds_tr_read<3584>(addr);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
                a[0].h[0] = r[0]; a[0].h[1] = r[1]; a[1].h[0] = r[2]; a[1].h[1] = r[3];
                b[0].h[0] = r[4]; b[0].h[1] = r[5]; b[1].h[0] = r[6]; b[1].h[1] = r[7];
            }
            if (MODE & 1) {
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i * 2 + j], 0, 0, 0);
            } else {
                for (int i = 0; i < 4; ++i) acc[i][0] += (float)(a[i >> 1].h[0][0] ^ b[i & 1].h[1][1]);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int MODE> void run(const char *name, int blocks_per_cu, int steps, float *out, unsigned long long *cyc)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 16384, 0, steps, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 16384, 0, steps, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double us = ms * 1e3, per_step_ns = us * 1e3 / steps;
    const double tflops = (MODE & 1) ? (double)grid * 4 * steps * 8 * 32768.0 / (us * 1e-6) / 1e12 : 0.0;
    printf("%-28s WG/CU %d  %8.1f us  %7.1f ns/step  clock64 %6.1f /step  %7.1f TFLOP/s\n", name, blocks_per_cu, us, per_step_ns,
           (double)c / steps, tflops);
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 8);
    const int steps = 2000;
    for (int b = 1; b <= 4; ++b) {
        run<1>("mfma", b, steps, out, cyc);
        run<5>("mfma + barrier", b, steps, out, cyc);
        run<2>("tr reads", b, steps, out, cyc);
        run<6>("tr reads + barrier", b, steps, out, cyc);
        run<3>("reads + mfma", b, steps, out, cyc);
        run<7>("reads + mfma + barrier", b, steps, out, cyc);
    }
    return 0;
}
