// Microbenchmark: LDS atomic throughput on gfx950 (profiling aid for msda_window.hip's design).
// Each block hammers an 80 KB LDS buffer with one kind of update; prints lane-updates per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kFloats = 20480;   // 80 KB

template <int MODE, int PATTERN>
__global__ __launch_bounds__(1024) void k(float *out, int iters)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < kFloats; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    unsigned rng = tid * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        int idx;
        if (PATTERN == 0) idx = (tid + it * 1024) % kFloats;                       // lane-linear, conflict-free
        else if (PATTERN == 1) { rng = rng * 1664525u + 1013904223u; idx = (rng >> 8) % kFloats; }   // random
        else {   // msda pattern: quad -> pixel p (neighbouring pixels per quad), lane -> channel 4c + sub, rotated
            const int quad = tid >> 2, sub = tid & 3;
            rng = rng * 1664525u + 1013904223u;
            const int p = (quad + (it * 37) + ((rng >> 20) & 3)) % 640;
            idx = p * 32 + ((4 * (it & 7) + 4 * (p & 7) + sub) & 31);
        }
        const float v = 1.0f + (float)(it & 3);
        if (MODE == 0) atomicAdd(&lds[idx], v);                                       // ds_add_f32
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(lds) + idx, (unsigned)(it & 3) + 1u);   // ds_add_u32
        else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long *>(lds) + (idx >> 1), 1ull);    // ds_add_u64
        else if (MODE == 3) lds[idx] += v;                                            // non-atomic read-modify-write
        else if (MODE == 4) lds[idx] = v;                                             // plain store
        else if (MODE == 5) acc += lds[idx];                                          // plain load
        else if (MODE == 6) acc += atomicAdd(&lds[idx], v);                           // ds_add_rtn_f32
        else if (MODE == 7) atomicAdd(reinterpret_cast<double *>(lds) + (idx >> 1), (double)v);          // ds_add_f64
        else if (MODE == 8) { unsigned old = atomicMax(reinterpret_cast<unsigned *>(lds) + idx, (unsigned)it); acc += old; }
        else if (MODE == 9) atomicAdd(reinterpret_cast<unsigned long long *>(lds) + (idx >> 1), (unsigned long long)(long long)(v * 4294967296.0f));
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = lds[0] + acc;
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int PATTERN> void run(const char *name, float *d_out)
{
    const int iters = 2000, blocks = 512, threads = 1024;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(threads), kFloats * 4, 0, d_out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(threads), kFloats * 4, 0, d_out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double updates = (double)blocks * threads * iters;
    const double per_cu_per_clk = updates / (ms * 1e-3) / 256.0 / 2.1e9;
    printf("%-34s pattern %d: %8.3f ms  %7.2f G lane-updates/s  %6.2f lanes/clk/CU (at 2.1 GHz)\n", name, PATTERN, ms,
           updates / (ms * 1e-3) / 1e9, per_cu_per_clk);
}

int main()
{
    float *d_out; hipMalloc(&d_out, 4096 * 4);
#define ALL(M, NAME) run<M, 0>(NAME, d_out); run<M, 1>(NAME, d_out); run<M, 2>(NAME, d_out);
    ALL(7, "ds_add_f64 (atomicAdd double)")
    ALL(9, "u64 fixed point incl. conversion")
    ALL(1, "ds_add_u32 (atomicAdd uint)")
    ALL(2, "ds_add_u64 (atomicAdd u64)")
    ALL(3, "non-atomic += (ds_read + ds_write)")
    ALL(4, "plain ds_write_b32")
    ALL(5, "plain ds_read_b32")
    return 0;
}
