// Microbenchmark: global (L2) atomic throughput on gfx950 for the access patterns the MSDA
// backward can produce.  Profiling aid for msda_window.hip's design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// buffer of `rows` rows of 32 floats (128 B); every "update" adds into row r (pseudo-random)
template <int PATTERN, int KIND>
__global__ __launch_bounds__(256) void k(float *buf, int rows, int iters)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned rng = (tid / 64) * 2654435761u + 777u;      // wave-uniform stream
    for (int it = 0; it < iters; ++it) {
        rng = rng * 1664525u + 1013904223u;
        float *p = nullptr;
        bool active = true;
        if (PATTERN == 0) {            // 64 lanes, 2 random rows x 32 contiguous floats (256 B per instr, 8 sectors)
            const unsigned r = ((rng >> 8) + (lane >> 5) * 7919u) % rows;
            p = buf + (size_t)r * 32 + (lane & 31);
        } else if (PATTERN == 1) {     // 32 active lanes, 1 random row (generic kernel: 4 sectors per instr)
            const unsigned r = (rng >> 8) % rows;
            p = buf + (size_t)r * 32 + (lane & 31);
            active = lane < 32;
        } else if (PATTERN == 2) {     // 8 groups x 8 lanes, each group one sector of a different random row
            const unsigned r = ((rng >> 8) + (lane >> 3) * 7919u) % rows;
            p = buf + (size_t)r * 32 + (it & 3) * 8 + (lane & 7);
        } else if (PATTERN == 3) {     // only 8 active lanes: one sector of one row
            const unsigned r = (rng >> 8) % rows;
            p = buf + (size_t)r * 32 + (it & 3) * 8 + (lane & 7);
            active = lane < 8;
        } else if (PATTERN == 4) {     // 16 quads x 4 lanes, stride 32 B inside the quad (quad kernel: 64 sectors)
            const unsigned r = ((rng >> 8) + (lane >> 2) * 7919u) % rows;
            p = buf + (size_t)r * 32 + (lane & 3) * 8 + (it & 7);
        } else if (PATTERN == 5) {     // 64 lanes fully contiguous, sequential rows (streaming flush)
            const unsigned r = ((unsigned)(tid / 64) * iters + it) * 2 % rows;
            p = buf + (size_t)r * 32 + lane;
        }
        if (active) {
            if (KIND == 0) unsafeAtomicAdd(p, 1.0f);
            else if (KIND == 1) atomicAdd(reinterpret_cast<unsigned *>(p), 1u);
            else if (KIND == 2) *p = 1.0f;                               // plain store (reference point)
            else if (KIND == 3) atomicAdd(p, 1.0f);                     // safe atomicAdd (may be a CAS loop)
        }
    }
}

template <int PATTERN, int KIND> void run(const char *name, float *buf, int rows)
{
    const int blocks = 256 * 8, threads = 256, iters = 400;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<PATTERN, KIND>), dim3(blocks), dim3(threads), 0, 0, buf, rows, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<PATTERN, KIND>), dim3(blocks), dim3(threads), 0, 0, buf, rows, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instrs = (double)blocks * threads / 64 * iters;
    static const int sect[6] = {8, 4, 8, 1, 64, 8};
    printf("%-26s %-14s %8.3f ms  %7.2f G wave-instr/s  %8.2f G sectors/s  clk/instr/CU %7.1f\n", name,
           KIND == 0 ? "f32 atomic" : KIND == 1 ? "u32 atomic" : KIND == 2 ? "plain store" : "safe f32", ms,
           instrs / (ms * 1e-3) / 1e9, instrs * sect[PATTERN] / (ms * 1e-3) / 1e9,
           (ms * 1e-3) * 2.1e9 * 256 / instrs);
}

int main()
{
    const int rows = 22223 * 8 * 4;     // batch-4 encoder grad_value: 711 K rows of 128 B = 91 MB
    float *buf; (void)hipMalloc(&buf, (size_t)rows * 128);
    (void)hipMemset(buf, 0, (size_t)rows * 128);
#define ALLK(P, NAME) run<P, 0>(NAME, buf, rows); run<P, 1>(NAME, buf, rows); run<P, 2>(NAME, buf, rows);
    ALLK(0, "64 lanes: 2 rows x 128 B")
    ALLK(1, "32 lanes: 1 row x 128 B")
    ALLK(2, "8 groups x 32 B")
    ALLK(3, "8 lanes: 1 x 32 B")
    ALLK(4, "16 quads strided (64 sect)")
    ALLK(5, "64 lanes sequential 256 B")
    run<1, 3>("32 lanes: 1 row x 128 B", buf, rows);
    return 0;
}
