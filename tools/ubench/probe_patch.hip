// probe_patch.hip -- semantics and rates of the instructions the patch-MFMA grad_value pass (csrc/msda_patch.hip)
// and the 4x4x4-MFMA K1 are built on, measured on gfx950:
//   1. ds_read_b64_tr_b16: which element lands in which lane for per-lane addresses
//   2. v_mfma_f32_16x16x32_bf16 operand / result layout (checked against a host product)
//   3. v_mfma_f32_4x4x4_16B_bf16 operand / result layout (16 independent 4x4x4 blocks, one per DPP quad)
//   4. issue rates: v_fma_f32, v_lshlrev_b32, v_and_b32, v_mov_b32 dpp, v_pk_fma_f32, v_cvt_pk_bf16_f32, both MFMAs
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/probe_patch tools/ubench/probe_patch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- 1. transpose read -------------------------------------------------------------------------------------------
__global__ void tr_kernel(const int *addr_in, uint16_t *out)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)lds;
    const unsigned a = base + (unsigned)addr_in[threadIdx.x];
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

// ---- 2. / 3. MFMA layouts ----------------------------------------------------------------------------------------
__global__ void mfma16_kernel(const uint16_t *a, const uint16_t *b, float *d)
{
    // operands handed over per lane: a[lane][8], b[lane][8]; result d[lane][4]
    union { bf16x8 v; uint16_t h[8]; } ua, ub;
    for (int j = 0; j < 8; ++j) { ua.h[j] = a[threadIdx.x * 8 + j]; ub.h[j] = b[threadIdx.x * 8 + j]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) d[threadIdx.x * 4 + j] = c[j];
}
__global__ void mfma4_kernel(const uint16_t *a, const uint16_t *b, float *d)
{
    union { s16x4 v; uint16_t h[4]; } ua, ub;
    for (int j = 0; j < 4; ++j) { ua.h[j] = a[threadIdx.x * 4 + j]; ub.h[j] = b[threadIdx.x * 4 + j]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ua.v, ub.v, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) d[threadIdx.x * 4 + j] = c[j];
}

// ---- 4. rates ----------------------------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(int steps, float *out, unsigned long long *cycles)
{
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 2654435761u + i;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    union { bf16x8 v; s16x4 h[2]; } opa, opb;
    opa.h[0] = opa.h[1] = s16x4{1, 2, 3, 4}; opb.h[0] = opb.h[1] = s16x4{1, 2, 3, 4};
    const float k = out[0];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (OP == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(k));
            } else if (OP == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
            } else if (OP == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
            } else if (OP == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
            } else if (OP == 4) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<double *>(&x[i])) : "v"(*reinterpret_cast<const double *>(&x[(i + 2) & 6])));
            } else if (OP == 5) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7]));
            } else if (OP == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(opa.v, opb.v, acc[i], 0, 0, 0);
            } else if (OP == 7) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(opa.h[0], opb.h[0], acc[i], 0, 0, 0);
            } else if (OP == 8) {     // 4x4x4 MFMAs on ONE accumulator (dependent chain, as in a channel dot product)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(opa.h[0], opb.h[0], acc[0], 0, 0, 0);
            } else if (OP == 9) {     // v_dot2_f32_bf16
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += x[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[1 + blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int OP> void run_rate(const char *name, int per_step, int waves_per_simd, float *out, unsigned long long *cyc)
{
    const int steps = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * waves_per_simd;
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, steps, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, steps, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // per SIMD: waves_per_simd waves, each issuing steps * per_step instructions
    const double inst_per_simd = (double)steps * per_step * waves_per_simd;
    printf("%-34s waves/SIMD %d  wall %8.1f us  -> %6.2f ns/inst/SIMD (%.2f clk at 2.4 GHz)  wave0 clock %6.2f per inst\n", name,
           waves_per_simd, ms * 1e3, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4, (double)c / (steps * per_step));
}

int main()
{
    // ---- 1 ----
    {
        int h_addr[64];
        // lane p of 16-lane group g: row (p >> 2) of the group's block, 8-byte piece (p & 3); rows 64 B apart, groups 1 KB apart
        for (int l = 0; l < 64; ++l) { const int g = l >> 4, p = l & 15; h_addr[l] = g * 1024 + (p >> 2) * 64 + (p & 3) * 8; }
        int *d_addr; uint16_t *d_out; hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
        hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        uint16_t h_out[256]; hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
        printf("== ds_read_b64_tr_b16: lane p of a group reads row p>>2 (64 B apart), piece p&3; element index = byte/2\n");
        int ok = 1;
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, i = l & 15;
            for (int j = 0; j < 4; ++j) {
                const int expect = (g * 1024 + j * 64) / 2 + i;      // row j, column i of the group's [4][16] block
                if (h_out[l * 4 + j] != expect) ok = 0;
            }
        }
        printf("   expectation 'lane i gets column i, element j = row j': %s\n", ok ? "CONFIRMED" : "WRONG");
        if (!ok) for (int l = 0; l < 32; ++l) printf("   lane %2d: %5u %5u %5u %5u\n", l, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
    // ---- 2 ----
    {
        std::vector<float> A(16 * 32), B(32 * 16);
        srand(1);
        for (auto &v : A) v = (float)(rand() % 9 - 4);
        for (auto &v : B) v = (float)(rand() % 9 - 4);
        std::vector<uint16_t> ha(64 * 8), hb(64 * 8);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * (l >> 4) + j;
                ha[l * 8 + j] = f2bf(A[(l & 15) * 32 + k]);       // A[i = l % 16][k]
                hb[l * 8 + j] = f2bf(B[k * 16 + (l & 15)]);       // B[k][n = l % 16]
            }
        uint16_t *da, *db; float *dd; hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
        hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma16_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
        float hd[256]; hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * (l >> 4) + j, n = l & 15;
                float ref = 0.f;
                for (int k = 0; k < 32; ++k) ref += A[i * 32 + k] * B[k * 16 + n];
                if (fabsf(ref - hd[l * 4 + j]) > 1e-3f) ok = 0;
            }
        printf("== v_mfma_f32_16x16x32_bf16: A lane l = row l%%16, k = 8(l/16)+j; B lane l = col l%%16, same k; D lane l = col l%%16, rows 4(l/16)+j: %s\n",
               ok ? "CONFIRMED" : "WRONG");
    }
    // ---- 3 ----
    {
        std::vector<float> A(16 * 4 * 4), B(16 * 4 * 4);       // [block][i][k], [block][k][j]
        srand(2);
        for (auto &v : A) v = (float)(rand() % 9 - 4);
        for (auto &v : B) v = (float)(rand() % 9 - 4);
        std::vector<uint16_t> ha(64 * 4), hb(64 * 4);
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 4; ++k) {
                const int blk = l >> 2, r = l & 3;
                ha[l * 4 + k] = f2bf(A[(blk * 4 + r) * 4 + k]);       // lane r of the quad: row r of A
                hb[l * 4 + k] = f2bf(B[(blk * 4 + k) * 4 + r]);       // lane r of the quad: column r of B
            }
        uint16_t *da, *db; float *dd; hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 1024);
        hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma4_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
        float hd[256]; hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int blk = l >> 2, j = l & 3;
                float ref = 0.f;
                for (int k = 0; k < 4; ++k) ref += A[(blk * 4 + i) * 4 + k] * B[(blk * 4 + k) * 4 + j];
                if (fabsf(ref - hd[l * 4 + i]) > 1e-3f) ok = 0;
            }
        printf("== v_mfma_f32_4x4x4_16B_bf16: block = quad, A lane r = row r, B lane r = column r, D lane j reg i = D[i][j]: %s\n",
               ok ? "CONFIRMED" : "WRONG");
        if (!ok) {
            // try the alternative: blocks striped over lanes (block = l % 16, row = l / 16)
            for (int l = 0; l < 8; ++l) printf("   lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
            printf("   block 0 reference D:\n");
            for (int i = 0; i < 4; ++i) {
                printf("   ");
                for (int j = 0; j < 4; ++j) { float r = 0.f; for (int k = 0; k < 4; ++k) r += A[i * 4 + k] * B[k * 4 + j]; printf("%g ", r); }
                printf("\n");
            }
        }
    }
    // ---- 4 ----
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (1 + 256 * 8 * 256) * 4); hipMalloc(&cyc, 8);
    hipMemset(out, 0, 4);
    for (int w = 1; w <= 4; w *= 2) {
        run_rate<0>("v_fma_f32", 32, w, out, cyc);
        run_rate<1>("v_lshlrev_b32", 32, w, out, cyc);
        run_rate<2>("v_and_b32 (literal)", 32, w, out, cyc);
        run_rate<3>("v_mov_b32 dpp quad_perm", 32, w, out, cyc);
        run_rate<4>("v_pk_fma_f32", 16, w, out, cyc);
        run_rate<5>("v_cvt_pk_bf16_f32", 32, w, out, cyc);
        run_rate<9>("v_dot2_f32_bf16", 32, w, out, cyc);
        run_rate<6>("v_mfma_f32_16x16x32_bf16 (4 acc)", 16, w, out, cyc);
        run_rate<7>("v_mfma_f32_4x4x4_16B_bf16 (4 acc)", 16, w, out, cyc);
        run_rate<8>("v_mfma_f32_4x4x4_16B_bf16 (chain)", 16, w, out, cyc);
    }
    return 0;
}
