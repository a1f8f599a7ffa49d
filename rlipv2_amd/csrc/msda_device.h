// msda_device.h -- device-side helpers shared by the gfx950 MSDA kernels.
//
// Written for CDNA4 only: 64-lane wavefronts, DPP quad permutes, hardware float atomics.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// The few constructs a host compiler cannot take (inline assembly, dynamic LDS declarations, LDS byte addresses, LDS-DMA) go
// through these macros: tools/emu/ compiles the kernel FILES for the CPU against a lane-level model of the workgroup
// (-DMSDA_EMU, tools/emu/stub/hip/hip_runtime.h defines them there) to check the kernels' logic without a GPU.  For hipcc
// they expand to exactly what stood in the kernels before (device assembly unchanged).
#ifndef MSDA_EMU
#define MSDA_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#define MSDA_DYNAMIC_LDS_ALIGNED(type, name, n) extern __shared__ __attribute__((aligned(n))) type name[]
#define MSDA_DYNAMIC_LDS_PLAIN(type, name) extern __shared__ type name[]
#define MSDA_LDS_BYTE_ADDR(p) ((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(p))
#define MSDA_ASM_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MSDA_ASM_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define MSDA_ASM_WAIT_LGKM(v) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v))
#define MSDA_ASM_FENCE() asm volatile("" ::: "memory")
#define MSDA_ASM_OPAQUE(x) asm volatile("" : "+v"(x))
// one LDS-DMA instruction: lane i copies 16 bytes from ITS global address to lds_base (wave-uniform) + 16 i
#define MSDA_GLOBAL_LOAD_LDS16(src, lds_base)                                                                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),                          \
                                     (__attribute__((address_space(3))) void *)(lds_base), 16, 0, 0)
// where the lanes of ONE wave hand data to each other through LDS with nothing but program order in between (the wave runs
// in lock step and its LDS operations complete in order): nothing to do on the hardware, a wave barrier in the host model
#define MSDA_WAVE_LDS_SYNC() do { } while (0)
#endif

namespace msda {

typedef uint16_t bf16_t;  // raw bfloat16 storage

// ---- bfloat16 <-> float -------------------------------------------------------------------
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_float(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN kept quiet
__device__ __forceinline__ uint32_t float_to_bf16_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    const uint32_t rounded = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    const uint32_t quiet = (u >> 16) | 0x40u;
    return (u & 0x7fffffffu) > 0x7f800000u ? quiet : rounded;   // branch-free select
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return float_to_bf16_bits(lo) | (float_to_bf16_bits(hi) << 16);
}

// ---- storage-type traits: 8 consecutive channels per lane -----------------------------------
// load8 / store8 move 8 channels (32 B of f32 or 16 B of bf16) as 16-byte vectors.
template <typename VT> struct Vec8;

template <> struct Vec8<float> {
    struct raw { float4 a, b; };
    static __device__ __forceinline__ raw load_raw(const float *p)
    {
        raw r;
        r.a = *reinterpret_cast<const float4 *>(p);
        r.b = *reinterpret_cast<const float4 *>(p + 4);
        return r;
    }
    // acc[k] += w * channel k
    static __device__ __forceinline__ void fma(float w, const raw &r, float (&acc)[8])
    {
        acc[0] = fmaf(w, r.a.x, acc[0]); acc[1] = fmaf(w, r.a.y, acc[1]);
        acc[2] = fmaf(w, r.a.z, acc[2]); acc[3] = fmaf(w, r.a.w, acc[3]);
        acc[4] = fmaf(w, r.b.x, acc[4]); acc[5] = fmaf(w, r.b.y, acc[5]);
        acc[6] = fmaf(w, r.b.z, acc[6]); acc[7] = fmaf(w, r.b.w, acc[7]);
    }
    static __device__ __forceinline__ void load(const float *p, float (&v)[8])
    {
        const float4 a = *reinterpret_cast<const float4 *>(p);
        const float4 b = *reinterpret_cast<const float4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[8])
    {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

template <> struct Vec8<bf16_t> {
    typedef uint4 raw;
    static __device__ __forceinline__ raw load_raw(const bf16_t *p) { return *reinterpret_cast<const uint4 *>(p); }
    static __device__ __forceinline__ void fma(float w, const raw &r, float (&acc)[8])
    {
        acc[0] = fmaf(w, bf16_lo(r.x), acc[0]); acc[1] = fmaf(w, bf16_hi(r.x), acc[1]);
        acc[2] = fmaf(w, bf16_lo(r.y), acc[2]); acc[3] = fmaf(w, bf16_hi(r.y), acc[3]);
        acc[4] = fmaf(w, bf16_lo(r.z), acc[4]); acc[5] = fmaf(w, bf16_hi(r.z), acc[5]);
        acc[6] = fmaf(w, bf16_lo(r.w), acc[6]); acc[7] = fmaf(w, bf16_hi(r.w), acc[7]);
    }
    static __device__ __forceinline__ void load(const bf16_t *p, float (&v)[8])
    {
        const uint4 a = *reinterpret_cast<const uint4 *>(p);
        v[0] = bf16_lo(a.x); v[1] = bf16_hi(a.x);
        v[2] = bf16_lo(a.y); v[3] = bf16_hi(a.y);
        v[4] = bf16_lo(a.z); v[5] = bf16_hi(a.z);
        v[6] = bf16_lo(a.w); v[7] = bf16_hi(a.w);
    }
    static __device__ __forceinline__ void store(bf16_t *p, const float (&v)[8])
    {
        uint4 a;
        a.x = pack_bf16x2(v[0], v[1]);
        a.y = pack_bf16x2(v[2], v[3]);
        a.z = pack_bf16x2(v[4], v[5]);
        a.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4 *>(p) = a;
    }
};

// scalar element access for the generic kernels
template <typename T, typename VT> struct Elem;
template <> struct Elem<float, float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Elem<double, double> {
    static __device__ __forceinline__ double ld(const double *p) { return *p; }
    static __device__ __forceinline__ void st(double *p, double v) { *p = v; }
};
template <> struct Elem<float, bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t *p) { return bf16_to_float(*p); }
    static __device__ __forceinline__ void st(bf16_t *p, float v) { *p = (bf16_t)float_to_bf16_bits(v); }
};

// ---- DPP quad helpers (4 adjacent lanes) ----------------------------------------------------
// quad_perm control word: lane i of each quad reads lane sel_i.
#define MSDA_QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

template <int CTRL> __device__ __forceinline__ float dpp_quad(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// broadcast lane J of every quad to the whole quad
template <int J> __device__ __forceinline__ float quad_bcast(float v)
{
    return dpp_quad<MSDA_QUAD_PERM(J, J, J, J)>(v);
}
// sum over the 4 lanes of a quad, result in every lane
__device__ __forceinline__ float quad_sum(float v)
{
    v += dpp_quad<MSDA_QUAD_PERM(1, 0, 3, 2)>(v);
    v += dpp_quad<MSDA_QUAD_PERM(2, 3, 0, 1)>(v);
    return v;
}

// ---- wave64 reductions ----------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// hardware float / double atomic add, no return value needed
__device__ __forceinline__ void atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }

}  // namespace msda
