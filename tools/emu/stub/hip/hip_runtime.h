// Stand-in for <hip/hip_runtime.h> when a kernel FILE of rlipv2_amd/csrc is compiled for the CPU (-DMSDA_EMU,
// -I tools/emu/stub): a lane-level model of a gfx950 workgroup.  Test infrastructure, not product.
//
// One host thread per lane.  Workgroups run one after the other, so `__shared__` is `static` and the dynamic LDS is one
// arena.  Cross-lane operations are rendezvous of the lanes that take part -- a quad for quad_perm DPP moves and the 4x4x4
// MFMA (block = quad), a 16-lane row for row shifts and the transposing LDS read, the wave for everything else -- so a
// kernel may run them in control flow that is uniform only at that granularity (e.g. "a quad is live or dead as a whole").
// Semantics as measured on the hardware (profiles/r03_probe_mfma_tr_rates.txt):
//   update_dpp      quad_perm 0x00-0xff; row_shr:n 0x111-0x11f; row_newbcast:n 0x150-0x15f; row_bcast15 0x142; row_bcast31 0x143;
//                   row_mask / bank_mask;
//                   bound_ctrl: an invalid source gives 0 instead of `old`
//   ds_read_b64_tr_b16   lane p of a 16-lane group ADDRESSES row p >> 2, 8-byte piece p & 3; lane i RECEIVES column i, 4 rows
//   v_mfma_f32_4x4x4_16B_bf16    block = 4 lanes; A lane r = row r (4 k), B lane j = column j, D lane j reg i = D[i][j]
//   v_mfma_f32_16x16x32_bf16     A lane l = row l % 16, k = 8 (l / 16) + e; B lane l = column l % 16, same k; D lane l holds
//                                column l % 16, rows 4 (l / 16) + i in register i
//   raw_buffer_load_b128         offset >= num_records returns zeros
#pragma once
#include <atomic>
#include <barrier>
#include <memory>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace emu {

struct Barrier {                                   // std::barrier (futex-based, no thundering-herd mutex) behind reset()
    std::unique_ptr<std::barrier<>> b;
    void reset(int n) { b = std::make_unique<std::barrier<>>(n); }
    void wait() { b->arrive_and_wait(); }
};

constexpr int kMaxThreads = 1024, kLdsBytes = 160 * 1024;

struct Wave {
    Barrier quad[16], row[4], half[2], all;
    uint64_t slot[4][2][64][8];                     // [scope][parity][lane]: see publish()
    Wave() { for (auto &b : quad) b.reset(4); for (auto &b : row) b.reset(16); for (auto &b : half) b.reset(32); all.reset(64); }
};

struct Group {
    // (guards: lanes whose results are discarded -- the dead quads of a ragged cell -- may compute LDS addresses outside the
    //  allocation; the hardware returns zeros / garbage for them without a fault, here they must not leave the process)
    alignas(16) unsigned char guard_lo[16 << 20];
    alignas(16) unsigned char lds[kLdsBytes];
    alignas(16) unsigned char guard_hi[16 << 20];
    Barrier bar;
    Wave waves[kMaxThreads / 64];
    int block_idx = 0, block_idx_y = 0, block_dim = 0, grid_dim = 0, grid_dim_y = 1;
};

inline Group *&group() { static Group *g = new Group; return g; }
inline int &tid_ref() { thread_local int t = 0; return t; }
inline int lane() { return tid_ref() & 63; }
inline Wave &wave() { return group()->waves[tid_ref() >> 6]; }

enum Scope { QUAD, ROW, HALF, WAVE };
inline Barrier &barrier_of(Scope s)
{
    Wave &w = wave();
    return s == QUAD ? w.quad[lane() >> 2] : s == ROW ? w.row[lane() >> 4] : s == HALF ? w.half[lane() >> 5] : w.all;
}
// A collective = every lane of the scope's group publishes its operands, ONE rendezvous, every lane reads what it needs.  The
// slots are double-buffered per scope: the next collective of the same scope writes the other buffer, and the one after that can
// only be reached through the next rendezvous of the same group -- which every lane enters after it has finished reading this
// one.  (One barrier per collective instead of two: these waits are where the host model spends its time.)
inline int &parity_ref(Scope s) { thread_local int p[4] = {0, 0, 0, 0}; return p[s]; }
inline uint64_t (*cur(Scope s))[8] { return wave().slot[s][parity_ref(s)]; }
inline uint64_t *publish(Scope s, uint64_t a, uint64_t b = 0, uint64_t c = 0, uint64_t d = 0)
{
    uint64_t *sl = cur(s)[lane()];
    sl[0] = a; sl[1] = b; sl[2] = c; sl[3] = d;
    barrier_of(s).wait();
    return sl;
}
inline void done(Scope s) { parity_ref(s) ^= 1; }

inline int update_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    const Scope s = ctrl <= 0xff ? QUAD : ((ctrl >= 0x111 && ctrl <= 0x11f) || (ctrl >= 0x150 && ctrl <= 0x15f)) ? ROW : WAVE;
    publish(s, (uint32_t)v);
    const int l = lane(), row = l >> 4, in_row = l & 15;
    int src = -1;
    if (ctrl >= 0 && ctrl <= 0xff) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; src = in_row >= n ? l - n : -1; }
    else if (ctrl >= 0x150 && ctrl <= 0x15f) src = (l & ~15) | (ctrl - 0x150);          // row_newbcast:n
    else if (ctrl == 0x142) src = row >= 1 ? row * 16 - 1 : -1;
    else if (ctrl == 0x143) src = row >= 2 ? 31 : -1;
    else { std::fprintf(stderr, "emu: DPP control %#x not modelled\n", ctrl); std::abort(); }
    int r = old;
    const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> (in_row >> 2)) & 1);
    if (enabled) r = src >= 0 ? (int)(uint32_t)cur(s)[src][0] : (bound_ctrl ? 0 : old);
    done(s);
    return r;
}
inline int readfirstlane(int v)
{
    publish(WAVE, (uint32_t)v);
    const int r = (int)(uint32_t)cur(WAVE)[0][0];
    done(WAVE);
    return r;
}
inline int readlane(int v, int l)
{
    publish(WAVE, (uint32_t)v);
    const int r = (int)(uint32_t)cur(WAVE)[l & 63][0];
    done(WAVE);
    return r;
}
inline unsigned long long ballot(bool p)
{
    publish(WAVE, p ? 1 : 0);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (unsigned long long)(cur(WAVE)[i][0] & 1) << i;
    done(WAVE);
    return m;
}
template <typename T> inline T shfl_xor(T v, int mask)
{
    // a butterfly that stays inside a quad / row / half-wave only needs those lanes (kernels run it in control flow that is
    // uniform per half-wave: two rows of a LayerNorm per wave with different trip counts)
    const Scope s = mask < 4 ? QUAD : mask < 16 ? ROW : mask < 32 ? HALF : WAVE;
    uint64_t u = 0;
    std::memcpy(&u, &v, sizeof(T));
    publish(s, u);
    const uint64_t o = cur(s)[lane() ^ mask][0];
    T r;
    std::memcpy(&r, &o, sizeof(T));
    done(s);
    return r;
}
inline unsigned char *lds_ptr(unsigned addr)
{
    if (addr >= (unsigned)kLdsBytes) { std::fprintf(stderr, "emu: LDS address %u out of range\n", addr); std::abort(); }
    return group()->lds + addr;
}
inline float bf16f(uint32_t bits16) { const uint32_t u = bits16 << 16; float f; std::memcpy(&f, &u, 4); return f; }

// launch: every block in turn, block_dim host threads each
template <typename F> inline void launch(unsigned grid, unsigned grid_y, unsigned block, F body)
{
    Group *g = group();
    if (block > (unsigned)kMaxThreads) { std::fprintf(stderr, "emu: block of %u threads\n", block); std::abort(); }
    for (unsigned by = 0; by < grid_y; ++by)
    for (unsigned b = 0; b < grid; ++b) {
        std::memset(g->lds, 0xa5, sizeof(g->lds));                 // LDS starts as garbage
        g->block_idx = (int)b; g->block_idx_y = (int)by; g->block_dim = (int)block; g->grid_dim = (int)grid;
        g->grid_dim_y = (int)grid_y;
        g->bar.reset((int)block);
        std::vector<std::thread> th;
        th.reserve(block);
        for (unsigned t = 0; t < block; ++t) th.emplace_back([&, t] { tid_ref() = (int)t; body(); });
        for (auto &x : th) x.join();
    }
}

}  // namespace emu

// ---- language ---------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define MSDA_DYNAMIC_LDS(type, name) type *name = reinterpret_cast<type *>(emu::group()->lds)
#define MSDA_DYNAMIC_LDS_ALIGNED(type, name, n) MSDA_DYNAMIC_LDS(type, name)
#define MSDA_DYNAMIC_LDS_PLAIN(type, name) MSDA_DYNAMIC_LDS(type, name)
#define MSDA_ASM_OPAQUE(x) do { } while (0)
#define MSDA_GLOBAL_LOAD_LDS16(src, lds_base)                                                                        \
    std::memcpy(emu::lds_ptr(MSDA_LDS_BYTE_ADDR(lds_base) + 16u * (unsigned)emu::lane()), (src), 16)
#define MSDA_LDS_BYTE_ADDR(p) ((unsigned)((const unsigned char *)(p) - emu::group()->lds))
#define MSDA_ASM_WAIT_VM() do { } while (0)
#define MSDA_ASM_WAIT_VMCNT(n) do { } while (0)
#define MSDA_ASM_WAIT_LGKM(v) do { } while (0)
#define __builtin_amdgcn_s_barrier() emu::group()->bar.wait()
#define MSDA_ASM_FENCE() do { } while (0)
#define MSDA_WAVE_LDS_SYNC() emu::barrier_of(emu::WAVE).wait()

struct emu_idx { int x, y, z; };
#define threadIdx (emu_idx{emu::tid_ref(), 0, 0})
#define blockIdx (emu_idx{emu::group()->block_idx, emu::group()->block_idx_y, 0})
#define blockDim (emu_idx{emu::group()->block_dim, 1, 1})
#define gridDim (emu_idx{emu::group()->grid_dim, emu::group()->grid_dim_y, 1})
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
typedef void *hipStream_t;
typedef int hipError_t;
constexpr int hipSuccess = 0;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 0;
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
#define HIP_SYMBOL(x) x
template <typename T> inline hipError_t hipMemcpyToSymbol(T &sym, const void *src, size_t n) { std::memcpy(&sym, src, n); return 0; }
template <typename T> inline hipError_t hipMemcpyFromSymbol(void *dst, const T &sym, size_t n) { std::memcpy(dst, &sym, n); return 0; }
#define hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, ...)                                              \
    emu::launch(dim3(grid).x, dim3(grid).y, dim3(block).x, [&] { kernel(__VA_ARGS__); })

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
inline int2 make_int2(int x, int y) { return {x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
struct alignas(16) uint4 { uint32_t x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }

// v_med3_f32: the median; with a NaN among the inputs the minimum of the others (what the clamp modifier it compiles to gives)
inline float __builtin_amdgcn_fmed3f(float a, float b, float c)
{
    if (a != a || b != b || c != c) return std::fmin(std::fmin(a, b), c);
    return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c));
}
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned)((uint64_t)(a & 0xffffffu) * (b & 0xffffffu)); }
inline int __mul24(int a, int b) { return (int)((int64_t)((a << 8) >> 8) * ((b << 8) >> 8)); }
inline int __popc(uint32_t v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(uint32_t v) { return __builtin_ffs((int)v); }
inline unsigned long long clock64() { return 0; }
template <typename T> inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> inline T max(T a, T b) { return a > b ? a : b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }
inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
inline float __expf(float x) { return expf(x); }      // (the device's fast exp: a few ulp apart; tolerances of the tests cover it)
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long min(long a, long b) { return a < b ? a : b; }
inline long max(long a, long b) { return a > b ? a : b; }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline void __syncthreads() { emu::group()->bar.wait(); }
template <typename T> inline T __shfl_xor(T v, int mask, int = 64) { return emu::shfl_xor(v, mask); }
template <typename T> inline T emu_shfl_from(T v, int src)      // every lane names its source lane (out of range: itself)
{
    uint64_t u = 0;
    std::memcpy(&u, &v, sizeof(T));
    emu::publish(emu::WAVE, u);
    const uint64_t o = emu::cur(emu::WAVE)[(src >= 0 && src < 64) ? src : emu::lane()][0];
    T r;
    std::memcpy(&r, &o, sizeof(T));
    emu::done(emu::WAVE);
    return r;
}
template <typename T> inline T __shfl(T v, int src, int = 64) { return emu_shfl_from(v, src & 63); }
template <typename T> inline T __shfl_up(T v, unsigned d, int = 64) { return emu_shfl_from(v, emu::lane() - (int)d); }
template <typename T> inline T __shfl_down(T v, unsigned d, int = 64) { return emu_shfl_from(v, emu::lane() + (int)d); }

template <typename T> inline T emu_atomic_rmw(T *p, T v, T (*op)(T, T))
{
    std::atomic_ref<T> r(*p);
    T cur = r.load();
    while (!r.compare_exchange_weak(cur, op(cur, v))) {}
    return cur;
}
inline int atomicMin(int *p, int v) { return emu_atomic_rmw<int>(p, v, [](int a, int b) { return a < b ? a : b; }); }
inline int atomicMax(int *p, int v) { return emu_atomic_rmw<int>(p, v, [](int a, int b) { return a > b ? a : b; }); }
inline int atomicOr(int *p, int v) { return emu_atomic_rmw<int>(p, v, [](int a, int b) { return a | b; }); }
inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return emu_atomic_rmw<uint32_t>(p, v, [](uint32_t a, uint32_t b) { return a | b; }); }
inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return emu_atomic_rmw<uint32_t>(p, v, [](uint32_t a, uint32_t b) { return a + b; }); }
inline uint32_t atomicMax(uint32_t *p, uint32_t v) { return emu_atomic_rmw<uint32_t>(p, v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }
inline uint32_t atomicMin(uint32_t *p, uint32_t v) { return emu_atomic_rmw<uint32_t>(p, v, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
inline int atomicAdd(int *p, int v) { return emu_atomic_rmw<int>(p, v, [](int a, int b) { return a + b; }); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v)
{
    return emu_atomic_rmw<unsigned long long>(p, v, [](unsigned long long a, unsigned long long b) { return a + b; });
}
inline float atomicAdd(float *p, float v) { return emu_atomic_rmw<float>(p, v, [](float a, float b) { return a + b; }); }
inline void unsafeAtomicAdd(float *p, float v) { atomicAdd(p, v); }
inline void unsafeAtomicAdd(double *p, double v) { emu_atomic_rmw<double>(p, v, [](double a, double b) { return a + b; }); }

// ---- builtins ---------------------------------------------------------------------------------------------------------
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
struct __amdgpu_buffer_rsrc_t { const unsigned char *base; uint32_t bytes; };
inline __amdgpu_buffer_rsrc_t emu_make_rsrc(void *p, int, uint32_t bytes, int) { return {(const unsigned char *)p, bytes}; }
inline emu_u32x4 emu_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned off, int, int)
{
    emu_u32x4 v = {0u, 0u, 0u, 0u};
    if ((uint64_t)off + 16 <= r.bytes) std::memcpy(&v, r.base + off, 16);
    return v;
}
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, bytes, flags) emu_make_rsrc(p, stride, bytes, flags)
#define __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, aux) emu_buffer_load_b128(r, off, soff, aux)
#define __builtin_amdgcn_readfirstlane(x) emu::readfirstlane(x)
#define __builtin_amdgcn_readlane(x, l) emu::readlane(x, l)
#define __builtin_amdgcn_ballot_w64(p) emu::ballot(p)
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) emu::update_dpp(old, v, ctrl, rm, bm, bc)
#define __builtin_amdgcn_sched_barrier(x) do { } while (0)
#define __builtin_amdgcn_wave_barrier() emu::barrier_of(emu::WAVE).wait()

inline emu_s16x4 emu_tr_read(unsigned addr, unsigned offset)
{
    emu::publish(emu::ROW, addr);
    const int l = emu::lane(), base = l & ~15, i = l & 15;
    emu_s16x4 v;
    for (int e = 0; e < 4; ++e) {
        const unsigned a = (unsigned)emu::cur(emu::ROW)[base + 4 * e + (i >> 2)][0] + offset + (unsigned)(i & 3) * 2u;
        short x;
        std::memcpy(&x, emu::lds_ptr(a), 2);
        v[e] = x;
    }
    emu::done(emu::ROW);
    return v;
}
inline emu_f32x4 emu_mfma444(emu_s16x4 a, emu_s16x4 b, emu_f32x4 c)
{
    uint64_t ua, ub;
    std::memcpy(&ua, &a, 8); std::memcpy(&ub, &b, 8);
    emu::publish(emu::QUAD, ua, ub);
    const int l = emu::lane(), blk = l & ~3, j = l & 3;
    const uint64_t bj = emu::cur(emu::QUAD)[blk + j][1];
    emu_f32x4 d = c;
    for (int i = 0; i < 4; ++i) {
        const uint64_t ai = emu::cur(emu::QUAD)[blk + i][0];
        float acc = c[i];
        for (int k = 0; k < 4; ++k) acc += emu::bf16f((uint32_t)((ai >> (16 * k)) & 0xffff)) * emu::bf16f((uint32_t)((bj >> (16 * k)) & 0xffff));
        d[i] = acc;
    }
    emu::done(emu::QUAD);
    return d;
}
#define __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, x, y, z) emu_mfma444(a, b, c)

// 16x16x32: operands are 8 bf16 per lane (any 16-byte vector type)
template <typename V> inline emu_f32x4 emu_mfma16(V a, V b, emu_f32x4 c)
{
    static_assert(sizeof(V) == 16, "8 bfloat16 per lane");
    uint64_t ua[2], ub[2];
    std::memcpy(ua, &a, 16); std::memcpy(ub, &b, 16);
    emu::publish(emu::WAVE, ua[0], ua[1], ub[0], ub[1]);
    const int l = emu::lane(), col = l & 15, rg = l >> 4;
    emu_f32x4 d = c;
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * rg + i;
        float acc = c[i];
        for (int kg = 0; kg < 4; ++kg) {                         // k = 8 kg + e: A from lane (row, kg), B from lane (col, kg)
            const uint64_t *sa = emu::cur(emu::WAVE)[kg * 16 + row], *sb = emu::cur(emu::WAVE)[kg * 16 + col];
            for (int e = 0; e < 8; ++e) {
                const uint32_t x = (uint32_t)((sa[e >> 2] >> (16 * (e & 3))) & 0xffff);
                const uint32_t y = (uint32_t)((sb[2 + (e >> 2)] >> (16 * (e & 3))) & 0xffff);
                acc += emu::bf16f(x) * emu::bf16f(y);
            }
        }
        d[i] = acc;
    }
    emu::done(emu::WAVE);
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma16(a, b, c)

// the two inline-assembly helpers of msda_patch.hip
inline emu_s16x4 lds_tr_read(unsigned addr) { return emu_tr_read(addr, 0); }
inline emu_s16x4 lds_tr_read32(unsigned addr) { return emu_tr_read(addr, 32); }

// 32x32x16: A lane l = row l % 32, k = 8 (l / 32) + e; B lane l = column l % 32, same k; D: lane l holds column l % 32,
// register r = row (r & 3) + 8 (r >> 2) + 4 (l / 32)   (the layout alif_attention.hip documents and a GPU has validated)
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
template <typename V> inline emu_f32x16 emu_mfma32(V a, V b, emu_f32x16 c)
{
    static_assert(sizeof(V) == 16, "8 bfloat16 per lane");
    uint64_t ua[2], ub[2];
    std::memcpy(ua, &a, 16); std::memcpy(ub, &b, 16);
    emu::publish(emu::WAVE, ua[0], ua[1], ub[0], ub[1]);
    const int l = emu::lane(), col = l & 31, hi = l >> 5;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kg = 0; kg < 2; ++kg) {
            const uint64_t *sa = emu::cur(emu::WAVE)[kg * 32 + row], *sb = emu::cur(emu::WAVE)[kg * 32 + col];
            for (int e = 0; e < 8; ++e) {
                const uint32_t x = (uint32_t)((sa[e >> 2] >> (16 * (e & 3))) & 0xffff);
                const uint32_t y = (uint32_t)((sb[2 + (e >> 2)] >> (16 * (e & 3))) & 0xffff);
                acc += emu::bf16f(x) * emu::bf16f(y);
            }
        }
        d[r] = acc;
    }
    emu::done(emu::WAVE);
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma32(a, b, c)
