// cell_forward_emu.cpp -- runs the DEVICE SOURCE of cell_forward_kernel (rlipv2_amd/csrc/msda_cell_forward.inc) on the host
// against a lane-level model of a gfx950 workgroup.  Test infrastructure (tests/test_cell_forward_emulated.py builds and
// runs it); nothing here is part of the product.
//
// Model: one host thread per lane, 512 per workgroup; LDS is a byte array shared by the workgroup's threads (addresses =
// offsets into it); __syncthreads is a barrier; every cross-lane operation of the kernel (DPP moves, readfirstlane, the
// transposing LDS read, the 4x4x4 MFMA) executes in wave-uniform control flow in this kernel, so it is modelled as a
// rendezvous of the wave's 64 threads: everybody publishes its operands, waits, and computes its own result from the
// published ones with the semantics measured on the hardware (profiles/r03_probe_mfma_tr_rates.txt):
//   update_dpp   quad_perm 0x00-0xff, row_shr:n 0x111-0x11f, row_bcast15 0x142, row_bcast31 0x143; row_mask, bound_ctrl = 0
//                (a lane whose source is out of range or whose row is masked keeps `old`)
//   ds_read_b64_tr_b16   16-lane group: lane p ADDRESSES row p >> 2, 8-byte piece p & 3; lane i RECEIVES column i (4 rows)
//   v_mfma_f32_4x4x4_16B_bf16   block = 4 lanes; A: lane r = row r (4 k), B: lane j = column j, D: lane j reg i = D[i][j];
//                products of two bfloat16 are exact in float32, the 4 products of a row are added to the accumulator in
//                order k = 0..3 (the hardware's internal order is not documented; the tolerance of the test covers it)
// usage: cell_forward_emu <problem.bin> <out.bin>      (format: see main)
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace emu {

struct Barrier {                                   // reusable barrier (C++17: no std::barrier)
    std::mutex m;
    std::condition_variable cv;
    int n, count = 0, gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const int g = gen;
        if (++count == n) { count = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

constexpr int kThreads = 512, kLdsBytes = 160 * 1024, kStaticBytes = 4096;

struct Wave {
    Barrier bar{64};
    uint64_t slot[64][4];                          // published operands of the current collective
};

struct Group {                                     // one workgroup
    alignas(16) unsigned char lds[kLdsBytes];      // [static __shared__ | dynamic]
    Barrier bar{kThreads};
    Wave waves[kThreads / 64];
    int block_idx;
};

thread_local Group *g_group;
thread_local int g_tid;

inline int lane() { return g_tid & 63; }
inline Wave &wave() { return g_group->waves[g_tid >> 6]; }

// all-lane exchange: publish up to 4 words, return after every lane of the wave has published; the caller reads the
// others' slots and then calls done() so that nobody overwrites a slot that is still being read
inline void publish(uint64_t a, uint64_t b = 0, uint64_t c = 0, uint64_t d = 0)
{
    Wave &w = wave();
    uint64_t *s = w.slot[lane()];
    s[0] = a; s[1] = b; s[2] = c; s[3] = d;
    w.bar.wait();
}
inline void done() { wave().bar.wait(); }

inline int readfirstlane(int v)
{
    publish((uint32_t)v);
    const int r = (int)(uint32_t)wave().slot[0][0];
    done();
    return r;
}

inline int update_dpp(int old, int v, int ctrl, int row_mask)
{
    publish((uint32_t)v);
    const int l = lane(), row = l >> 4, in_row = l & 15;
    int src = -1;                                  // source lane, -1: invalid (keep old)
    if (ctrl >= 0 && ctrl <= 0xff) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; src = in_row >= n ? l - n : -1; }
    else if (ctrl == 0x142) src = row >= 1 ? row * 16 - 1 : -1;       // row_bcast15: lane 15 of the previous row
    else if (ctrl == 0x143) src = row >= 2 ? 31 : -1;                 // row_bcast31: lane 31 to rows 2 and 3
    else { std::fprintf(stderr, "emu: DPP control %#x not modelled\n", ctrl); std::abort(); }
    int r = old;
    if (((row_mask >> row) & 1) && src >= 0) r = (int)(uint32_t)wave().slot[src][0];
    done();
    return r;
}

}  // namespace emu

// ---- what the include file expects from its surroundings ------------------------------------------------------------------
typedef unsigned short bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct float2 { float x, y; };
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
static inline float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
static inline uint32_t rne_bf16(float f)          // v_cvt_pk_bf16_f32: round to nearest even
{
    const uint32_t u = float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline uint32_t cvt_pk_bf16(float a, float b) { return rne_bf16(a) | (rne_bf16(b) << 16); }
static inline void split_pair(float e0, float e1, uint32_t &hi, uint32_t &lo)       // as in msda_patch.hip
{
    hi = cvt_pk_bf16(e0, e1);
    lo = cvt_pk_bf16(e0 - bf16_lo(hi), e1 - bf16_hi(hi));
}

constexpr int kL = 4, kP = 4, kD = 32, kCellQ = 340, kCellThreads = 512;
struct PatchPlan { int H[kL], W[kL]; int CY, CX; };      // (the fields the forward kernel reads)

static inline s16x4 tr_read(unsigned addr, int offset)
{
    emu::publish(addr);
    const int l = emu::lane(), base = l & ~15, i = l & 15;
    s16x4 v;
    for (int e = 0; e < 4; ++e) {                  // row e is addressed by lanes 4 e .. 4 e + 3 of the group, 8 bytes each
        const unsigned a = (unsigned)emu::wave().slot[base + 4 * e + (i >> 2)][0] + (unsigned)offset + (unsigned)(i & 3) * 2u;
        if (a + 2 > (unsigned)emu::kLdsBytes) { std::fprintf(stderr, "emu: transposing read outside LDS (%u)\n", a); std::abort(); }
        short x;
        std::memcpy(&x, emu::g_group->lds + a, 2);
        v[e] = x;
    }
    emu::done();
    return v;
}

static inline f32x4 mfma444(s16x4 a, s16x4 b, f32x4 c)
{
    uint64_t ua, ub;
    std::memcpy(&ua, &a, 8); std::memcpy(&ub, &b, 8);
    emu::publish(ua, ub);
    const int l = emu::lane(), blk = l & ~3, j = l & 3;
    uint64_t bj = emu::wave().slot[blk + j][1];
    f32x4 d = c;
    for (int i = 0; i < 4; ++i) {                  // D[i][j] = sum_k A[i][k] B[k][j], A row i from lane blk + i
        const uint64_t ai = emu::wave().slot[blk + i][0];
        float acc = c[i];
        for (int k = 0; k < 4; ++k) {
            const float x = __uint_as_float((uint32_t)((ai >> (16 * k)) & 0xffff) << 16);
            const float y = __uint_as_float((uint32_t)((bj >> (16 * k)) & 0xffff) << 16);
            acc += x * y;                          // (product exact in float32)
        }
        d[i] = acc;
    }
    emu::done();
    return d;
}

static inline int lds_atomic_min(int *p, int v)
{
    std::atomic_ref<int> r(*p);
    int cur = r.load();
    while (v < cur && !r.compare_exchange_weak(cur, v)) {}
    return cur;
}

#define MSDA_DEVFN static inline
#define MSDA_KERNEL_BOUNDS(T, W) static
#define MSDA_LDS_DYNAMIC(name) unsigned char *name = emu::g_group->lds + emu::kStaticBytes
#define MSDA_LDS_STATIC(type, name, dims) type(&name) dims = *reinterpret_cast<type(*) dims>(emu::g_group->lds + static_offset(#name))
#define MSDA_TID emu::g_tid
#define MSDA_BID emu::g_group->block_idx
#define MSDA_READFIRSTLANE(x) emu::readfirstlane(x)
#define MSDA_UPDATE_DPP(old, v, ctrl, row_mask) emu::update_dpp(old, v, ctrl, row_mask)
#define MSDA_SYNCTHREADS() emu::g_group->bar.wait()
#define MSDA_LDS_ATOMIC_MIN(p, v) lds_atomic_min(p, v)
#define MSDA_LDS_ADDR(p) ((unsigned)((unsigned char *)(p) - emu::g_group->lds))
#define MSDA_WAVE_FENCE() emu::wave().bar.wait()      /* lock step on the hardware: the wave's LDS writes precede its reads */
#define MSDA_TR_READ_PAIR(b0, b1, addr) do { b0 = tr_read(addr, 0); b1 = tr_read(addr, 32); } while (0)
#define MSDA_MFMA444(a, b, c) mfma444(a, b, c)

static inline int static_offset(const char *name)      // the kernel's two __shared__ arrays, 16-byte aligned
{
    if (!std::strcmp(name, "box")) return 0;
    if (!std::strcmp(name, "winfo")) return 256;
    std::fprintf(stderr, "emu: unknown __shared__ array %s\n", name);
    std::abort();
}

#include "../../rlipv2_amd/csrc/msda_cell_forward.inc"

// problem file: int32 header {N, S, M, Lq, H0, W0, H1, W1, H2, W2, H3, W3}, then value bf16 [N, S, M, 32], starts int64 [4],
// loc float32 [N, Lq, M, 4, 4, 2], aw float32 [N, Lq, M, 4, 4]; output: bf16 [N, Lq, M, 32]
int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s problem.bin out.bin\n", argv[0]); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[12];
    if (std::fread(h, 4, 12, f) != 12) return 2;
    const int N = h[0], S = h[1], M = h[2], Lq = h[3];
    PatchPlan pl{};
    pl.CY = 1; pl.CX = 1;
    for (int l = 0; l < kL; ++l) {
        pl.H[l] = h[4 + 2 * l]; pl.W[l] = h[5 + 2 * l];
        const int cs = 16 >> l;                     // make_patch_plan (msda_patch.hip)
        pl.CY = max(pl.CY, (pl.H[l] + cs - 1) / cs);
        pl.CX = max(pl.CX, (pl.W[l] + cs - 1) / cs);
    }
    std::vector<bf16_t> value((size_t)N * S * M * kD), out((size_t)N * Lq * M * kD, 0x7fc0);
    std::vector<int64_t> starts(kL);
    std::vector<float> loc((size_t)N * Lq * M * 32), aw((size_t)N * Lq * M * 16);
    if (std::fread(value.data(), 2, value.size(), f) != value.size()) return 2;
    if (std::fread(starts.data(), 8, kL, f) != (size_t)kL) return 2;
    if (std::fread(loc.data(), 4, loc.size(), f) != loc.size()) return 2;
    if (std::fread(aw.data(), 4, aw.size(), f) != aw.size()) return 2;
    std::fclose(f);
    static_assert(kFwdLds + emu::kStaticBytes <= emu::kLdsBytes, "LDS model");
    const int blocks = N * M * pl.CY * pl.CX;
    emu::Group *grp = new emu::Group;
    for (int b = 0; b < blocks; ++b) {
        std::memset(grp->lds, 0xa5, sizeof(grp->lds));          // LDS starts as garbage
        grp->block_idx = b;
        std::vector<std::thread> th;
        for (int t = 0; t < emu::kThreads; ++t)
            th.emplace_back([&, t] {
                emu::g_group = grp;
                emu::g_tid = t;
                cell_forward_kernel(pl, value.data(), starts.data(), loc.data(), aw.data(), N, S, M, Lq, out.data());
            });
        for (auto &x : th) x.join();
    }
    f = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), 2, out.size(), f);
    std::fclose(f);
    return 0;
}
