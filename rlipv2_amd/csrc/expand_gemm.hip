// expand_gemm.hip -- C[T, N] = epilogue(A[T, 256] . B[N, 256]^T) for the token-major Linears that widen the
// 256-channel encoder tensors (FFN first layer, input gradient of the FFN second layer), bf16 in/out, float32
// accumulation, for gfx950.
//
// Why not the library: at T = 88 892 these GEMMs are bound by the [T, N] output (364 MB for N = 2048), and the
// elementwise pass that follows them in PyTorch costs as much again:
//   * forward:  relu(x W1^T + b1)            hipBLASLt + epilogue 164 us (130 us with a tuned solution)
//   * backward: (dy W2) * (h > 0)            hipBLASLt 166-176 us  +  threshold_backward 172 us (1.1 GB of traffic)
// Here the ReLU mask (read from the saved activation h) or bias + ReLU is applied to the tile while it is
// still in the workgroup, so the masked gradient is written once and never re-read by an elementwise kernel.
//
// Design (MI355X, wave64):
//   * workgroup = 256 tokens x a quarter (or so) of the N columns, 4 waves, each owning 64 tokens; two workgroups
//     per CU.  K = 256 is fixed, so a wave keeps its 64 x 256 slice of A in registers for the whole kernel
//     (2 x 16 MFMA fragments = 128 VGPRs, loaded once): only B travels through LDS, one 16-byte fragment read per
//     two MFMAs -> the LDS pipe runs at half the MFMA rate instead of being the limit;
//   * B (the weight, L2-resident) is streamed in 64 x 256 tiles (32 KB) by LDS-DMA (global_load_lds_dwordx4) into
//     ONE buffer per workgroup: the next tile is requested as soon as every wave is done with the current one and
//     travels behind the rest of the epilogue and behind the sibling workgroup.  The DMA's LDS image is
//     lane-linear, so the bank skew is applied on the source side: 16-byte piece c of row n is stored at piece
//     c ^ (n & 31); the 32 rows a half-wave reads with one ds_read_b128 then hit 32 distinct pieces;
//   * the product is taken "swapped" (weights as the MFMA A operand, activations as B): a lane then owns one
//     token and 4 consecutive output columns per accumulator quad, so the bf16 results leave as 8-byte LDS
//     writes into a per-wave 64 x 64 staging tile (XOR-swizzled, no padding) and come back as 16-byte pieces of
//     whole 128-byte rows for the coalesced stores.  The 64 columns of a step are computed as two halves of 32
//     (2 accumulators = 32 VGPRs): all 64 at once left no registers for the epilogue;
//   * the mask rows of a step arrive by LDS-DMA in that same staging tile (same swizzle) while the first MFMA loop
//     runs, are read back in the accumulator layout and applied to the packed results before these overwrite them:
//     no registers are held across the MFMA loops (32 VGPRs of prefetched mask spilled), and the mask's HBM latency
//     (~10 000 cycles per step when loaded next to its use, tools/expand_timeline.py) overlaps the MFMAs;
//   * ALL LDS traffic is inline asm: hipcc makes every LDS access it can see wait for vmcnt(0) while an LDS-DMA may
//     be in flight -- in the epilogue that meant waiting for the previous rows' global stores on every row.
// Measured (tools/expand_bench.py, tools/expand_ablate.sh; T = 88 892, N = 2048): masked input gradient 215 us
// against 360 us for the library GEMM + threshold_backward; bias + ReLU forward 150 us against 135-145 us for the
// tuned library GEMM, so the forward stays on the library.  Ablation: skeleton 45 us + MFMA 34 + output stores 33 +
// A loads 33 + DMA 7 add up; the cycle-stamp timeline (tools/expand_timeline.py, -DXDBG=64) shows why: every
// vector-memory instruction takes hundreds of cycles to ISSUE (8 DMA instructions: 800-3000 cycles) -- the memory
// system is saturated by 128-byte pieces at 4 KB stride (one output / mask row segment per wave and step), at about
// half the HBM rate.  Next: 256-512-byte runs per row (waves of a workgroup side by side in N instead of in T).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/rlipv2_linear.h"
#include "once_per_device.h"
#include "../../include/rlipv2_msda.h"
#include "msda_device.h"
#ifdef MSDA_EMU
#include <cstring>
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int XK = 256;                     // reduction width (fixed: the encoder's d_model)
constexpr int XBM = 256;                    // tokens per workgroup (4 waves x 64)
constexpr int XBN = 64;                     // output columns per step
constexpr int XTHREADS = 256;
constexpr int XWAVES = XTHREADS / 64;
#ifndef XPF
#define XPF 3
#endif
#ifndef XDBG
#define XDBG 0   // ablation build switches (tools/expand_ablate.sh); 0 in the product
#endif
#if XDBG & 64
__device__ unsigned long long xdbg_ts[4 * 16 * 10];      // [probe][step][point] cycle stamps (timeline builds only)
#define XTS(k)                                                                                             \
    do {                                                                                                   \
        if ((blockIdx.x == 0 || blockIdx.x == 301) && blockIdx.y == 0 && (tid == 0 || tid == 192))        \
            xdbg_ts[(((blockIdx.x != 0) * 2 + (tid != 0)) * 16 + (i - step0)) * 10 + (k)] = clock64();    \
    } while (0)
#else
#define XTS(k) do { } while (0)
#endif
constexpr int BTILE = XBN * XK * 2;         // 32 KB
constexpr int CSTAGE = 64 * 128;            // per wave: 64 tokens x 64 columns bf16 = 8 KB
constexpr int X_LDS = BTILE + XWAVES * CSTAGE;       // 64 KB (+ bias slice): two workgroups per CU

template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr)
{
    u32x4 v;
#ifndef MSDA_EMU
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
#else
    std::memcpy(&v, emu::lds_ptr(addr + OFF), 16);     // (host model, tools/emu/)
#endif
    return v;
}

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ void lds_write_b64(unsigned addr, u32x2 v)
{
#ifndef MSDA_EMU
    asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(v) : "memory");
#else
    std::memcpy(emu::lds_ptr(addr), &v, 8);
#endif
}

__device__ __forceinline__ u32x2 lds_read_b64(unsigned addr)
{
    u32x2 v;
#ifndef MSDA_EMU
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
#else
    std::memcpy(&v, emu::lds_ptr(addr), 8);
#endif
    return v;
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// v_cvt_pk_bf16_f32: round-to-nearest-even, NaN-preserving, one instruction for two values
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    union { bf16x2 b; uint32_t u; } cv;
    cv.b = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
    return cv.u;
}

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// keep the halves of `v` whose mask element is > 0 (bf16 compare done on the raw bits: positive and non-zero)
__device__ __forceinline__ uint32_t keep_positive(uint32_t v, uint32_t h)
{
    const uint32_t lo = ((int32_t)(h << 16) > 0) ? 0x0000ffffu : 0u;
    const uint32_t hi = ((int32_t)(h & 0xffff0000u) > 0) ? 0xffff0000u : 0u;
    return v & (lo | hi);
}

template <bool MASK, bool BIAS, bool RELU>
__global__ __launch_bounds__(XTHREADS, 2) void expand_kernel(const uint16_t *__restrict__ a, const uint16_t *__restrict__ b,
                                                             const uint16_t *__restrict__ bias,
                                                             const uint16_t *__restrict__ mask, int T, int N,
                                                             int steps_per_split, uint16_t *__restrict__ c)
{
    MSDA_DYNAMIC_LDS_ALIGNED(char, smem, 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform values stay in scalar registers
    const int l32 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * XBM + wave * 64;            // this wave's first token
    const int step0 = blockIdx.y * steps_per_split;
    const int step1 = min(N / XBN, step0 + steps_per_split);

    // B tile DMA: instruction j of thread tid fills LDS slot j * 256 + tid (16 bytes);
    // slot -> row n = slot >> 5, physical piece pc = slot & 31, which holds logical piece pc ^ (n & 31)
    const int drow = tid >> 5, dpc = tid & 31;
    auto issue = [&](int step) {
        const uint16_t *tile = b + (size_t)step * XBN * XK;
        char *dst = smem + wave * 1024;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = j * 8 + drow;
            const uint16_t *src = tile + (size_t)n * XK + ((dpc ^ (n & 31)) * 8);
            MSDA_GLOBAL_LOAD_LDS16(src, dst + j * 4096);
        }
    };
    if (step0 < step1) issue(step0);
    if (BIAS) {   // this workgroup's bias slice -> LDS (read back per step by ds_read_b64)
        uint2 *dst = reinterpret_cast<uint2 *>(smem + BTILE + XWAVES * CSTAGE);
        const uint2 *src = reinterpret_cast<const uint2 *>(bias + (size_t)step0 * XBN);
        for (int k = tid; k < (step1 - step0) * XBN / 4; k += XTHREADS) dst[k] = src[k];
    }

    // ---- A: this wave's 64 tokens x 256, as MFMA B-operand fragments (token = lane % 32, 8 k per lane) ----------
    bf16x8 af[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = min(t0 + mt * 32 + l32, T - 1);
        const u32x4 *src = reinterpret_cast<const u32x4 *>(a + (size_t)row * XK + 8 * hi);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            union { u32x4 u; bf16x8 v; } cv;
            if (XDBG & 32) cv.u = u32x4{(unsigned)lane, (unsigned)s, 1u, 2u}; else
            cv.u = src[2 * s];
            af[mt][s] = cv.v;
        }
    }
    // retire the A loads (and the first tile's DMA) HERE: otherwise the compiler's vmcnt(0) for them sits in front
    // of the first MFMA of the column loop and waits for the freshly issued B-tile DMA on every iteration
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int s = 0; s < 16; ++s) MSDA_ASM_OPAQUE(af[mt][s]);
    MSDA_ASM_WAIT_VM();

    // fragment addresses: row n = nt * 32 + l32 of the tile, k-step s -> logical piece 2 s + hi
    const unsigned lds0 = MSDA_LDS_BYTE_ADDR(smem);
    const unsigned frag_row = lds0 + (unsigned)l32 * 512u;
    const unsigned stage_lds = lds0 + BTILE + wave * CSTAGE;
    const unsigned bias_lds = lds0 + BTILE + XWAVES * CSTAGE;

    // One B buffer per workgroup and TWO workgroups per CU: while this one waits for its next tile (and for its
    // output stores to drain) the sibling computes.  (An 8-wave workgroup with a double-buffered B ran every phase
    // of all its waves in lockstep: MFMA time, A loads and output stores simply added up, tools/expand_ablate.sh.)
    auto step_body = [&](int i) {
        constexpr int BUF = 0;
        __builtin_amdgcn_s_barrier();          // tile i is in LDS, for every wave
        MSDA_ASM_FENCE();
        XTS(0);

        // The 64-column tile is taken as two 32-column halves, one after the other: a half needs 2 accumulators
        // (32 VGPRs) next to the 128 of the resident A fragments; both halves at once (64) left the epilogue
        // without registers and the spill reloads (vector-memory operations) waited for the output stores.
        const int ncol = i * XBN;
        unsigned swz = (unsigned)l32;              // opaque copy of the lane id: keeps the 16 swizzled addresses
        MSDA_ASM_OPAQUE(swz);                      // from being hoisted out of the column loop (16 registers)
        // staging tile of this wave: [64 tokens][128 bytes], 16-byte piece c of row r stored at piece c ^ ((r >> 1) & 7)
        // (the same map for the mask that arrives by DMA and for the results that replace it)
        const unsigned s3 = ((unsigned)(l32 >> 1) & 7u) << 4;
        const unsigned wbase = stage_lds + (unsigned)l32 * 128u + (unsigned)hi * 8u;
        const int q8 = lane & 7, r0 = lane >> 3;
        const unsigned off0 = ((unsigned)(t0 + r0) * (unsigned)N + (unsigned)(ncol + q8 * 8)) * 2u, rstep = 16u * (unsigned)N;
        char *cbytes = reinterpret_cast<char *>(c);
        if (MASK) {
            // the mask rows of this step travel straight into the staging tile (no registers) while the first MFMA
            // loop runs: loaded next to their use (timeline build, tools/expand_timeline.py) the wave sat ~10 000
            // cycles per step waiting for HBM.  DMA instruction j, lane l -> slot 64 j + l: row 8 j + (l >> 3),
            // physical piece l & 7.  (32-bit byte offsets: the host checks T * N * 2 < 2^32.)
            const char *mbytes = reinterpret_cast<const char *>(mask);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * j + r0;
                const unsigned row = (unsigned)min(t0 + r, T - 1);
                const unsigned src = (row * (unsigned)N + (unsigned)(ncol + ((q8 ^ ((r >> 1) & 7)) << 3))) * 2u;
                MSDA_GLOBAL_LOAD_LDS16(mbytes + src, smem + BTILE + wave * CSTAGE + j * 1024);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 acc[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mt][q] = 0.f;
            // 16 k-steps, one fragment read per two MFMAs, reads EIGHT steps ahead of their use: with 8 waves on the
            // LDS pipe a read takes several hundred cycles to come back, and two steps (4 MFMAs = 128 clk) of cover
            // left the loop latency-bound at 2.3x the MFMA time (tools/expand_bench.py)
            constexpr int PF = XPF;              // fragment reads in flight ahead of the MFMAs
            u32x4 f[PF];
            auto read_step = [&](int s) {
                if (XDBG & 2) return u32x4{swz, (unsigned)s, 3u, 4u};
                return lds_read_b128<BUF * BTILE>(frag_row + nt * 32 * 512 + (((unsigned)(2 * s + hi) ^ swz) << 4));
            };
#pragma unroll
            for (int s = 0; s < PF; ++s) f[s] = read_step(s);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                // reads issued after read(s) and still allowed in flight: min(PF - 1, 15 - s)
                constexpr int dummy = 0; (void)dummy;
                const int allow = (15 - s < PF - 1) ? 15 - s : PF - 1;
#ifndef MSDA_EMU
                switch (allow) {
                case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[s % PF])); break;
                case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f[s % PF])); break;
                case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[s % PF])); break;
                case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f[s % PF])); break;
                case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[s % PF])); break;
                case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(f[s % PF])); break;
                case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f[s % PF])); break;
                default: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(f[s % PF])); break;
                }
#else
                (void)allow;
#endif
                union { u32x4 u; bf16x8 v; } w;
                w.u = f[s % PF];
                if (XDBG & 1) { acc[0][s] += __uint_as_float(w.u[0]); acc[1][s] += __uint_as_float(w.u[1]); continue; }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.v, af[0][s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.v, af[1][s], acc[1], 0, 0, 0);
                if (s + PF < 16) f[s % PF] = read_step(s + PF);
            }

            // accumulator register q of lane (l32, hi) is column 8 (q / 4) + 4 hi + q % 4 of this half, token l32 of
            // token sub-tile mt.  All LDS traffic below is inline asm as well: a plain LDS access would make the
            // compiler wait for vmcnt(0), i.e. for the previous rows' global stores, every time.
            XTS(1 + 2 * nt);
            u32x2 mraw[2][4];
            if (MASK) {
                if (nt == 0) {
                    MSDA_ASM_WAIT_VM();      // the mask tile has landed
                    MSDA_WAVE_LDS_SYNC();    // (every lane's slice of it, for the lanes that read it below)
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) mraw[mt][g] = lds_read_b64(wbase + mt * 32 * 128 + (s3 ^ ((4 * nt + g) << 4)));
#ifndef MSDA_EMU
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(mraw[0][0]), "+v"(mraw[0][1]), "+v"(mraw[0][2]), "+v"(mraw[0][3]), "+v"(mraw[1][0]),
                               "+v"(mraw[1][1]), "+v"(mraw[1][2]), "+v"(mraw[1][3]));
#endif
            }
            u32x2 braw[4];
            if (BIAS) {
                const unsigned baddr = bias_lds + (unsigned)((i - step0) * XBN + nt * 32 + 4 * hi) * 2u;
#pragma unroll
                for (int g = 0; g < 4; ++g) braw[g] = lds_read_b64(baddr + 8 * g * 2);
#ifndef MSDA_EMU
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(braw[0]), "+v"(braw[1]), "+v"(braw[2]), "+v"(braw[3]));
#endif
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * g + e];
                    if (BIAS) {
                        v[0] += bf16_lo(braw[g][0]); v[1] += bf16_hi(braw[g][0]);
                        v[2] += bf16_lo(braw[g][1]); v[3] += bf16_hi(braw[g][1]);
                    }
                    if (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x2 o;
                    o[0] = pack_bf16(v[0], v[1]);
                    o[1] = pack_bf16(v[2], v[3]);
                    if (MASK) {
                        o[0] = keep_positive(o[0], mraw[mt][g][0]);
                        o[1] = keep_positive(o[1], mraw[mt][g][1]);
                    }
                    lds_write_b64(wbase + mt * 32 * 128 + (s3 ^ ((4 * nt + g) << 4)), o);
                }
            XTS(2 + 2 * nt);
        }
        // every wave is done with tile i -> the next tile may overwrite it; its latency hides behind the rest of
        // the epilogue and, beyond that, behind the sibling workgroup
        __builtin_amdgcn_s_barrier();
        MSDA_ASM_FENCE();
        XTS(5);
        if (!(XDBG & 16) && i + 1 < step1) issue(i + 1);
        // read back whole rows: lane -> row (lane >> 3) + 8 jj, 16-byte slot lane & 7 (LDS runs a wave's
        // instructions in order, so the reads see the writes above without a wait in between)
        XTS(6);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 rv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned r = (unsigned)(r0 + 8 * (4 * half + j));
                rv[j] = lds_read_b128<0>(stage_lds + r * 128u + (((unsigned)q8 ^ ((r >> 1) & 7u)) << 4));
            }
#ifndef MSDA_EMU
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]));
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int jj = 4 * half + j, r = r0 + 8 * jj;
                const uint4 v = make_uint4(rv[j][0], rv[j][1], rv[j][2], rv[j][3]);
                if ((XDBG & 4) ? (v.x == 0x12345u && t0 + r < T) : (t0 + r < T))
                    *reinterpret_cast<uint4 *>(cbytes + (off0 + jj * rstep)) = v;
            }
        }
        XTS(8);
        MSDA_ASM_WAIT_VM();      // tile i+1 has landed (and this step's stores have left)
        XTS(9);
    };

    for (int i = step0; i < step1; ++i) step_body(i);
}

}  // namespace

#if XDBG & 64
extern "C" int linear_expand_debug_read(void *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(xdbg_ts), sizeof(unsigned long long) * 4 * 16 * 10);
}
#endif

extern "C" int linear_expand_supported(int T, int N, int K)
{
    // (32-bit byte offsets inside the [T, N] tensors)
    return (T >= 1 && K == XK && N >= XBN && N % XBN == 0 && (size_t)T * N * 2 < ((size_t)1 << 32)) ? 1 : 0;
}

extern "C" int linear_expand_bf16(const void *a, const void *b, const void *bias, const void *mask, int T, int N, int K,
                                  int relu, void *c, void *stream_)
{
    if (!linear_expand_supported(T, N, K)) return MSDA_ERR_BAD_SHAPE;
    if (!a || !b || !c) return MSDA_ERR_NULL_POINTER;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
         reinterpret_cast<uintptr_t>(mask)) & 15u)
        return MSDA_ERR_ALIGNMENT;
    if (reinterpret_cast<uintptr_t>(bias) & 7u) return MSDA_ERR_ALIGNMENT;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint16_t *a16 = static_cast<const uint16_t *>(a), *b16 = static_cast<const uint16_t *>(b);
    const uint16_t *bias16 = static_cast<const uint16_t *>(bias), *mask16 = static_cast<const uint16_t *>(mask);
    uint16_t *c16 = static_cast<uint16_t *>(c);
    // split the columns over blockIdx.y so that the grid fills the 256 CUs (two 4-wave workgroups each): cost model
    // = rounds * (steps per workgroup + 2 for loading its A slice), minimised over the split
    const int steps = N / XBN, row_blocks = (T + XBM - 1) / XBM;
    int best_split = 1;
    long best_cost = -1;
    for (int split = 1; split <= steps; ++split) {
        const int per = (steps + split - 1) / split, used = (steps + per - 1) / per;
        const long blocks = (long)row_blocks * used, cost = ((blocks + 511) / 512) * (per + 2);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_split = split; }
    }
    const int steps_per_split = (steps + best_split - 1) / best_split;
    const dim3 grid(row_blocks, (steps + steps_per_split - 1) / steps_per_split), block(XTHREADS);
    const size_t lds_bytes = X_LDS + (size_t)steps_per_split * XBN * 2;      // + the bias slice
    if (lds_bytes > 160 * 1024 || (size_t)T * N * 2 >= ((size_t)1 << 32)) return MSDA_ERR_BAD_SHAPE;
#define XLAUNCH(M_, B_, R_)                                                                                              \
    do {                                                                                                                 \
        RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void *>(&expand_kernel<M_, B_, R_>),     \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));       \
        hipLaunchKernelGGL((expand_kernel<M_, B_, R_>), grid, block, lds_bytes, stream, a16, b16, bias16, mask16, T, N,     \
                           steps_per_split, c16);                                                                        \
    } while (0)
    const bool m = mask != nullptr, bb = bias != nullptr, r = relu != 0;
    if (m) {
        if (bb) { if (r) XLAUNCH(true, true, true); else XLAUNCH(true, true, false); }
        else    { if (r) XLAUNCH(true, false, true); else XLAUNCH(true, false, false); }
    } else {
        if (bb) { if (r) XLAUNCH(false, true, true); else XLAUNCH(false, true, false); }
        else    { if (r) XLAUNCH(false, false, true); else XLAUNCH(false, false, false); }
    }
#undef XLAUNCH
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
