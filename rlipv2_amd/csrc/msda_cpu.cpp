// CPU twins of the MSDA operator (include/rlipv2_msda_cpu.h): what `MultiScaleDeformableAttention` does when it is handed
// CPU tensors ("CPU twins msda_forward_cpu / msda_backward_cpu"; BASELINE config 1 runs the model on
// the CPU).  The reference has no native CPU implementation (models/ops/src/cpu/ms_deform_attn_cpu.cpp:24,40 raise); its CPU
// arithmetic is the per-level grid_sample formulation of models/ops/functions/ms_deform_attn_func.py:45-65, whose sampling
// rule (pixel = loc * size - 0.5, zero padding, samples with a coordinate <= -1 or >= size dropped) and gradient formulas
// are those of models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-159, 282-291.  This file follows the CUDA kernel's rule
// at the drop boundary (the two differ on a measure-zero set, tests/conftest.py: boundary_samples).
//
// Built with g++ -fopenmp into librlipv2_msda_cpu.so: no HIP, no GPU, loads anywhere.  It serves CPU tensors ONLY -- a CUDA
// tensor never comes here, and a missing HIP library still raises for CUDA tensors (rlipv2_amd/_lib.py).
//
// Work partition.  Forward: one (image, query) row per task, heads and channels inside.  Backward: one (image, head, LEVEL)
// per task (levels with overlapping row ranges: one (image, head) per task, see backward()) -- the task owns the grad_value rows of its level and head (zero-fills them, then accumulates: no atomics, the
// order of the sums is fixed by the query order, so results are bit-repeatable for any thread count) and the
// grad_sampling_loc / grad_attn_weight entries of its level.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rlipv2_msda_cpu.h"

namespace {

template <typename T> struct Corner {
    int64_t row[4];      // pixel index inside the level (y * W + x), -1 = outside the level
    T weight[4];         // bilinear weight of TL, TR, BL, BR
    T lw, lh;            // fractional parts
};

// geometry of one sample: false = dropped (ms_deform_im2col_cuda.cuh:285-288)
template <typename T> inline bool sample_geometry(T loc_x, T loc_y, int H, int W, Corner<T> &c)
{
    const T x = loc_x * T(W) - T(0.5), y = loc_y * T(H) - T(0.5);
    if (!(y > T(-1) && x > T(-1) && y < T(H) && x < T(W)))
        return false;
    const T fx = std::floor(x), fy = std::floor(y);
    const int x0 = int(fx), y0 = int(fy);
    c.lw = x - fx;
    c.lh = y - fy;
    const T hw = T(1) - c.lw, hh = T(1) - c.lh;
    const bool top = y0 >= 0, bottom = y0 + 1 <= H - 1, left = x0 >= 0, right = x0 + 1 <= W - 1;
    c.row[0] = (top && left) ? int64_t(y0) * W + x0 : -1;
    c.row[1] = (top && right) ? int64_t(y0) * W + x0 + 1 : -1;
    c.row[2] = (bottom && left) ? int64_t(y0 + 1) * W + x0 : -1;
    c.row[3] = (bottom && right) ? int64_t(y0 + 1) * W + x0 + 1 : -1;
    c.weight[0] = hh * hw;
    c.weight[1] = hh * c.lw;
    c.weight[2] = c.lh * hw;
    c.weight[3] = c.lh * c.lw;
    return true;
}

template <typename T>
void forward(const T *value, const int64_t *shapes, const int64_t *starts, const T *loc, const T *aw, int N, int S, int M, int D,
             int L, int Lq, int P, T *out)
{
    const int64_t rows = int64_t(N) * Lq;
#pragma omp parallel
    {
        std::vector<T> acc(size_t(M) * D);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < rows; ++r) {
            const int n = int(r / Lq);
            const T *v_img = value + int64_t(n) * S * M * D;
            std::fill(acc.begin(), acc.end(), T(0));
            for (int m = 0; m < M; ++m) {
                T *a = acc.data() + size_t(m) * D;
                const T *loc_h = loc + ((r * M + m) * L) * P * 2;
                const T *aw_h = aw + ((r * M + m) * L) * P;
                for (int l = 0; l < L; ++l) {
                    const int H = int(shapes[2 * l]), W = int(shapes[2 * l + 1]);
                    const T *v_lvl = v_img + (starts[l] * M + m) * D;
                    for (int p = 0; p < P; ++p) {
                        Corner<T> c;
                        if (!sample_geometry(loc_h[(l * P + p) * 2], loc_h[(l * P + p) * 2 + 1], H, W, c))
                            continue;
                        const T w = aw_h[l * P + p];
                        for (int k = 0; k < 4; ++k) {
                            if (c.row[k] < 0)
                                continue;
                            const T wk = w * c.weight[k];
                            const T *v = v_lvl + c.row[k] * M * D;
                            for (int d = 0; d < D; ++d)
                                a[d] += wk * v[d];
                        }
                    }
                }
            }
            std::memcpy(out + r * M * D, acc.data(), sizeof(T) * size_t(M) * D);
        }
    }
}

// `disjoint`: the levels' row ranges do not overlap (every real pyramid).  Then a task = (image, head, level) owns its rows.  With
// overlapping level_start_index ranges (the reference's atomicAdd formulation tolerates any layout, ms_deform_im2col_cuda.cuh:
// 142-156) two level tasks would zero and accumulate the same rows: there a task = (image, head) walks all levels itself, over a
// grad_value the caller has zeroed.
template <typename T>
void backward(const T *value, const int64_t *shapes, const int64_t *starts, const T *loc, const T *aw, const T *grad_out, int N,
              int S, int M, int D, int L, int Lq, int P, T *g_value, T *g_loc, T *g_aw, bool disjoint)
{
    const int groups = disjoint ? L : 1;                 // level groups per (image, head)
    const int64_t tasks = int64_t(N) * M * groups;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t t = 0; t < tasks; ++t) {
        const int g = int(t % groups), m = int((t / groups) % M), n = int(t / (int64_t(groups) * M));
        const int l_begin = disjoint ? g : 0, l_end = disjoint ? g + 1 : L;
        for (int l = l_begin; l < l_end; ++l) {
        const int H = int(shapes[2 * l]), W = int(shapes[2 * l + 1]);
        const T *v_lvl = value + ((int64_t(n) * S + starts[l]) * M + m) * D;
        T *gv_lvl = g_value + ((int64_t(n) * S + starts[l]) * M + m) * D;
        if (disjoint)
            for (int64_t s = 0; s < int64_t(H) * W; ++s)                    // this task's rows of grad_value
                std::memset(gv_lvl + s * M * D, 0, sizeof(T) * D);
        for (int q = 0; q < Lq; ++q) {
            const int64_t head = (int64_t(n) * Lq + q) * M + m;
            const T *go = grad_out + head * D;
            for (int p = 0; p < P; ++p) {
                const int64_t s_idx = (head * L + l) * P + p;
                Corner<T> c;
                if (!sample_geometry(loc[s_idx * 2], loc[s_idx * 2 + 1], H, W, c)) {
                    g_aw[s_idx] = g_loc[s_idx * 2] = g_loc[s_idx * 2 + 1] = T(0);
                    continue;
                }
                const T w = aw[s_idx];
                // <grad_out, corner k> for the in-level corners; grad_value rows in the same sweep
                T dot[4] = {T(0), T(0), T(0), T(0)};
                for (int k = 0; k < 4; ++k) {
                    if (c.row[k] < 0)
                        continue;
                    const T *v = v_lvl + c.row[k] * M * D;
                    T *gv = gv_lvl + c.row[k] * M * D;
                    const T wk = w * c.weight[k];
                    T sum = T(0);
                    for (int d = 0; d < D; ++d) {
                        sum += go[d] * v[d];
                        gv[d] += wk * go[d];
                    }
                    dot[k] = sum;
                }
                const T hw = T(1) - c.lw, hh = T(1) - c.lh;
                // d(bilinear)/dx = hh (v_TR - v_TL) + lh (v_BR - v_BL), d/dy = hw (v_BL - v_TL) + lw (v_BR - v_TR), each dotted
                // with grad_out (ms_deform_im2col_cuda.cuh:114-158); in pixels -> normalised coordinates: x W, x H
                g_aw[s_idx] = c.weight[0] * dot[0] + c.weight[1] * dot[1] + c.weight[2] * dot[2] + c.weight[3] * dot[3];
                g_loc[s_idx * 2] = T(W) * w * (hh * (dot[1] - dot[0]) + c.lh * (dot[3] - dot[2]));
                g_loc[s_idx * 2 + 1] = T(H) * w * (hw * (dot[2] - dot[0]) + c.lw * (dot[3] - dot[1]));
            }
        }
        }
    }
}

// do the row ranges [start, start + H W) of the levels overlap?  (L is small: the quadratic test is the clear one)
bool levels_disjoint(const int64_t *shapes, const int64_t *starts, int L)
{
    for (int a = 0; a < L; ++a)
        for (int b = a + 1; b < L; ++b) {
            const int64_t a0 = starts[a], a1 = a0 + shapes[2 * a] * shapes[2 * a + 1];
            const int64_t b0 = starts[b], b1 = b0 + shapes[2 * b] * shapes[2 * b + 1];
            if (a0 < b1 && b0 < a1)
                return false;
        }
    return true;
}

int check(int dtype, const void *const *ptrs, int n_ptrs, const int64_t *shapes, const int64_t *starts, int N, int S, int M, int D,
          int L, int Lq, int P)
{
    if (dtype != MSDA_CPU_F32 && dtype != MSDA_CPU_F64)
        return MSDA_CPU_ERR_DTYPE;
    if (N < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0)
        return MSDA_CPU_ERR_DIMS;
    // operand order: value, shapes, starts, loc, aw, then per-query tensors (out | grad_out, g_value, g_loc, g_aw); an empty
    // tensor may come with a null pointer
    for (int k = 0; k < n_ptrs; ++k) {
        const bool is_value = k == 0 || (n_ptrs == 9 && k == 6);
        const bool empty = (k == 1 || k == 2) ? false : (is_value ? (N == 0 || S == 0) : (N == 0 || Lq == 0));
        if (!ptrs[k] && !empty)
            return MSDA_CPU_ERR_NULL;
    }
    int64_t total = 0;
    for (int l = 0; l < L; ++l) {
        if (shapes[2 * l] <= 0 || shapes[2 * l + 1] <= 0 || starts[l] < 0 || starts[l] + shapes[2 * l] * shapes[2 * l + 1] > S)
            return MSDA_CPU_ERR_LEVELS;          // a level must lie inside value's S rows (the kernels index with it)
        total += shapes[2 * l] * shapes[2 * l + 1];
    }
    (void)total;
    return 0;
}

}  // namespace

extern "C" {

int msda_forward_cpu(int dtype, const void *value, const int64_t *shapes, const int64_t *starts, const void *loc, const void *aw,
                     int N, int S, int M, int D, int L, int Lq, int P, void *out)
{
    const void *ptrs[] = {value, shapes, starts, loc, aw, out};
    if (!shapes || !starts)
        return MSDA_CPU_ERR_NULL;
    if (int st = check(dtype, ptrs, 6, shapes, starts, N, S, M, D, L, Lq, P))
        return st;
    if (dtype == MSDA_CPU_F32)
        forward<float>((const float *)value, shapes, starts, (const float *)loc, (const float *)aw, N, S, M, D, L, Lq, P, (float *)out);
    else
        forward<double>((const double *)value, shapes, starts, (const double *)loc, (const double *)aw, N, S, M, D, L, Lq, P,
                        (double *)out);
    return 0;
}

int msda_backward_cpu(int dtype, const void *value, const int64_t *shapes, const int64_t *starts, const void *loc, const void *aw,
                      const void *grad_out, int N, int S, int M, int D, int L, int Lq, int P, void *g_value, void *g_loc, void *g_aw)
{
    const void *ptrs[] = {value, shapes, starts, loc, aw, grad_out, g_value, g_loc, g_aw};
    if (!shapes || !starts)
        return MSDA_CPU_ERR_NULL;
    if (int st = check(dtype, ptrs, 9, shapes, starts, N, S, M, D, L, Lq, P))
        return st;
    // rows of `value` that belong to no level (S larger than the pyramid) receive no gradient: zero them here, the tasks
    // zero the rows they own.  Overlapping levels: nobody owns a row, everything is zeroed here.
    const size_t elt = dtype == MSDA_CPU_F32 ? sizeof(float) : sizeof(double);
    const bool disjoint = levels_disjoint(shapes, starts, L);
    int64_t covered = 0;
    for (int l = 0; l < L; ++l)
        covered += shapes[2 * l] * shapes[2 * l + 1];
    if ((covered != S || !disjoint) && g_value)
        std::memset(g_value, 0, elt * size_t(N) * S * M * D);
    if (dtype == MSDA_CPU_F32)
        backward<float>((const float *)value, shapes, starts, (const float *)loc, (const float *)aw, (const float *)grad_out, N, S, M,
                        D, L, Lq, P, (float *)g_value, (float *)g_loc, (float *)g_aw, disjoint);
    else
        backward<double>((const double *)value, shapes, starts, (const double *)loc, (const double *)aw, (const double *)grad_out, N,
                         S, M, D, L, Lq, P, (double *)g_value, (double *)g_loc, (double *)g_aw, disjoint);
    return 0;
}

const char *msda_cpu_strerror(int status)
{
    switch (status) {
    case 0: return "ok";
    case MSDA_CPU_ERR_DTYPE: return "unsupported dtype (float32 / float64)";
    case MSDA_CPU_ERR_NULL: return "null pointer argument";
    case MSDA_CPU_ERR_DIMS: return "bad dimensions";
    case MSDA_CPU_ERR_LEVELS: return "a level of spatial_shapes / level_start_index does not lie inside value's rows";
    default: return "unknown status";
    }
}

int msda_cpu_abi_version(void) { return 1; }

}  // extern "C"
