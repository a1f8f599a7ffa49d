// hoi_assign.hip -- host-side batched rectangular assignment (include/rlipv2_matcher.h).  No device code: it lives in the
// HIP library only so that the train step's native pieces load as one shared object.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/rlipv2_matcher.h"

namespace {

struct Solver {
    std::vector<double> cost, u, v, shortest;
    std::vector<long> path, col4row, row4col, remaining, order;
    std::vector<char> SR, SC;

    // one augmenting path from row i; returns the sink column or -1 (infeasible)
    long augment(long nc, long i, double *p_min)
    {
        double min_val = 0;
        long num_remaining = nc;
        for (long it = 0; it < nc; ++it) remaining[it] = nc - it - 1;     // reverse fill: lower columns win ties
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(shortest.begin(), shortest.end(), std::numeric_limits<double>::infinity());
        long sink = -1;
        while (sink == -1) {
            long index = -1;
            double lowest = std::numeric_limits<double>::infinity();
            SR[i] = 1;
            for (long it = 0; it < num_remaining; ++it) {
                const long j = remaining[it];
                const double r = min_val + cost[i * nc + j] - u[i] - v[j];
                if (r < shortest[j]) { path[j] = i; shortest[j] = r; }
                // among equal minima prefer a column that is still free (a new sink)
                if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) { lowest = shortest[j]; index = it; }
            }
            min_val = lowest;
            if (min_val == std::numeric_limits<double>::infinity()) return -1;
            const long j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        *p_min = min_val;
        return sink;
    }

    // c: [nr0, nc0] with row pitch `pitch` (float32).  Writes min(nr0, nc0) pairs (row, col) ordered by row.
    // Returns the number of pairs or -1.
    long solve(const float *c, long nr0, long nc0, long pitch, int64_t *a, int64_t *b)
    {
        if (nr0 == 0 || nc0 == 0) return 0;
        const bool transpose = nc0 < nr0;
        const long nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
        cost.resize((size_t)nr * nc);
        for (long i = 0; i < nr0; ++i)
            for (long j = 0; j < nc0; ++j) {
                const double x = (double)c[i * pitch + j];
                if (std::isnan(x) || x == -std::numeric_limits<double>::infinity()) return -1;
                if (transpose) cost[(size_t)j * nc + i] = x; else cost[(size_t)i * nc + j] = x;
            }
        u.assign(nr, 0.0); v.assign(nc, 0.0); shortest.resize(nc);
        path.assign(nc, -1); col4row.assign(nr, -1); row4col.assign(nc, -1);
        SR.resize(nr); SC.resize(nc); remaining.resize(nc);
        for (long cur = 0; cur < nr; ++cur) {
            double min_val;
            const long sink = augment(nc, cur, &min_val);
            if (sink < 0) return -1;
            u[cur] += min_val;
            for (long i = 0; i < nr; ++i)
                if (SR[i] && i != cur) u[i] += min_val - shortest[col4row[i]];
            for (long j = 0; j < nc; ++j)
                if (SC[j]) v[j] -= min_val - shortest[j];
            long j = sink;
            for (;;) {
                const long i = path[j];
                row4col[j] = i;
                std::swap(col4row[i], j);
                if (i == cur) break;
            }
        }
        if (transpose) {
            order.resize(nr);
            std::iota(order.begin(), order.end(), 0L);
            std::sort(order.begin(), order.end(), [&](long x, long y) { return col4row[x] < col4row[y]; });
            for (long i = 0; i < nr; ++i) { a[i] = col4row[order[i]]; b[i] = order[i]; }
        } else {
            for (long i = 0; i < nr; ++i) { a[i] = i; b[i] = col4row[i]; }
        }
        return nr;
    }
};

}  // namespace

extern "C" long hoi_assign_batch(const float *cost, int K, int bs, int nq, const int *sizes, int64_t *rows, int64_t *cols,
                                 long capacity)
{
    if (!cost || !sizes || !rows || !cols || K < 0 || bs < 0 || nq < 0) return -2;
    long T = 0;
    for (int i = 0; i < bs; ++i) {
        if (sizes[i] < 0) return -2;
        T += sizes[i];
    }
    Solver s;
    long n = 0;
    for (int k = 0; k < K; ++k) {
        long start = 0;
        for (int i = 0; i < bs; ++i) {
            const long nt = sizes[i], pairs = std::min<long>(nq, nt);
            if (n + pairs > capacity) return -2;
            const float *c = cost + ((size_t)(k * bs + i) * nq) * T + start;
            const long got = s.solve(c, nq, nt, T, rows + n, cols + n);
            if (got < 0) return -1;
            for (long p = 0; p < got; ++p) {
                rows[n + p] += (int64_t)(k * bs + i) * nq;
                cols[n + p] += start;
            }
            n += got;
            start += nt;
        }
    }
    return n;
}
