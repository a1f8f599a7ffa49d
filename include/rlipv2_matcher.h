/* rlipv2_matcher.h -- C ABI of the host-side assignment step of HungarianMatcherHOI (plain C++, no device code).
 *
 * The reference solves one rectangular linear-sum-assignment problem per image and decoder layer with
 * scipy.optimize.linear_sum_assignment on the host copy of the cost matrix (models/matcher.py:91, :193, :258:
 * `[linear_sum_assignment(c[i]) for i, c in enumerate(C.split(sizes, -1))]`).  The GPU idles while that runs -- between
 * the forward and the backward graph of the train step -- so the K * bs problems of a step are solved here in one call,
 * without Python in the loop.
 *
 * hoi_assign_batch: cost [K, bs, nq, T] float32, contiguous, on the HOST; image i owns the columns
 *   [start_i, start_i + sizes[i]) with start_i = sizes[0] + .. + sizes[i-1], T = sum(sizes).  For every (k, i) the
 *   min(nq, sizes[i]) matched pairs are appended to rows / cols: rows = (k * bs + i) * nq + query, cols = start_i + target,
 *   pairs ordered by query (scipy's row_ind order).  Returns the number of pairs written (<= capacity), -1 if a cost is
 *   NaN or -inf or a problem is infeasible (scipy raises ValueError there), -2 on bad arguments.
 *
 * The solver is the shortest-augmenting-path algorithm scipy uses (Crouse, "On implementing 2D rectangular assignment
 * algorithms", 2016) restated in double precision with the same tie-breaking (lower-numbered columns preferred, a free
 * column preferred among equals), so the assignments are the ones scipy returns -- tests/test_matcher_native.py compares
 * them on random, integer-valued (tied) and degenerate matrices.
 */
#ifndef RLIPV2_MATCHER_H
#define RLIPV2_MATCHER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

long hoi_assign_batch(const float *cost, int K, int bs, int nq, const int *sizes, int64_t *rows, int64_t *cols,
                      long capacity);

#ifdef __cplusplus
}
#endif
#endif
