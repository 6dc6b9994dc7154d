// Linear sum assignment on the device: the Hungarian matching of the training criterion without the host round trip
// (reference: HungarianMatcher.forward_seg_single, model/matcher.py:79-126, which moves the [n_queries, n_instances]
// cost matrix to the host and calls scipy.optimize.linear_sum_assignment per scene and step).
//
// Same algorithm as scipy's solver (the shortest-augmenting-path method of D. F. Crouse, "On implementing 2D
// rectangular assignment algorithms", IEEE TAES 52(4), 2016, as scipy/optimize/rectangular_lsap implements it), in
// float64 like scipy, with its tie-breaking reproduced: the column scan runs over `remaining` initialised in
// descending order, and among equal shortest-path costs the LAST unassigned column of the scan wins, else the FIRST
// column of the scan.  The problem is transposed when there are fewer instances than queries, as scipy does.
// One wave per problem: the column scan is lane-parallel (argmin by reduction), the bookkeeping between scans is a
// handful of elements.  The columns that take part are the instances marked present, in ascending order.
#include "common.h"

#define LSAP_MAX_ROWS 512   // smaller side of the problem
#define LSAP_MAX_COLS 1024  // larger side

namespace {

__device__ __forceinline__ double wave_min(double x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = fmin(x, __shfl_xor(x, d, 64));
    return x;
}
__device__ __forceinline__ int wave_max_i(int x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = max(x, __shfl_xor(x, d, 64));
    return x;
}
__device__ __forceinline__ int wave_min_i(int x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = min(x, __shfl_xor(x, d, 64));
    return x;
}

// cost: fp32 [nq, K] row-major; present[K]; out match_q[K] (query of instance k, -1 = absent or left over),
// match_of_q[nq] (instance of query q or -1), n_match
__global__ __launch_bounds__(64) void k_lsap(const float* __restrict__ cost, int nq, int K,
                                             const int32_t* __restrict__ present, int32_t* __restrict__ match_q,
                                             int32_t* __restrict__ match_of_q, int32_t* __restrict__ n_match,
                                             int32_t* __restrict__ status) {
    __shared__ double u[LSAP_MAX_ROWS], v[LSAP_MAX_COLS], spc[LSAP_MAX_COLS];
    __shared__ int path[LSAP_MAX_COLS], row4col[LSAP_MAX_COLS], remaining[LSAP_MAX_COLS], col4row[LSAP_MAX_ROWS];
    __shared__ int pcol[LSAP_MAX_COLS];  // present instances, ascending
    __shared__ unsigned char SR[LSAP_MAX_ROWS], SC[LSAP_MAX_COLS];
    __shared__ int s_P;
    const int lane = threadIdx.x;
    for (int k = lane; k < K; k += 64) match_q[k] = -1;
    for (int q = lane; q < nq; q += 64) match_of_q[q] = -1;
    // compact the present instances (ascending): ballot prefix over chunks of 64
    int P = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const bool p = k < K && present[k] != 0;
        const unsigned long long m = __ballot(p);
        if (p) {
            const int pos = P + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < LSAP_MAX_COLS) pcol[pos] = k;
        }
        P += __popcll(m);
    }
    if (lane == 0) {
        *n_match = 0;
        *status = 0;
    }
    if (P == 0 || nq == 0) return;
    // scipy: cost has shape (nq, P); transposed when P < nq, so that rows are the smaller side
    const bool transposed = P < nq;
    const int nr = transposed ? P : nq, nc = transposed ? nq : P;
    if (nr > LSAP_MAX_ROWS || nc > LSAP_MAX_COLS) {
        if (lane == 0) *status = 1;  // larger than the kernel's tables
        return;
    }
    __syncthreads();
    // C(i, j) of the (possibly transposed) problem
    auto C = [&](int i, int j) -> double {
        return transposed ? (double)cost[(size_t)j * K + pcol[i]] : (double)cost[(size_t)i * K + pcol[j]];
    };
    for (int i = lane; i < nr; i += 64) {
        u[i] = 0.0;
        col4row[i] = -1;
    }
    for (int j = lane; j < nc; j += 64) {
        v[j] = 0.0;
        path[j] = -1;
        row4col[j] = -1;
    }
    __syncthreads();
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    for (int curRow = 0; curRow < nr; curRow++) {
        // ---- augmenting path from curRow ----
        double minVal = 0.0;
        int num_remaining = nc;
        for (int it = lane; it < nc; it += 64) {
            remaining[it] = nc - it - 1;
            SC[it] = 0;
            spc[it] = INF;
        }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        __syncthreads();
        int sink = -1, i = curRow;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            double best = INF;
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double r = minVal + C(i, j) - ui - v[j];
                if (r < spc[j]) {
                    path[j] = i;
                    spc[j] = r;
                }
                best = fmin(best, spc[j]);
            }
            const double lowest = wave_min(best);
            if (lowest == INF) {  // infeasible (scipy raises); report and stop
                if (lane == 0) *status = 2;
                return;
            }
            int lastFree = -1, firstAny = 0x7fffffff;
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                if (spc[j] == lowest) {
                    firstAny = min(firstAny, it);
                    if (row4col[j] == -1) lastFree = max(lastFree, it);
                }
            }
            lastFree = wave_max_i(lastFree);
            firstAny = wave_min_i(firstAny);
            const int index = lastFree >= 0 ? lastFree : firstAny;
            minVal = lowest;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            __syncthreads();  // every lane has read remaining[] / row4col[] of this round
            if (lane == 0) {
                SC[j] = 1;
                remaining[index] = remaining[num_remaining - 1];
            }
            num_remaining--;
            __syncthreads();
        }
        // ---- dual variables ----
        if (lane == 0) u[curRow] += minVal;
        for (int r = lane; r < nr; r += 64)
            if (SR[r] && r != curRow) u[r] += minVal - spc[col4row[r]];
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
        __syncthreads();
        // ---- augment ----
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int r = path[j];
                row4col[j] = r;
                const int t = col4row[r];
                col4row[r] = j;
                j = t;
                if (r == curRow) break;
            }
        }
        __syncthreads();
    }
    for (int r = lane; r < nr; r += 64) {
        const int c = col4row[r];
        const int q = transposed ? c : r, inst = pcol[transposed ? r : c];
        match_q[inst] = q;
        match_of_q[q] = inst;
    }
    if (lane == 0) *n_match = nr;
}

}  // namespace

extern "C" int gf_lsap(const float* cost, int nq, int K, const int32_t* present, int32_t* match_q, int32_t* match_of_q,
                       int32_t* n_match, int32_t* status, void* stream) {
    GF_CHECK_ARG(cost && present && match_q && match_of_q && n_match && status, "gf_lsap: null argument");
    GF_CHECK_ARG(nq >= 0 && K >= 0, "gf_lsap: bad sizes");
    GF_CHECK_ARG((nq <= LSAP_MAX_ROWS || K <= LSAP_MAX_ROWS) && nq <= LSAP_MAX_COLS && K <= LSAP_MAX_COLS,
                 "gf_lsap: %d x %d exceeds the kernel's tables (%d on the smaller side, %d on the larger)", nq, K,
                 LSAP_MAX_ROWS, LSAP_MAX_COLS);
    hipLaunchKernelGGL(k_lsap, dim3(1), dim3(64), 0, (hipStream_t)stream, cost, nq, K, present, match_q, match_of_q, n_match,
                       status);
    GF_CHECK_LAUNCH("gf_lsap");
    return GF_OK;
}
