// Geodesic BFS, multi-source form: all nq queries of a scene advance together, one launch per hop over the whole chip.
//
//   reference: model/geoformer/geodesic_utils.py:91-164 (cal_geodesic_vectorize), unique_with_inds :4-8
//
// The reference's hop keeps, for every (point u, query q) reached in a hop, the FIRST entry of the flattened
// (frontier vertex ascending, neighbour rank ascending) candidate list: the lowest-index parent v of the previous
// hop's frontier of q that lists u inside the radius, then its lowest rank r; geo[q][u] = geo[q][v] + D[v][r].
// The per-query kernels of geodesic.hip walk each query's frontier top-down and resolve that minimum with one 8-byte
// atomicMin per edge and query (4 GB of scattered traffic for 0.24 GB of algorithmic bytes, 1 workgroup per query
// walking <= 256 hops of ~5 us).  Here the search is turned around (bottom-up) and the queries are bit lanes:
//
//   F[h][u]   W = ceil(nq/32) words: bit q set iff u is in query q's frontier of hop h
//   vis[u]    W words: bit q set iff geo[q][u] is assigned
//   in-list   of u: every (v, r) with I[v][r] = u, r >= 1, D[v][r] <= radius, sorted by (v, r) -- a reverse CSR of the
//             kNN graph, built once per scene (count / scan / fill / rank-sort), the edge length stored beside it
//   hop h     one thread per (u, word w): walk u's in-list in order; new = F[h-1][v].w & open; every new bit q gets
//             dist[u][q] = dist[v][q] + edge and closes; F[h][u].w = all bits taken, vis[u].w |= taken
//
// "First in-neighbour in ascending (v, r) whose frontier bit is set" is exactly the entry the reference keeps, so the
// distances are the same fp32 sums along the same parent chains: bit-identical (tests/test_gpu_geodesic.py against
// the oracle, which the reference's own Python pins).  Every graph row is fetched once per hop for ALL queries, no
// atomics on the data path, and a hop is n*W independent threads (480 000 for the eval forward) instead of 256
// workgroups in lockstep with their own rings.  Distances are kept vertex-major ([n][32 W]: a vertex's words sit in
// one line next to its masks' order) during the search and transposed to the caller's [nq][n] at the end, where the
// unreached entries get their -1 from the vis masks -- no 61 MB fill before the search.
#include "common.h"

#define MS_THREADS 256
#define MS_ELL 16          // in-neighbours per vertex kept in fixed-width rows (the rest, rare, is read from the CSR)
#define MS_FLAG_SHARDS 64  // "this hop reached something" flag, sharded so the stores of a hop do not queue on one word

// ------------------------------------------------------------------------------------
// reverse CSR
// ------------------------------------------------------------------------------------
// one wave per row v, lane = column r
__global__ __launch_bounds__(MS_THREADS) void k_ms_count(const float* __restrict__ D, const int32_t* __restrict__ I,
                                                         int n, int K, float radius, int32_t* __restrict__ rcount) {
    const int v = blockIdx.x * (MS_THREADS / 64) + (threadIdx.x >> 6);
    const int r = threadIdx.x & 63;
    if (v >= n || r >= K || r == 0) return;
    const int u = I[(size_t)v * K + r];
    const float d = D[(size_t)v * K + r];
    if (u >= 0 && u < n && d <= radius) atomicAdd(&rcount[u], 1);
}

__global__ __launch_bounds__(MS_THREADS) void k_ms_fill(const float* __restrict__ D, const int32_t* __restrict__ I,
                                                        int n, int K, float radius, int32_t* __restrict__ rcur,
                                                        uint32_t* __restrict__ tkey, float* __restrict__ tdist) {
    const int v = blockIdx.x * (MS_THREADS / 64) + (threadIdx.x >> 6);
    const int r = threadIdx.x & 63;
    if (v >= n || r >= K || r == 0) return;
    const int u = I[(size_t)v * K + r];
    const float d = D[(size_t)v * K + r];
    if (u >= 0 && u < n && d <= radius) {
        const int pos = atomicAdd(&rcur[u], 1);
        tkey[pos] = ((uint32_t)v << 6) | (uint32_t)r;
        tdist[pos] = d;
    }
}

// rank-sort every in-list by key (keys are unique): 16 lanes per vertex; lists of up to 16 entries (nearly all) are
// ranked with shuffles, longer ones by counting over the list in memory
__global__ __launch_bounds__(MS_THREADS) void k_ms_sort(const int32_t* __restrict__ rstart, int n,
                                                        const uint32_t* __restrict__ tkey,
                                                        const float* __restrict__ tdist, uint32_t* __restrict__ rkey,
                                                        float* __restrict__ rdist, uint32_t* __restrict__ ell_v,
                                                        float* __restrict__ ell_d) {
    const int g = (blockIdx.x * MS_THREADS + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const bool live = g < n;
    const int beg = live ? rstart[g] : 0;
    const int d = live ? rstart[g + 1] - beg : 0;
    // (all 64 lanes take part in the shuffles: no early return)
    const bool mine = l < d;
    const uint32_t k0 = mine ? tkey[beg + l] : 0xffffffffu;
    const float d0 = mine ? tdist[beg + l] : 0.f;
    if (__builtin_expect(__all(d <= 16), 1)) {
        int rank = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t ki = (uint32_t)__shfl((int)k0, i, 16);
            rank += (ki < k0) ? 1 : 0;
        }
        if (mine) {
            rkey[beg + rank] = k0;
            rdist[beg + rank] = d0;
            ell_v[(size_t)g * MS_ELL + rank] = k0 >> 6;
            ell_d[(size_t)g * MS_ELL + rank] = d0;
        } else if (live) {
            ell_v[(size_t)g * MS_ELL + l] = 0xffffffffu;  // (l >= d: the padding)
            ell_d[(size_t)g * MS_ELL + l] = 0.f;
        }
        return;
    }
    if (d <= 16) {
        int rank = 0;
        for (int i = 0; i < d; i++) rank += (tkey[beg + i] < k0) ? 1 : 0;
        if (mine) {
            rkey[beg + rank] = k0;
            rdist[beg + rank] = d0;
            ell_v[(size_t)g * MS_ELL + rank] = k0 >> 6;
            ell_d[(size_t)g * MS_ELL + rank] = d0;
        } else if (live) {
            ell_v[(size_t)g * MS_ELL + l] = 0xffffffffu;
            ell_d[(size_t)g * MS_ELL + l] = 0.f;
        }
        return;
    }
    for (int j = l; j < d; j += 16) {
        const uint32_t kj = tkey[beg + j];
        const float dj = tdist[beg + j];
        int rank = 0;
        for (int i = 0; i < d; i++) rank += (tkey[beg + i] < kj) ? 1 : 0;
        rkey[beg + rank] = kj;
        rdist[beg + rank] = dj;
        if (rank < MS_ELL) {
            ell_v[(size_t)g * MS_ELL + rank] = kj >> 6;
            ell_d[(size_t)g * MS_ELL + rank] = dj;
        }
    }
}

// ------------------------------------------------------------------------------------
// the search
// ------------------------------------------------------------------------------------
// hop 0: F[0][u].w = the queries whose source is u; vis = the same (+ the bits beyond nq, so that a finished word
// reads "nothing open"); dist[src[q]][q] = 0
__global__ __launch_bounds__(MS_THREADS) void k_ms_init(const int32_t* __restrict__ src, int nq, int n, int W, int S,
                                                        uint32_t* __restrict__ F0, uint32_t* __restrict__ vis,
                                                        float* __restrict__ dist_t) {
    const int t = blockIdx.x * MS_THREADS + threadIdx.x;
    const int u = t / W, w = t - u * W;
    uint32_t bits = 0;
    const int q0 = w * 32;
    if (t < n * W) {
        for (int b = 0; b < 32; b++) {
            const int q = q0 + b;
            if (q < nq && src[q] == u) {
                bits |= 1u << b;
                dist_t[(size_t)u * S + q] = 0.0f;
            }
        }
        const int rem = nq - q0;  // >= 1: W = ceil(nq / 32)
        const uint32_t invalid = rem >= 32 ? 0u : ~((1u << rem) - 1u);
        F0[t] = bits;
        vis[t] = bits | invalid;
    }
}

template <int WC>  // WC > 0: W known at compile time
__global__ __launch_bounds__(MS_THREADS) void k_ms_hop(const int32_t* __restrict__ rstart,
                                                       const uint32_t* __restrict__ rkey,
                                                       const float* __restrict__ rdist,
                                                       const uint32_t* __restrict__ Fcur, uint32_t* __restrict__ Fnext,
                                                       uint32_t* __restrict__ vis, float* __restrict__ dist_t, int n,
                                                       int Wrt, int S, const int32_t* __restrict__ flag_prev,
                                                       int32_t* __restrict__ flag_cur) {
    const int W = WC > 0 ? WC : Wrt;
    if (flag_prev) {  // the previous hop reached nothing: the search is over (uniform over the launch)
        const int f = flag_prev[threadIdx.x & (MS_FLAG_SHARDS - 1)];
        if (!__any(f != 0)) return;
    }
    // blocks b, b + 8, b + 16, ... share an XCD (and its L2): give each XCD one contiguous range of vertices, whose
    // in-neighbours are mostly its own, instead of every eighth block of the whole scene (speed only)
    const int per_xcd = gridDim.x >> 3;
    const int vb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int t = vb * MS_THREADS + threadIdx.x;
    uint32_t taken = 0;
    if (t < n * W) {
        const int u = t / W, w = t - u * W;
        uint32_t open = ~vis[t];
        if (open) {
            int e = rstart[u];
            const int end = rstart[u + 1];
            float* du = dist_t + (size_t)u * S + w * 32;
            constexpr int B = 8;
            for (; e < end; e += B) {
                uint32_t key[B], f[B];
#pragma unroll
#ifdef MS_EXP_NOKEY
                for (int j = 0; j < B; j++) key[j] = (uint32_t)(e + j) << 6;
#else
                for (int j = 0; j < B; j++) key[j] = rkey[min(e + j, end - 1)];
#endif
#pragma unroll
#ifdef MS_EXP_NOF
                for (int j = 0; j < B; j++) f[j] = (key[j] == 0xfffffff0u) ? Fcur[0] : 0u;
#else
                for (int j = 0; j < B; j++) f[j] = Fcur[(size_t)(key[j] >> 6) * W + w];
#endif
#pragma unroll
                for (int j = 0; j < B; j++) {
                    uint32_t nw = (e + j < end) ? (f[j] & open) : 0u;
                    if (nw) {
                        const float ed = rdist[e + j];
                        const float* dv = dist_t + (size_t)(key[j] >> 6) * S + w * 32;
                        open &= ~nw;
                        taken |= nw;
                        do {
                            const int b = __builtin_ctz(nw);
                            nw &= nw - 1;
                            du[b] = ed + dv[b];
                        } while (nw);
                    }
                }
                if (!open) break;
            }
            if (taken) vis[t] = ~open;
        }
        Fnext[t] = taken;
    }
    if (__any(taken != 0) && (threadIdx.x & 63) == 0) flag_cur[vb & (MS_FLAG_SHARDS - 1)] = 1;
}

// The same hop over fixed-width in-lists (W = 4 or 8 words).  What a hop costs is the DEPTH of its chain of dependent
// loads, each a trip to the memory side (the masks were written by the previous launch, on other XCDs): in k_ms_hop
// vis / rstart -> keys -> masks -> distances; here the 16 parents of a vertex sit at a fixed address, so vis and the
// parents are requested together and the masks follow: two levels for the (vertex, word) pairs -- most of them -- that
// take nothing in this hop.  Every lane of a vertex's group reads the same 64-byte parent row (four 16-byte loads).
template <int W>
__global__ __launch_bounds__(MS_THREADS) void k_ms_hop_ell(const uint32_t* __restrict__ ell_v,
                                                           const float* __restrict__ ell_d,
                                                           const int32_t* __restrict__ rstart,
                                                           const uint32_t* __restrict__ rkey,
                                                           const float* __restrict__ rdist,
                                                           const uint32_t* __restrict__ Fcur,
                                                           uint32_t* __restrict__ Fnext, uint32_t* __restrict__ vis,
                                                           float* __restrict__ dist_t, int n, int S,
                                                           const int32_t* __restrict__ flag_prev,
                                                           int32_t* __restrict__ flag_cur) {
    if (flag_prev) {
        const int f = flag_prev[threadIdx.x & (MS_FLAG_SHARDS - 1)];
        if (!__any(f != 0)) return;
    }
    const int per_xcd = gridDim.x >> 3;
    const int vb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int t = vb * MS_THREADS + threadIdx.x;
    uint32_t taken = 0;
    if (t < n * W) {
        const int u = t / W, w = t - u * W;
        // level 1: the word's visited bits and the vertex's 16 parents, requested together (no branch may separate the
        // requests: the compiler would wait for the first before it issues the second)
        const uint4* ev = reinterpret_cast<const uint4*>(ell_v + (size_t)u * MS_ELL);
        const uint32_t vw = vis[t];
        const uint4 p0 = ev[0], p1 = ev[1], p2 = ev[2], p3 = ev[3];
        uint32_t pv[MS_ELL] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w,
                               p2.x, p2.y, p2.z, p2.w, p3.x, p3.y, p3.z, p3.w};
        // level 2: the parents' frontier words, all in flight at once; a missing parent reads the vertex's own word
        // (an address that is valid and warm) and is masked afterwards
        uint32_t f[MS_ELL];
#ifdef MS_EXP_STATIC
        const uint32_t* Fsrc = reinterpret_cast<const uint32_t*>(ell_d);
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = Fsrc[(size_t)(pv[j] != 0xffffffffu ? pv[j] : (uint32_t)u) * W + w] & (pv[j] == 0xfffffff0u ? 1u : 0u);
#else
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = Fcur[(size_t)(pv[j] != 0xffffffffu ? pv[j] : (uint32_t)u) * W + w];
#endif
        const bool more = __any(pv[8] != 0xffffffffu);  // (wave-uniform: most waves stop at 8 parents)
        if (more) {
#pragma unroll
            for (int j = 8; j < 16; j++) f[j] = Fcur[(size_t)(pv[j] != 0xffffffffu ? pv[j] : (uint32_t)u) * W + w];
        } else {
#pragma unroll
            for (int j = 8; j < 16; j++) f[j] = 0u;
        }
        // which bits each parent hands over: pure ALU, in list order (an earlier parent closes the bit for the later
        // ones).  The first two parents that hand something over are kept as items (v, bits, slot) in plain registers
        // (compile-time slots only: a lane-dependent index into pv[] would move the array to scratch memory).
        uint32_t open = ~vw;
        uint32_t nwj[MS_ELL];
        uint32_t m = 0;  // bit j: parent j hands something over
#pragma unroll
        for (int j = 0; j < MS_ELL; j++) {
            const uint32_t nw = (pv[j] != 0xffffffffu ? f[j] : 0u) & open;
            open &= ~nw;
            taken |= nw;
            nwj[j] = nw;
            m |= (nw != 0 ? 1u : 0u) << j;
        }
        const uint32_t m1 = m & (m - 1);
        const uint32_t rest = m1 & (m1 - 1);  // a third, fourth ... handing parent (rare), done one by one below
        const int j0 = m ? __builtin_ctz(m) : MS_ELL, j1 = m1 ? __builtin_ctz(m1) : MS_ELL;
        const int cnt = (int)m;
        uint32_t v0 = (uint32_t)u, v1 = (uint32_t)u, n0 = 0, n1 = 0;
#pragma unroll
        for (int j = 0; j < MS_ELL; j++) {
            v0 = (j == j0) ? pv[j] : v0;
            n0 = (j == j0) ? nwj[j] : n0;
            v1 = (j == j1) ? pv[j] : v1;
            n1 = (j == j1) ? nwj[j] : n1;
        }
        // the distances of the new bits.  A wave's lanes that took something did so from different parents: walking
        // the 16 parents with a load -> wait -> store inside each step costs a trip to memory per step (that WAS the
        // hop: 5 of its 7 us).  Here the wave requests every lane's first two items together.
#if defined(MS_EXP_SMALLWR)
        float* du = dist_t + (size_t)(u & 1023) * S + w * 32;
#else
        float* du = dist_t + (size_t)u * S + w * 32;
#endif
#if defined(MS_DBG_A)
        if (cnt != 0) {
#else
        if (__any(cnt != 0)) {
#endif
            const int b0 = n0 ? __builtin_ctz(n0) : 0;
            const int b1 = n1 ? __builtin_ctz(n1) : 0;
            const float e0 = ell_d[(size_t)u * MS_ELL + (j0 & (MS_ELL - 1))], e1 = ell_d[(size_t)u * MS_ELL + (j1 & (MS_ELL - 1))];
            // (a lane without an item reads dist_t[0]: one line for the whole wave instead of a line of its own)
#if defined(MS_EXP_SMALLRD)
            const size_t a0 = n0 ? (size_t)(v0 & 1023u) * S + w * 32 + b0 : (size_t)0;
            const size_t a1 = n1 ? (size_t)(v1 & 1023u) * S + w * 32 + b1 : a0;
#else
            const size_t a0 = n0 ? (size_t)v0 * S + w * 32 + b0 : (size_t)0;
            const size_t a1 = n1 ? (size_t)v1 * S + w * 32 + b1 : a0;
#endif
#if defined(MS_EXP_NORD)
            const float d0 = (float)a0, d1 = (float)a1;
#else
            const float d0 = dist_t[a0], d1 = dist_t[a1];
#endif
#if defined(MS_DBG_B)
            if (n0) du[b0] = ell_d[(size_t)u * MS_ELL + (j0 & (MS_ELL - 1))] + dist_t[(size_t)v0 * S + w * 32 + b0];
            if (n1) du[b1] = ell_d[(size_t)u * MS_ELL + (j1 & (MS_ELL - 1))] + dist_t[(size_t)v1 * S + w * 32 + b1];
#elif defined(MS_EXP_NOWR)
            if (n0 && e0 + d0 == 12345.f) du[b0] = e0 + d0;
            if (n1 && e1 + d1 == 12345.f) du[b1] = e1 + d1;
#else
            if (n0) du[b0] = e0 + d0;
            if (n1) du[b1] = e1 + d1;
#endif
            // further bits of the same two parents (rare: two queries of one word reach u from the same parent)
            uint32_t r0 = n0 & (n0 - 1), r1 = n1 & (n1 - 1);
            while (r0) {
                const int b = __builtin_ctz(r0);
                r0 &= r0 - 1;
                du[b] = e0 + dist_t[(size_t)v0 * S + w * 32 + b];
            }
            while (r1) {
                const int b = __builtin_ctz(r1);
                r1 &= r1 - 1;
                du[b] = e1 + dist_t[(size_t)v1 * S + w * 32 + b];
            }
#if defined(MS_DBG_C)
            if (rest != 0) {
#else
            if (__any(rest != 0)) {
#endif
                uint32_t op2 = ~vw;  // replay the list: the bits parent j hands over are f[j] & (what was open before j)
#pragma unroll
                for (int j = 0; j < MS_ELL; j++) {
                    uint32_t nw = (pv[j] != 0xffffffffu ? f[j] : 0u) & op2;
                    op2 &= ~nw;
                    if ((rest >> j) & 1u) {
                        const float ed = ell_d[(size_t)u * MS_ELL + j];
                        const float* dv = dist_t + (size_t)pv[j] * S + w * 32;
                        do {
                            const int b = __builtin_ctz(nw);
                            nw &= nw - 1;
                            du[b] = ed + dv[b];
                        } while (nw);
                    }
                }
            }
        }
        if (open && pv[MS_ELL - 1] != 0xffffffffu) {  // a long in-list: the rest from the CSR
            int e = rstart[u] + MS_ELL;
            const int end = rstart[u + 1];
            for (; e < end && open; e++) {
                const uint32_t v = rkey[e] >> 6;
                uint32_t nw = Fcur[(size_t)v * W + w] & open;
                if (nw) {
                    const float ed = rdist[e];
                    const float* dv = dist_t + (size_t)v * S + w * 32;
                    open &= ~nw;
                    taken |= nw;
                    do {
                        const int b = __builtin_ctz(nw);
                        nw &= nw - 1;
                        du[b] = ed + dv[b];
                    } while (nw);
                }
            }
        }
        if (taken) vis[t] = ~open;
        Fnext[t] = taken;
    }
#if !defined(MS_EXP_NOFLAG)
    if (__any(taken != 0) && (threadIdx.x & 63) == 0) flag_cur[vb & (MS_FLAG_SHARDS - 1)] = 1;
#endif
}

// geo[q][u] = vis[u] bit q ? dist[u][q] : -1 : 64 vertices x 64 queries per block through LDS
__global__ __launch_bounds__(MS_THREADS) void k_ms_transpose(const float* __restrict__ dist_t,
                                                             const uint32_t* __restrict__ vis, int n, int nq, int W,
                                                             int S, float* __restrict__ geo) {
    __shared__ float tile[64][65];
    const int u0 = blockIdx.x * 64, q0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int u = u0 + r, q = q0 + tx;
        float val = -1.0f;
        if (u < n && q < nq) {
            const uint32_t m = vis[(size_t)u * W + (q >> 5)];
            if ((m >> (q & 31)) & 1u) val = dist_t[(size_t)u * S + q];
        }
        tile[r][tx] = val;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int q = q0 + r, u = u0 + tx;
        if (q < nq && u < n) geo[(size_t)q * n + u] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------
static size_t ms_align(size_t b) { return (b + 255) & ~(size_t)255; }
struct MsLayout {
    size_t rcount, rstart, rcur, bsum, tkey, tdist, rkey, rdist, ell_v, ell_d, F0, F1, vis, flags, dist_t, total;
    int W, S;
    size_t E;
};
static MsLayout ms_layout(int n, int K, int nq, int max_step) {
    MsLayout L;
    L.W = (nq + 31) / 32;
    if (L.W < 1) L.W = 1;
    L.S = L.W * 32;
    L.E = (size_t)n * (size_t)(K - 1);
    size_t o = 0;
    L.rcount = o; o += ms_align((size_t)(n + 1) * 4);
    L.rstart = o; o += ms_align((size_t)(n + 2) * 4);
    L.rcur = o; o += ms_align((size_t)(n + 1) * 4);
    L.bsum = o; o += ms_align((size_t)2 * gf_iscan_blocks(n) * 4 + 64);
    L.tkey = o; o += ms_align(L.E * 4);
    L.tdist = o; o += ms_align(L.E * 4);
    L.rkey = o; o += ms_align(L.E * 4);
    L.rdist = o; o += ms_align(L.E * 4);
    L.ell_v = o; o += ms_align((size_t)n * MS_ELL * 4);
    L.ell_d = o; o += ms_align((size_t)n * MS_ELL * 4);
    L.F0 = o; o += ms_align((size_t)n * L.W * 4);
    L.F1 = o; o += ms_align((size_t)n * L.W * 4);
    L.vis = o; o += ms_align((size_t)n * L.W * 4);
    L.flags = o; o += ms_align((size_t)(max_step + 2) * MS_FLAG_SHARDS * 4);
    L.dist_t = o; o += ms_align((size_t)n * L.S * 4);
    L.total = o;
    return L;
}

extern "C" size_t gf_geodesic_ms_scratch_bytes(int n, int K, int nq, int max_step) {
    if (n < 1 || K < 2 || nq < 1 || max_step < 0) return 0;
    return ms_layout(n, K, nq, max_step).total;
}

extern "C" int gf_geodesic_bfs_ms(const float* D, const int32_t* I, int n, int K, const int32_t* src, int nq,
                                  float radius, int max_step, float* geo, void* scratch, size_t scratch_bytes,
                                  void* stream) {
    GF_CHECK_ARG(n >= 1 && K >= 2 && K <= 64 && nq >= 0 && max_step >= 0, "gf_geodesic_bfs_ms: bad arguments");
    GF_CHECK_ARG(n < (1 << 26), "gf_geodesic_bfs_ms: n=%d exceeds the 26-bit parent field", n);
    if (nq == 0) return GF_OK;
    const MsLayout L = ms_layout(n, K, nq, max_step);
    GF_CHECK_ARG(D && I && src && geo && scratch, "gf_geodesic_bfs_ms: null pointer");
    GF_CHECK_ARG(scratch_bytes >= L.total, "gf_geodesic_bfs_ms: scratch of %zu bytes, %zu needed", scratch_bytes, L.total);
    GF_CHECK_ARG((size_t)n * L.W < (size_t)0x7fffffff, "gf_geodesic_bfs_ms: n * words overflows");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)scratch;
    int32_t* rcount = (int32_t*)(base + L.rcount);
    int32_t* rstart = (int32_t*)(base + L.rstart);
    int32_t* rcur = (int32_t*)(base + L.rcur);
    int32_t* bsum = (int32_t*)(base + L.bsum);
    uint32_t* tkey = (uint32_t*)(base + L.tkey);
    float* tdist = (float*)(base + L.tdist);
    uint32_t* rkey = (uint32_t*)(base + L.rkey);
    float* rdist = (float*)(base + L.rdist);
    uint32_t* ell_v = (uint32_t*)(base + L.ell_v);
    float* ell_d = (float*)(base + L.ell_d);
    uint32_t* F[2] = {(uint32_t*)(base + L.F0), (uint32_t*)(base + L.F1)};
    uint32_t* vis = (uint32_t*)(base + L.vis);
    int32_t* flags = (int32_t*)(base + L.flags);
    float* dist_t = (float*)(base + L.dist_t);
    const int W = L.W, S = L.S;

    GF_TRY(hipMemsetAsync(rcount, 0, (size_t)(n + 1) * 4, st));
    GF_TRY(hipMemsetAsync(flags, 0, (size_t)(max_step + 2) * MS_FLAG_SHARDS * 4, st));
    const int rows_grid = gf_div_up(n, MS_THREADS / 64);
    hipLaunchKernelGGL(k_ms_count, dim3(rows_grid), dim3(MS_THREADS), 0, st, D, I, n, K, radius, rcount);
    gf_iscan(rcount, n, rstart, rcur, bsum, bsum + gf_iscan_blocks(n), st);
    hipLaunchKernelGGL(k_ms_fill, dim3(rows_grid), dim3(MS_THREADS), 0, st, D, I, n, K, radius, rcur, tkey, tdist);
    hipLaunchKernelGGL(k_ms_sort, dim3(gf_div_up((long long)n * 16, MS_THREADS)), dim3(MS_THREADS), 0, st, rstart, n,
                       tkey, tdist, rkey, rdist, ell_v, ell_d);
    const int grid = (gf_div_up((long long)n * W, MS_THREADS) + 7) & ~7;  // (a multiple of 8: k_ms_hop's XCD mapping)
    hipLaunchKernelGGL(k_ms_init, dim3(grid), dim3(MS_THREADS), 0, st, src, nq, n, W, S, F[0], vis, dist_t);
    for (int h = 1; h <= max_step; h++) {
        const int32_t* fp = h >= 2 ? flags + (size_t)(h - 1) * MS_FLAG_SHARDS : nullptr;
        int32_t* fc = flags + (size_t)h * MS_FLAG_SHARDS;
        const uint32_t* Fc = F[(h - 1) & 1];
        uint32_t* Fn = F[h & 1];
        if (W == 8)
            hipLaunchKernelGGL(k_ms_hop_ell<8>, dim3(grid), dim3(MS_THREADS), 0, st, ell_v, ell_d, rstart, rkey, rdist, Fc,
                               Fn, vis, dist_t, n, S, fp, fc);
        else if (W == 4)
            hipLaunchKernelGGL(k_ms_hop_ell<4>, dim3(grid), dim3(MS_THREADS), 0, st, ell_v, ell_d, rstart, rkey, rdist, Fc,
                               Fn, vis, dist_t, n, S, fp, fc);
        else
            hipLaunchKernelGGL(k_ms_hop<0>, dim3(grid), dim3(MS_THREADS), 0, st, rstart, rkey, rdist, Fc, Fn, vis, dist_t,
                               n, W, S, fp, fc);
    }
    hipLaunchKernelGGL(k_ms_transpose, dim3(gf_div_up(n, 64), gf_div_up(nq, 64)), dim3(MS_THREADS), 0, st, dist_t, vis,
                       n, nq, W, S, geo);
    GF_CHECK_LAUNCH("gf_geodesic_bfs_ms");
    return GF_OK;
}
