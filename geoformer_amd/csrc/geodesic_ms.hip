// Geodesic BFS, multi-source form: all nq queries of a scene advance together, one launch per hop over the whole chip.
//
//   reference: model/geoformer/geodesic_utils.py:91-164 (cal_geodesic_vectorize), unique_with_inds :4-8
//
// The reference's hop keeps, for every (point u, query q) reached in a hop, the FIRST entry of the flattened
// (frontier vertex ascending, neighbour rank ascending) candidate list: the lowest-index parent v of the previous
// hop's frontier of q that lists u inside the radius, then its lowest rank r; geo[q][u] = geo[q][v] + D[v][r].
// The per-query kernels of geodesic.hip walk each query's frontier top-down and resolve that minimum with one 8-byte
// atomicMin per edge and query (4 GB of scattered traffic for 0.24 GB of algorithmic bytes, 1 workgroup per query
// walking <= 256 hops of ~4 us).  Here the search is turned around (bottom-up) and the queries are bit lanes:
//
//   R[h][u]   ceil(nq/64) 64-bit words: bit q set iff query q has reached u within h hops (cumulative)
//   in-list   of u: every (v, r) with I[v][r] = u, r >= 1, D[v][r] <= radius, sorted by (v, r) -- a reverse CSR of the
//             kNN graph, built once per scene (count / scan / fill / rank-sort), the edge length stored beside it; the
//             first 16 entries also as fixed-width rows (parent ids and edge lengths), padded with the id n, whose
//             mask row is all zero
//   hop h     one thread per (u, word): R[h][u] = R[h-1][u] | OR of the parents' R[h-1] words.  A bit that a parent has
//             and u has not is NEW, and that parent reached it exactly in hop h-1 (had it been earlier, u would have it
//             by now): so "the first in-neighbour, ascending (v, r), that has the bit" is exactly the entry the
//             reference keeps, and dist[u][q] = dist[v][q] + edge for it.  No frontier and no visited array: one
//             cumulative mask, double-buffered.
//
// The distances are the same fp32 sums along the same parent chains: bit-identical (tests/test_gpu_geodesic.py against
// the oracle, which the reference's own Python pins).  Every graph row is fetched once per hop for ALL queries, there
// is no atomic on the data path, and a hop is n * ceil(nq/64) independent threads.  Distances are kept vertex-major
// ([n][64 W]) during the search and transposed to the caller's [nq][n] at the end, where the unreached entries get
// their -1 from the masks -- no 61 MB fill before the search.
//
// What a hop costs (profiles/r5_bfs_ms_notes.md): 1.5 us of launch boundary + the kernel; the kernel was first bound by
// VALU issue, not by memory (L1-miss latency 360 cycles by TCP_TCC_READ_REQ_LATENCY, 1.36 M vector instructions per hop:
// 16 unrolled parent slots x 8 word lanes per vertex of address arithmetic and select chains).  Hence: 64-bit words
// (half the lanes), 32-bit offsets from uniform bases, the plain OR as the common path, and the search for the
// contributing parent only in lanes that took a bit.
#include <cstdlib>
#include "common.h"

#define MS_THREADS 256
#define MS_ELL 16          // in-neighbours per vertex kept in fixed-width rows (the rest, rare, is read from the CSR)
// Is the search over?  No hop sets a flag (7 500 waves storing into a few words is a hot spot of its own); a small
// kernel compares the two mask buffers every MS_CHECK_EVERY hops and leaves `alive` at zero when they are equal: the
// hops behind it return at once.
#define MS_CHECK_EVERY 16

// ------------------------------------------------------------------------------------
// spatial working order (optional: when the caller hands the coordinates over)
// ------------------------------------------------------------------------------------
// The tile form of the hop wants the vertices of a tile to be neighbours in space: vertices are ordered by the Morton
// code of their 0.2 m cell (6 bits per axis over the scene's bounding box, clamped beyond 12.8 m), any order inside a
// cell (an atomic cursor: the results do not depend on it).  perm[position] = vertex, inv[vertex] = position.
#define MS_MBITS 6
#define MS_MCELLS (1 << (3 * MS_MBITS))
__device__ __forceinline__ int ms_ord(float x) {  // float -> int with the same order
    const int i = __float_as_int(x);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__global__ __launch_bounds__(MS_THREADS) void k_ms_bbox(const float* __restrict__ xyz, int n, int32_t* __restrict__ lo) {
    const int i = blockIdx.x * MS_THREADS + threadIdx.x;
    int a = 0x7fffffff, b = 0x7fffffff, c = 0x7fffffff;
    if (i < n) {
        a = ms_ord(xyz[i * 3 + 0]);
        b = ms_ord(xyz[i * 3 + 1]);
        c = ms_ord(xyz[i * 3 + 2]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a = min(a, __shfl_xor(a, d, 64));
        b = min(b, __shfl_xor(b, d, 64));
        c = min(c, __shfl_xor(c, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&lo[0], a);
        atomicMin(&lo[1], b);
        atomicMin(&lo[2], c);
    }
}
__device__ __forceinline__ uint32_t ms_spread3(uint32_t v) {  // 6 bits -> every third bit
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < MS_MBITS; b++) r |= ((v >> b) & 1u) << (3 * b);
    return r;
}
__global__ __launch_bounds__(MS_THREADS) void k_ms_mkey(const float* __restrict__ xyz, int n,
                                                       const int32_t* __restrict__ lo, uint32_t* __restrict__ mkey,
                                                       int32_t* __restrict__ mcount) {
    const int i = blockIdx.x * MS_THREADS + threadIdx.x;
    if (i >= n) return;
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int o = lo[a];
        const float base = __int_as_float(o >= 0 ? o : o ^ 0x7fffffff);
        const float t = (xyz[i * 3 + a] - base) * 5.0f;  // 0.2 m cells
        c[a] = (uint32_t)fminf(fmaxf(t, 0.0f), (float)((1 << MS_MBITS) - 1));
    }
    const uint32_t key = ms_spread3(c[0]) | (ms_spread3(c[1]) << 1) | (ms_spread3(c[2]) << 2);
    mkey[i] = key;
    atomicAdd(&mcount[key], 1);
}
__global__ __launch_bounds__(MS_THREADS) void k_ms_mfill(const uint32_t* __restrict__ mkey, int n,
                                                        int32_t* __restrict__ mcur, int32_t* __restrict__ perm,
                                                        int32_t* __restrict__ inv) {
    const int i = blockIdx.x * MS_THREADS + threadIdx.x;
    if (i >= n) return;
    const int pos = atomicAdd(&mcur[mkey[i]], 1);
    perm[pos] = i;
    inv[i] = pos;
}

// ------------------------------------------------------------------------------------
// reverse CSR
// ------------------------------------------------------------------------------------
// one wave per row v, lane = column r
// (inv: vertex -> its position in the spatial order the search works in, or nullptr = scene order)
__global__ __launch_bounds__(MS_THREADS) void k_ms_count(const float* __restrict__ D, const int32_t* __restrict__ I,
                                                         int n, int K, float radius, const int32_t* __restrict__ inv,
                                                         int32_t* __restrict__ rcount) {
    const int v = blockIdx.x * (MS_THREADS / 64) + (threadIdx.x >> 6);
    const int r = threadIdx.x & 63;
    if (v >= n || r >= K || r == 0) return;
    const int u = I[(size_t)v * K + r];
    const float d = D[(size_t)v * K + r];
    if (u >= 0 && u < n && d <= radius) atomicAdd(&rcount[inv ? inv[u] : u], 1);
}

__global__ __launch_bounds__(MS_THREADS) void k_ms_fill(const float* __restrict__ D, const int32_t* __restrict__ I,
                                                        int n, int K, float radius, const int32_t* __restrict__ inv,
                                                        int32_t* __restrict__ rcur, uint32_t* __restrict__ tkey,
                                                        float* __restrict__ tdist) {
    const int v = blockIdx.x * (MS_THREADS / 64) + (threadIdx.x >> 6);
    const int r = threadIdx.x & 63;
    if (v >= n || r >= K || r == 0) return;
    const int u = I[(size_t)v * K + r];
    const float d = D[(size_t)v * K + r];
    if (u >= 0 && u < n && d <= radius) {
        const int pos = atomicAdd(&rcur[inv ? inv[u] : u], 1);
        tkey[pos] = ((uint32_t)v << 6) | (uint32_t)r;  // (the ORIGINAL parent id: the order the reference resolves ties in)
        tdist[pos] = d;
    }
}

// rank-sort every in-list by key (keys are unique): 16 lanes per vertex; lists of up to 16 entries (nearly all) are
// ranked with shuffles, longer ones by counting over the list in memory
__global__ __launch_bounds__(MS_THREADS) void k_ms_sort(const int32_t* __restrict__ rstart, int n,
                                                        const uint32_t* __restrict__ tkey,
                                                        const float* __restrict__ tdist, uint32_t* __restrict__ rkey,
                                                        float* __restrict__ rdist, uint32_t* __restrict__ ell_v,
                                                        float* __restrict__ ell_d, const int32_t* __restrict__ inv) {
    // sorted by the original (parent, rank); what is STORED is the parent's position in the working order
    auto tr = [&](uint32_t key) -> uint32_t { return inv ? (((uint32_t)inv[key >> 6] << 6) | (key & 63u)) : key; };
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
            const uint32_t kt = tr(k0);
            rkey[beg + rank] = kt;
            rdist[beg + rank] = d0;
            ell_v[(size_t)g * MS_ELL + rank] = kt >> 6;
            ell_d[(size_t)g * MS_ELL + rank] = d0;
        } else if (live) {
            ell_v[(size_t)g * MS_ELL + l] = (uint32_t)n;  // (l >= d: the padding: the id whose mask row is zero)
            ell_d[(size_t)g * MS_ELL + l] = 0.f;
        }
        return;
    }
    if (d <= 16) {
        int rank = 0;
        for (int i = 0; i < d; i++) rank += (tkey[beg + i] < k0) ? 1 : 0;
        if (mine) {
            const uint32_t kt = tr(k0);
            rkey[beg + rank] = kt;
            rdist[beg + rank] = d0;
            ell_v[(size_t)g * MS_ELL + rank] = kt >> 6;
            ell_d[(size_t)g * MS_ELL + rank] = d0;
        } else if (live) {
            ell_v[(size_t)g * MS_ELL + l] = (uint32_t)n;
            ell_d[(size_t)g * MS_ELL + l] = 0.f;
        }
        return;
    }
    for (int j = l; j < d; j += 16) {
        const uint32_t kj = tkey[beg + j];
        const float dj = tdist[beg + j];
        int rank = 0;
        for (int i = 0; i < d; i++) rank += (tkey[beg + i] < kj) ? 1 : 0;
        const uint32_t kt = tr(kj);
        rkey[beg + rank] = kt;
        rdist[beg + rank] = dj;
        if (rank < MS_ELL) {
            ell_v[(size_t)g * MS_ELL + rank] = kt >> 6;
            ell_d[(size_t)g * MS_ELL + rank] = dj;
        }
    }
}

// ------------------------------------------------------------------------------------
// the search
// ------------------------------------------------------------------------------------
typedef unsigned long long ms_word;

// hop 0: R[0][u] = the queries whose source is u (+ the bits beyond nq, which then never read as new);
// dist[src[q]][q] = 0; row n (the padding parent) stays zero in both buffers
// (nsets source sets: bit q has one source per set -- the scenes of a batch, whose graphs do not touch)
__global__ __launch_bounds__(MS_THREADS) void k_ms_init(const int32_t* __restrict__ src, int nq, int nsets, int n, int W,
                                                        int S, const int32_t* __restrict__ perm,
                                                        ms_word* __restrict__ R0, ms_word* __restrict__ R1,
                                                        float* __restrict__ dist_t) {
    const int t = blockIdx.x * MS_THREADS + threadIdx.x;
    if (t >= (n + 1) * W) return;
    const int u = t / W, w = t - u * W;
    ms_word bits = 0;
    if (u < n) {
        const int q0 = w * 64;
        const int uo = perm ? perm[u] : u;  // the vertex this working position holds
        for (int b = 0; b < 64; b++) {
            const int q = q0 + b;
            bool hit = false;
            if (q < nq)
                for (int k = 0; k < nsets; k++) hit = hit || src[(size_t)k * nq + q] == uo;
            if (hit) {
                bits |= 1ull << b;
                dist_t[(size_t)u * S + q] = 0.0f;
            }
        }
        const int rem = nq - q0;  // >= 1: W = ceil(nq / 64)
        if (rem < 64) bits |= ~((1ull << rem) - 1ull);
    }
    R0[t] = bits;
    if (u == n) R1[t] = 0;
}

// One hop.  Thread t = (vertex u, 64-bit word w).  W64 > 0: words per vertex known at compile time.
template <int W64>
__global__ __launch_bounds__(MS_THREADS) void k_ms_hop(const uint32_t* __restrict__ ell_v,
                                                       const float* __restrict__ ell_d,
                                                       const int32_t* __restrict__ rstart,
                                                       const uint32_t* __restrict__ rkey,
                                                       const float* __restrict__ rdist,
                                                       const ms_word* __restrict__ Rcur, ms_word* __restrict__ Rnext,
                                                       float* __restrict__ dist_t, int n, int Wrt,
                                                       const int32_t* __restrict__ alive
#ifdef MS_TRACE
                                                       , unsigned long long* __restrict__ trace
#endif
                                                       ) {
#ifdef MS_TRACE
#define MS_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (trace && (threadIdx.x & 63) == 0) trace[(size_t)(blockIdx.x * (MS_THREADS / 64) + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MS_STAMP(i) do {} while (0)
#endif
    MS_STAMP(0);
    if (alive && *alive == 0) return;  // an earlier k_ms_check found the last hop empty: the search is over
    const uint32_t W = W64 > 0 ? (uint32_t)W64 : (uint32_t)Wrt;
    const uint32_t S = W * 64u;
    // blocks b, b + 8, b + 16, ... share an XCD (and its L2): give each XCD one contiguous range of vertices, whose
    // in-neighbours are mostly its own, instead of every eighth block of the whole scene (speed only)
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t vb = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t t = vb * MS_THREADS + threadIdx.x;
    if (t >= (uint32_t)n * W) return;
    const uint32_t u = t / W, w = t - u * W;
    // level 1: the word itself and the vertex's first 8 parents, requested together
    const uint4* ev = reinterpret_cast<const uint4*>(ell_v + u * MS_ELL);
    const ms_word own = Rcur[t];
    const uint4 p0 = ev[0], p1 = ev[1];
    uint32_t pv[MS_ELL];
    pv[0] = p0.x; pv[1] = p0.y; pv[2] = p0.z; pv[3] = p0.w;
    pv[4] = p1.x; pv[5] = p1.y; pv[6] = p1.z; pv[7] = p1.w;
    MS_STAMP(1);
    // level 2: the parents' words, all in flight at once (a padding parent reads row n: zero)
    ms_word f[MS_ELL];
#pragma unroll
#if defined(MS_EXP_OWNF)
    for (int j = 0; j < 8; j++) f[j] = Rcur[(pv[j] != (uint32_t)n ? u : (uint32_t)n) * W + w];
#elif defined(MS_EXP_STATICF)
    for (int j = 0; j < 8; j++) f[j] = reinterpret_cast<const ms_word*>(ell_d)[(pv[j] >> 1) * W + w] & 1ull;
#else
    for (int j = 0; j < 8; j++) f[j] = Rcur[pv[j] * W + w];
#endif
    const bool more = __any(pv[7] != (uint32_t)n);  // (wave-uniform: most waves stop at 8 parents)
    if (more) {
        const uint4 p2 = ev[2], p3 = ev[3];
        pv[8] = p2.x; pv[9] = p2.y; pv[10] = p2.z; pv[11] = p2.w;
        pv[12] = p3.x; pv[13] = p3.y; pv[14] = p3.z; pv[15] = p3.w;
#pragma unroll
        for (int j = 8; j < 16; j++) f[j] = Rcur[pv[j] * W + w];
    } else {
#pragma unroll
        for (int j = 8; j < 16; j++) {
            pv[j] = (uint32_t)n;
            f[j] = 0;
        }
    }
    MS_STAMP(2);
    ms_word acc = own;
#pragma unroll
    for (int j = 0; j < MS_ELL; j++) acc |= f[j];
    ms_word taken = acc & ~own;
    float* du = dist_t + (size_t)u * S + w * 64u;
    if (__any(taken != 0)) {
        // the first parent (ascending) that has one of the new bits, and what it hands over; found by a select chain in
        // descending order over registers (a lane-dependent index into pv[] / f[] would move them to scratch memory)
        uint32_t v0 = u, j0 = 0;
        ms_word n0 = 0;
        if (more) {
#pragma unroll
            for (int j = MS_ELL - 1; j >= 8; j--) {
                const ms_word c = f[j] & taken;
                v0 = c ? pv[j] : v0;
                j0 = c ? (uint32_t)j : j0;
                n0 = c ? c : n0;
            }
        }
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            const ms_word c = f[j] & taken;
            v0 = c ? pv[j] : v0;
            j0 = c ? (uint32_t)j : j0;
            n0 = c ? c : n0;
        }
        // its distance: edge length and the parent's value requested together (a lane without an item reads element 0
        // of each array: one line for the whole wave)
        MS_STAMP(3);
        const uint32_t b0 = n0 ? (uint32_t)__builtin_ctzll(n0) : 0u;
        const float e0 = ell_d[n0 ? u * MS_ELL + j0 : 0u];
        const float d0 = dist_t[n0 ? (size_t)v0 * S + w * 64u + b0 : (size_t)0];
        MS_STAMP(4);
        if (n0) du[b0] = e0 + d0;
        ms_word r0 = n0 & (n0 - 1);  // further bits of the same parent (two queries of one word reach u from it: rare)
        while (r0) {
            const uint32_t b = (uint32_t)__builtin_ctzll(r0);
            r0 &= r0 - 1;
            du[b] = e0 + dist_t[(size_t)v0 * S + w * 64u + b];
        }
        // bits handed over by later parents (rare): in list order, one by one
        ms_word rem = taken & ~n0;
        if (__any(rem != 0)) {
#pragma unroll
            for (int j = 1; j < MS_ELL; j++) {
                ms_word c = f[j] & rem;
                if (c) {
                    rem &= ~c;
                    const float ed = ell_d[u * MS_ELL + j];
                    const float* dv = dist_t + (size_t)pv[j] * S + w * 64u;
                    do {
                        const uint32_t b = (uint32_t)__builtin_ctzll(c);
                        c &= c - 1;
                        du[b] = ed + dv[b];
                    } while (c);
                }
            }
        }
    }
    MS_STAMP(5);
    if (~acc != 0 && pv[MS_ELL - 1] != (uint32_t)n) {  // a long in-list: the rest from the CSR
        int e = rstart[u] + MS_ELL;
        const int end = rstart[u + 1];
        for (; e < end && ~acc != 0; e++) {
            const uint32_t v = rkey[e] >> 6;
            ms_word c = Rcur[v * W + w] & ~acc;
            if (c) {
                const float ed = rdist[e];
                const float* dv = dist_t + (size_t)v * S + w * 64u;
                acc |= c;
                do {
                    const uint32_t b = (uint32_t)__builtin_ctzll(c);
                    c &= c - 1;
                    du[b] = ed + dv[b];
                } while (c);
            }
        }
    }
#if defined(MS_EXP_NOSTORE)
    if (taken) Rnext[t] = acc;
#else
    Rnext[t] = acc;
#endif
    MS_STAMP(6);
}

// ------------------------------------------------------------------------------------
// tiles: MS_TV consecutive vertices per workgroup, the parents' words staged in LDS
// ------------------------------------------------------------------------------------
// What the hop above waits for is the texture-address unit: 8 parents x 4 word lanes per vertex of 8-byte gathers, 14
// waves per compute unit all at it at once (cycle stamps: 1 900 cycles for the first level of loads, 3 300 for the
// parents' words, against a mean L1-miss latency of 360).  Parents of neighbouring vertices are mostly the same
// vertices, so a workgroup that owns MS_TV consecutive vertices fetches every row it needs ONCE -- its own rows as one
// contiguous block, the rows of parents outside the tile (its "halo", a static list built once per scene) by one
// gather -- into LDS, and the per-parent reads become LDS reads through 16-bit local slots.
#define MS_TV 256
#define MS_HCAP 768                  // halo rows a tile holds in LDS; parents beyond it are read from memory (slot 0xfffe)
#define MS_ZROW (MS_TV + MS_HCAP)    // the LDS row that stays zero: the padding parent
#define MS_HASH 8192                 // >= 2 x the most parents a tile can have outside itself (MS_TV x 16)
#define MS_SLOT_FAR 0xfffeu

// per tile: the distinct out-of-tile parents of its vertices (halo_gid, nhalo) and every vertex's 16 local slots
#define MS_NBMAX 64      // neighbour tiles listed per tile (more: the tile waits for every tile, nnbr = -1)
// adj (optional, for k_ms_persist): [tiles][ceil(tiles/32)] bits, zeroed by the caller; bit (a, b) and bit (b, a) are set
// when tile a reads a row of tile b -- a tile must wait for the tiles it reads AND for the tiles that read it (they
// must be done with a published buffer before it is overwritten two hops later)
__global__ __launch_bounds__(MS_TV) void k_ms_tiles(const uint32_t* __restrict__ ell_v, int n,
                                                    const int32_t* __restrict__ rstart,
                                                    const uint32_t* __restrict__ rkey, uint16_t* __restrict__ slot16,
                                                    uint32_t* __restrict__ halo_gid, int32_t* __restrict__ nhalo,
                                                    uint32_t* __restrict__ adj, int adj_words) {
    __shared__ uint32_t s_key[MS_HASH];
    __shared__ uint16_t s_idx[MS_HASH];
    __shared__ int s_cnt;
    const int tile = blockIdx.x, lu = threadIdx.x;
    const uint32_t base = (uint32_t)tile * MS_TV, u = base + lu;
    for (int i = lu; i < MS_HASH; i += MS_TV) s_key[i] = 0xffffffffu;
    if (lu == 0) s_cnt = 0;
    __syncthreads();
    uint32_t pv[MS_ELL];
#pragma unroll
    for (int j = 0; j < MS_ELL; j++) pv[j] = u < (uint32_t)n ? ell_v[(size_t)u * MS_ELL + j] : (uint32_t)n;
#pragma unroll
    for (int j = 0; j < MS_ELL; j++) {
        const uint32_t v = pv[j];
        if (v != (uint32_t)n && (v < base || v >= base + MS_TV)) {
            if (adj) {
                const uint32_t vt = v / MS_TV;
                atomicOr(&adj[(size_t)tile * adj_words + (vt >> 5)], 1u << (vt & 31));
                atomicOr(&adj[(size_t)vt * adj_words + ((uint32_t)tile >> 5)], 1u << (tile & 31));
            }
            uint32_t hpos = (v * 2654435761u) >> 19;  // 13 bits
            for (;;) {
                const uint32_t old = atomicCAS(&s_key[hpos], 0xffffffffu, v);
                if (old == 0xffffffffu || old == v) break;
                hpos = (hpos + 1) & (MS_HASH - 1);
            }
        }
    }
    __syncthreads();
    for (int i = lu; i < MS_HASH; i += MS_TV) {
        const uint32_t key = s_key[i];
        if (key != 0xffffffffu) {
            const int idx = atomicAdd(&s_cnt, 1);
            s_idx[i] = idx < MS_HCAP ? (uint16_t)(MS_TV + idx) : (uint16_t)MS_SLOT_FAR;
            if (idx < MS_HCAP) halo_gid[(size_t)tile * MS_HCAP + idx] = key;
        }
    }
    // parents beyond the 16 fixed slots (read from the CSR by the hop): their tiles are neighbours too
    if (adj && u < (uint32_t)n && pv[MS_ELL - 1] != (uint32_t)n) {
        for (int e = rstart[u] + MS_ELL; e < rstart[u + 1]; e++) {
            const uint32_t vt = (rkey[e] >> 6) / MS_TV;
            if (vt != (uint32_t)tile) {
                atomicOr(&adj[(size_t)tile * adj_words + (vt >> 5)], 1u << (vt & 31));
                atomicOr(&adj[(size_t)vt * adj_words + ((uint32_t)tile >> 5)], 1u << (tile & 31));
            }
        }
    }
    __syncthreads();
    if (lu == 0) nhalo[tile] = s_cnt < MS_HCAP ? s_cnt : MS_HCAP;
    uint16_t sl[MS_ELL];
#pragma unroll
    for (int j = 0; j < MS_ELL; j++) {
        const uint32_t v = pv[j];
        uint16_t r = (uint16_t)MS_ZROW;
        if (v != (uint32_t)n) {
            if (v >= base && v < base + MS_TV) {
                r = (uint16_t)(v - base);
            } else {
                uint32_t hpos = (v * 2654435761u) >> 19;
                while (s_key[hpos] != v) hpos = (hpos + 1) & (MS_HASH - 1);
                r = s_idx[hpos];
            }
        }
        sl[j] = r;
    }
    uint4* out = reinterpret_cast<uint4*>(slot16 + (size_t)u * MS_ELL);  // (slot16 has tiles * MS_TV rows)
    out[0] = make_uint4(sl[0] | (sl[1] << 16), sl[2] | (sl[3] << 16), sl[4] | (sl[5] << 16), sl[6] | (sl[7] << 16));
    out[1] = make_uint4(sl[8] | (sl[9] << 16), sl[10] | (sl[11] << 16), sl[12] | (sl[13] << 16), sl[14] | (sl[15] << 16));
}

// the neighbour lists of k_ms_persist from the adjacency bits: one thread per tile
__global__ __launch_bounds__(MS_THREADS) void k_ms_nbr(const uint32_t* __restrict__ adj, int adj_words, int ntiles,
                                                      int32_t* __restrict__ nbr, int32_t* __restrict__ nnbr) {
    const int t = blockIdx.x * MS_THREADS + threadIdx.x;
    if (t >= ntiles) return;
    int cnt = 0;
    for (int i = 0; i < adj_words; i++) {
        uint32_t word = adj[(size_t)t * adj_words + i];
        while (word) {
            const int b = __builtin_ctz(word);
            word &= word - 1;
            const int o = i * 32 + b;
            if (o == t) continue;
            if (cnt < MS_NBMAX) nbr[(size_t)t * MS_NBMAX + cnt] = o;
            cnt++;
        }
    }
    nnbr[t] = cnt > MS_NBMAX ? -1 : cnt;
}

// One hop over tiles.  Workgroup = tile, thread = (local vertex, 64-bit word); W words per vertex (1..4).
template <int W>
__global__ __launch_bounds__(MS_TV* W) void k_ms_hop_tile(const uint16_t* __restrict__ slot16,
                                                          const uint32_t* __restrict__ halo_gid,
                                                          const int32_t* __restrict__ nhalo,
                                                          const uint32_t* __restrict__ ell_v,
                                                          const float* __restrict__ ell_d,
                                                          const int32_t* __restrict__ rstart,
                                                          const uint32_t* __restrict__ rkey,
                                                          const float* __restrict__ rdist,
                                                          const ms_word* __restrict__ Rcur, ms_word* __restrict__ Rnext,
                                                          float* __restrict__ dist_t, int n,
                                                          const int32_t* __restrict__ alive) {
    constexpr int THREADS = MS_TV * W;
    constexpr int HIT = (MS_HCAP * W + THREADS - 1) / THREADS;  // halo (row, word) items per thread
    extern __shared__ ms_word s_rows[];                          // [(MS_ZROW + 1) * W] words, then the halo's vertex ids
    uint32_t* s_gid = reinterpret_cast<uint32_t*>(s_rows + (size_t)(MS_ZROW + 1) * W);
    if (alive && *alive == 0) return;
    constexpr uint32_t S = W * 64u;
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t lu = tid / W, w = tid - lu * W;
    const uint32_t base = tile * MS_TV, u = base + lu;
    const bool live = u < (uint32_t)n;
    const int nh = nhalo[tile];
    // level 1: own word (one contiguous block per tile), the vertex's 16 slots, the halo's vertex ids
    const ms_word own = live ? Rcur[u * W + w] : 0ull;
    const uint4* sp = reinterpret_cast<const uint4*>(slot16 + (size_t)u * MS_ELL);
    const uint4 q0 = sp[0], q1 = sp[1];
    uint32_t hg[HIT];
#pragma unroll
    for (int i = 0; i < HIT; i++) {
        const uint32_t item = tid + i * THREADS, h = item / W;
        hg[i] = h < (uint32_t)nh ? halo_gid[(size_t)tile * MS_HCAP + h] : (uint32_t)n;
    }
    // level 2: the halo's rows (a row is W consecutive lanes' words)
    ms_word hr[HIT];
#pragma unroll
    for (int i = 0; i < HIT; i++) {
        const uint32_t item = tid + i * THREADS, ww = item - (item / W) * W;
        hr[i] = Rcur[hg[i] * W + ww];  // (beyond the halo: row n, zero)
    }
    s_rows[lu * W + w] = own;
    if (tid < W) s_rows[(size_t)MS_ZROW * W + tid] = 0ull;
#pragma unroll
    for (int i = 0; i < HIT; i++) {
        const uint32_t item = tid + i * THREADS, h = item / W, ww = item - h * W;
        if (h < (uint32_t)nh) {
            s_rows[(MS_TV + h) * W + ww] = hr[i];
            if (ww == 0) s_gid[h] = hg[i];
        }
    }
    __syncthreads();
    if (!live) return;
    uint32_t sl[MS_ELL];
    sl[0] = q0.x & 0xffffu; sl[1] = q0.x >> 16; sl[2] = q0.y & 0xffffu; sl[3] = q0.y >> 16;
    sl[4] = q0.z & 0xffffu; sl[5] = q0.z >> 16; sl[6] = q0.w & 0xffffu; sl[7] = q0.w >> 16;
    sl[8] = q1.x & 0xffffu; sl[9] = q1.x >> 16; sl[10] = q1.y & 0xffffu; sl[11] = q1.y >> 16;
    sl[12] = q1.z & 0xffffu; sl[13] = q1.z >> 16; sl[14] = q1.w & 0xffffu; sl[15] = q1.w >> 16;
    ms_word f[MS_ELL];
    bool far = false;
#pragma unroll
    for (int j = 0; j < MS_ELL; j++) {
        far = far || sl[j] == MS_SLOT_FAR;
        f[j] = s_rows[(sl[j] == MS_SLOT_FAR ? (uint32_t)MS_ZROW : sl[j]) * W + w];
    }
    if (__any(far)) {  // parents beyond the halo capacity (a tile with > MS_HCAP outside parents): from memory
#pragma unroll
        for (int j = 0; j < MS_ELL; j++)
            if (sl[j] == MS_SLOT_FAR) f[j] = Rcur[ell_v[u * MS_ELL + j] * W + w];
    }
    ms_word acc = own;
#pragma unroll
    for (int j = 0; j < MS_ELL; j++) acc |= f[j];
    const ms_word taken = acc & ~own;
    float* du = dist_t + (size_t)u * S + w * 64u;
    if (__any(taken != 0)) {
        // the first two parents (ascending) that hand a new bit over, and what they hand over; select chains in
        // descending order over registers (a lane-dependent index into f[] would move it to scratch memory)
        uint32_t s0 = MS_ZROW, j0 = 0, s1 = MS_ZROW, j1 = 0;
        ms_word n0 = 0, n1 = 0;
#pragma unroll
        for (int j = MS_ELL - 1; j >= 0; j--) {
            const ms_word c = f[j] & taken;
            // (j becomes the first; the previous first becomes the second)
            s1 = c ? s0 : s1; j1 = c ? j0 : j1; n1 = c ? n0 : n1;
            s0 = c ? sl[j] : s0; j0 = c ? (uint32_t)j : j0; n0 = c ? c : n0;
        }
        n1 &= ~n0;  // (what the second hands over excludes the first's bits; it may be empty: then a later one may own
                    //  bits -- the loop below)
        // the parents' vertex ids: own tile or halo (LDS), beyond the halo from memory
        const uint32_t v0 = s0 < MS_TV ? base + s0 : (s0 == MS_SLOT_FAR ? ell_v[u * MS_ELL + j0] : s_gid[s0 < MS_ZROW ? s0 - MS_TV : 0]);
        const uint32_t v1 = s1 < MS_TV ? base + s1 : (s1 == MS_SLOT_FAR ? ell_v[u * MS_ELL + j1] : s_gid[s1 < MS_ZROW ? s1 - MS_TV : 0]);
        const uint32_t b0 = n0 ? (uint32_t)__builtin_ctzll(n0) : 0u, b1 = n1 ? (uint32_t)__builtin_ctzll(n1) : 0u;
        // edge lengths and the parents' values requested together (a lane without an item reads element 0)
        const float e0 = ell_d[n0 ? u * MS_ELL + j0 : 0u], e1 = ell_d[n1 ? u * MS_ELL + j1 : 0u];
        const float d0 = dist_t[n0 ? (size_t)v0 * S + w * 64u + b0 : (size_t)0];
        const float d1 = dist_t[n1 ? (size_t)v1 * S + w * 64u + b1 : (size_t)0];
        if (n0) du[b0] = e0 + d0;
        if (n1) du[b1] = e1 + d1;
        ms_word r0 = n0 & (n0 - 1), r1 = n1 & (n1 - 1);  // further bits of the same parents (rare)
        while (r0) {
            const uint32_t b = (uint32_t)__builtin_ctzll(r0);
            r0 &= r0 - 1;
            du[b] = e0 + dist_t[(size_t)v0 * S + w * 64u + b];
        }
        while (r1) {
            const uint32_t b = (uint32_t)__builtin_ctzll(r1);
            r1 &= r1 - 1;
            du[b] = e1 + dist_t[(size_t)v1 * S + w * 64u + b];
        }
        // bits handed over by a third, fourth ... parent (rare): in list order, one by one
        ms_word rem = taken & ~(n0 | n1);
        if (__any(rem != 0)) {
#pragma unroll
            for (int j = 2; j < MS_ELL; j++) {
                ms_word c = f[j] & rem;
                if (c) {
                    rem &= ~c;
                    const float ed = ell_d[u * MS_ELL + j];
                    const float* dv = dist_t + (size_t)ell_v[u * MS_ELL + j] * S + w * 64u;
                    do {
                        const uint32_t b = (uint32_t)__builtin_ctzll(c);
                        c &= c - 1;
                        du[b] = ed + dv[b];
                    } while (c);
                }
            }
        }
    }
    if (~acc != 0 && sl[MS_ELL - 1] != MS_ZROW) {  // a long in-list: the rest from the CSR
        int e = rstart[u] + MS_ELL;
        const int end = rstart[u + 1];
        for (; e < end && ~acc != 0; e++) {
            const uint32_t v = rkey[e] >> 6;
            ms_word c = Rcur[v * W + w] & ~acc;
            if (c) {
                const float ed = rdist[e];
                const float* dv = dist_t + (size_t)v * S + w * 64u;
                acc |= c;
                do {
                    const uint32_t b = (uint32_t)__builtin_ctzll(c);
                    c &= c - 1;
                    du[b] = ed + dv[b];
                } while (c);
            }
        }
    }
    Rnext[u * W + w] = acc;
}

// ------------------------------------------------------------------------------------
// the whole search in ONE launch: tiles keep their rows in LDS, neighbours exchange through memory
// ------------------------------------------------------------------------------------
// A launch per hop pays, besides its ~2.6 us, for cold caches: every launch re-fetches the masks, the in-lists and the
// distance lines it touches in 64-byte granules (~7 MB per hop at the start of a search, ~15 MB in the middle: that
// traffic, not the arithmetic, is the 5-9 us a hop takes in either form above).  Here one workgroup per tile stays
// resident for all hops: its rows, slots and halo ids live in LDS / registers.  Iteration h of a tile:
//   1. wait until every neighbour tile's counter is >= h - 1: their rows of hop h - 1 and the distances of every bit
//      they took up to hop h - 2 are visible
//   2. request the parents' distances for the bits THIS tile took in hop h - 1 (items saved by the previous iteration;
//      those parents took the bits in hop h - 2) -- not waited for
//   3. pull the halo rows of hop h - 1 into LDS
//   4. rows of hop h = own | OR of the parents' rows; for every new bit the first parent (list order) that has it:
//      saved as items for the next iteration (two per lane in registers, the rest in a per-tile list in memory)
//   5. store the distances of the hop h - 1 bits (the loads of step 2 have arrived behind steps 3-4), publish the rows of
//      hop h, drain the stores, raise the counter to h.
// The distance of a bit thus follows its mask bit one iteration later, off the critical path.  Cross-workgroup data
// (published rows, distances, counters) is written and read with agent-scope (sc1) accesses only -- MI355X_MICROARCH.md
// "Valid forms", first row: every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier,
// ONE lane stores the counter; the consumer polls the counter with sc1 loads, the workgroup meets at a barrier, then sc1
// loads.  Published rows are double-buffered by hop parity; a tile waits for the tiles it reads AND for the tiles that
// read it, so a buffer is never overwritten while a reader is on it.  Every wait is bounded (MS_SPIN_TICKS of the
// 100 MHz clock): on a timeout -- a workgroup that never became resident -- or a full item list the launch sets *err
// (1 / 2) and every tile leaves; nothing can hang, and the caller must discard the result.
#define MS_SPIN_TICKS 20000000ull  // 0.2 s
#define MS_ITEMS 16384             // extra items (third, fourth ... parent of a word, further bits of one parent) per tile and hop
// an item = one 64-bit word, written and read with agent-scope accesses (the writer and the reader are different waves of
// one workgroup an iteration apart: nothing may be served from a stale L1 line)
//   low word:  parent vertex, or 0x80000000 | index of a CSR entry (an in-list entry beyond the 16 slots)
//   high word: (local vertex * W + word) << 10 | bit << 4 | in-list slot
typedef unsigned long long MsItem;
__device__ __forceinline__ MsItem ms_item(uint32_t v, uint32_t pos) { return ((unsigned long long)pos << 32) | v; }
template <int W>
__global__ __launch_bounds__(MS_TV* W, 2 * W) void k_ms_persist(const uint16_t* __restrict__ slot16,
                                                                const uint32_t* __restrict__ halo_gid,
                                                                const int32_t* __restrict__ nhalo,
                                                                const int32_t* __restrict__ nbr,
                                                                const int32_t* __restrict__ nnbr,
                                                                const uint32_t* __restrict__ ell_v,
                                                                const float* __restrict__ ell_d,
                                                                const int32_t* __restrict__ rstart,
                                                                const uint32_t* __restrict__ rkey,
                                                                const float* __restrict__ rdist,
                                                                ms_word* __restrict__ R0, ms_word* __restrict__ R1,
                                                                float* __restrict__ dist_t, int n, int ntiles,
                                                                int max_step, int32_t* __restrict__ hopc,
                                                                MsItem* __restrict__ items, int32_t* __restrict__ err
#ifdef MS_TRACE
                                                                , unsigned long long* __restrict__ ptrace
#endif
                                                                ) {
#ifdef MS_TRACE
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define MS_PT(i) do { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[i] += tn - tprev; tprev = tn; } while (0)
#else
#define MS_PT(i) do {} while (0)
#endif
    constexpr int THREADS = MS_TV * W;
    constexpr int HIT = (MS_HCAP * W + THREADS - 1) / THREADS;
    constexpr uint32_t S = W * 64u;
    extern __shared__ ms_word s_rows[];
    uint32_t* s_gid = reinterpret_cast<uint32_t*>(s_rows + (size_t)(MS_ZROW + 1) * W);
    __shared__ uint16_t s_slot[MS_TV * MS_ELL];  // the vertices' local slots (LDS: 8 registers per lane less)
    __shared__ int s_abort, s_nitems[2];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t lu = tid / W, w = tid - lu * W;
    const uint32_t base = tile * MS_TV, u = base + lu;
    const bool live = u < (uint32_t)n;
    const int nh = nhalo[tile];
    const int nn = nnbr[tile];
    MsItem* my_items = items + (size_t)tile * 2 * MS_ITEMS;
    // static state in LDS: the vertices' 16 slots, the halo's vertex ids, the zero row
    for (int i = tid; i < MS_TV * MS_ELL / 2; i += THREADS)
        reinterpret_cast<uint32_t*>(s_slot)[i] = reinterpret_cast<const uint32_t*>(slot16 + (size_t)base * MS_ELL)[i];
    for (int i = tid; i < nh; i += THREADS) s_gid[i] = halo_gid[(size_t)tile * MS_HCAP + i];
    if (tid < W) s_rows[(size_t)MS_ZROW * W + tid] = 0ull;
    if (tid == 0) s_abort = s_nitems[0] = s_nitems[1] = 0;
    ms_word own = live ? R0[u * W + w] : 0ull;  // hop 0, written by k_ms_init (an earlier launch)
    s_rows[lu * W + w] = own;
    float* du = dist_t + (size_t)u * S + w * 64u;
    const uint16_t* my_slot = s_slot + lu * MS_ELL;
    // the items of the previous hop held by this lane: parent, bit, edge slot (bit = 64: none)
    uint32_t pv0 = 0, pv1 = 0, pb0 = 64, pb1 = 64, pj0 = 0, pj1 = 0;
    __syncthreads();
    bool far_any = false;
#pragma unroll
    for (int j = 0; j < MS_ELL; j++) far_any = far_any || my_slot[j] == MS_SLOT_FAR;
    const bool tail = live && my_slot[MS_ELL - 1] != MS_ZROW;
    for (int h = 1; h <= max_step + 1; h++) {
        const bool last = h > max_step;  // one more round for the distances of the last hop's bits
        const ms_word* Rc = (h - 1) & 1 ? R1 : R0;
        ms_word* Rn = h & 1 ? R1 : R0;
        // (1) every neighbour has published hop h - 1
        if (tid < 64) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            const int cnt = nn >= 0 ? nn : ntiles;
            for (int j = (int)tid; j < cnt; j += 64) {
                const int t = nn >= 0 ? nbr[(size_t)tile * MS_NBMAX + j] : j;
                while (__hip_atomic_load(&hopc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < h - 1) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > MS_SPIN_TICKS ||
                        __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        atomicCAS(err, 0, 1);
                        s_abort = 1;
                        break;
                    }
                }
            }
        }
        __syncthreads();
        MS_PT(0);
        if (s_abort) return;
        // (2) the distances the bits of hop h - 1 wait for: requested now, used in (5)
        const float e0 = ell_d[pb0 < 64 ? u * MS_ELL + pj0 : 0u], e1 = ell_d[pb1 < 64 ? u * MS_ELL + pj1 : 0u];
        const float d0 = __hip_atomic_load(&dist_t[pb0 < 64 ? (size_t)pv0 * S + w * 64u + pb0 : (size_t)0], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        const float d1 = __hip_atomic_load(&dist_t[pb1 < 64 ? (size_t)pv1 * S + w * 64u + pb1 : (size_t)0], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        // (the tile's extra items of hop h - 1: one per thread and round)
        const int nit = s_nitems[(h - 1) & 1];
        const MsItem* it_in = my_items + (size_t)((h - 1) & 1) * MS_ITEMS;
        uint32_t xv = 0, xpos = 0xffffffffu;
        float xe = 0.f, xd = 0.f;
        if ((int)tid < nit) {
            const MsItem xi = __hip_atomic_load(&it_in[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xv = (uint32_t)xi;
            xpos = (uint32_t)(xi >> 32);
            if (!(xv & 0x80000000u)) {
                const uint32_t xlw = xpos >> 10, xb = (xpos >> 4) & 63u, xj = xpos & 15u;
                xe = ell_d[(base + xlw / W) * MS_ELL + xj];
                xd = __hip_atomic_load(&dist_t[(size_t)xv * S + (xlw % W) * 64u + xb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        ms_word acc = own;
        uint32_t cv0 = 0, cv1 = 0, cb0 = 64, cb1 = 64, cj0 = 0, cj1 = 0;
        if (!last) {
            // (3) the halo's rows of hop h - 1
#pragma unroll
            for (int i = 0; i < HIT; i++) {
                const uint32_t item = tid + i * THREADS, hh = item / W, ww = item - hh * W;
                if (hh < (uint32_t)nh)
                    s_rows[(MS_TV + hh) * W + ww] = __hip_atomic_load(&Rc[s_gid[hh] * W + ww], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) s_nitems[h & 1] = 0;
            __syncthreads();
            MS_PT(1);
            // (4) OR over the parents; who hands the new bits over
            if (live) {
#pragma unroll
                for (int j = 0; j < MS_ELL; j++) {
                    const uint32_t sl = my_slot[j];
                    acc |= s_rows[(sl == MS_SLOT_FAR ? (uint32_t)MS_ZROW : sl) * W + w];
                }
                if (far_any) {
#pragma unroll
                    for (int j = 0; j < MS_ELL; j++) {
                        const uint32_t sl = my_slot[j];
                        if (sl == MS_SLOT_FAR)
                            acc |= __hip_atomic_load(&Rc[ell_v[u * MS_ELL + j] * W + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                ms_word rem = acc & ~own;
                int taken_items = 0;
                if (rem) {
#pragma unroll 1
                    for (int j = 0; j < MS_ELL && rem; j++) {
                        const uint32_t sl = my_slot[j];
                        if (sl == MS_ZROW) break;
                        uint32_t v;
                        ms_word f;
                        if (sl == MS_SLOT_FAR) {
                            v = ell_v[u * MS_ELL + j];
                            f = __hip_atomic_load(&Rc[v * W + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            v = sl < MS_TV ? base + sl : s_gid[sl - MS_TV];
                            f = s_rows[sl * W + w];
                        }
                        ms_word c = f & rem;
                        rem &= ~c;
                        while (c) {
                            const uint32_t b = (uint32_t)__builtin_ctzll(c);
                            c &= c - 1;
                            if (taken_items == 0) {
                                cv0 = v; cb0 = b; cj0 = (uint32_t)j;
                            } else if (taken_items == 1) {
                                cv1 = v; cb1 = b; cj1 = (uint32_t)j;
                            } else {
                                const int pos = atomicAdd(&s_nitems[h & 1], 1);
                                if (pos < MS_ITEMS) {
                                    __hip_atomic_store(&my_items[(size_t)(h & 1) * MS_ITEMS + pos],
                                                       ms_item(v, ((lu * W + w) << 10) | (b << 4) | (uint32_t)j), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
                                } else {
                                    atomicCAS(err, 0, 2);
                                }
                            }
                            taken_items++;
                        }
                    }
                }
                if (tail && ~acc != 0) {  // a long in-list: the rest from the CSR; its bits go to the item list
                    const int end = rstart[u + 1];
                    for (int e = rstart[u] + MS_ELL; e < end && ~acc != 0; e++) {
                        const uint32_t v = rkey[e] >> 6;
                        ms_word c = __hip_atomic_load(&Rc[v * W + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~acc;
                        acc |= c;
                        while (c) {
                            const uint32_t b = (uint32_t)__builtin_ctzll(c);
                            c &= c - 1;
                            // (not one of the 16 slots: the item names the CSR entry, which has the parent and the edge)
                            const int pos = atomicAdd(&s_nitems[h & 1], 1);
                            if (pos < MS_ITEMS)
                                __hip_atomic_store(&my_items[(size_t)(h & 1) * MS_ITEMS + pos],
                                                   ms_item((uint32_t)e | 0x80000000u, ((lu * W + w) << 10) | (b << 4)), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                            else
                                atomicCAS(err, 0, 2);
                        }
                    }
                }
            }
            MS_PT(2);
            __syncthreads();  // (every read of the tile's own rows of hop h - 1 is done; the item list is complete)
        }
        // (5) the distances of the hop h - 1 bits; the tile's rows of hop h; drain; counter
        if (pb0 < 64) __hip_atomic_store(&du[pb0], e0 + d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pb1 < 64) __hip_atomic_store(&du[pb1], e1 + d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (xpos != 0xffffffffu) {
            const uint32_t xlw = xpos >> 10, xb = (xpos >> 4) & 63u;
            float* xdu = dist_t + (size_t)(base + xlw / W) * S + (xlw % W) * 64u;
            if (xv & 0x80000000u) {  // an in-list entry beyond the 16 slots: parent and edge from the CSR
                const uint32_t e = xv & 0x7fffffffu, v = rkey[e] >> 6;
                const float d = __hip_atomic_load(&dist_t[(size_t)v * S + (xlw % W) * 64u + xb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&xdu[xb], rdist[e] + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(&xdu[xb], xe + xd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        for (int i = (int)tid + THREADS; i < nit; i += THREADS) {  // (more extra items than threads: rare, synchronous)
            const MsItem y = __hip_atomic_load(&it_in[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t yv = (uint32_t)y, ypos = (uint32_t)(y >> 32);
            const uint32_t ylw = ypos >> 10, yb = (ypos >> 4) & 63u, yj = ypos & 15u;
            float* ydu = dist_t + (size_t)(base + ylw / W) * S + (ylw % W) * 64u;
            if (yv & 0x80000000u) {
                const uint32_t e = yv & 0x7fffffffu, v = rkey[e] >> 6;
                const float d = __hip_atomic_load(&dist_t[(size_t)v * S + (ylw % W) * 64u + yb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ydu[yb], rdist[e] + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const float d = __hip_atomic_load(&dist_t[(size_t)yv * S + (ylw % W) * 64u + yb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ydu[yb], ell_d[(base + ylw / W) * MS_ELL + yj] + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (last) break;
        own = acc;
        s_rows[lu * W + w] = own;
        if (live) __hip_atomic_store(&Rn[u * W + w], own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&hopc[tile], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pv0 = cv0; pb0 = cb0; pj0 = cj0;
        pv1 = cv1; pb1 = cb1; pj1 = cj1;
        MS_PT(3);
    }
#ifdef MS_TRACE
    if (ptrace && (tid & 63) == 0)
        for (int i = 0; i < 4; i++) ptrace[((size_t)tile * (THREADS / 64) + (tid >> 6)) * 4 + i] = tacc[i];
#endif
}

// alive_out = (the two mask buffers differ: the last hop reached something); runs only while alive_in says the search
// was on at the previous check
__global__ __launch_bounds__(MS_THREADS) void k_ms_check(const ms_word* __restrict__ Ra, const ms_word* __restrict__ Rb,
                                                         int nwords, const int32_t* __restrict__ alive_in,
                                                         int32_t* __restrict__ alive_out) {
    if (alive_in && *alive_in == 0) return;
    ms_word acc = 0;
    for (int i = blockIdx.x * MS_THREADS + threadIdx.x; i < nwords; i += gridDim.x * MS_THREADS) acc |= Ra[i] ^ Rb[i];
    if (__any(acc != 0) && (threadIdx.x & 63) == 0) *alive_out = 1;
}

// geo[q][u] = R[u] bit q ? dist[u][q] : -1 : 64 vertices x 64 queries per block through LDS
__global__ __launch_bounds__(MS_THREADS) void k_ms_transpose(const float* __restrict__ dist_t,
                                                             const ms_word* __restrict__ R, int n, int nq, int W,
                                                             int S, const int32_t* __restrict__ inv, int u_off,
                                                             float* __restrict__ geo) {
    __shared__ float tile[64][65];
    const int u0 = blockIdx.x * 64, q0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int u = u0 + r, q = q0 + tx;
        float val = -1.0f;
        if (u < n && q < nq) {
            const size_t row = inv ? (size_t)inv[u + u_off] : (size_t)(u + u_off);  // (n = this set's vertices)
            const ms_word m = R[row * W + (q >> 6)];
            if ((m >> (q & 63)) & 1ull) val = dist_t[row * S + q];
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
    size_t rcount, rstart, rcur, bsum, tkey, tdist, rkey, rdist, ell_v, ell_d, slot16, halo_gid, nhalo, R0, R1, flags,
        dist_t, total;
    size_t mlo, mcount, mstart, mcur, mbsum, mkey, perm, inv;
    size_t nbr, nnbr, hopc, err, adj, items;
    int adj_words;
    int tiles;
    int W, S;
    size_t E;
};
static MsLayout ms_layout(int n, int K, int nq, int max_step) {
    MsLayout L;
    L.W = (nq + 63) / 64;
    if (L.W < 1) L.W = 1;
    L.S = L.W * 64;
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
    L.tiles = (n + MS_TV - 1) / MS_TV;
    L.slot16 = o; o += ms_align((size_t)L.tiles * MS_TV * MS_ELL * 2);
    L.halo_gid = o; o += ms_align((size_t)L.tiles * MS_HCAP * 4);
    L.nhalo = o; o += ms_align((size_t)L.tiles * 4);
    L.nbr = o; o += ms_align((size_t)L.tiles * MS_NBMAX * 4);
    L.nnbr = o; o += ms_align((size_t)L.tiles * 4);
    L.hopc = o; o += ms_align((size_t)L.tiles * 4);
    L.err = o; o += ms_align(16);
    L.adj_words = (L.tiles + 31) / 32;
    L.adj = o; o += ms_align((size_t)L.tiles * L.adj_words * 4);
    L.items = o; o += ms_align((size_t)L.tiles * 2 * MS_ITEMS * sizeof(unsigned long long));
    L.mlo = o; o += ms_align(16);
    L.mcount = o; o += ms_align((size_t)(MS_MCELLS + 1) * 4);
    L.mstart = o; o += ms_align((size_t)(MS_MCELLS + 2) * 4);
    L.mcur = o; o += ms_align((size_t)(MS_MCELLS + 1) * 4);
    L.mbsum = o; o += ms_align((size_t)2 * gf_iscan_blocks(MS_MCELLS) * 4 + 64);
    L.mkey = o; o += ms_align((size_t)n * 4);
    L.perm = o; o += ms_align((size_t)n * 4);
    L.inv = o; o += ms_align((size_t)n * 4);
    L.R0 = o; o += ms_align((size_t)(n + 1) * L.W * 8);
    L.R1 = o; o += ms_align((size_t)(n + 1) * L.W * 8);
    L.flags = o; o += ms_align((size_t)(max_step / MS_CHECK_EVERY + 2) * 4);
    L.dist_t = o; o += ms_align((size_t)n * L.S * 4);
    L.total = o;
    return L;
}

// dev knob: GF_BFS_MS_TILES=1 takes the LDS-tile form of the hop (k_ms_hop_tile).  Measured slower than the gather
// form on vertices in scene order (8.7 against 7.3 us per hop, S150k foreground, 256 queries): a tile of 256
// consecutive vertices is a strip, its halo ~4 tiles' worth of rows, so the tile fetches MORE bytes per hop than the
// gather's L1-shared lines; it would need spatially compact tiles (a Morton re-ordering of the vertices) to pay.
// GF_BFS_MS_PERSIST=1 (with coordinates): the whole search as ONE launch (k_ms_persist)
static int g_ms_persist = -1;
static bool ms_persist_on() {
    if (g_ms_persist < 0) {
        const char* e = getenv("GF_BFS_MS_PERSIST");
        g_ms_persist = e ? (atoi(e) != 0) : 0;
    }
    return g_ms_persist != 0;
}
extern "C" int gf_dev_bfs_ms_persist(int on) {
    g_ms_persist = on < 0 ? -1 : (on != 0);
    return GF_OK;
}
extern "C" const int32_t* gf_geodesic_ms_error_flag(void* scratch, int n, int K, int nq, int max_step);
static int g_ms_tiles = -2;  // -2: read the environment; -1: by the caller's coordinates; 0 / 1: forced
static int g_ms_tiles_knob() {
    if (g_ms_tiles == -2) {
        const char* e = getenv("GF_BFS_MS_TILES");
        g_ms_tiles = e ? (atoi(e) != 0) : -1;
    }
    return g_ms_tiles;
}
extern "C" int gf_dev_bfs_ms_tiles(int on) {
    g_ms_tiles = on < 0 ? -2 : (on != 0);
    return GF_OK;
}
#ifdef MS_TRACE
static unsigned long long* g_ms_trace = nullptr;
static int g_ms_trace_hop = -1;
extern "C" int gf_dev_ms_trace(void* buf, int hop) {
    g_ms_trace = (unsigned long long*)buf;
    g_ms_trace_hop = hop;
    return 0;
}
#define MS_TRACE_ARG , (h == g_ms_trace_hop ? g_ms_trace : (unsigned long long*)nullptr)
#define MS_PTRACE_ARG , g_ms_trace
#else
#define MS_TRACE_ARG
#define MS_PTRACE_ARG
#endif
// device word that k_ms_persist sets when a wait timed out (a workgroup never became resident): the results of that
// call are then invalid
extern "C" const int32_t* gf_geodesic_ms_error_flag(void* scratch, int n, int K, int nq, int max_step) {
    if (!scratch || n < 1 || K < 2 || nq < 1 || max_step < 0) return nullptr;
    return (const int32_t*)((char*)scratch + ms_layout(n, K, nq, max_step).err);
}
extern "C" size_t gf_geodesic_ms_scratch_bytes(int n, int K, int nq, int max_step) {
    if (n < 1 || K < 2 || nq < 1 || max_step < 0) return 0;
    return ms_layout(n, K, nq, max_step).total;
}

static int ms_run(const float* D, const int32_t* I, const float* xyz, int n, int K, const int32_t* src, int nq, int nsets,
                  const int32_t* set_off, float* const* geos, float radius, int max_step, void* scratch,
                  size_t scratch_bytes, void* stream);

extern "C" int gf_geodesic_bfs_ms(const float* D, const int32_t* I, const float* xyz, int n, int K, const int32_t* src,
                                  int nq, float radius, int max_step, float* geo, void* scratch, size_t scratch_bytes,
                                  void* stream) {
    const int32_t off[2] = {0, n};
    float* geos[1] = {geo};
    return ms_run(D, I, xyz, n, K, src, nq, 1, off, geos, radius, max_step, scratch, scratch_bytes, stream);
}

// Several graphs that do not touch (the scenes of a batch) searched together: D / I are the scenes' rows concatenated,
// the indices in I GLOBAL (scene offset added); src int32 [nsets][nq] global vertex ids; set_off host int32 [nsets + 1];
// geos host array of nsets device pointers, geos[k] fp32 [nq][set_off[k+1] - set_off[k]].  One hop launch serves all.
extern "C" int gf_geodesic_bfs_ms_sets(const float* D, const int32_t* I, int n, int K, const int32_t* src, int nq, int nsets,
                                       const int32_t* set_off, float* const* geos, float radius, int max_step,
                                       void* scratch, size_t scratch_bytes, void* stream) {
    GF_CHECK_ARG(nsets >= 1 && nsets <= 64 && set_off && geos && set_off[0] == 0 && set_off[nsets] == n,
                 "gf_geodesic_bfs_ms_sets: bad set table");
    return ms_run(D, I, nullptr, n, K, src, nq, nsets, set_off, geos, radius, max_step, scratch, scratch_bytes, stream);
}

static int ms_run(const float* D, const int32_t* I, const float* xyz, int n, int K, const int32_t* src, int nq, int nsets,
                  const int32_t* set_off, float* const* geos, float radius, int max_step, void* scratch,
                  size_t scratch_bytes, void* stream) {
    float* geo = geos[0];
    GF_CHECK_ARG(n >= 1 && K >= 2 && K <= 64 && nq >= 0 && max_step >= 0, "gf_geodesic_bfs_ms: bad arguments");
    GF_CHECK_ARG(n < (1 << 26), "gf_geodesic_bfs_ms: n=%d exceeds the 26-bit parent field", n);
    if (nq == 0) return GF_OK;
    const MsLayout L = ms_layout(n, K, nq, max_step);
    GF_CHECK_ARG(D && I && src && geo && scratch, "gf_geodesic_bfs_ms: null pointer");
    GF_CHECK_ARG(scratch_bytes >= L.total, "gf_geodesic_bfs_ms: scratch of %zu bytes, %zu needed", scratch_bytes, L.total);
    // 32-bit offsets inside the kernels: mask words, in-list rows
    GF_CHECK_ARG((size_t)(n + 1) * L.W < ((size_t)1 << 29) && (size_t)n * MS_ELL < ((size_t)1 << 30),
                 "gf_geodesic_bfs_ms: n * words overflows the 32-bit offsets");
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
    uint16_t* slot16 = (uint16_t*)(base + L.slot16);
    uint32_t* halo_gid = (uint32_t*)(base + L.halo_gid);
    int32_t* nhalo = (int32_t*)(base + L.nhalo);
    ms_word* R[2] = {(ms_word*)(base + L.R0), (ms_word*)(base + L.R1)};
    int32_t* flags = (int32_t*)(base + L.flags);
    float* dist_t = (float*)(base + L.dist_t);
    const int W = L.W, S = L.S;
    // the tile form of the hop goes with the spatial working order (coordinates given); GF_BFS_MS_TILES forces either
    const bool tiled = W >= 1 && W <= 4 && (g_ms_tiles_knob() < 0 ? xyz != nullptr : g_ms_tiles_knob() != 0);
    int32_t *perm = nullptr, *inv = nullptr;
    if (xyz && tiled) {
        int32_t* mlo = (int32_t*)(base + L.mlo);
        int32_t* mcount = (int32_t*)(base + L.mcount);
        int32_t* mstart = (int32_t*)(base + L.mstart);
        int32_t* mcur = (int32_t*)(base + L.mcur);
        int32_t* mbsum = (int32_t*)(base + L.mbsum);
        uint32_t* mkey = (uint32_t*)(base + L.mkey);
        perm = (int32_t*)(base + L.perm);
        inv = (int32_t*)(base + L.inv);
        GF_TRY(hipMemsetAsync(mlo, 0x7f, 16, st));
        GF_TRY(hipMemsetAsync(mcount, 0, (size_t)(MS_MCELLS + 1) * 4, st));
        const int g = gf_div_up(n, MS_THREADS);
        hipLaunchKernelGGL(k_ms_bbox, dim3(g), dim3(MS_THREADS), 0, st, xyz, n, mlo);
        hipLaunchKernelGGL(k_ms_mkey, dim3(g), dim3(MS_THREADS), 0, st, xyz, n, (const int32_t*)mlo, mkey, mcount);
        gf_iscan(mcount, MS_MCELLS, mstart, mcur, mbsum, mbsum + gf_iscan_blocks(MS_MCELLS), st);
        hipLaunchKernelGGL(k_ms_mfill, dim3(g), dim3(MS_THREADS), 0, st, (const uint32_t*)mkey, n, mcur, perm, inv);
    }

    GF_TRY(hipMemsetAsync(rcount, 0, (size_t)(n + 1) * 4, st));
    GF_TRY(hipMemsetAsync(flags, 0, (size_t)(max_step / MS_CHECK_EVERY + 2) * 4, st));
    const int rows_grid = gf_div_up(n, MS_THREADS / 64);
    hipLaunchKernelGGL(k_ms_count, dim3(rows_grid), dim3(MS_THREADS), 0, st, D, I, n, K, radius, (const int32_t*)inv, rcount);
    gf_iscan(rcount, n, rstart, rcur, bsum, bsum + gf_iscan_blocks(n), st);
    hipLaunchKernelGGL(k_ms_fill, dim3(rows_grid), dim3(MS_THREADS), 0, st, D, I, n, K, radius, (const int32_t*)inv, rcur,
                       tkey, tdist);
    hipLaunchKernelGGL(k_ms_sort, dim3(gf_div_up((long long)n * 16, MS_THREADS)), dim3(MS_THREADS), 0, st, rstart, n,
                       tkey, tdist, rkey, rdist, ell_v, ell_d, (const int32_t*)inv);
    int32_t* nbr = (int32_t*)(base + L.nbr);
    int32_t* nnbr = (int32_t*)(base + L.nnbr);
    int32_t* hopc = (int32_t*)(base + L.hopc);
    int32_t* errw = (int32_t*)(base + L.err);
    uint32_t* adj = (uint32_t*)(base + L.adj);
    const bool persist = tiled && xyz != nullptr && ms_persist_on();
    if (persist) GF_TRY(hipMemsetAsync(adj, 0, (size_t)L.tiles * L.adj_words * 4, st));
    if (tiled)
        hipLaunchKernelGGL(k_ms_tiles, dim3(L.tiles), dim3(MS_TV), 0, st, ell_v, n, (const int32_t*)rstart,
                           (const uint32_t*)rkey, slot16, halo_gid, nhalo, persist ? adj : (uint32_t*)nullptr, L.adj_words);
    if (persist)
        hipLaunchKernelGGL(k_ms_nbr, dim3(gf_div_up(L.tiles, MS_THREADS)), dim3(MS_THREADS), 0, st, (const uint32_t*)adj,
                           L.adj_words, L.tiles, nbr, nnbr);
    const size_t tile_lds = (size_t)(MS_ZROW + 1) * W * sizeof(ms_word) + (size_t)MS_HCAP * sizeof(uint32_t);
    hipLaunchKernelGGL(k_ms_init, dim3(gf_div_up((long long)(n + 1) * W, MS_THREADS)), dim3(MS_THREADS), 0, st, src, nq,
                       nsets, n, W, S, (const int32_t*)perm, R[0], R[1], dist_t);
    const int grid = (gf_div_up((long long)n * W, MS_THREADS) + 7) & ~7;  // (a multiple of 8: k_ms_hop's XCD mapping)
    if (persist) {
        GF_TRY(hipMemsetAsync(hopc, 0, (size_t)L.tiles * 4, st));
        GF_TRY(hipMemsetAsync(errw, 0, 16, st));
#define MS_LAUNCH_PERSIST(WW)                                                                                            \
    hipLaunchKernelGGL(k_ms_persist<WW>, dim3(L.tiles), dim3(MS_TV* WW), tile_lds, st, slot16, halo_gid, nhalo, nbr, nnbr,    \
                       ell_v, ell_d, rstart, rkey, rdist, R[0], R[1], dist_t, n, L.tiles, max_step, hopc,                   \
                       (MsItem*)(base + L.items), errw MS_PTRACE_ARG)
        if (W == 4) MS_LAUNCH_PERSIST(4);
        else if (W == 3) MS_LAUNCH_PERSIST(3);
        else if (W == 2) MS_LAUNCH_PERSIST(2);
        else MS_LAUNCH_PERSIST(1);
    }
    for (int h = 1; h <= max_step && !persist; h++) {
        // flags[c]: the search was still on at check c (after hop c * MS_CHECK_EVERY); the hops up to the first check run
        // unconditionally
        const int c = (h - 1) / MS_CHECK_EVERY;
        const int32_t* alive = c >= 1 ? flags + c : nullptr;
        const ms_word* Rc = R[(h - 1) & 1];
        ms_word* Rn = R[h & 1];
        if (tiled) {
#define MS_LAUNCH_TILE(WW)                                                                                              \
    hipLaunchKernelGGL(k_ms_hop_tile<WW>, dim3(L.tiles), dim3(MS_TV* WW), tile_lds, st, slot16, halo_gid, nhalo, ell_v,  \
                       ell_d, rstart, rkey, rdist, Rc, Rn, dist_t, n, alive)
            if (W == 4) MS_LAUNCH_TILE(4);
            else if (W == 3) MS_LAUNCH_TILE(3);
            else if (W == 2) MS_LAUNCH_TILE(2);
            else MS_LAUNCH_TILE(1);
        } else if (W == 4)
            hipLaunchKernelGGL(k_ms_hop<4>, dim3(grid), dim3(MS_THREADS), 0, st, ell_v, ell_d, rstart, rkey, rdist, Rc, Rn,
                               dist_t, n, W, alive MS_TRACE_ARG);
        else if (W == 2)
            hipLaunchKernelGGL(k_ms_hop<2>, dim3(grid), dim3(MS_THREADS), 0, st, ell_v, ell_d, rstart, rkey, rdist, Rc, Rn,
                               dist_t, n, W, alive MS_TRACE_ARG);
        else
            hipLaunchKernelGGL(k_ms_hop<0>, dim3(grid), dim3(MS_THREADS), 0, st, ell_v, ell_d, rstart, rkey, rdist, Rc, Rn,
                               dist_t, n, W, alive MS_TRACE_ARG);
        if (h % MS_CHECK_EVERY == 0 && h < max_step) {
            const int cc = h / MS_CHECK_EVERY;
            hipLaunchKernelGGL(k_ms_check, dim3(64), dim3(MS_THREADS), 0, st, Rc, (const ms_word*)Rn, n * W,
                               cc >= 2 ? flags + cc - 1 : nullptr, flags + cc);
        }
    }
    // (a search that ended early left both buffers equal; otherwise the last hop wrote R[max_step & 1])
    for (int k = 0; k < nsets; k++) {
        const int nk = set_off[k + 1] - set_off[k];
        if (nk <= 0) continue;
        GF_CHECK_ARG(geos[k] != nullptr, "gf_geodesic_bfs_ms: null output");
        hipLaunchKernelGGL(k_ms_transpose, dim3(gf_div_up(nk, 64), gf_div_up(nq, 64)), dim3(MS_THREADS), 0, st, dist_t,
                           (const ms_word*)R[max_step & 1], nk, nq, W, S, (const int32_t*)inv, set_off[k], geos[k]);
    }
    (void)geo;
    GF_CHECK_LAUNCH("gf_geodesic_bfs_ms");
    return GF_OK;
}
