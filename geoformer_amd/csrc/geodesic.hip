// Geodesic stage on gfx950: radius-limited kNN graph + hop-synchronous frontier BFS.
//
//   reference: model/geoformer/geodesic_utils.py:11-24 (find_knn over faiss GpuIndexFlatL2)
//              model/geoformer/geodesic_utils.py:91-164 (cal_geodesic_vectorize)
//
// kNN.  The BFS only ever follows edges with sqrt(d2) <= radius (geodesic_utils.py:123,152), and
// "the k nearest, then keep those within the radius" equals "the k nearest among those within the
// radius" because the test is monotone in d2.  So instead of faiss' O(N^2) brute force the graph is
// built from a spatial-hash grid with cell >= radius: 27 buckets per query point, exact fp32
// d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)), neighbours ordered by (d2, index) -- the oracle's order
// (oracle/gf_oracle.c orc_knn).  Rows are padded with (inf, -1).  One wave per point; candidates
// inside the radius are compacted into LDS and ranked by counting (typical count ~15).
//
// BFS.  One workgroup per query walks its own frontier; per hop (a) every frontier vertex pushes
// (parent<<6 | rank)+1 to its unvisited in-radius neighbours with atomicMin -- the minimum is the
// lowest-index parent, then the lowest neighbour rank, which is exactly the entry the reference's
// sort-based de-duplication keeps -- and the first toucher appends the vertex to the next frontier;
// (b) after a barrier the new frontier commits geo[v] = geo[parent] + D[parent][rank].  Distances
// are fp32 sums along that parent chain, so results are bit-identical to the reference's.
#include <cstdlib>
#include "common.h"

// Scope of the atomics on a query's key row: AGENT.  Every access to the row of query q -- the bids' atomic minima, the
// commit's loads -- comes from the ONE workgroup that owns q, so workgroup scope would be what the algorithm needs, but
// it is only correct while all waves of a workgroup share one coherent L1 (not in tgsplit mode), and it buys nothing:
// measured (profiles/r3_pmc_bfs.md), no difference in time or traffic -- a 256-query launch moves 2.4 GB of fetches +
// 1.4 GB of writes through the fabric (TCC hit rate 12 %) against ~0.24 GB of algorithmic traffic, at either scope: it is
// the ~76 M scattered 8-byte minima themselves (~5 bids per reached vertex, each a 32-byte sector each way), which the
// XCD's L2 does not hold on to (32 queries x 480 KB of keys per XCD against 4 MB).
// -DBFS_KEY_SCOPE=__HIP_MEMORY_SCOPE_WORKGROUP is a dev knob.
#ifndef BFS_KEY_SCOPE
#define BFS_KEY_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#include <type_traits>

// ------------------------------------------------------------------------------------
// spatial hash grid
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int3 cell_of(float x, float y, float z, float inv_cell) {
    return make_int3((int)floorf(x * inv_cell), (int)floorf(y * inv_cell), (int)floorf(z * inv_cell));
}
__device__ __forceinline__ unsigned bucket_of(int cx, int cy, int cz, unsigned tmask) {
    return (((unsigned)cx * 73856093u) ^ ((unsigned)cy * 19349663u) ^ ((unsigned)cz * 83492791u)) & tmask;
}

__global__ void k_grid_count(const float* __restrict__ xyz, int n, float inv_cell, unsigned tmask,
                             int32_t* __restrict__ counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int3 c = cell_of(xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], inv_cell);
    atomicAdd(&counts[bucket_of(c.x, c.y, c.z, tmask)], 1);
}

#define ISCAN_IPT 4
#define ISCAN_IPB (SCAN_THREADS * ISCAN_IPT)
__global__ void k_iscan_block_sums(const int32_t* __restrict__ v, int n, int32_t* __restrict__ block_sums) {
    int base = blockIdx.x * ISCAN_IPB + threadIdx.x * ISCAN_IPT;
    int s = 0;
#pragma unroll
    for (int j = 0; j < ISCAN_IPT; j++)
        if (base + j < n) s += v[base + j];
    int tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
__global__ void k_iscan_top(const int32_t* __restrict__ block_sums, int nblocks, int32_t* __restrict__ block_off) {
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += SCAN_THREADS) {
        int i = base + threadIdx.x;
        int val = i < nblocks ? block_sums[i] : 0;
        int tot;
        int ex = block_excl_scan(val, &tot);
        int carry = carry_s;
        if (i < nblocks) block_off[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
}
// start[i] = exclusive prefix of counts; start[n] = total; cursor[i] = start[i]
__global__ void k_iscan_apply(const int32_t* __restrict__ v, int n, const int32_t* __restrict__ block_off,
                              int32_t* __restrict__ start, int32_t* __restrict__ cursor) {
    int base = blockIdx.x * ISCAN_IPB + threadIdx.x * ISCAN_IPT;
    int c[ISCAN_IPT];
    int s = 0;
#pragma unroll
    for (int j = 0; j < ISCAN_IPT; j++) {
        c[j] = (base + j < n) ? v[base + j] : 0;
        s += c[j];
    }
    int tot;
    int ex = block_excl_scan(s, &tot) + block_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < ISCAN_IPT; j++) {
        if (base + j < n) {
            start[base + j] = ex;
            cursor[base + j] = ex;
        }
        ex += c[j];
        if (base + j == n - 1) start[n] = ex;
    }
}

__global__ void k_grid_fill(const float* __restrict__ xyz, int n, float inv_cell, unsigned tmask,
                            int32_t* __restrict__ cursor, float4* __restrict__ sorted) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    int3 c = cell_of(x, y, z, inv_cell);
    int pos = atomicAdd(&cursor[bucket_of(c.x, c.y, c.z, tmask)], 1);
    sorted[pos] = make_float4(x, y, z, __int_as_float(i));
}

// exclusive scan of v[0..n) -> start[0..n] (start[n] = total) and cursor = start; block_sums / block_off: gf_iscan_blocks(n)
// words each.  Three launches.  (common.h; also used by geodesic_ms.hip)
int gf_iscan_blocks(int n) { return (n + ISCAN_IPB - 1) / ISCAN_IPB; }
void gf_iscan(const int32_t* v, int n, int32_t* start, int32_t* cursor, int32_t* block_sums, int32_t* block_off,
              hipStream_t st) {
    const int nb = gf_iscan_blocks(n);
    hipLaunchKernelGGL(k_iscan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, st, v, n, block_sums);
    hipLaunchKernelGGL(k_iscan_top, dim3(1), dim3(SCAN_THREADS), 0, st, block_sums, nb, block_off);
    hipLaunchKernelGGL(k_iscan_apply, dim3(nb), dim3(SCAN_THREADS), 0, st, v, n, block_off, start, cursor);
}

#define KNN_WAVES 4
#define KNN_CAP 1024  // in-radius candidates per query point held in LDS
__global__ __launch_bounds__(KNN_WAVES * 64) void k_knn_radius(const float* __restrict__ xyz, int n, int k,
                                                               float radius, float inv_cell, unsigned tmask,
                                                               const int32_t* __restrict__ start,
                                                               const float4* __restrict__ sorted, int sqrt_out,
                                                               float* __restrict__ D, int32_t* __restrict__ I,
                                                               int32_t* __restrict__ deg, int* __restrict__ err) {
    __shared__ unsigned long long s_keys[KNN_WAVES][KNN_CAP];
    __shared__ int s_off[KNN_WAVES][28];
    __shared__ int s_beg[KNN_WAVES][27];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = blockIdx.x * KNN_WAVES + wid;
    if (i >= n) return;  // whole wave exits together (no block-level barrier below)
    const float qx = xyz[i * 3 + 0], qy = xyz[i * 3 + 1], qz = xyz[i * 3 + 2];
    const int3 qc = cell_of(qx, qy, qz, inv_cell);
    // lanes 0..26 fetch the 27 bucket ranges
    int beg = 0, cnt = 0;
    int ncx = 0, ncy = 0, ncz = 0;
    if (lane < 27) {
        ncx = qc.x + lane / 9 - 1;
        ncy = qc.y + (lane / 3) % 3 - 1;
        ncz = qc.z + lane % 3 - 1;
        const unsigned h = bucket_of(ncx, ncy, ncz, tmask);
        beg = start[h];
        cnt = start[h + 1] - beg;
    }
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane < 27) {
        s_off[wid][lane + 1] = inc;
        s_beg[wid][lane] = beg;
    }
    if (lane == 0) s_off[wid][0] = 0;
    const int total = __shfl(inc, 26, 64);
    __builtin_amdgcn_wave_barrier();
    // sweep the concatenated candidate ranges with all 64 lanes
    int nin = 0;
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        bool hit = false;
        unsigned long long key = 0;
        if (t < total) {
            int c = 0;
#pragma unroll
            for (int s = 1; s < 27; s++) c += (s_off[wid][s] <= t) ? 1 : 0;
            const float4 p = sorted[s_beg[wid][c] + (t - s_off[wid][c])];
            // a bucket may hold several cells (hash collisions) and several of the 27 cells may share a
            // bucket: keep the candidate only when it really lies in the cell this slot stands for
            const int3 pc = cell_of(p.x, p.y, p.z, inv_cell);
            const int ex = qc.x + c / 9 - 1, ey = qc.y + (c / 3) % 3 - 1, ez = qc.z + c % 3 - 1;
            if (pc.x == ex && pc.y == ey && pc.z == ez) {
                const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
                const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (sqrtf(d2) <= radius) {
                    hit = true;
                    key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(p.w);
                }
            }
        }
        const unsigned long long bal = __ballot(hit);
        const int pos = nin + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < KNN_CAP) s_keys[wid][pos] = key;
        nin += __popcll(bal);
    }
    if (nin > KNN_CAP) {
        if (lane == 0) *err = 1;  // more in-radius points than the LDS list holds: reported to the host
        nin = KNN_CAP;
    }
    __builtin_amdgcn_wave_barrier();
    // rank by counting: keys are unique (index in the low word), so ranks are a permutation
    for (int t = lane; t < nin; t += 64) {
        const unsigned long long key = s_keys[wid][t];
        int rank = 0;
        for (int j = 0; j < nin; j++) rank += (s_keys[wid][j] < key) ? 1 : 0;
        if (rank < k) {
            const float d2 = __uint_as_float((unsigned)(key >> 32));
            D[(size_t)i * k + rank] = sqrt_out ? sqrtf(d2) : d2;
            I[(size_t)i * k + rank] = (int)(unsigned)(key & 0xffffffffu);
        }
    }
    for (int t = nin + lane; t < k; t += 64) {
        D[(size_t)i * k + t] = __builtin_inff();
        I[(size_t)i * k + t] = -1;
    }
    if (lane == 0 && deg) deg[i] = (nin < k ? nin : k) - 1;
}

static unsigned knn_table_size(int n) {
    unsigned t = 1024;
    while (t < 4u * (unsigned)n && t < (1u << 26)) t <<= 1;
    return t;
}
extern "C" size_t gf_knn_scratch_bytes(int n) {
    const size_t T = knn_table_size(n);
    const size_t nb = (T + ISCAN_IPB - 1) / ISCAN_IPB;
    // counts[T] start[T+1] cursor[T] block_sums[nb] block_off[nb] err[4] sorted[n] float4
    return (3 * T + 8 + 2 * nb + 8) * sizeof(int32_t) + (size_t)(n + 1) * sizeof(float4) + 64;
}

struct PointGrid {
    const int32_t* start;
    const float4* sorted;
    int* err;
    unsigned tmask;
    float inv_cell;
};

// hash grid of one point set with cells of `radius` (x 1.001): counts -> exclusive scan -> (xyz, index) records
// sorted by bucket.  scratch: gf_knn_scratch_bytes(n).
static int build_point_grid(const float* xyz, int n, float radius, void* scratch, hipStream_t st, PointGrid* G,
                            bool ready = false) {
    const unsigned T = knn_table_size(n);
    const int nb = (int)((T + ISCAN_IPB - 1) / ISCAN_IPB);
    int32_t* counts = (int32_t*)scratch;
    int32_t* start = counts + T;
    int32_t* cursor = start + T + 1;
    int32_t* block_sums = cursor + T;
    int32_t* block_off = block_sums + nb;
    int* err = (int*)(block_off + nb);
    uintptr_t sp = ((uintptr_t)(err + 8) + 63) & ~(uintptr_t)63;
    float4* sorted = (float4*)sp;
    const float cell = radius * 1.001f;
    const float inv_cell = 1.0f / cell;
    *G = {start, sorted, err, T - 1, inv_cell};
    if (ready) return GF_OK;  // built by an earlier gf_point_grid_build over the same points and radius
    GF_TRY(hipMemsetAsync(counts, 0, (size_t)T * sizeof(int32_t), st));
    GF_TRY(hipMemsetAsync(err, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_grid_count, dim3(gf_div_up(n, 256)), dim3(256), 0, st, xyz, n, inv_cell, T - 1, counts);
    hipLaunchKernelGGL(k_iscan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, st, counts, (int)T, block_sums);
    hipLaunchKernelGGL(k_iscan_top, dim3(1), dim3(SCAN_THREADS), 0, st, block_sums, nb, block_off);
    hipLaunchKernelGGL(k_iscan_apply, dim3(nb), dim3(SCAN_THREADS), 0, st, counts, (int)T, block_off, start, cursor);
    hipLaunchKernelGGL(k_grid_fill, dim3(gf_div_up(n, 256)), dim3(256), 0, st, xyz, n, inv_cell, T - 1, cursor, sorted);
    return GF_OK;
}

// the grid alone (five small launches that only need the points): a caller with something else on the critical
// path builds it early, on another stream, and hands gf_ball_query_grid the scratch with grid_ready = 1
extern "C" int gf_point_grid_build(const float* xyz, int n, float radius, void* scratch, void* stream) {
    GF_CHECK_ARG(xyz && scratch && n >= 1 && radius > 0.f, "gf_point_grid_build: bad arguments");
    PointGrid G;
    if (int rc = build_point_grid(xyz, n, radius, scratch, (hipStream_t)stream, &G)) return rc;
    GF_CHECK_LAUNCH("gf_point_grid_build");
    return GF_OK;
}

extern "C" int gf_knn_radius(const float* xyz, int n, int k, float radius, int sqrt_out, float* D, int32_t* I,
                             int32_t* deg, void* scratch, void* stream) {
    GF_CHECK_ARG(n >= 0 && k >= 1 && k <= 64 && radius > 0.f, "gf_knn_radius: bad arguments n=%d k=%d r=%g", n, k,
                 radius);
    if (n == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    PointGrid G;
    if (int rc = build_point_grid(xyz, n, radius, scratch, st, &G)) return rc;
    hipLaunchKernelGGL(k_knn_radius, dim3(gf_div_up(n, KNN_WAVES)), dim3(KNN_WAVES * 64), 0, st, xyz, n, k, radius,
                       G.inv_cell, G.tmask, G.start, G.sorted, sqrt_out, D, I, deg, G.err);
    GF_CHECK_LAUNCH("gf_knn_radius");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// ball query over the hash grid (pointnet2 ball_query, ball_query_gpu.cu:10-46: the first `nsample` points in
// INDEX order with d2 < r^2, the rest of the row padded with the first hit, all zeros without any).
// The brute-force kernel (pointops.hip) tests every centre against every point: 10^8 tests for the eval forward's
// 2048 x 50 000, 78 us.  Here a wave walks the 27 buckets around its centre (~270 candidates), keeps the hits'
// indices in LDS and ranks them by counting -- the rank of an index among the hits IS its output column.  A centre
// with more than BQG_CAP hits (never seen at 2 cm voxels and r = 0.2) falls back to the linear scan inside the same
// wave, so the result never depends on the capacity.
// ------------------------------------------------------------------------------------
#define BQG_WAVES 4
#define BQG_CAP 1024
__global__ __launch_bounds__(BQG_WAVES * 64) void k_ball_query_grid(const float* __restrict__ xyz, int n,
                                                                    const int32_t* __restrict__ centre_idx,
                                                                    const float* __restrict__ centres, int m,
                                                                    float radius2, int nsample, PointGrid G,
                                                                    float* __restrict__ new_xyz_out,
                                                                    int32_t* __restrict__ idx) {
    __shared__ __attribute__((aligned(16))) unsigned s_hit[BQG_WAVES][BQG_CAP + 16];
    __shared__ int s_off[BQG_WAVES][28];
    __shared__ int s_beg[BQG_WAVES][27];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int j = blockIdx.x * BQG_WAVES + wid;
    if (j >= m) return;  // whole wave exits together (no block-level barrier below)
    const float* c = centre_idx ? xyz + (size_t)centre_idx[j] * 3 : centres + (size_t)j * 3;
    const float qx = c[0], qy = c[1], qz = c[2];
    if (new_xyz_out && lane < 3) new_xyz_out[(size_t)j * 3 + lane] = lane == 0 ? qx : lane == 1 ? qy : qz;
    const int3 qc = cell_of(qx, qy, qz, G.inv_cell);
    int beg = 0, cnt = 0;
    if (lane < 27) {
        const unsigned h = bucket_of(qc.x + lane / 9 - 1, qc.y + (lane / 3) % 3 - 1, qc.z + lane % 3 - 1, G.tmask);
        beg = G.start[h];
        cnt = G.start[h + 1] - beg;
    }
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane < 27) {
        s_off[wid][lane + 1] = inc;
        s_beg[wid][lane] = beg;
    }
    if (lane == 0) s_off[wid][0] = 0;
    const int total = __shfl(inc, 26, 64);
    __builtin_amdgcn_wave_barrier();
    int nin = 0;
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        bool hit = false;
        unsigned pid = 0;
        if (t < total) {
            int cc = 0;
#pragma unroll
            for (int s = 1; s < 27; s++) cc += (s_off[wid][s] <= t) ? 1 : 0;
            const float4 p = G.sorted[s_beg[wid][cc] + (t - s_off[wid][cc])];
            // several cells may share a bucket and several of the 27 cells may hash alike: count a point only in the
            // slot of the cell it really lies in
            const int3 pc = cell_of(p.x, p.y, p.z, G.inv_cell);
            if (pc.x == qc.x + cc / 9 - 1 && pc.y == qc.y + (cc / 3) % 3 - 1 && pc.z == qc.z + cc % 3 - 1) {
                const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
                hit = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < radius2;
                pid = (unsigned)__float_as_int(p.w);
            }
        }
        const unsigned long long bal = __ballot(hit);
        const int pos = nin + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < BQG_CAP) s_hit[wid][pos] = pid;
        nin += __popcll(bal);
    }
    int32_t* row = idx + (size_t)j * nsample;
    if (nin > BQG_CAP) {
        // more hits than the list holds: the linear scan in index order (same result, just slower)
        int got = 0, first = -1;
        for (int s = 0; s < n && got < nsample; s += 64) {
            const int p = s + lane;
            bool hit = false;
            if (p < n) {
                const float dx = qx - xyz[(size_t)p * 3 + 0], dy = qy - xyz[(size_t)p * 3 + 1],
                            dz = qz - xyz[(size_t)p * 3 + 2];
                hit = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < radius2;
            }
            const unsigned long long bal = __ballot(hit);
            if (bal) {
                if (first < 0) first = s + __builtin_ctzll(bal);
                const int pos = got + __popcll(bal & ((1ull << lane) - 1ull));
                if (hit && pos < nsample) row[pos] = p;
                got += __popcll(bal);
            }
        }
        for (int l = min(got, nsample) + lane; l < nsample; l += 64) row[l] = first < 0 ? 0 : first;
        return;
    }
    // pad the list to a multiple of 16 with a value no index is below: the counting loop reads it 16 at a time
    {
        const int padded = (nin + 15) & ~15;
        if (nin + lane < padded) s_hit[wid][nin + lane] = 0xffffffffu;
    }
    __builtin_amdgcn_wave_barrier();
    // rank by counting: indices are unique, so the ranks are a permutation of 0..nin-1.  Four ds_read_b128 per
    // step, issued together (one key at a time with a dependent LDS read each was 60 us for a centre with 300 hits)
    unsigned lowest = 0xffffffffu;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* list = reinterpret_cast<const u32x4*>(s_hit[wid]);
    const int n16 = (nin + 15) >> 4;
    for (int t = lane; t < nin; t += 64) {
        const unsigned key = s_hit[wid][t];
        int rank = 0;
        for (int u = 0; u < n16; u++) {
            const u32x4 a = list[4 * u + 0], b = list[4 * u + 1], c2 = list[4 * u + 2], d = list[4 * u + 3];
            rank += (a[0] < key) + (a[1] < key) + (a[2] < key) + (a[3] < key) + (b[0] < key) + (b[1] < key) +
                    (b[2] < key) + (b[3] < key) + (c2[0] < key) + (c2[1] < key) + (c2[2] < key) + (c2[3] < key) +
                    (d[0] < key) + (d[1] < key) + (d[2] < key) + (d[3] < key);
        }
        if (rank < nsample) row[rank] = (int)key;
        lowest = min(lowest, key);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lowest = min(lowest, (unsigned)__shfl_xor((int)lowest, d, 64));
    const int pad = nin > 0 ? (int)lowest : 0;
    for (int l = min(nin, nsample) + lane; l < nsample; l += 64) row[l] = pad;
}

// one point set (b = 1): xyz fp32 [n,3]; centres given as indices into xyz (new_xyz [m,3] then receives their
// coordinates) or as coordinates.  scratch: gf_knn_scratch_bytes(n).
extern "C" int gf_ball_query_grid(const float* xyz, int n, const int32_t* centre_idx, const float* centres, int m,
                                  float radius, int nsample, void* scratch, int grid_ready, float* new_xyz,
                                  int32_t* idx, void* stream) {
    GF_CHECK_ARG(xyz && scratch && idx && (centre_idx || centres), "gf_ball_query_grid: null argument");
    GF_CHECK_ARG(n >= 1 && m >= 0 && nsample >= 1 && radius > 0.f, "gf_ball_query_grid: bad sizes");
    if (m == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    PointGrid G;
    if (int rc = build_point_grid(xyz, n, radius, scratch, st, &G, grid_ready != 0)) return rc;
    hipLaunchKernelGGL(k_ball_query_grid, dim3(gf_div_up(m, BQG_WAVES)), dim3(BQG_WAVES * 64), 0, st, xyz, n, centre_idx,
                       centres, m, radius * radius, nsample, G, new_xyz, idx);
    GF_CHECK_LAUNCH("gf_ball_query_grid");
    return GF_OK;
}

// device flag set when a point had more than KNN_CAP in-radius neighbours (result rows truncated)
extern "C" const int32_t* gf_knn_error_flag(void* scratch, int n) {
    const unsigned T = knn_table_size(n);
    const int nb = (int)((T + ISCAN_IPB - 1) / ISCAN_IPB);
    return (const int32_t*)scratch + 3 * (size_t)T + 1 + 2 * (size_t)nb;
}

// ------------------------------------------------------------------------------------
// frontier BFS
// ------------------------------------------------------------------------------------
#define BFS_THREADS 1024
// One thread per frontier vertex: its row segment (<= 16 entries per pass, typical in-radius degree ~14)
// is fetched with 16-byte loads issued together, then the visited probes of all its neighbours are
// issued together, then the atomics -- so a hop costs a handful of dependent memory latencies for up to
// 1024 vertices at once instead of one vertex per 16-lane group.  Queue entries carry the vertex degree
// (fetched during the commit of the previous hop) to take one more load off the chain.
__global__ __launch_bounds__(BFS_THREADS) void k_geodesic_bfs(const float* __restrict__ D, const int32_t* __restrict__ I,
                                                              const int32_t* __restrict__ deg, int n, int K,
                                                              const int32_t* __restrict__ src, float radius,
                                                              int max_step, float* __restrict__ geo,
                                                              unsigned* __restrict__ keys,
                                                              int32_t* __restrict__ queues) {
    __shared__ int s_cnt;
    const int q = blockIdx.x;
    float* g = geo + (size_t)q * n;
    unsigned* key = keys + (size_t)q * n;
    int2* cur = reinterpret_cast<int2*>(queues + (size_t)q * 4 * n);
    int2* nxt = cur + n;
    const int tid = threadIdx.x;
    for (int t = tid; t < n; t += BFS_THREADS) {
        g[t] = -1.0f;
        key[t] = 0xffffffffu;
    }
    const int s = src[q];
    __syncthreads();
    if (tid == 0) {
        g[s] = 0.0f;
        cur[0] = make_int2(s, deg ? deg[s] : K - 1);
        s_cnt = 0;
    }
    __syncthreads();
    int ncur = 1;
    const bool vec = (K & 3) == 0;
    for (int step = 0; step < max_step && ncur > 0; step++) {
        for (int f = tid; f < ncur; f += BFS_THREADS) {
            const int2 e = cur[f];
            const int u = e.x, du = e.y;
            const int32_t* Iu = I + (size_t)u * K;
            const float* Du = D + (size_t)u * K;
            for (int r0 = 0; r0 <= du; r0 += 16) {  // entries r0 .. r0+15 (column 0 = self is skipped)
                int v[16];
                float d[16];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int rb = r0 + 4 * c;
                    if (rb <= du && vec) {
                        const int4 vi = *reinterpret_cast<const int4*>(Iu + rb);
                        const float4 di = *reinterpret_cast<const float4*>(Du + rb);
                        v[4 * c + 0] = vi.x; v[4 * c + 1] = vi.y; v[4 * c + 2] = vi.z; v[4 * c + 3] = vi.w;
                        d[4 * c + 0] = di.x; d[4 * c + 1] = di.y; d[4 * c + 2] = di.z; d[4 * c + 3] = di.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int rr = rb + j;
                            const bool ok = rr <= du && rr < K;
                            v[4 * c + j] = ok ? Iu[rr] : -1;
                            d[4 * c + j] = ok ? Du[rr] : 0.f;
                        }
                    }
                }
                float gv[16];
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int rr = r0 + j;
                    const bool ok = rr >= 1 && rr <= du && v[j] >= 0 && d[j] <= radius;
                    gv[j] = ok ? g[v[j]] : 0.0f;  // 0 = "visited": nothing to do
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (gv[j] < 0.0f) {
                        const unsigned cand = (((unsigned)u << 6) | (unsigned)(r0 + j)) + 1u;
                        const unsigned old = __hip_atomic_fetch_min(&key[v[j]], cand, __ATOMIC_RELAXED, BFS_KEY_SCOPE);
                        if (old == 0xffffffffu) {
                            const int pos = atomicAdd(&s_cnt, 1);
                            nxt[pos].x = v[j];
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int nn = s_cnt;
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        for (int t = tid; t < nn; t += BFS_THREADS) {
            const int v = nxt[t].x;
            const unsigned kk = __hip_atomic_load(&key[v], __ATOMIC_RELAXED, BFS_KEY_SCOPE) - 1u;
            const int u = (int)(kk >> 6), r = (int)(kk & 63u);
            const int dv = deg ? deg[v] : K - 1;
            g[v] = D[(size_t)u * K + r] + g[u];
            nxt[t].y = dv;
        }
        __syncthreads();
        int2* tmp = cur;
        cur = nxt;
        nxt = tmp;
        ncur = nn;
    }
}

// LDS-resident variant (n <= BFS_LDS_MAX_N): the per-query "visited" and "touched" sets are bitmaps in
// LDS and the frontier queues (vertex, distance) live in LDS too (spilling to global memory past the
// LDS capacity), so a hop's dependent chain is: row fetch -> fire-and-forget 64-bit atomicMin carrying
// (parent<<6|rank, distance) -> barrier -> one load of the winning pair -> store.  The 64-bit minimum
// orders by (parent, rank) first, so the distance that rides in the low word is the winner's.
#define BFS_LDS_MAX_N (1 << 19)
#ifndef BFS_LDS_BYTES
#define BFS_LDS_BYTES (150 * 1024)
#endif
// workgroup barrier that orders LDS traffic only: global stores (the distances, never read back by this kernel) and
// loads requested ahead of their use stay in flight across it -- __syncthreads() waits for every one (vmcnt(0))
__device__ __forceinline__ void bfs_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_geodesic_bfs_lds(const float* __restrict__ D,
                                                                  const int32_t* __restrict__ I, int n, int K,
                                                                  const int32_t* __restrict__ src, float radius,
                                                                  int max_step, float* __restrict__ geo,
                                                                  unsigned long long* __restrict__ keys,
                                                                  int2* __restrict__ queues, int qcap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_cnt[2];  // next-frontier counters of even / odd hops (reset a hop ahead: one barrier less)
#ifdef BFS_PRIO
    __builtin_amdgcn_s_setprio(BFS_PRIO);
#endif
    const int nw = (n + 31) >> 5;
    unsigned* visited = reinterpret_cast<unsigned*>(smem);
    unsigned* touched = visited + nw;
    int2* q0 = reinterpret_cast<int2*>(touched + nw + ((2 * nw) & 1));
    int2* q1 = q0 + qcap;
    const int q = blockIdx.x;
    float* g = geo + (size_t)q * n;
    unsigned long long* key = keys + (size_t)q * n;
    int2* gq0 = queues + (size_t)q * 2 * n;  // overflow space of the two queues
    int2* gq1 = gq0 + n;
    const int tid = threadIdx.x;
    for (int t = tid; t < n; t += THREADS) {
        g[t] = -1.0f;
        key[t] = ~0ull;
    }
    for (int t = tid; t < nw; t += THREADS) {
        visited[t] = 0u;
        touched[t] = 0u;
    }
    const int s = src[q];
    __syncthreads();
    if (tid == 0) {
        g[s] = 0.0f;
        visited[s >> 5] = 1u << (s & 31);
        touched[s >> 5] = 1u << (s & 31);
        q0[0] = make_int2(s, __float_as_int(0.0f));
        s_cnt[0] = 0;
        s_cnt[1] = 0;
    }
    __syncthreads();
    int ncur = 1;
    int2 *cl = q0, *cg = gq0, *nl = q1, *ng = gq1;
    // One LANE per row entry: 16 lanes share a frontier vertex and read 16 consecutive entries of its row (one 64-byte
    // segment of I and of D per group), probe the visited bitmap, bid.  A thread per vertex walking 16 entries in
    // registers was bound by instruction issue (9500 of a hop's 14 000 cycles in ~1000 instructions on the one wave
    // its SIMD had; 7-11 of a row's 64 entries are inside the radius); a ring of 220 vertices is 3500 items over all
    // the workgroup's lanes.  What bounds a hop then is the row fetch (~2000 cycles, the graph is larger than L2), so
    // BFS_B items per lane go together (one round trip for a ring of up to BFS_B * THREADS / 16 vertices) and the
    // lanes that expand entry f of the next frontier are the lanes that commit it: they request their row entry right
    // there, and the fetch runs under the commit's own round trip and the barrier instead of after them.
    constexpr int BFS_B = 4;
    const int l16 = tid & 15;
    const int gsh = (tid & 48) | 15;  // the wave's lane that holds this group's last entry
    int pv[BFS_B];
    float pd[BFS_B];
    bool have_pf = false;
#ifdef BFS_TRACE
    // dev build (tools/trace_bfs.py): cycle stamps of the first thread of every query, summed over the hops --
    // 0 expand, 1 atomics done, 2 barrier A, 3 commit, 4 barrier B, 5 hops, 6 sum of ring sizes, 7 largest ring,
    // 8 batches of thread 0, 9 bids of its group, first batch of a hop: 10 entries in registers, 11 bids placed
    unsigned long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define BT() __builtin_amdgcn_s_memtime()
#endif
    for (int step = 0; step < max_step && ncur > 0; step++) {
#ifdef BFS_TRACE
        const unsigned long long t0_ = BT();
        tr[5]++; tr[6] += ncur; tr[7] = tr[7] > (unsigned long long)ncur ? tr[7] : (unsigned long long)ncur;
#endif
        int* cnt = &s_cnt[step & 1];
        auto bid = [&](int u, float gu, int r, int v, float d) {
            const unsigned bit = 1u << (v & 31);
            if (!(visited[v >> 5] & bit)) {
#ifdef BFS_TRACE
                if ((tid >> 4) == 0) tr[9]++;
#endif
                const unsigned cand = (((unsigned)u << 6) | (unsigned)r) + 1u;
                __hip_atomic_fetch_min(&key[v], ((unsigned long long)cand << 32) | (unsigned)__float_as_int(d + gu),
                                       __ATOMIC_RELAXED, BFS_KEY_SCOPE);
                const unsigned old = atomicOr(&touched[v >> 5], bit);
                if (!(old & bit)) {
                    const int pos = atomicAdd(cnt, 1);
                    if (pos < qcap) nl[pos].x = v;
                    else ng[pos - qcap].x = v;
                }
            }
        };
        const int kc = l16 < K ? l16 : K - 1;
        // (the frontier spills out of LDS into global memory only on graphs far larger than a scene's: own instance of the loop)
        auto batch = [&](int base, auto spill, auto pf) {
            unsigned cand[BFS_B];  // (parent << 6 | rank) + 1 of this lane's entry
            int v[BFS_B];
            float d[BFS_B];
            // branch-free (clamped indices, results masked afterwards) so that the loads of the whole batch are
            // requested back to back; the queue is read as LDS or as global memory explicitly -- a pointer that may be
            // either makes every access a flat one, which waits for all outstanding memory traffic
            {
                float gu[BFS_B];
#pragma unroll
                for (int k = 0; k < BFS_B; k++) {
                    const int f = (base + k * THREADS + tid) >> 4;
                    const int fc = f < ncur ? f : ncur - 1;
                    int2 e = cl[fc < qcap ? fc : 0];
                    if constexpr (decltype(spill)::value)
                        if (fc >= qcap) e = cg[fc - qcap];
                    cand[k] = (((unsigned)e.x << 6) | (unsigned)l16) + 1u;
                    gu[k] = __int_as_float(e.y);
                    if constexpr (decltype(pf)::value) {  // requested by the commit of the hop before
                        v[k] = pv[k];
                        d[k] = pd[k];
                    } else {
                        v[k] = I[(size_t)e.x * K + kc];
                        d[k] = D[(size_t)e.x * K + kc];
                    }
                }
#ifdef BFS_TRACE
                if (tid == 0) tr[8]++;
#endif
                // every entry of the batch in its register before the first bid: one round trip for the batch
                // (otherwise the compiler sinks each load to its use, behind the previous item's atomics)
#pragma unroll
                for (int k = 0; k < BFS_B; k++) asm volatile("" : "+v"(v[k]), "+v"(d[k]));
#ifdef BFS_TRACE
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (base == 0) tr[10] += BT() - t0_;
#endif
#pragma unroll
                for (int k = 0; k < BFS_B; k++) {  // v = -1: not an entry inside the radius; d: the distance it bids
                    const int f = (base + k * THREADS + tid) >> 4;
                    if (!(f < ncur && l16 < K && d[k] <= radius)) v[k] = -1;
                    d[k] += gu[k];
                }
            }
#ifdef BFS_TRACE
            const unsigned long long tb0_ = BT();
#endif
            // the batch's LDS traffic in three rounds instead of three per item: the probes of the visited bitmap, the
            // marks in the touched bitmap (returning: the first to mark a vertex queues it), one add to the queue's
            // counter per wave for all the first marks of the batch
            unsigned long long mm = 0ull;  // bit k: this group's entry 15 of item k is inside the radius
            unsigned w[BFS_B];
#pragma unroll
            for (int k = 0; k < BFS_B; k++) {
                mm |= ((__ballot(v[k] >= 0) >> gsh) & 1ull) << k;
                w[k] = visited[v[k] >= 0 ? (v[k] >> 5) : 0];
            }
#pragma unroll
            for (int k = 0; k < BFS_B; k++) {  // w becomes: nonzero = not the first to mark (or no bid at all)
                const unsigned bit = 1u << (v[k] & 31);
                const bool bids = v[k] >= 0 && l16 >= 1 && !(w[k] & bit);
#ifdef BFS_TRACE
                if ((tid >> 4) == 0) tr[9] += bids ? 1 : 0;
#endif
                w[k] = 1u;
                if (bids) {
                    __hip_atomic_fetch_min(&key[v[k]], ((unsigned long long)cand[k] << 32) | (unsigned)__float_as_int(d[k]),
                                           __ATOMIC_RELAXED, BFS_KEY_SCOPE);
                    w[k] = atomicOr(&touched[v[k] >> 5], bit) & bit;
                }
            }
            unsigned long long fm[BFS_B];
            int nfirst = 0;
#pragma unroll
            for (int k = 0; k < BFS_B; k++) {
                fm[k] = __ballot(w[k] == 0u);
                nfirst += __popcll(fm[k]);
            }
            if (nfirst) {  // (uniform over the wave: every lane is here, those past the ring's end with v = -1)
                int pos = 0;
                if ((tid & 63) == 0) pos = atomicAdd(cnt, nfirst);
                pos = __builtin_amdgcn_readfirstlane(pos);
#pragma unroll
                for (int k = 0; k < BFS_B; k++) {
                    if (w[k] == 0u) {
                        const int p = pos + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(fm[k] >> 32),
                                                                             __builtin_amdgcn_mbcnt_lo((unsigned)fm[k], 0u));
                        if (p < qcap) nl[p].x = v[k];
                        else ng[p - qcap].x = v[k];
                    }
                    pos += __popcll(fm[k]);
                }
            }
#ifdef BFS_TRACE
            if (base == 0) tr[11] += BT() - tb0_;
#endif
            // rows are sorted by distance and padded with (inf,-1): a group goes on to the next 16 entries only while
            // its last one is still inside the radius (1-8 % of the rows)
            if (mm) {
#pragma unroll
                for (int k = 0; k < BFS_B; k++) {
                    if ((mm >> k) & 1ull) {
                        const int f = (base + k * THREADS + tid) >> 4;  // (< ncur: the entry was a real one)
                        int2 e = cl[f < qcap ? f : 0];
                        if constexpr (decltype(spill)::value)
                            if (f >= qcap) e = cg[f - qcap];
                        for (int r0 = 16; r0 < K; r0 += 16) {
                            const int r = r0 + l16;
                            int vv = -1;
                            float dd = 0.f;
                            if (r < K) {
                                vv = I[(size_t)e.x * K + r];
                                dd = D[(size_t)e.x * K + r];
                            }
                            const bool in2 = vv >= 0 && dd <= radius;
                            if (in2) bid(e.x, __int_as_float(e.y), r, vv, dd);
                            if (!((__ballot(in2) >> gsh) & 1ull)) break;
                        }
                    }
                }
            }
        };
        auto expand = [&](auto spill) {
            int base = 0;
            if (have_pf) {
                batch(0, spill, std::true_type{});
                base = BFS_B * THREADS;
            }
            for (; (base >> 4) < ncur; base += BFS_B * THREADS) batch(base, spill, std::false_type{});
        };
        if (ncur > qcap) expand(std::true_type{});
        else expand(std::false_type{});
#ifdef BFS_TRACE
        const unsigned long long t1_ = BT();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every atomicMin of this wave has been performed at L2
#ifdef BFS_TRACE
        const unsigned long long t2_ = BT();
#endif
        __syncthreads();
#ifdef BFS_TRACE
        const unsigned long long t3_ = BT();
#endif
        const int nn = *cnt;
        if (tid == 0) s_cnt[(step + 1) & 1] = 0;  // read last a hop ago, two barriers back
        const bool pf_now = step + 1 < max_step;
        auto commit_batch = [&](int base, auto spill, auto pf) {
            int v[BFS_B];
            unsigned kk[BFS_B];  // the distance half of the key (the low word)
#pragma unroll
            for (int k = 0; k < BFS_B; k++) {  // the keys first: loads return in order, and the rows are not waited for
                const int t = (base + k * THREADS + tid) >> 4;
                const int tc = t < nn ? t : nn - 1;
                v[k] = nl[tc < qcap ? tc : 0].x;
                if constexpr (decltype(spill)::value)
                    if (tc >= qcap) v[k] = ng[tc - qcap].x;
                kk[k] = __hip_atomic_load(reinterpret_cast<const unsigned*>(&key[v[k]]), __ATOMIC_RELAXED, BFS_KEY_SCOPE);
            }
            if constexpr (decltype(pf)::value) {
#pragma unroll
                for (int k = 0; k < BFS_B; k++) {
                    pv[k] = I[(size_t)v[k] * K + kc];
                    pd[k] = D[(size_t)v[k] * K + kc];
                }
            }
#pragma unroll
            for (int k = 0; k < BFS_B; k++) {
                const int t = (base + k * THREADS + tid) >> 4;
                if (t < nn && l16 == 0) {
                    const int di = (int)kk[k];
                    g[v[k]] = __int_as_float(di);
                    if (!decltype(spill)::value || t < qcap) nl[t].y = di;
                    else ng[t - qcap].y = di;
                    atomicOr(&visited[v[k] >> 5], 1u << (v[k] & 31));
                }
            }
        };
        auto commit = [&](auto spill) {
            int base = 0;
            if (pf_now && nn > 0) {  // (a join of a path with and one without the row requests would wait for them)
                commit_batch(0, spill, std::true_type{});
                base = BFS_B * THREADS;
            }
            for (; (base >> 4) < nn; base += BFS_B * THREADS) commit_batch(base, spill, std::false_type{});
        };
        if (nn > qcap) commit(std::true_type{});
        else commit(std::false_type{});
        have_pf = pf_now;
        // the next hop reads the queue and the bitmaps (LDS); only a frontier that spilled into global memory needs
        // the stores themselves to have landed
#ifdef BFS_TRACE
        const unsigned long long t4_ = BT();
#endif
        if (nn > qcap) __syncthreads();
        else bfs_lds_barrier();
#ifdef BFS_TRACE
        const unsigned long long t5_ = BT();
        tr[0] += t1_ - t0_; tr[1] += t2_ - t1_; tr[2] += t3_ - t2_; tr[3] += t4_ - t3_; tr[4] += t5_ - t4_;
#endif
        int2* t1 = cl; cl = nl; nl = t1;
        int2* t2 = cg; cg = ng; ng = t2;
        ncur = nn;
    }
#ifdef BFS_TRACE
    if (tid == 0 && n >= 12)  // parked in the query's queue overflow space (unused by then)
        for (int i = 0; i < 12; i++) reinterpret_cast<unsigned long long*>(gq0)[i] = tr[i];
#endif
}

template <int THREADS>
static void launch_bfs_lds(int nq, size_t lds, hipStream_t st, const float* D, const int32_t* I, int n, int K,
                           const int32_t* src, float radius, int max_step, float* geo, void* keys_ws, void* queue_ws,
                           int qcap) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_geodesic_bfs_lds<THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  BFS_LDS_BYTES);
        attr_set = true;
    }
    GF_LAUNCH_OP(GF_OP_BFS, k_geodesic_bfs_lds<THREADS>, dim3(nq), dim3(THREADS), lds, st, D, I, n, K, src, radius, max_step,
                 geo, (unsigned long long*)keys_ws, (int2*)queue_ws, qcap);
}
// dev knob (tests): upper bound of the LDS queue capacity, so that small graphs exercise the global overflow
static int g_bfs_qcap_max = 0;
extern "C" int gf_dev_bfs_qcap_max(int qcap) {
    g_bfs_qcap_max = qcap > 0 ? qcap : 0;
    return GF_OK;
}
// int32 words of queue_ws per query: two queues of (vertex, distance) overflow
extern "C" size_t gf_geodesic_bfs_queue_words(int n) { return (size_t)4 * (size_t)(n > 0 ? n : 0); }

// wg_threads: threads (and, in proportion, LDS) per query.  The kernel spreads a ring's row entries over the lanes, so
// more threads per query is faster when the launch has the chip to itself (S150k eval graphs, 256 queries: 1.06 ms at
// 1024, 1.25 ms at 512, 2.0 ms at 256).  Beside furthest point sampling (13 compute units busy) the 256 queries at
// 1024 threads -- one per compute unit -- need a second round, and 512 is the fastest (two queries can share a unit).
// (Rounds 4-5 also measured a 768-thread form with capped LDS, a distance-pipelined kernel, a launch gated on the
// sampler beside it and six multi-source forms: none faster in the forward, HISTORY.md 4.3 / 7; removed in round 6.)
extern "C" int gf_geodesic_bfs_cfg(const float* D, const int32_t* I, const int32_t* deg, int n, int K,
                                   const int32_t* src, int nq, float radius, int max_step, float* geo, void* keys_ws,
                                   void* queue_ws, size_t queue_words, int wg_threads, void* stream) {
    GF_CHECK_ARG(n >= 1 && K >= 2 && K <= 64 && nq >= 0 && max_step >= 0, "gf_geodesic_bfs: bad arguments");
    GF_CHECK_ARG(n < (1 << 26), "gf_geodesic_bfs: n=%d exceeds the 26-bit parent field", n);
    GF_CHECK_ARG(wg_threads == 1024 || wg_threads == 512 || wg_threads == 256, "gf_geodesic_bfs: wg_threads=%d (256, 512 or 1024)",
                 wg_threads);
    if (nq == 0) return GF_OK;
    // queue_words: int32 words of queue_ws PER QUERY as the caller allocated them (gf_geodesic_bfs_queue_words): checked here
    GF_CHECK_ARG(queue_words >= (size_t)4 * (size_t)n, "gf_geodesic_bfs: queue workspace of %zu words per query, %zu needed",
                 queue_words, (size_t)4 * (size_t)n);
    const int nw = (n + 31) / 32;
    const size_t bm = ((size_t)2 * nw + ((2 * nw) & 1)) * sizeof(unsigned);
    // the two bitmaps must fit the workgroup's LDS share: a large scene moves to the next larger workgroup (and
    // share) before it gives up the LDS-resident kernel altogether
    while (wg_threads < 1024 && bm + 256 * 2 * sizeof(int2) > (size_t)BFS_LDS_BYTES * wg_threads / 1024) wg_threads *= 2;
    const size_t budget = (size_t)BFS_LDS_BYTES * wg_threads / 1024;
    if (n <= BFS_LDS_MAX_N && (K & 3) == 0 && bm + 64 * 2 * sizeof(int2) <= budget) {
        // rows must be distance-sorted and padded with (inf,-1) (gf_knn_radius / faiss order): the
        // LDS variant relies on that to stop scanning a row early
        int qcap = (int)((budget - bm) / (2 * sizeof(int2)));
        if (qcap > n) qcap = n;
        if (g_bfs_qcap_max > 0 && qcap > g_bfs_qcap_max) qcap = g_bfs_qcap_max < 64 ? 64 : g_bfs_qcap_max;
        const size_t lds = bm + (size_t)qcap * 2 * sizeof(int2);
        hipStream_t st = (hipStream_t)stream;
        if (wg_threads == 256)
            launch_bfs_lds<256>(nq, lds, st, D, I, n, K, src, radius, max_step, geo, keys_ws, queue_ws, qcap);
        else if (wg_threads == 512)
            launch_bfs_lds<512>(nq, lds, st, D, I, n, K, src, radius, max_step, geo, keys_ws, queue_ws, qcap);
        else
            launch_bfs_lds<1024>(nq, lds, st, D, I, n, K, src, radius, max_step, geo, keys_ws, queue_ws, qcap);
        GF_CHECK_LAUNCH("gf_geodesic_bfs");
        return GF_OK;
    }
    hipLaunchKernelGGL(k_geodesic_bfs, dim3(nq), dim3(BFS_THREADS), 0, (hipStream_t)stream, D, I, deg, n, K, src,
                       radius, max_step, geo, (unsigned*)keys_ws, (int32_t*)queue_ws);
    GF_CHECK_LAUNCH("gf_geodesic_bfs");
    return GF_OK;
}

extern "C" int gf_geodesic_bfs(const float* D, const int32_t* I, const int32_t* deg, int n, int K, const int32_t* src,
                               int nq, float radius, int max_step, float* geo, void* keys_ws, void* queue_ws,
                               void* stream) {
    // (the legacy entry's contract: nq * 4 n words)
    return gf_geodesic_bfs_cfg(D, I, deg, n, K, src, nq, radius, max_step, geo, keys_ws, queue_ws, (size_t)4 * (size_t)(n > 0 ? n : 0),
                               1024, stream);
}

// ------------------------------------------------------------------------------------
// ballquery_batch_p (lib/pointgroup_ops/src/bfs_cluster/bfs_cluster.cu:15-89): per point, the indices of the
// points of its own batch segment with d2 < r^2 (at most 1000, ascending), packed CSR-like into idx with
// (start, len) per point; entries beyond n*meanActive are dropped and the caller retries with a larger
// buffer (pointgroup_ops.py:136-143).  The reference hands out `start` with an atomic cursor (arbitrary
// order); here a count pass + exclusive scan gives starts in point order -- a canonical instance of it.
// ------------------------------------------------------------------------------------
__global__ void k_bqb_count(const float* __restrict__ xyz, const int32_t* __restrict__ batch_idxs,
                            const int32_t* __restrict__ batch_offsets, int n, float radius2, int32_t* __restrict__ cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ox = xyz[i * 3 + 0], oy = xyz[i * 3 + 1], oz = xyz[i * 3 + 2];
    const int b = batch_idxs[i];
    int c = 0;
    for (int k = batch_offsets[b]; k < batch_offsets[b + 1]; k++) {
        const float dx = ox - xyz[k * 3 + 0], dy = oy - xyz[k * 3 + 1], dz = oz - xyz[k * 3 + 2];
        if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < radius2) {
            if (c >= 1000) break;
            c++;
        }
    }
    cnt[i] = c;
}
__global__ void k_bqb_fill(const float* __restrict__ xyz, const int32_t* __restrict__ batch_idxs,
                           const int32_t* __restrict__ batch_offsets, int n, float radius2,
                           const int32_t* __restrict__ start, const int32_t* __restrict__ cnt, long long thre,
                           int32_t* __restrict__ idx, int32_t* __restrict__ start_len) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = start[i];
    int c = cnt[i];
    start_len[i * 2 + 0] = s;
    start_len[i * 2 + 1] = c;
    if (s >= thre) return;
    if ((long long)s + c >= thre) c = (int)(thre - s);
    const float ox = xyz[i * 3 + 0], oy = xyz[i * 3 + 1], oz = xyz[i * 3 + 2];
    const int b = batch_idxs[i];
    int w = 0;
    for (int k = batch_offsets[b]; k < batch_offsets[b + 1] && w < c; k++) {
        const float dx = ox - xyz[k * 3 + 0], dy = oy - xyz[k * 3 + 1], dz = oz - xyz[k * 3 + 2];
        if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < radius2) idx[s + w++] = k;
    }
}
extern "C" size_t gf_ballquery_batch_p_scratch_bytes(int n) {
    const size_t nb = ((size_t)n + ISCAN_IPB - 1) / ISCAN_IPB;
    return (2 * (size_t)n + 4 + 2 * nb + 16) * sizeof(int32_t);
}
// d_cumsum (device int32) receives the total number of (point, neighbour) pairs (the reference's return value)
extern "C" int gf_ballquery_batch_p(const float* xyz, const int32_t* batch_idxs, const int32_t* batch_offsets, int n,
                                    int meanActive, float radius, int32_t* idx, int32_t* start_len, int32_t* d_cumsum,
                                    void* scratch, void* stream) {
    GF_CHECK_ARG(n >= 0 && meanActive >= 1, "gf_ballquery_batch_p: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        GF_TRY(hipMemsetAsync(d_cumsum, 0, sizeof(int32_t), st));
        return GF_OK;
    }
    const int nb = (n + ISCAN_IPB - 1) / ISCAN_IPB;
    int32_t* cnt = (int32_t*)scratch;
    int32_t* start = cnt + n;          // n + 1 entries
    int32_t* block_sums = start + n + 2;
    int32_t* block_off = block_sums + nb;
    const float r2 = radius * radius;
    hipLaunchKernelGGL(k_bqb_count, dim3(gf_div_up(n, 256)), dim3(256), 0, st, xyz, batch_idxs, batch_offsets, n, r2, cnt);
    hipLaunchKernelGGL(k_iscan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, st, cnt, n, block_sums);
    hipLaunchKernelGGL(k_iscan_top, dim3(1), dim3(SCAN_THREADS), 0, st, block_sums, nb, block_off);
    // start[] and a throw-away cursor copy (idx is large enough to take it: it is overwritten by the fill)
    hipLaunchKernelGGL(k_iscan_apply, dim3(nb), dim3(SCAN_THREADS), 0, st, cnt, n, block_off, start, start_len);
    GF_TRY(hipMemcpyAsync(d_cumsum, start + n, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_bqb_fill, dim3(gf_div_up(n, 256)), dim3(256), 0, st, xyz, batch_idxs, batch_offsets, n, r2,
                       start, cnt, (long long)n * meanActive, idx, start_len);
    GF_CHECK_LAUNCH("gf_ballquery_batch_p");
    return GF_OK;
}
