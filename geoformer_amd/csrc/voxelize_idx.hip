// voxelize_idx on the GPU (reference: lib/pointgroup_ops/src/voxelize/voxelize.cpp:10-152, a CPU hash
// map walked in point order inside the DataLoader workers).
//
// Same result, data-parallel:
//   1. open-addressing table keyed by the packed (b,x,y,z): every point claims / finds its voxel's slot and lowers
//      the slot's "first point" with an atomicMin;
//   2. a point is the first of its voxel iff first[slot] == i; an exclusive scan of those flags in POINT order is
//      the voxel id of the reference's insertion counter (voxel ids in order of first occurrence);
//   3. per-voxel counts, their scan, a scatter of the point ids and a tiny per-voxel sort give the rule rows
//      [count, point ids ascending, 0 padding]; the coordinates of rule[1] are the voxel's output coordinates.
// Two entry points because the caller sizes the outputs from (M, maxActive): *_count fills input_map and returns the two
// numbers in device memory, *_fill writes output_coords / output_map.
#include "common.h"

#define VX_EMPTY 0xffffffffffffffffull
#define VX_THREADS 256

struct VxScratch {
    unsigned long long* keys;  // [T]
    int* first;                // [T] lowest point index of the slot's voxel
    int* slot_of;              // [N]
    int* vid_of_point;         // [N] exclusive scan of the "first" flags
    int* cnt;                  // [N+1] per voxel
    int* offs;                 // [N+1] exclusive scan of cnt
    int* cursor;               // [N+1]
    int* tmp;                  // [N] point ids grouped by voxel
    int* block_sums;           // [nb] x 2
    int* block_off;
    int* err;                  // [4]: error flag, M, maxActive
    unsigned T;
};

static unsigned vx_table_size(int N) {
    unsigned t = 1024;
    while (t < 2u * (unsigned)(N > 0 ? N : 1) && t < (1u << 30)) t <<= 1;
    return t;
}
static int vx_nb(int N) { return (N + VX_THREADS - 1) / VX_THREADS + 1; }

extern "C" size_t gf_voxelize_idx_scratch_bytes(int N) {
    const size_t T = vx_table_size(N), n = (size_t)(N > 0 ? N : 0);
    return T * 8 + T * 4 + (6 * (n + 1) + 2 * vx_nb(N) + 16) * 4 + 256;
}

static VxScratch vx_carve(void* scratch, int N) {
    VxScratch s;
    s.T = vx_table_size(N);
    const size_t n1 = (size_t)(N > 0 ? N : 0) + 1;
    unsigned char* p = (unsigned char*)scratch;
    s.keys = (unsigned long long*)p;
    p += (size_t)s.T * 8;
    int* q = (int*)p;
    s.first = q;
    q += s.T;
    s.slot_of = q;
    q += n1;
    s.vid_of_point = q;
    q += n1;
    s.cnt = q;
    q += n1;
    s.offs = q;
    q += n1;
    s.cursor = q;
    q += n1;
    s.tmp = q;
    q += n1;
    s.block_sums = q;
    q += vx_nb(N);
    s.block_off = q;
    q += vx_nb(N);
    s.err = q;
    return s;
}

__device__ __forceinline__ unsigned long long vx_pack(const long long* c, int ncol, int* err) {
    const long long b = ncol == 4 ? c[0] : 0, x = c[ncol - 3], y = c[ncol - 2], z = c[ncol - 1];
    if ((unsigned long long)b > 0xffffull || (unsigned long long)x > 0xffffull || (unsigned long long)y > 0xffffull ||
        (unsigned long long)z > 0xffffull)
        *err = 1;  // outside the 16-bit fields of the packed key (negative or > 65535)
    return ((unsigned long long)(b & 0xffff) << 48) | ((unsigned long long)(x & 0xffff) << 32) |
           ((unsigned long long)(y & 0xffff) << 16) | (unsigned long long)(z & 0xffff);
}

__global__ void k_vx_insert(const long long* __restrict__ coords, int N, int ncol, VxScratch s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned long long key = vx_pack(coords + (size_t)i * ncol, ncol, s.err);
    unsigned long long h = key * 0x9E3779B97F4A7C15ull;
    unsigned slot = (unsigned)(h >> 32) & (s.T - 1);
    while (true) {
        const unsigned long long old = atomicCAS(&s.keys[slot], VX_EMPTY, key);
        if (old == VX_EMPTY || old == key) break;
        slot = (slot + 1) & (s.T - 1);
    }
    atomicMin(&s.first[slot], i);
    s.slot_of[i] = slot;
}

// exclusive scan of per-element values produced by `f` (three launches: block sums, top, apply)
template <int WHAT>
__device__ __forceinline__ int vx_value(const VxScratch& s, int i, int n) {
    if (i >= n) return 0;
    if (WHAT == 0) return s.first[s.slot_of[i]] == i ? 1 : 0;  // "first point of its voxel"
    return s.cnt[i];                                          // per-voxel counts
}
template <int WHAT>
__global__ __launch_bounds__(SCAN_THREADS) void k_vx_block_sums(VxScratch s, int n) {
    int total;
    block_excl_scan(vx_value<WHAT>(s, blockIdx.x * SCAN_THREADS + threadIdx.x, n), &total);
    if (threadIdx.x == 0) s.block_sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_vx_top(VxScratch s, int nb, int which) {
    // single block: exclusive scan of the block sums; the grand total goes to err[1 + which]
    int carry = 0;
    for (int base = 0; base < nb; base += SCAN_THREADS) {
        const int i = base + threadIdx.x;
        int total;
        const int ex = block_excl_scan(i < nb ? s.block_sums[i] : 0, &total);
        if (i < nb) s.block_off[i] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) s.err[1 + which] = carry;
}
template <int WHAT>
__global__ __launch_bounds__(SCAN_THREADS) void k_vx_apply(VxScratch s, int n) {
    const int i = blockIdx.x * SCAN_THREADS + threadIdx.x;
    int total;
    const int ex = block_excl_scan(vx_value<WHAT>(s, i, n), &total);
    if (i < n) (WHAT == 0 ? s.vid_of_point : s.offs)[i] = s.block_off[blockIdx.x] + ex;
}

__global__ void k_vx_map(int N, VxScratch s, int32_t* __restrict__ input_map) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int v = s.vid_of_point[s.first[s.slot_of[i]]];
    input_map[i] = v;
    atomicAdd(&s.cnt[v], 1);
}

__global__ void k_vx_max(VxScratch s, int mode) {
    // grid-stride maximum of the counts over the M voxels (M = err[1]); modes 0-2 keep one point per voxel
    const int M = s.err[1];
    int mx = 1;
    if (mode == 3 || mode == 4)
        for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < M; v += gridDim.x * blockDim.x) mx = max(mx, s.cnt[v]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&s.err[3], mx);
}

__global__ void k_vx_scatter(int N, VxScratch s, const int32_t* __restrict__ input_map) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int v = input_map[i];
    s.tmp[s.offs[v] + atomicAdd(&s.cursor[v], 1)] = i;
}

__global__ void k_vx_rows(const long long* __restrict__ coords, int ncol, int M, int maxActive, int mode, VxScratch s,
                          long long* __restrict__ out_coords, int32_t* __restrict__ out_map) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    const int c = s.cnt[v];
    int* p = s.tmp + s.offs[v];
    for (int a = 1; a < c; a++) {  // insertion sort: the scatter order is arbitrary, the rule rows are ascending
        const int x = p[a];
        int b = a - 1;
        while (b >= 0 && p[b] > x) {
            p[b + 1] = p[b];
            b--;
        }
        p[b + 1] = x;
    }
    int32_t* row = out_map + (size_t)v * (maxActive + 1);
    int lead;
    if (mode == 3 || mode == 4) {
        row[0] = c;
        for (int a = 0; a < maxActive; a++) row[1 + a] = a < c ? p[a] : 0;
        lead = p[0];
    } else {
        lead = mode == 2 ? p[c - 1] : p[0];  // voxelize.cpp:125-136: mode 2 keeps back(), modes 0/1 front()
        row[0] = 1;
        row[1] = lead;
    }
    for (int a = 0; a < ncol; a++) out_coords[(size_t)v * ncol + a] = coords[(size_t)lead * ncol + a];
}

extern "C" int gf_voxelize_idx_count(const long long* coords, int N, int ncol, int mode, void* scratch,
                                     int32_t* input_map, int32_t* d_M_maxActive, void* stream) {
    GF_CHECK_ARG(N >= 0 && (ncol == 3 || ncol == 4), "gf_voxelize_idx_count: coords must be [N,3] or [N,4]");
    GF_CHECK_ARG(mode >= 0 && mode <= 4, "gf_voxelize_idx_count: mode %d", mode);
    hipStream_t st = (hipStream_t)stream;
    VxScratch s = vx_carve(scratch, N);
    const size_t n1 = (size_t)N + 1;
    GF_TRY(hipMemsetAsync(s.keys, 0xff, (size_t)s.T * 8, st));
    GF_TRY(hipMemsetAsync(s.first, 0x7f, (size_t)s.T * 4, st));
    GF_TRY(hipMemsetAsync(s.cnt, 0, n1 * 4, st));
    GF_TRY(hipMemsetAsync(s.cursor, 0, n1 * 4, st));
    GF_TRY(hipMemsetAsync(s.err, 0, 4 * 4, st));
    if (N > 0) {
        const int nb = gf_div_up(N, SCAN_THREADS);
        hipLaunchKernelGGL(k_vx_insert, dim3(gf_div_up(N, VX_THREADS)), dim3(VX_THREADS), 0, st, coords, N, ncol, s);
        hipLaunchKernelGGL(k_vx_block_sums<0>, dim3(nb), dim3(SCAN_THREADS), 0, st, s, N);
        hipLaunchKernelGGL(k_vx_top, dim3(1), dim3(SCAN_THREADS), 0, st, s, nb, 0);
        hipLaunchKernelGGL(k_vx_apply<0>, dim3(nb), dim3(SCAN_THREADS), 0, st, s, N);
        hipLaunchKernelGGL(k_vx_map, dim3(gf_div_up(N, VX_THREADS)), dim3(VX_THREADS), 0, st, N, s, input_map);
        // counts are indexed by voxel id < M <= N: scanning all N entries is harmless (zeros past M)
        hipLaunchKernelGGL(k_vx_block_sums<1>, dim3(nb), dim3(SCAN_THREADS), 0, st, s, N);
        hipLaunchKernelGGL(k_vx_top, dim3(1), dim3(SCAN_THREADS), 0, st, s, nb, 1);
        hipLaunchKernelGGL(k_vx_apply<1>, dim3(nb), dim3(SCAN_THREADS), 0, st, s, N);
        hipLaunchKernelGGL(k_vx_max, dim3(64), dim3(256), 0, st, s, mode);
    }
    // d_M_maxActive[0] = M, [1] = maxActive, [2] = error flag (coordinate outside the packed key's range)
    GF_TRY(hipMemcpyAsync(d_M_maxActive, s.err + 1, 4, hipMemcpyDeviceToDevice, st));
    GF_TRY(hipMemcpyAsync(d_M_maxActive + 1, s.err + 3, 4, hipMemcpyDeviceToDevice, st));
    GF_TRY(hipMemcpyAsync(d_M_maxActive + 2, s.err, 4, hipMemcpyDeviceToDevice, st));
    GF_CHECK_LAUNCH("gf_voxelize_idx_count");
    return GF_OK;
}

extern "C" int gf_voxelize_idx_fill(const long long* coords, int N, int ncol, int mode, void* scratch,
                                    const int32_t* input_map, int M, int maxActive, long long* out_coords,
                                    int32_t* out_map, void* stream) {
    GF_CHECK_ARG(N >= 0 && M >= 0 && M <= N && maxActive >= 1, "gf_voxelize_idx_fill: bad sizes N=%d M=%d maxActive=%d",
                 N, M, maxActive);
    if (N == 0 || M == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    VxScratch s = vx_carve(scratch, N);
    hipLaunchKernelGGL(k_vx_scatter, dim3(gf_div_up(N, VX_THREADS)), dim3(VX_THREADS), 0, st, N, s, input_map);
    hipLaunchKernelGGL(k_vx_rows, dim3(gf_div_up(M, VX_THREADS)), dim3(VX_THREADS), 0, st, coords, ncol, M, maxActive,
                       mode, s, out_coords, out_map);
    GF_CHECK_LAUNCH("gf_voxelize_idx_fill");
    return GF_OK;
}
