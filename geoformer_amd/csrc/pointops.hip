// Point-set operators of the set-aggregation stage and the voxel mean (gfx950).
//   voxelize_fp/bp      <- lib/pointgroup_ops/src/voxelize/voxelize.cu:9-53
//   gather_points(+grad) <- lib/pointnet2/_ext_src/src/sampling_gpu.cu:11-60
//   group_points(+grad)  <- lib/pointnet2/_ext_src/src/group_points_gpu.cu:11-78
//   ball_query           <- lib/pointnet2/_ext_src/src/ball_query_gpu.cu:12-57
//   furthest_point_sampling <- lib/pointnet2/_ext_src/src/sampling_gpu.cu:72-232
// All are HBM/latency-bound integer or copy work; none is reshaped into a GEMM.
// Built with -ffp-contract=off: the fmaf() calls below are the only fused operations and
// match oracle/gf_oracle.c term by term.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// ------------------------------------------------------------------------------------
// voxel mean: one thread per (voxel, channel); terms added in rule order, product rounded
// before the add exactly like the reference's atomicAdd(&out, multiplier * inp).
// ------------------------------------------------------------------------------------
__global__ void k_voxelize_fp(const float* __restrict__ feats, const int32_t* __restrict__ rules, int M, int maxActive,
                              int C, int average, float* __restrict__ out) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)M * C) return;
    const int row = (int)(t / C), c = (int)(t - (long long)row * C);
    const int32_t* r = rules + (size_t)row * (maxActive + 1);
    const int n = r[0];
    const float mult = (average && n > 0) ? __fdiv_rn(1.0f, (float)n) : 1.0f;
    float acc = 0.f;
    for (int i = 1; i <= n; i++) acc = __fadd_rn(acc, __fmul_rn(mult, feats[(size_t)r[i] * C + c]));
    out[t] = acc;
}

__global__ void k_voxelize_bp(const float* __restrict__ d_out, const int32_t* __restrict__ rules, int M, int maxActive,
                              int C, int average, float* __restrict__ d_feats) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)M * C) return;
    const int row = (int)(t / C), c = (int)(t - (long long)row * C);
    const int32_t* r = rules + (size_t)row * (maxActive + 1);
    const int n = r[0];
    const float mult = (average && n > 0) ? __fdiv_rn(1.0f, (float)n) : 1.0f;
    const float g = __fmul_rn(mult, d_out[t]);
    // a point id may legally repeat in several rules (point_recover reuses this kernel), so add atomically
    for (int i = 1; i <= n; i++) atomicAdd(&d_feats[(size_t)r[i] * C + c], g);
}

extern "C" int gf_voxelize_fp(const float* feats, const int32_t* rules, int M, int maxActive, int C, int average,
                              float* out, void* stream) {
    GF_CHECK_ARG(M >= 0 && maxActive >= 0 && C >= 1, "gf_voxelize_fp: bad sizes");
    if (M == 0) return GF_OK;
    hipLaunchKernelGGL(k_voxelize_fp, dim3(gf_div_up((long long)M * C, 256)), dim3(256), 0, (hipStream_t)stream, feats,
                       rules, M, maxActive, C, average, out);
    GF_CHECK_LAUNCH("gf_voxelize_fp");
    return GF_OK;
}
extern "C" int gf_voxelize_bp(const float* d_out, const int32_t* rules, int M, int maxActive, int C, int average,
                              float* d_feats, void* stream) {
    GF_CHECK_ARG(M >= 0 && maxActive >= 0 && C >= 1, "gf_voxelize_bp: bad sizes");
    if (M == 0) return GF_OK;
    hipLaunchKernelGGL(k_voxelize_bp, dim3(gf_div_up((long long)M * C, 256)), dim3(256), 0, (hipStream_t)stream, d_out,
                       rules, M, maxActive, C, average, d_feats);
    GF_CHECK_LAUNCH("gf_voxelize_bp");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// gather / group (pure index copies) and their scatter-add gradients
// ------------------------------------------------------------------------------------
__global__ void k_gather_points(const float* __restrict__ points, const int32_t* __restrict__ idx, int b, int c, int n,
                                int m, float* __restrict__ out) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)b * c * m) return;
    const int j = (int)(t % m);
    const long long bc = t / m;
    const int bi = (int)(bc / c);
    out[t] = points[bc * n + idx[(size_t)bi * m + j]];
}
__global__ void k_gather_points_grad(const float* __restrict__ grad_out, const int32_t* __restrict__ idx, int b, int c,
                                     int n, int m, float* __restrict__ grad_points) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)b * c * m) return;
    const int j = (int)(t % m);
    const long long bc = t / m;
    const int bi = (int)(bc / c);
    atomicAdd(&grad_points[bc * n + idx[(size_t)bi * m + j]], grad_out[t]);
}
__global__ void k_group_points(const float* __restrict__ points, const int32_t* __restrict__ idx, int b, int c, int n,
                               int np, int ns, float* __restrict__ out) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)np * ns;
    if (t >= (long long)b * c * per) return;
    const long long js = t % per;
    const long long bc = t / per;
    const int bi = (int)(bc / c);
    out[t] = points[bc * n + idx[(size_t)bi * per + js]];
}
__global__ void k_group_points_grad(const float* __restrict__ grad_out, const int32_t* __restrict__ idx, int b, int c,
                                    int n, int np, int ns, float* __restrict__ grad_points) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)np * ns;
    if (t >= (long long)b * c * per) return;
    const long long js = t % per;
    const long long bc = t / per;
    const int bi = (int)(bc / c);
    atomicAdd(&grad_points[bc * n + idx[(size_t)bi * per + js]], grad_out[t]);
}

#define GF_ELEMENTWISE(fn, kern, total, ...)                                                                    \
    do {                                                                                                        \
        long long tot__ = (total);                                                                              \
        if (tot__ > 0) {                                                                                        \
            hipLaunchKernelGGL(kern, dim3(gf_div_up(tot__, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
            GF_CHECK_LAUNCH(fn);                                                                                \
        }                                                                                                       \
        return GF_OK;                                                                                           \
    } while (0)

extern "C" int gf_gather_points(const float* points, const int32_t* idx, int b, int c, int n, int m, float* out,
                                void* stream) {
    GF_ELEMENTWISE("gf_gather_points", k_gather_points, (long long)b * c * m, points, idx, b, c, n, m, out);
}
extern "C" int gf_gather_points_grad(const float* grad_out, const int32_t* idx, int b, int c, int n, int m,
                                     float* grad_points, void* stream) {
    GF_ELEMENTWISE("gf_gather_points_grad", k_gather_points_grad, (long long)b * c * m, grad_out, idx, b, c, n, m,
                   grad_points);
}
extern "C" int gf_group_points(const float* points, const int32_t* idx, int b, int c, int n, int np, int ns, float* out,
                               void* stream) {
    GF_ELEMENTWISE("gf_group_points", k_group_points, (long long)b * c * np * ns, points, idx, b, c, n, np, ns, out);
}
extern "C" int gf_group_points_grad(const float* grad_out, const int32_t* idx, int b, int c, int n, int np, int ns,
                                    float* grad_points, void* stream) {
    GF_ELEMENTWISE("gf_group_points_grad", k_group_points_grad, (long long)b * c * np * ns, grad_out, idx, b, c, n, np,
                   ns, grad_points);
}

// ------------------------------------------------------------------------------------
// ball query: the first nsample indices (ascending) with d2 < r^2, padded with the first hit;
// rows without a hit stay all-zero.  One wave per centre; the workgroup (8 waves) stages
// 512-point tiles of xyz through LDS so 8 centres share every fetched tile; a wave stops as
// soon as it holds nsample hits (ballot + popcount compaction keeps ascending order).
// ------------------------------------------------------------------------------------
// One wave per centre, 8 centres per workgroup sharing every tile of points.  The tile lives in LDS as three
// coordinate planes so a lane tests four consecutive points per ds_read_b128 triple (256 points per wave trip), and
// the next tile is already in flight in registers while this one is tested (one barrier per tile).  Hits are
// appended in index order: the four compare masks give each lane the number of earlier hits.
#ifndef BQ_WAVES
#define BQ_WAVES 8
#endif
#ifndef BQ_TILE
#define BQ_TILE 2048  // 49 KB of LDS per workgroup; 1024: 88 us, 2048: 78 us on the S150k stage
#endif
#define BQ_PLANE (BQ_TILE + 12)  // plane stride: 16-byte aligned rows, x/y/z of one point in different banks
#define BQ_LD ((BQ_TILE * 3 + BQ_WAVES * 64 - 1) / (BQ_WAVES * 64))
__global__ __launch_bounds__(BQ_WAVES * 64) void k_ball_query(const float* __restrict__ new_xyz,
                                                               const float* __restrict__ xyz, int n, int m,
                                                               float radius2, int nsample, int32_t* __restrict__ idx,
                                                               const int32_t* __restrict__ centre_idx,
                                                               float* __restrict__ new_xyz_out) {
    __shared__ __attribute__((aligned(16))) float tile[2][3 * BQ_PLANE];
    __shared__ int s_done[2][BQ_WAVES];
    const int bi = blockIdx.y;
    xyz += (size_t)bi * n * 3;
    idx += (size_t)bi * m * nsample;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int j = blockIdx.x * BQ_WAVES + wid;
    const bool valid = j < m;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (valid) {
        // centres given as coordinates, or as indices into xyz (then their coordinates are also written out)
        const float* c = centre_idx ? xyz + (size_t)centre_idx[(size_t)bi * m + j] * 3 : new_xyz + ((size_t)bi * m + j) * 3;
        nx = c[0];
        ny = c[1];
        nz = c[2];
        if (centre_idx && lane < 3) new_xyz_out[((size_t)bi * m + j) * 3 + lane] = lane == 0 ? nx : lane == 1 ? ny : nz;
    }
    int cnt = valid ? 0 : nsample;  // invalid waves are "done"
    int first = -1;
    // this thread's slots of a tile: flat float f = threadIdx.x + 512 i  ->  plane f % 3, point f / 3
    int slot[BQ_LD];
#pragma unroll
    for (int i = 0; i < BQ_LD; i++) {
        const unsigned f = threadIdx.x + BQ_WAVES * 64 * i;
        const unsigned p = __umulhi(f, 0xAAAAAAABu) >> 1;
        slot[i] = f < BQ_TILE * 3 ? (int)((f - 3u * p) * BQ_PLANE + p) : -1;
    }
    float nxt[BQ_LD];
    const int ntiles = (n + BQ_TILE - 1) / BQ_TILE;
    auto fetch = [&](int t) {
        const int base3 = t * BQ_TILE * 3, lim = n * 3;
#pragma unroll
        for (int i = 0; i < BQ_LD; i++) {
            const int f = base3 + (int)threadIdx.x + BQ_WAVES * 64 * i;
            nxt[i] = (f < lim && slot[i] >= 0) ? xyz[f] : 1e30f;  // padding points are never inside a ball
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BQ_LD; i++)
            if (slot[i] >= 0) tile[buf][slot[i]] = nxt[i];
    };
    if (ntiles > 0) {
        fetch(0);
        stage(0);
    }
    if (lane == 0) s_done[0][wid] = cnt >= nsample;
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        const int buf = t & 1;
        int alldone = 1;
#pragma unroll
        for (int w = 0; w < BQ_WAVES; w++) alldone &= s_done[buf][w];
        if (alldone) break;  // uniform: every wave of the block has its nsample hits
        if (t + 1 < ntiles) fetch(t + 1);
        if (cnt < nsample) {
            const float* tx = tile[buf];
            const int base = t * BQ_TILE;
            for (int s = 0; s < BQ_TILE && cnt < nsample; s += 256) {  // BQ_TILE is a multiple of 256
                const float4 X = *reinterpret_cast<const float4*>(tx + s + 4 * lane);
                const float4 Y = *reinterpret_cast<const float4*>(tx + BQ_PLANE + s + 4 * lane);
                const float4 Z = *reinterpret_cast<const float4*>(tx + 2 * BQ_PLANE + s + 4 * lane);
                const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
                bool hit[4];
                unsigned long long bal[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float dx = nx - xs[u], dy = ny - ys[u], dz = nz - zs[u];
                    hit[u] = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < radius2;
                    bal[u] = __ballot(hit[u]);
                }
                const unsigned long long any = bal[0] | bal[1] | bal[2] | bal[3];
                if (any) {
                    if (first < 0) {
                        const int l0 = __builtin_ctzll(any);
                        const int u0 = ((bal[0] >> l0) & 1) ? 0 : ((bal[1] >> l0) & 1) ? 1 : ((bal[2] >> l0) & 1) ? 2 : 3;
                        first = base + s + 4 * l0 + u0;
                    }
                    const unsigned long long below = (1ull << lane) - 1ull;
                    int pos = cnt + __popcll(bal[0] & below) + __popcll(bal[1] & below) + __popcll(bal[2] & below) +
                              __popcll(bal[3] & below);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (hit[u]) {
                            if (pos < nsample) idx[(size_t)j * nsample + pos] = base + s + 4 * lane + u;
                            pos++;
                        }
                    }
                    cnt += __popcll(bal[0]) + __popcll(bal[1]) + __popcll(bal[2]) + __popcll(bal[3]);
                }
            }
        }
        if (t + 1 < ntiles) stage(buf ^ 1);
        if (lane == 0) s_done[buf ^ 1][wid] = cnt >= nsample;
        __syncthreads();
    }
    if (valid) {
        const int filled = min(cnt, nsample);
        const int pad = first < 0 ? 0 : first;
        for (int l = filled + lane; l < nsample; l += 64) idx[(size_t)j * nsample + l] = pad;
    }
}

extern "C" int gf_ball_query(const float* new_xyz, const float* xyz, int b, int n, int m, float radius, int nsample,
                             int32_t* idx, void* stream) {
    GF_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample >= 1, "gf_ball_query: bad sizes");
    if (b == 0 || m == 0) return GF_OK;
    const float radius2 = radius * radius;  // fp32 like ball_query_gpu.cu:25
    hipLaunchKernelGGL(k_ball_query, dim3(gf_div_up(m, BQ_WAVES), b), dim3(BQ_WAVES * 64), 0, (hipStream_t)stream,
                       new_xyz, xyz, n, m, radius2, nsample, idx, nullptr, nullptr);
    GF_CHECK_LAUNCH("gf_ball_query");
    return GF_OK;
}

// ball query around xyz[centre_idx] (the gather_operation + ball_query pair of PointnetSAModuleVotes with given
// indices, pointnet2_modules.py:305-312); new_xyz [b,m,3] receives the centre coordinates.
extern "C" int gf_ball_query_centres(const float* xyz, const int32_t* centre_idx, int b, int n, int m, float radius,
                                     int nsample, float* new_xyz, int32_t* idx, void* stream) {
    GF_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample >= 1, "gf_ball_query_centres: bad sizes");
    GF_CHECK_ARG(centre_idx && new_xyz, "gf_ball_query_centres: null argument");
    if (b == 0 || m == 0) return GF_OK;
    const float radius2 = radius * radius;
    hipLaunchKernelGGL(k_ball_query, dim3(gf_div_up(m, BQ_WAVES), b), dim3(BQ_WAVES * 64), 0, (hipStream_t)stream,
                       nullptr, xyz, n, m, radius2, nsample, idx, centre_idx, new_xyz);
    GF_CHECK_LAUNCH("gf_ball_query_centres");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// furthest point sampling.
//
// Reference semantics (sampling_gpu.cu:72-176): start at index 0; each round updates
// temp[k] = min(temp[k], |p_k - p_old|^2) for points with |p_k|^2 > 1e-3 and picks the arg-max.
// Ties are decided by the reference's launch geometry (block size bs = largest power of two
// <= n, capped at 512): inside a thread the lowest k wins (strict '>'), and the shared-memory
// tree keeps the LOWER slot on ties, which makes the winner the candidate with the smallest
// bit-reversed (k mod bs), then the smallest k.  That total order is reproduced here with an
// explicit key, so the work can be laid out for MI355X instead of mimicking the CUDA block:
//
// The 2047 rounds are a serial chain, so what matters is the latency of ONE round.  NW independent
// single-wave workgroups (spread over the chip) keep ALL points and running distances in registers
// (P points per lane).  Per round a wave reduces its slice with DPP (no LDS, no barrier), publishes
// one 8-byte {round tag, distance, key} granule with a write-through store, and every lane polls one
// or two granules of the other waves (MI355X_MICROARCH.md hand-off row "handoff-1to1": a naturally
// aligned 8-byte sc1 store needs no fence).  A round is therefore: ~P*10 VALU ops, two DPP
// reductions, one cross-CU hop and one scalar load of the winner's coordinates.
// ------------------------------------------------------------------------------------
#define FPS_KEY_NONE 0x7fffffffu
#define FPS_SPIN_LIMIT (1 << 22)

__device__ __forceinline__ unsigned dpp_max_step(unsigned v, unsigned o) { return o > v ? o : v; }

// wave-uniform maximum of a 32-bit unsigned value: 4 DPP steps inside each row of 16, 4 readlanes
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    // (old = 0 with bound_ctrl: the identity of max, so the compiler folds the move into v_max_u32_dpp -- one
    // instruction per step instead of v_mov + v_mov_dpp + v_max)
    v = dpp_max_step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad xor 1
    v = dpp_max_step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad xor 2
    v = dpp_max_step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
    v = dpp_max_step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
    const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    const unsigned a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}

// arg-max over the wave of (dist desc, key asc); dist as non-negative float bits, "none" = (0, KEY_NONE).
// The maximum distance usually has a single owner: its key then comes from one readlane instead of a
// second reduction.
__device__ __forceinline__ void wave_best(unsigned& dbits, unsigned& key) {
    const unsigned md = wave_max_u32(dbits);
    const unsigned long long owners = __ballot(dbits == md);
    if (__popcll(owners) == 1) {
        key = (unsigned)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(owners));
    } else {
        const unsigned cand = dbits == md ? (FPS_KEY_NONE - key) : 0u;
        key = FPS_KEY_NONE - wave_max_u32(cand);
    }
    dbits = md;
}

// compile-time loop: the wave-wide reductions inside these bodies keep the optimiser from unrolling an ordinary
// loop, and a runtime-indexed candidate array then lands in LDS (64 KB per workgroup) instead of registers
template <int I, int N, typename F>
__device__ __forceinline__ void fps_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        fps_static_for<I + 1, N>(f);
    }
}

#ifndef FPS_WAVES
#define FPS_WAVES 16  // waves per workgroup
#endif
#ifndef FPS_MAXG
#define FPS_MAXG 16   // cooperating workgroups per point set
#endif
#ifndef FPS_KPUB
#define FPS_KPUB 4    // candidates a workgroup publishes per exchange
#endif
#ifndef FPS_K
#define FPS_K 16      // upper bound of the picks one exchange can deliver
#endif
#ifndef FPS_PIPE
#define FPS_PIPE 1    // 1: wave 0 of a workgroup replays the exchanges and the point-holding waves absorb the picks of an
#endif                //    exchange WHILE it does (below); the waves that share wave 0's SIMD (wave id % 4 == 0) hold no
                      //    points: anything they issued would delay the replay's dependent chain by an issue slot per
                      //    instruction (measured: 580 -> 950 cycles per pick).  0: every wave holds points, picks
                      //    absorbed after the exchange.
#define FPS_PW (FPS_PIPE ? FPS_WAVES - FPS_WAVES / 4 : FPS_WAVES)  // waves of a workgroup that hold points
#ifndef FPS_POLL_SLEEP
#define FPS_POLL_SLEEP 4  // x 64 cycles between two looks at the mailbox (1: the sampler alone 1 % faster, a search workgroup
                          // on the same compute unit 4 % slower -- and the search is the longer launch of the forward)
#endif
#ifndef FPS_PACKED
#define FPS_PACKED 0  // 1: the absorb step on packed fp32 instructions (two points per instruction)
#endif
#define FPS_NG (FPS_MAXG * FPS_KPUB / 64)  // granules a lane gathers
#define FPS_TAG (1ull << 63)

// 64-bit candidate code: [63] exchange tag, [62:32] distance bits (non-negative float), [31:1] KEY_NONE - key,
// [0] "last entry its source forwards"; larger = better; 0 = none
__device__ __forceinline__ unsigned long long fps_code(unsigned dbits, unsigned key) {
    return ((unsigned long long)dbits << 32) | ((unsigned long long)(FPS_KEY_NONE - key) << 1);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long c) {
    const unsigned hi = (unsigned)(c >> 32), md = wave_max_u32(hi);
    const unsigned long long owners = __ballot(hi == md);
    unsigned ml;
    if (__popcll(owners) == 1) {
        ml = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)c, __builtin_ctzll(owners));
    } else {
        ml = wave_max_u32(hi == md ? (unsigned)c : 0u);
    }
    return ((unsigned long long)md << 32) | ml;
}
__device__ __forceinline__ unsigned long long glt_lane_fix(const unsigned long long (&g)[8], int lane) {
    unsigned long long v = g[0];
#pragma unroll
    for (int t = 1; t < 8; t++) v = lane == t ? g[t] : v;
    return v;
}
__device__ __forceinline__ int fps_code_index(unsigned long long c) {
    return c ? (int)((FPS_KEY_NONE - (unsigned)((c >> 1) & 0x7fffffffull)) & 0x3fffffu) : 0;
}

#ifdef FPS_TRACE
// dev build: cycle stamps (s_memtime) of wave 0 of workgroup 0, summed per phase over the exchanges
__device__ unsigned long long* g_fps_trace = nullptr;
extern "C" int gf_dev_fps_trace(void* p) {
    GF_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_fps_trace), &p, sizeof(p)));
    return GF_OK;
}
#define FT() __builtin_amdgcn_s_memtime()
#define FTA(i, a, b) do { ftr[i] += (b) - (a); } while (0)
#else
#define FT() 0ull
#define FTA(i, a, b) do { } while (0)
#endif

// Several picks per exchange.  After the distances have absorbed the picks of the previous exchange,
// the best candidate c1 is the next pick by definition; the runner-up c2 is the pick after that iff
// c1 does not lower its distance (d(c2,c1) >= tmp[c2]): nobody else can then overtake it, ties
// included, because every other distance only decreases and c2 already preceded the rest in the
// (distance desc, key asc) order.  The same argument chains to c3, c4, ...  FPS picks are far apart by
// construction, so ~5.5 of 8 candidates are accepted on ScanNet-like scenes and the serial chain of
// 2047 cross-CU exchanges shrinks to ~370.  Each level (lane -> wave -> workgroup -> grid) forwards
// a sorted prefix of its candidates with the last one flagged, and merging stops after consuming a
// flagged entry (the source's next one is unknown), which keeps the result exact.
// one pick absorbed by a lane's P points: tmp = min(tmp, |p - a|^2) in the reference's operation order
// (dx*dx, then fma dy, then fma dz).  Two points per instruction where the ISA has packed fp32 (v_pk_add / v_pk_mul /
// v_pk_fma_f32 are element-wise IEEE operations: same bits), the minimum per element.
// LDS mailbox accesses of the pipelined absorb: relaxed workgroup-scope atomics keep the LDS address space (a volatile
// access through a cast pointer becomes a FLAT instruction with sc0 sc1: +360 cycles per pick in the replay), the empty asm
// keeps the compiler from reordering them, and the LDS executes one wave's instructions in order.
#define FPS_LDS_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FPS_LDS_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FPS_ORDER() asm volatile("" ::: "memory")
typedef float fps_f2 __attribute__((ext_vector_type(2)));
template <int P>
__device__ __forceinline__ void fps_absorb(const float (&px)[P], const float (&py)[P], const float (&pz)[P], float (&tmp)[P],
                                           float ax, float ay, float az) {
#if FPS_PACKED
    const fps_f2 a2x = {ax, ax}, a2y = {ay, ay}, a2z = {az, az};
#pragma unroll
    for (int i = 0; i + 1 < P; i += 2) {
        const fps_f2 dx = fps_f2{px[i], px[i + 1]} - a2x, dy = fps_f2{py[i], py[i + 1]} - a2y, dz = fps_f2{pz[i], pz[i + 1]} - a2z;
        const fps_f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        tmp[i] = fminf(d.x, tmp[i]);
        tmp[i + 1] = fminf(d.y, tmp[i + 1]);
    }
    if (P & 1) {
        const float dx = px[P - 1] - ax, dy = py[P - 1] - ay, dz = pz[P - 1] - az;
        tmp[P - 1] = fminf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), tmp[P - 1]);
    }
#else
#pragma unroll
    for (int i = 0; i < P; i++) {
        const float dx = px[i] - ax, dy = py[i] - ay, dz = pz[i] - az;
        tmp[i] = fminf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), tmp[i]);
    }
#endif
}

#ifdef FPS_NUM_VGPR  // dev knob: a register cap, so that the workgroup fits beside another kernel's on its compute unit
#define FPS_VGPR_CAP __attribute__((amdgpu_num_vgpr(FPS_NUM_VGPR)))
#else
#define FPS_VGPR_CAP
#endif
template <int P>
__global__ __launch_bounds__(FPS_WAVES * 64) FPS_VGPR_CAP void k_fps(const float* __restrict__ xyz, int n, int m, int m0, int G,
                                                        int bs_log2, int batch0,
                                                        unsigned long long* __restrict__ slots,
                                                        int32_t* __restrict__ idxs, int* __restrict__ err) {
    static_assert(FPS_NG >= 1 && FPS_NG * 64 == FPS_MAXG * FPS_KPUB && FPS_KPUB <= FPS_K && FPS_WAVES * 2 <= 64,
                  "lane mappings of the exchange");
    __shared__ unsigned long long s_part[2][FPS_WAVES * 2];
    __shared__ int s_pick[2][FPS_K + 1];
    __shared__ float s_xyz[2][FPS_K * 3];
    __shared__ unsigned s_prog[2];  // (FPS_PIPE) progress of the exchange being replayed: round << 8 | 0x80 (complete) | picks
    const int bi = batch0 + blockIdx.y, wg = blockIdx.x;
    xyz += (size_t)bi * n * 3;
    idxs += (size_t)bi * m;
    slots += (size_t)bi * 2 * FPS_MAXG * FPS_KPUB;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // FPS_PIPE: wave 0 is the workgroup's coordinator (merge, exchange, replay) and owns no points
    const bool holder = !FPS_PIPE || (wid & 3) != 0;
    const int hid = FPS_PIPE ? wid - 1 - (wid >> 2) : wid;  // index among the workgroup's point-holding waves
    const int gtid = (wg * FPS_PW + hid) * 64 + lane;
    const int stride = G * FPS_PW * 64;
    if (FPS_PIPE) {
        if (threadIdx.x < 2) s_prog[threadIdx.x] = 0u;
        if (!holder && lane < 4) s_part[lane >> 1][wid * 2 + (lane & 1)] = 0ull;  // these waves never forward a candidate
    }

    float px[P], py[P], pz[P], tmp[P];
    unsigned key[P];
    unsigned elig = 0;
    const unsigned bs_mask = (1u << bs_log2) - 1u;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int k = gtid + i * stride;
        px[i] = py[i] = pz[i] = 0.f;
        tmp[i] = 1e10f;
        key[i] = FPS_KEY_NONE;
        if (holder && k < n) {
            px[i] = xyz[(size_t)k * 3 + 0];
            py[i] = xyz[(size_t)k * 3 + 1];
            pz[i] = xyz[(size_t)k * 3 + 2];
            const float mag = fmaf(pz[i], pz[i], fmaf(py[i], py[i], px[i] * px[i]));
            if (!((double)mag <= 1e-3)) elig |= 1u << i;
            const unsigned rev = bs_log2 ? (__brev((unsigned)k & bs_mask) >> (32 - bs_log2)) : 0u;
            key[i] = (rev << 22) | (unsigned)k;
        }
    }
    if (m0 <= 0 && wg == 0 && threadIdx.x == 0) idxs[0] = 0;
    // picks of the previous exchange (uniform): count + coordinates
    int nnew = 1;
    float nx[FPS_K], ny[FPS_K], nz[FPS_K];
    nx[0] = xyz[0]; ny[0] = xyz[1]; nz[0] = xyz[2];
#pragma unroll
    for (int a = 1; a < FPS_K; a++) nx[a] = ny[a] = nz[a] = 0.f;
    int done = 1;  // picks written so far
    if (m0 > 1) {
        // resume: idxs[0..m0) are the first m0 picks of this very sequence (an earlier launch); the state of the
        // algorithm is a function of the pick SET, so absorbing them in any grouping reproduces it exactly
        for (int base = 0; base < m0 - 1; base += FPS_K) {
            const int cntk = min(FPS_K, m0 - 1 - base);
#pragma unroll
            for (int a = 0; a < FPS_K; a++) {
                const int pi = a < cntk ? idxs[base + a] : 0;
                nx[a] = xyz[(size_t)pi * 3 + 0];
                ny[a] = xyz[(size_t)pi * 3 + 1];
                nz[a] = xyz[(size_t)pi * 3 + 2];
            }
#pragma unroll
            for (int i = 0; i < P; i++) {
                if (elig & (1u << i)) {
                    float d2 = tmp[i];
#pragma unroll
                    for (int a = 0; a < FPS_K; a++) {
                        if (a < cntk) {
                            const float dx = px[i] - nx[a], dy = py[i] - ny[a], dz = pz[i] - nz[a];
                            d2 = fminf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), d2);
                        }
                    }
                    tmp[i] = d2;
                }
            }
        }
        // the last known pick is absorbed by the first round like a fresh one
        const int pl = idxs[m0 - 1];
        nx[0] = xyz[(size_t)pl * 3 + 0];
        ny[0] = xyz[(size_t)pl * 3 + 1];
        nz[0] = xyz[(size_t)pl * 3 + 2];
        done = m0;
    }
#ifdef FPS_TRACE
    unsigned long long ftr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
#if FPS_PIPE
    // The picks of an exchange are absorbed WHILE wave 0 replays it: wave 0 posts every accepted pick to LDS (coordinates,
    // then s_prog) and the point-holding waves, which used to idle at a barrier until the whole exchange was known and
    // then absorbed its ~12 picks on the critical path (17 % of the kernel + 20 % waiting for the slowest wave, by cycle
    // stamps), follow it pick by pick.  Only the pick before the loop (the start point / the resumed sequence's last one)
    // is absorbed the old way.
    if (holder) {
#pragma unroll
        for (int i = 0; i < P; i++) {
            if (elig & (1u << i)) {
                const float dx = px[i] - nx[0], dy = py[i] - ny[0], dz = pz[i] - nz[0];
                tmp[i] = fminf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), tmp[i]);
            }
        }
    } else {
        __builtin_amdgcn_s_setprio(3);  // the replay is the kernel's serial chain
    }
    __syncthreads();  // (s_prog zeroed)
#endif
    for (int round = 1; done < m; round++) {
        const int par = round & 1;
        const unsigned long long ft0 = FT();
        // 1) absorb the new picks (FPS_PIPE: done already), track this lane's best
        unsigned bd = 0u, bk = FPS_KEY_NONE;
#pragma unroll
        for (int i = 0; i < P; i++) {
            if (elig & (1u << i)) {
                float d2 = tmp[i];
#if !FPS_PIPE
#pragma unroll
                for (int a = 0; a < FPS_K; a++) {
                    if (a < nnew) {
                        const float dx = px[i] - nx[a], dy = py[i] - ny[a], dz = pz[i] - nz[a];
                        d2 = fminf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), d2);
                    }
                }
                tmp[i] = d2;
#endif
                const unsigned db = __float_as_uint(d2);
                if (db > bd || (db == bd && key[i] < bk)) {
                    bd = db;
                    bk = key[i];
                }
            }
        }
        const unsigned long long ft1 = FT();
        FTA(0, ft0, ft1);
        // 2) wave top-2: best, then the owner of the best exposes its runner-up
        if (holder) {
        unsigned d1 = bd, k1 = bk;
        wave_best(d1, k1);
        unsigned cd = bd, ck = bk;
        if (bk == k1 && k1 != FPS_KEY_NONE) {
            cd = 0u;
            ck = FPS_KEY_NONE;
#pragma unroll
            for (int i = 0; i < P; i++) {
                if ((elig & (1u << i)) && key[i] != k1) {
                    const unsigned db = __float_as_uint(tmp[i]);
                    if (db > cd || (db == cd && key[i] < ck)) {
                        cd = db;
                        ck = key[i];
                    }
                }
            }
        }
        unsigned d2w = cd, k2w = ck;
        wave_best(d2w, k2w);
        if (lane == 0) {
            unsigned long long c1 = fps_code(d1, k1), c2 = fps_code(d2w, k2w);
            if (c2 != 0ull) c2 |= 1ull;  // the last entry this wave forwards
            else if (c1 != 0ull) c1 |= 1ull;
            s_part[par][wid * 2 + 0] = c1;
            s_part[par][wid * 2 + 1] = c2;
        }
        }
        const unsigned long long ft2 = FT();
        FTA(1, ft1, ft2);
        __syncthreads();
        const unsigned long long ft3 = FT();
        FTA(2, ft2, ft3);
        if (wid == 0) {
            // 3) merge the wave candidates into the workgroup's sorted prefix (<= FPS_KPUB entries)
            unsigned long long mine = lane < FPS_WAVES * 2 ? s_part[par][lane] : 0ull;
            unsigned long long wgc[FPS_KPUB];
            bool stop = false;
            int cnt = 0;
            fps_static_for<0, FPS_KPUB>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const unsigned long long best = stop ? 0ull : wave_max_u64(mine);
                wgc[t] = best & ~1ull;
                if (best != 0ull) {
                    if (mine == best) mine = 0ull;
                    if (best & 1ull) stop = true;
                    cnt = t + 1;
                } else {
                    stop = true;
                }
            });
#pragma unroll
            for (int t = 0; t < FPS_KPUB; t++)
                if (t == cnt - 1) wgc[t] |= 1ull;  // ... and the last one this workgroup forwards
            const unsigned long long ft4 = FT();
            FTA(3, ft3, ft4);
            // 4) grid level: publish FPS_KPUB granules, gather everybody's (FPS_NG per lane)
            unsigned long long v[FPS_NG];
            v[0] = wgc[0];
#pragma unroll
            for (int t = 1; t < FPS_KPUB; t++) v[0] = lane == t ? wgc[t] : v[0];
            if (lane >= FPS_KPUB) v[0] = 0ull;
#pragma unroll
            for (int j = 1; j < FPS_NG; j++) v[j] = 0ull;
            if (G > 1) {
                unsigned long long* rs = slots + (size_t)par * FPS_MAXG * FPS_KPUB;
                const unsigned long long tag = ((unsigned long long)((round + 1) >> 1) & 1ull) << 63;
                if (lane < FPS_KPUB)
                    __hip_atomic_store(rs + wg * FPS_KPUB + lane, tag | v[0], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                bool ok[FPS_NG];
                bool all = true;
#pragma unroll
                for (int j = 0; j < FPS_NG; j++) {
                    ok[j] = !(lane + 64 * j < G * FPS_KPUB);
                    all = all && ok[j];
                    v[j] = 0ull;
                }
                int spins = 0;
                while (!all) {
                    all = true;
#pragma unroll
                    for (int j = 0; j < FPS_NG; j++) {
                        if (!ok[j]) {
                            v[j] = __hip_atomic_load(rs + lane + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok[j] = (v[j] & FPS_TAG) == tag;
                            all = all && ok[j];
                        }
                    }
                    if (++spins > FPS_SPIN_LIMIT) {
                        *err = 1;
                        break;
                    }
                }
#pragma unroll
                for (int j = 0; j < FPS_NG; j++) v[j] = (lane + 64 * j < G * FPS_KPUB) ? (v[j] & ~FPS_TAG) : 0ull;
            }
            const unsigned long long ft5 = FT();
            FTA(4, ft4, ft5);
            // coordinates of every gathered candidate, in flight while the merge runs
            float vx[FPS_NG], vy[FPS_NG], vz[FPS_NG];
#pragma unroll
            for (int j = 0; j < FPS_NG; j++) {
                const int ci = fps_code_index(v[j]);
                vx[j] = xyz[(size_t)ci * 3 + 0];
                vy[j] = xyz[(size_t)ci * 3 + 1];
                vz[j] = xyz[(size_t)ci * 3 + 2];
            }
            // 5) picks of this exchange.  Every lane holds one gathered candidate with its coordinates; B is the
            //    best code any source may still be hiding (its last forwarded entry, original value: distances only
            //    fall).  Repeatedly: the best candidate under its UPDATED distance is the next pick as long as its
            //    code is >= B -- every hidden point is below B, every known one below the best -- then all lanes
            //    lower their candidate's distance by the new pick.  A candidate that an accepted pick pulled down
            //    is simply re-ranked, so the exchange keeps going where the prefix rule had to stop (simulated on
            //    the S150k foreground: 174 exchanges for 2048 picks instead of 370).
            static_assert(FPS_NG == 1, "one gathered candidate per lane");
#ifdef FPS_TRACE
            {
                float probe = vx[0];
                asm volatile("v_mov_b32 %0, %0" : "+v"(probe));
                vx[0] = probe;
            }
#endif
            const unsigned long long ft6 = FT();
            FTA(5, ft5, ft6);
            const unsigned long long mine0 = v[0];
            const unsigned long long Bcode = wave_max_u64((mine0 & 1ull) ? (mine0 & ~1ull) : 0ull);
            unsigned ckey = (unsigned)((mine0 >> 1) & 0x7fffffffull);  // KEY_NONE - key
            float cdist = __uint_as_float((unsigned)(mine0 >> 32));
            bool alive = mine0 != 0ull;
            int nacc = 0;
            // lane t keeps pick t (coordinates, index): written out once after the loop instead of four single-lane
            // stores per accepted pick inside the serial chain
            float kx = 0.f, ky = 0.f, kz = 0.f;
            int kidx = 0;
            bool direct = false;
            const unsigned clo = ckey << 1;  // low word of this lane's code (the high word is its current distance)
            const unsigned Bhi = (unsigned)(Bcode >> 32), Blo = (unsigned)Bcode;
            auto take = [&](int ol, unsigned best_lo, int t) {  // the candidate of lane ol is pick t of this exchange
                const float bx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx[0]), ol));
                const float by = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy[0]), ol));
                const float bz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz[0]), ol));
                const bool keeper = lane == t;
                kx = keeper ? bx : kx;
                ky = keeper ? by : ky;
                kz = keeper ? bz : kz;
                kidx = keeper ? (int)((FPS_KEY_NONE - ((best_lo >> 1) & 0x7fffffffu)) & 0x3fffffu) : kidx;
                if (lane == ol) alive = false;
#if FPS_PIPE
                if (lane == 0) {  // posted at once: the other waves absorb it while the replay goes on (LDS keeps a wave's order)
                    FPS_LDS_ST(&s_xyz[par][t * 3 + 0], bx);
                    FPS_LDS_ST(&s_xyz[par][t * 3 + 1], by);
                    FPS_LDS_ST(&s_xyz[par][t * 3 + 2], bz);
                    FPS_ORDER();
                    FPS_LDS_ST(&s_prog[par], ((unsigned)round << 8) | (unsigned)(t + 1));
                }
#endif
                const float dx = vx[0] - bx, dy = vy[0] - by, dz = vz[0] - bz;
                cdist = fminf(cdist, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            };
            {
                // first pick of the exchange: the best code whatever its distance (a live candidate at distance 0 counts)
                const unsigned long long my = alive ? (((unsigned long long)__float_as_uint(cdist) << 32) | clo) : 0ull;
                const unsigned long long best = wave_max_u64(my);
                if (best == 0ull) {  // nothing eligible anywhere: index 0 like the reference
                    if (lane == 0) {
                        FPS_LDS_ST(&s_xyz[par][0], xyz[0]);
                        FPS_LDS_ST(&s_xyz[par][1], xyz[1]);
                        FPS_LDS_ST(&s_xyz[par][2], xyz[2]);
                        if (wg == 0) idxs[done] = 0;
                    }
                    direct = true;
                } else {
                    take(__builtin_ctzll(__ballot(alive && my == best)), (unsigned)best, 0);
                }
                nacc = 1;
            }
            if (!direct) {
                // further picks: the best candidate under its UPDATED distance while that is positive (a candidate at
                // distance 0 never overtakes the accepted picks, which sit at 0 with lower keys) and its code is not
                // below B.  The reduction runs on the 32-bit distances; the key word is read from the owner's lane
                // (a second reduction only when two candidates share the largest distance).
#pragma unroll 1
                for (int t = 1; t < FPS_K && done + t < m; t++) {
                    const unsigned hi = alive ? __float_as_uint(cdist) : 0u;
                    const unsigned md = wave_max_u32(hi);
                    if (md == 0u) break;
                    const unsigned long long owners = __ballot(hi == md);
                    int ol;
                    unsigned blo;
                    if (__popcll(owners) == 1) {
                        ol = __builtin_ctzll(owners);
                        blo = (unsigned)__builtin_amdgcn_readlane((int)clo, ol);
                    } else {
                        blo = wave_max_u32(hi == md ? clo : 0u);
                        ol = __builtin_ctzll(__ballot(hi == md && clo == blo));
                    }
                    if (md < Bhi || (md == Bhi && blo < Blo)) break;
                    take(ol, blo, t);
                    nacc = t + 1;
                }
            }
#if FPS_PIPE
            if (!direct && lane < nacc && wg == 0) idxs[done + lane] = kidx;
            FPS_ORDER();
            if (lane == 0) FPS_LDS_ST(&s_prog[par], ((unsigned)round << 8) | 0x80u | (unsigned)nacc);
            nnew = nacc;
            (void)kx; (void)ky; (void)kz;
#else
            if (!direct && lane < nacc) {
                s_xyz[par][lane * 3 + 0] = kx;
                s_xyz[par][lane * 3 + 1] = ky;
                s_xyz[par][lane * 3 + 2] = kz;
                if (wg == 0) idxs[done + lane] = kidx;
            }
            if (lane == 0) s_pick[par][0] = nacc;
#endif
            const unsigned long long ft7 = FT();
            FTA(6, ft6, ft7);
        }
#if FPS_PIPE
        else if (holder) {
            // the point-holding waves follow the replay: absorb every pick as soon as wave 0 has posted it.  s_prog
            // carries the round, so a value two rounds old reads as "nothing yet"; wave 0 cannot start the next replay
            // into this buffer before every wave has left this loop (the barrier of the round in between).
            // (no eligibility test in here: an ineligible slot's distance is never read)
            const unsigned tag = (unsigned)round << 8;
            int seen = 0;
            for (;;) {
                const unsigned c = FPS_LDS_LD(&s_prog[par]);
                FPS_ORDER();
                const bool cur = (c & 0xffffff00u) == tag;
                const int cnt = cur ? (int)(c & 0x7fu) : 0;
                for (int a = seen; a < cnt; a++) {
                    const float ax = FPS_LDS_LD(&s_xyz[par][a * 3 + 0]), ay = FPS_LDS_LD(&s_xyz[par][a * 3 + 1]),
                                az = FPS_LDS_LD(&s_xyz[par][a * 3 + 2]);
#ifndef FPS_EXP_NOABSORB  // dev experiment (wrong picks): what the absorbing waves cost a neighbour on the compute unit
                    fps_absorb<P>(px, py, pz, tmp, ax, ay, az);
#else
                    (void)ax; (void)ay; (void)az;
#endif
                }
                seen = cnt;
                if (cur && (c & 0x80u)) break;
                __builtin_amdgcn_s_sleep(FPS_POLL_SLEEP);
            }
            nnew = seen;
        } else {
            // a wave on wave 0's SIMD: sleeps through the replay, only needs the number of picks
            const unsigned tag = (unsigned)round << 8;
            unsigned c;
            for (;;) {
                c = FPS_LDS_LD(&s_prog[par]);
                if ((c & 0xffffff80u) == (tag | 0x80u)) break;
                __builtin_amdgcn_s_sleep(32);
            }
            nnew = (int)(c & 0x7fu);
        }
        (void)s_pick;
#else
        const unsigned long long ft8 = FT();
        __syncthreads();
        const unsigned long long ft9 = FT();
        FTA(7, ft8, ft9);
        nnew = s_pick[par][0];
#pragma unroll
        for (int a = 0; a < FPS_K; a++) {
            nx[a] = s_xyz[par][a * 3 + 0];
            ny[a] = s_xyz[par][a * 3 + 1];
            nz[a] = s_xyz[par][a * 3 + 2];
        }
#endif
        done += nnew;
#ifdef FPS_TRACE
        ftr[8] += 1;
        ftr[9] += nnew;
#endif
    }
#ifdef FPS_TRACE
    if (g_fps_trace && wg == 0 && threadIdx.x == 0)
        for (int i = 0; i < 10; i++) g_fps_trace[i] = ftr[i];
#endif
}

template <int P>
static void launch_fps(int G, int nb, hipStream_t st, const float* xyz, int n, int m, int m0, int bs_log2, int batch0,
                       unsigned long long* slots, int32_t* idxs, int* err) {
    GF_LAUNCH_OP(GF_OP_FPS, (k_fps<P>), dim3(G, nb), dim3(FPS_WAVES * 64), 0, st, xyz, n, m, m0, G, bs_log2, batch0, slots,
                 idxs, err);
}

extern "C" size_t gf_fps_scratch_bytes(int b) {
    return ((size_t)b * 2 * FPS_MAXG * FPS_KPUB + 8) * sizeof(unsigned long long);
}

extern "C" int gf_furthest_point_sampling_resume(const float* xyz, int b, int n, int m, int m_known, int32_t* idxs,
                                                 void* scratch, void* stream);

extern "C" int gf_furthest_point_sampling(const float* xyz, int b, int n, int m, int32_t* idxs, void* scratch,
                                          void* stream) {
    return gf_furthest_point_sampling_resume(xyz, b, n, m, 0, idxs, scratch, stream);
}

// (Round 5 also had a form of this launch that told a search kernel beside it when the first picks were stored, and an
// LDS pad that kept other kernels off the sampler's compute units: no gain in the forward, HISTORY.md 7; removed.)
extern "C" int gf_furthest_point_sampling_resume(const float* xyz, int b, int n, int m, int m_known, int32_t* idxs,
                                                 void* scratch, void* stream) {
    GF_CHECK_ARG(b >= 0 && n >= 1 && m >= 0, "gf_furthest_point_sampling: bad sizes b=%d n=%d m=%d", b, n, m);
    GF_CHECK_ARG(m_known >= 0 && m_known <= m, "gf_furthest_point_sampling_resume: m_known=%d not in [0, m=%d]", m_known,
                 m);
    if (m_known == m && m > 0) return GF_OK;
    GF_CHECK_ARG(n < (1 << 22), "gf_furthest_point_sampling: n=%d exceeds the 22-bit index of the tie-break key", n);
    if (b == 0 || m == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    // reference launch geometry decides ties: bs = largest power of two <= n, capped at 512
    int bs_log2 = 0;
    while ((2 << bs_log2) <= n && bs_log2 < 9) bs_log2++;
    const int per_wg = FPS_PW * 64;  // lanes that hold points
    int G = (2 * n + per_wg * 5 - 1) / (per_wg * 5);  // ~2.5 points per lane (40 000 points: 16 workgroups; 13 at 3 per lane cost the forward 1.8 %)
    if (G < 1) G = 1;
    if (G > FPS_MAXG) G = FPS_MAXG;
    const int P = (n + G * per_wg - 1) / (G * per_wg);
    GF_CHECK_ARG(P <= 22, "gf_furthest_point_sampling: n=%d too large (max %d)", n, FPS_MAXG * per_wg * 22);
    unsigned long long* slots = (unsigned long long*)scratch;
    int* err = (int*)(slots + (size_t)b * 2 * FPS_MAXG * FPS_KPUB);
    GF_TRY(hipMemsetAsync(scratch, 0, gf_fps_scratch_bytes(b), st));
    const int per_launch = 1024 / (G * FPS_WAVES) > 0 ? 1024 / (G * FPS_WAVES) : 1;  // all cooperating waves resident
    for (int b0 = 0; b0 < b; b0 += per_launch) {
        const int nb = (b - b0) < per_launch ? (b - b0) : per_launch;
        if (P <= 1) launch_fps<1>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 2) launch_fps<2>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 3) launch_fps<3>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 4) launch_fps<4>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 5) launch_fps<5>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 6) launch_fps<6>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 8) launch_fps<8>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 12) launch_fps<12>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 16) launch_fps<16>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else if (P <= 20) launch_fps<20>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
        else launch_fps<22>(G, nb, st, xyz, n, m, m_known, bs_log2, b0, slots, idxs, err);
    }
    GF_CHECK_LAUNCH("gf_furthest_point_sampling");
    return GF_OK;
}
