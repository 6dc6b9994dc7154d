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
#define BQ_WAVES 8
#define BQ_TILE 512
__global__ __launch_bounds__(BQ_WAVES * 64) void k_ball_query(const float* __restrict__ new_xyz,
                                                               const float* __restrict__ xyz, int n, int m,
                                                               float radius2, int nsample, int32_t* __restrict__ idx) {
    __shared__ float tile[BQ_TILE * 3];
    __shared__ int done_count;
    const int bi = blockIdx.y;
    xyz += (size_t)bi * n * 3;
    new_xyz += (size_t)bi * m * 3;
    idx += (size_t)bi * m * nsample;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int j = blockIdx.x * BQ_WAVES + wid;
    const bool valid = j < m;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (valid) {
        nx = new_xyz[j * 3 + 0];
        ny = new_xyz[j * 3 + 1];
        nz = new_xyz[j * 3 + 2];
    }
    int cnt = valid ? 0 : nsample;  // invalid waves are "done"
    int first = -1;
    for (int base = 0; base < n; base += BQ_TILE) {
        if (threadIdx.x == 0) done_count = 0;
        __syncthreads();
        const int tn = min(BQ_TILE, n - base);
        for (int t = threadIdx.x; t < tn * 3; t += BQ_WAVES * 64) tile[t] = xyz[(size_t)base * 3 + t];
        if (lane == 0 && cnt >= nsample) atomicAdd(&done_count, 1);
        __syncthreads();
        if (done_count == BQ_WAVES) break;  // uniform: every wave of the block has its nsample hits
        if (cnt < nsample) {
            for (int s = 0; s < tn && cnt < nsample; s += 64) {
                const int p = s + lane;
                bool hit = false;
                if (p < tn) {
                    const float dx = nx - tile[p * 3 + 0], dy = ny - tile[p * 3 + 1], dz = nz - tile[p * 3 + 2];
                    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    hit = d2 < radius2;
                }
                const unsigned long long bal = __ballot(hit);
                if (bal) {
                    if (first < 0) first = base + s + __builtin_ctzll(bal);
                    const int pos = cnt + __popcll(bal & ((1ull << lane) - 1ull));
                    if (hit && pos < nsample) idx[(size_t)j * nsample + pos] = base + p;
                    cnt += __popcll(bal);
                }
            }
        }
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
                       new_xyz, xyz, n, m, radius2, nsample, idx);
    GF_CHECK_LAUNCH("gf_ball_query");
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
    v = dpp_max_step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad xor 1
    v = dpp_max_step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad xor 2
    v = dpp_max_step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = dpp_max_step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
    const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    const unsigned a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}

// arg-max over the wave of (dist desc, key asc); dist as non-negative float bits, "none" = (0, KEY_NONE)
__device__ __forceinline__ void wave_best(unsigned& dbits, unsigned& key) {
    const unsigned md = wave_max_u32(dbits);
    const unsigned cand = dbits == md ? (FPS_KEY_NONE - key) : 0u;
    const unsigned mk = wave_max_u32(cand);
    dbits = md;
    key = FPS_KEY_NONE - mk;
}

#define FPS_WAVES 16
#define FPS_MAXG 16
#define FPS_K 4  // candidates exchanged per round and workgroup

// 62-bit candidate code: [61:31] distance bits, [30:0] KEY_NONE - key; larger = better; 0 = none
__device__ __forceinline__ unsigned long long fps_code(unsigned dbits, unsigned key) {
    return ((unsigned long long)dbits << 31) | (unsigned long long)(FPS_KEY_NONE - key);
}
__device__ __forceinline__ unsigned long long wave_max_u62(unsigned long long c) {
    unsigned d = (unsigned)(c >> 31), k = FPS_KEY_NONE - (unsigned)(c & 0x7fffffffull);
    wave_best(d, k);
    return fps_code(d, k);
}

// Several picks per exchange.  After the distances have absorbed the picks of the previous exchange,
// the best candidate c1 is the next pick by definition; the runner-up c2 is the pick after that iff
// c1 does not lower its distance (d(c2,c1) >= tmp[c2]): nobody else can then overtake it, ties
// included, because every other distance only decreases and c2 already preceded the rest in the
// (distance desc, key asc) order.  The same argument chains to c3, c4.  FPS picks are far apart by
// construction, so ~3.5 of 4 candidates are accepted on ScanNet-like scenes and the serial chain of
// 2047 cross-CU exchanges shrinks to ~600.  Each level (lane -> wave -> workgroup -> grid) forwards
// a sorted prefix of its candidates and merging stops after consuming the LAST entry a source
// forwarded (its next one is unknown), which keeps the result exact.
template <int P>
__global__ __launch_bounds__(FPS_WAVES * 64) void k_fps(const float* __restrict__ xyz, int n, int m, int G,
                                                        int bs_log2, int batch0,
                                                        unsigned long long* __restrict__ slots,
                                                        int32_t* __restrict__ idxs, int* __restrict__ err) {
    __shared__ unsigned long long s_part[2][FPS_WAVES * 2];
    __shared__ int s_pick[2][FPS_K + 1];
    __shared__ float s_xyz[2][FPS_K * 3];
    const int bi = batch0 + blockIdx.y, wg = blockIdx.x;
    xyz += (size_t)bi * n * 3;
    idxs += (size_t)bi * m;
    slots += (size_t)bi * 2 * FPS_MAXG * FPS_K;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int gtid = wg * (FPS_WAVES * 64) + threadIdx.x;
    const int stride = G * FPS_WAVES * 64;

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
        if (k < n) {
            px[i] = xyz[(size_t)k * 3 + 0];
            py[i] = xyz[(size_t)k * 3 + 1];
            pz[i] = xyz[(size_t)k * 3 + 2];
            const float mag = fmaf(pz[i], pz[i], fmaf(py[i], py[i], px[i] * px[i]));
            if (!((double)mag <= 1e-3)) elig |= 1u << i;
            const unsigned rev = bs_log2 ? (__brev((unsigned)k & bs_mask) >> (32 - bs_log2)) : 0u;
            key[i] = (rev << 22) | (unsigned)k;
        }
    }
    if (wg == 0 && threadIdx.x == 0) idxs[0] = 0;
    // picks of the previous exchange (uniform): count + coordinates
    int nnew = 1;
    float nx[FPS_K], ny[FPS_K], nz[FPS_K];
    nx[0] = xyz[0]; ny[0] = xyz[1]; nz[0] = xyz[2];
#pragma unroll
    for (int a = 1; a < FPS_K; a++) nx[a] = ny[a] = nz[a] = 0.f;
    int done = 1;  // picks written so far
    for (int round = 1; done < m; round++) {
        const int par = round & 1;
        // 1) absorb the new picks, track this lane's best
        unsigned bd = 0u, bk = FPS_KEY_NONE;
#pragma unroll
        for (int i = 0; i < P; i++) {
            if (elig & (1u << i)) {
                float d2 = tmp[i];
#pragma unroll
                for (int a = 0; a < FPS_K; a++) {
                    if (a < nnew) {
                        const float dx = px[i] - nx[a], dy = py[i] - ny[a], dz = pz[i] - nz[a];
                        d2 = fminf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), d2);
                    }
                }
                tmp[i] = d2;
                const unsigned db = __float_as_uint(d2);
                if (db > bd || (db == bd && key[i] < bk)) {
                    bd = db;
                    bk = key[i];
                }
            }
        }
        // 2) wave top-2: best, then the owner of the best exposes its runner-up
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
            s_part[par][wid * 2 + 0] = fps_code(d1, k1);
            s_part[par][wid * 2 + 1] = fps_code(d2w, k2w);
        }
        __syncthreads();
        // 3) wave 0: merge the 32 wave candidates into the workgroup's sorted prefix (<= FPS_K entries)
        if (wid == 0) {
            unsigned long long mine = lane < FPS_WAVES * 2 ? s_part[par][lane] : 0ull;
            unsigned long long wgc[FPS_K];
            bool stop = false;
#pragma unroll
            for (int t = 0; t < FPS_K; t++) {
                unsigned long long best = stop ? 0ull : wave_max_u62(mine);
                wgc[t] = best;
                if (best != 0ull) {
                    // owner lane retires its entry; consuming a wave's second (= last forwarded) entry, or a
                    // first entry whose successor is empty, ends the prefix
                    const bool own = mine == best;
                    const unsigned long long bal = __ballot(own);
                    const int ol = __builtin_ctzll(bal);
                    const unsigned long long succ = __shfl(mine, ol | 1, 64);
                    if (own) mine = 0ull;
                    if ((ol & 1) || succ == 0ull) stop = true;
                } else {
                    stop = true;
                }
            }
            // 4) grid level: publish FPS_K granules, gather everybody's, merge the same way
            unsigned long long gl[FPS_K];
            if (G > 1) {
                unsigned long long* rs = slots + (size_t)par * FPS_MAXG * FPS_K;
                const unsigned long long tag = (unsigned long long)(round & 3) << 62;
                if (lane < FPS_K) {
                    unsigned long long v = wgc[0];
#pragma unroll
                    for (int t = 1; t < FPS_K; t++) v = lane == t ? wgc[t] : v;
                    __hip_atomic_store(rs + wg * FPS_K + lane, tag | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                unsigned long long v = 0ull;
                if (lane < G * FPS_K) {
                    int spins = 0;
                    while (true) {
                        v = __hip_atomic_load(rs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((v >> 62) == (tag >> 62)) break;
                        if (++spins > FPS_SPIN_LIMIT) {
                            *err = 1;
                            break;
                        }
                    }
                    v &= (1ull << 62) - 1ull;
                }
                bool gstop = false;
#pragma unroll
                for (int t = 0; t < FPS_K; t++) {
                    unsigned long long best = gstop ? 0ull : wave_max_u62(v);
                    gl[t] = best;
                    if (best != 0ull) {
                        const bool own = v == best;
                        const int ol = __builtin_ctzll(__ballot(own));
                        const unsigned long long succ = __shfl(v, ol + 1, 64);
                        if (own) v = 0ull;
                        if ((ol % FPS_K) == FPS_K - 1 || succ == 0ull) gstop = true;
                    } else {
                        gstop = true;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < FPS_K; t++) gl[t] = wgc[t];
            }
            // 5) validate the chain c1, c2, ... (uniform): c_i is a pick iff no accepted c_a lowers its distance
            int cidx[FPS_K];
            float cx[FPS_K], cy[FPS_K], cz[FPS_K], ct[FPS_K];
            int nacc = 0;
#pragma unroll
            for (int t = 0; t < FPS_K; t++) {
                const unsigned kk = FPS_KEY_NONE - (unsigned)(gl[t] & 0x7fffffffull);
                const bool none = gl[t] == 0ull || kk == FPS_KEY_NONE;
                cidx[t] = none ? 0 : (int)(kk & 0x3fffffu);
                ct[t] = __uint_as_float((unsigned)(gl[t] >> 31));
                const int ci = __builtin_amdgcn_readfirstlane(cidx[t]);
                cx[t] = xyz[(size_t)ci * 3 + 0];
                cy[t] = xyz[(size_t)ci * 3 + 1];
                cz[t] = xyz[(size_t)ci * 3 + 2];
                if (t == 0) {
                    nacc = 1;  // c1 is always the next pick ("none" resolves to index 0 like the reference)
                } else if (nacc == t && !none && done + t < m) {
                    // the accepted picks drop to distance 0 themselves: a candidate at distance 0 can never
                    // overtake them (they precede it in key order), e.g. in the m > n padding regime
                    bool ok = ct[t] > 0.f;
#pragma unroll
                    for (int a = 0; a < FPS_K; a++) {
                        if (a < t) {
                            const float dx = cx[t] - cx[a], dy = cy[t] - cy[a], dz = cz[t] - cz[a];
                            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                            ok = ok && !(d < ct[t]);
                        }
                    }
                    if (ok) nacc = t + 1;
                }
            }
            if (lane == 0) {
                s_pick[par][0] = nacc;
#pragma unroll
                for (int t = 0; t < FPS_K; t++) {
                    s_pick[par][1 + t] = cidx[t];
                    s_xyz[par][t * 3 + 0] = cx[t];
                    s_xyz[par][t * 3 + 1] = cy[t];
                    s_xyz[par][t * 3 + 2] = cz[t];
                }
                if (wg == 0)
                    for (int t = 0; t < nacc; t++) idxs[done + t] = cidx[t];
            }
        }
        __syncthreads();
        nnew = s_pick[par][0];
#pragma unroll
        for (int a = 0; a < FPS_K; a++) {
            nx[a] = s_xyz[par][a * 3 + 0];
            ny[a] = s_xyz[par][a * 3 + 1];
            nz[a] = s_xyz[par][a * 3 + 2];
        }
        done += nnew;
    }
}

template <int P>
static void launch_fps(int G, int nb, hipStream_t st, const float* xyz, int n, int m, int bs_log2, int batch0,
                       unsigned long long* slots, int32_t* idxs, int* err) {
    hipLaunchKernelGGL((k_fps<P>), dim3(G, nb), dim3(FPS_WAVES * 64), 0, st, xyz, n, m, G, bs_log2, batch0, slots,
                       idxs, err);
}

extern "C" size_t gf_fps_scratch_bytes(int b) {
    return ((size_t)b * 2 * FPS_MAXG * FPS_K + 8) * sizeof(unsigned long long);
}

extern "C" int gf_furthest_point_sampling(const float* xyz, int b, int n, int m, int32_t* idxs, void* scratch,
                                          void* stream) {
    GF_CHECK_ARG(b >= 0 && n >= 1 && m >= 0, "gf_furthest_point_sampling: bad sizes b=%d n=%d m=%d", b, n, m);
    GF_CHECK_ARG(n < (1 << 22), "gf_furthest_point_sampling: n=%d exceeds the 22-bit index of the tie-break key", n);
    if (b == 0 || m == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    // reference launch geometry decides ties: bs = largest power of two <= n, capped at 512
    int bs_log2 = 0;
    while ((2 << bs_log2) <= n && bs_log2 < 9) bs_log2++;
    const int per_wg = FPS_WAVES * 64;
    int G = (n + per_wg * 3 - 1) / (per_wg * 3);
    if (const char* e = getenv("GF_FPS_G")) G = atoi(e);
    if (G < 1) G = 1;
    if (G > FPS_MAXG) G = FPS_MAXG;
    const int P = (n + G * per_wg - 1) / (G * per_wg);
    GF_CHECK_ARG(P <= 16, "gf_furthest_point_sampling: n=%d too large (max %d)", n, FPS_MAXG * per_wg * 16);
    unsigned long long* slots = (unsigned long long*)scratch;
    int* err = (int*)(slots + (size_t)b * 2 * FPS_MAXG * FPS_K);
    hipMemsetAsync(scratch, 0, gf_fps_scratch_bytes(b), st);
    const int per_launch = 128 / G > 0 ? 128 / G : 1;  // keep every cooperating workgroup resident
    for (int b0 = 0; b0 < b; b0 += per_launch) {
        const int nb = (b - b0) < per_launch ? (b - b0) : per_launch;
        if (P <= 1) launch_fps<1>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else if (P <= 2) launch_fps<2>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else if (P <= 4) launch_fps<4>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else if (P <= 6) launch_fps<6>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else if (P <= 8) launch_fps<8>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else if (P <= 12) launch_fps<12>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else launch_fps<16>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
    }
    GF_CHECK_LAUNCH("gf_furthest_point_sampling");
    return GF_OK;
}
