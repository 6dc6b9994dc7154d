// Point-set operators of the set-aggregation stage and the voxel mean (gfx950).
//   voxelize_fp/bp      <- lib/pointgroup_ops/src/voxelize/voxelize.cu:9-53
//   gather_points(+grad) <- lib/pointnet2/_ext_src/src/sampling_gpu.cu:11-60
//   group_points(+grad)  <- lib/pointnet2/_ext_src/src/group_points_gpu.cu:11-78
//   ball_query           <- lib/pointnet2/_ext_src/src/ball_query_gpu.cu:12-57
//   furthest_point_sampling <- lib/pointnet2/_ext_src/src/sampling_gpu.cu:72-232
// All are HBM/latency-bound integer or copy work; none is reshaped into a GEMM.
// Built with -ffp-contract=off: the fmaf() calls below are the only fused operations and
// match oracle/gf_oracle.c term by term.
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
// G cooperating workgroups of 1024 threads keep ALL points and running distances in registers
// (P = ceil(n / (G*1024)) points per thread).  Per round each workgroup reduces its slice
// (DPP/shuffle arg-max per wave, 16 partials through LDS), publishes one 8-byte
// {distance, key|round-tag} granule with a write-through store, and polls the other G-1
// granules (MI355X_MICROARCH.md hand-off row "handoff-1to1": one naturally aligned 8-byte sc1
// store needs no fence).  The serial chain per round is therefore one cross-CU hop instead of
// a sweep of the whole point set through one CU's L2 port.
// ------------------------------------------------------------------------------------
#define FPS_THREADS 1024
#define FPS_MAXG 32
#define FPS_KEY_NONE 0x7fffffffu
#define FPS_SPIN_LIMIT (1 << 24)

// candidate order: larger distance first, then smaller key.  u62 = dist31 << 31 | (NONE - key), max wins.
__device__ __forceinline__ unsigned long long fps_pack(float d, unsigned key) {
    return ((unsigned long long)__float_as_uint(d) << 31) | (unsigned long long)(FPS_KEY_NONE - key);
}

template <int P>
__global__ __launch_bounds__(FPS_THREADS) void k_fps(const float* __restrict__ xyz, int n, int m, int G, int bs_log2,
                                                     int batch0, unsigned long long* __restrict__ slots,
                                                     int32_t* __restrict__ idxs, int* __restrict__ err) {
    __shared__ unsigned long long s_part[FPS_THREADS / 64];
    __shared__ int s_old;
    const int bi = batch0 + blockIdx.y, wg = blockIdx.x;
    xyz += (size_t)bi * n * 3;
    idxs += (size_t)bi * m;
    slots += (size_t)bi * 2 * FPS_MAXG;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gtid = wg * FPS_THREADS + tid;
    const int stride = G * FPS_THREADS;

    float px[P], py[P], pz[P], tmp[P];
    unsigned key[P];
    unsigned elig = 0;
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
            const unsigned low = (unsigned)k & ((1u << bs_log2) - 1u);
            const unsigned rev = bs_log2 ? (__brev(low) >> (32 - bs_log2)) : 0u;
            key[i] = (rev << 22) | (unsigned)k;
        }
    }
    int old = 0;
    if (wg == 0 && tid == 0) idxs[0] = 0;
    for (int j = 1; j < m; j++) {
        const int so = __builtin_amdgcn_readfirstlane(old);
        const float x1 = xyz[(size_t)so * 3 + 0], y1 = xyz[(size_t)so * 3 + 1], z1 = xyz[(size_t)so * 3 + 2];
        // "no eligible point" (reference: best = -1, besti = 0) is the smallest possible candidate
        unsigned long long best = fps_pack(0.f, FPS_KEY_NONE);
#pragma unroll
        for (int i = 0; i < P; i++) {
            if (elig & (1u << i)) {
                const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                const float d2 = fminf(d, tmp[i]);
                tmp[i] = d2;
                const unsigned long long c = fps_pack(d2, key[i]);
                best = c > best ? c : best;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        if (lane == 0) s_part[wid] = best;
        __syncthreads();
        if (wid == 0) {
            unsigned long long c = lane < FPS_THREADS / 64 ? s_part[lane] : 0ull;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                const unsigned long long o = __shfl_xor(c, off, 64);
                c = o > c ? o : c;
            }
            c = __shfl(c, 0, 64);
            if (G > 1) {
                // one 8-byte write-through granule per workgroup and round: [63:62] round tag, [61:0] candidate
                unsigned long long* rs = slots + (size_t)(j & 1) * FPS_MAXG;
                const unsigned long long tag = (unsigned long long)(j & 3) << 62;
                if (lane == 0) __hip_atomic_store(rs + wg, tag | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long v = 0ull;
                if (lane < G) {
                    int spins = 0;
                    while (true) {
                        v = __hip_atomic_load(rs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((v >> 62) == (unsigned long long)(j & 3)) break;
                        if (++spins > FPS_SPIN_LIMIT) {
                            *err = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    v &= (1ull << 62) - 1ull;
                }
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) {
                    const unsigned long long o = __shfl_xor(v, off, 64);
                    v = o > v ? o : v;
                }
                c = __shfl(v, 0, 64);
            }
            if (lane == 0) {
                const unsigned kk = FPS_KEY_NONE - (unsigned)(c & 0x7fffffffull);
                const int pick = kk == FPS_KEY_NONE ? 0 : (int)(kk & 0x3fffffu);
                s_old = pick;
                if (wg == 0) idxs[j] = pick;
            }
        }
        __syncthreads();
        old = s_old;
    }
}

template <int P>
static void launch_fps(int G, int nb, hipStream_t st, const float* xyz, int n, int m, int bs_log2, int batch0,
                       unsigned long long* slots, int32_t* idxs, int* err) {
    hipLaunchKernelGGL((k_fps<P>), dim3(G, nb), dim3(FPS_THREADS), 0, st, xyz, n, m, G, bs_log2, batch0, slots, idxs,
                       err);
}

extern "C" size_t gf_fps_scratch_bytes(int b) { return ((size_t)b * 2 * FPS_MAXG + 8) * sizeof(unsigned long long); }

extern "C" int gf_furthest_point_sampling(const float* xyz, int b, int n, int m, int32_t* idxs, void* scratch,
                                          void* stream) {
    GF_CHECK_ARG(b >= 0 && n >= 1 && m >= 0, "gf_furthest_point_sampling: bad sizes b=%d n=%d m=%d", b, n, m);
    GF_CHECK_ARG(n < (1 << 22), "gf_furthest_point_sampling: n=%d exceeds the 22-bit index of the tie-break key", n);
    if (b == 0 || m == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    // reference launch geometry decides ties: bs = largest power of two <= n, capped at 512
    int bs_log2 = 0;
    while ((2 << bs_log2) <= n && bs_log2 < 9) bs_log2++;
    int G = (n + FPS_THREADS * 4 - 1) / (FPS_THREADS * 4);
    if (G < 1) G = 1;
    if (G > FPS_MAXG) G = FPS_MAXG;
    int P = (n + G * FPS_THREADS - 1) / (G * FPS_THREADS);
    GF_CHECK_ARG(P <= 16, "gf_furthest_point_sampling: n=%d too large (max %d)", n, FPS_MAXG * FPS_THREADS * 16);
    unsigned long long* slots = (unsigned long long*)scratch;
    int* err = (int*)(slots + (size_t)b * 2 * FPS_MAXG);
    hipMemsetAsync(scratch, 0, gf_fps_scratch_bytes(b), st);
    const int per_launch = 256 / G > 0 ? 256 / G : 1;  // keep every cooperating workgroup resident
    for (int b0 = 0; b0 < b; b0 += per_launch) {
        const int nb = (b - b0) < per_launch ? (b - b0) : per_launch;
        if (P <= 1) launch_fps<1>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else if (P <= 2) launch_fps<2>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else if (P <= 4) launch_fps<4>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else if (P <= 8) launch_fps<8>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
        else launch_fps<16>(G, nb, st, xyz, n, m, bs_log2, b0, slots, idxs, err);
    }
    GF_CHECK_LAUNCH("gf_furthest_point_sampling");
    return GF_OK;
}
