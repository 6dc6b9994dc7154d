// Training-mode BatchNorm1d (+ ReLU) over voxel rows x[M, C], forward and backward (gfx950).
//
// The reference's sparse U-Net is pre-activation: every sparse convolution is fed by BatchNorm1d(eps 1e-4, momentum
// 0.1) + ReLU over the [M_L, C] rows of its level (geoformer_modules.py:10-35,52-129; geoformer.py:39) -- 67 such
// pairs per training forward.  Through the framework each pair is 4 launches forward (statistics, normalise, counter,
// ReLU) and 3-4 backward, and its two reduction kernels take 33 us each on a 523k x 16 batch (4 TB/s would be 8 us).
// Here a pair is THREE launches per direction:
//   forward   k_bn_stats      per-channel sum / sum of squares of (x - pivot) over row slabs -> per-workgroup partials
//             k_bn_finalize   one workgroup: the partials in a fixed order -> mean / invstd, running statistics
//             k_bn_apply      y = max(0, (x - mean) * invstd * gamma + beta)
//   backward  k_bn_bwd_reduce g = dy * [y > 0]; per-channel sum g, sum g * xhat -> partials
//             k_bn_finalize   dbeta, dgamma and the two means the input gradient needs
//             k_bn_bwd_apply  dx = (g - mean(g) - xhat * mean(g * xhat)) * gamma * invstd
// (A two-launch form in which the LAST workgroup to arrive reduced the partials was built first: every workgroup then
// needs a device-scope fence before its arrival -- an L2 write-back / invalidate each on this chip -- and the
// statistics pass ran at 0.6 TB/s, 41-56 us for 523k x 16; the kernel boundary orders the partials for free.)
// Rows are read as float4 (C is a multiple of 4: the U-Net's widths are multiples of 16); a thread owns one float4
// column and walks rows; partial sums are combined in a fixed order (deterministic, unlike atomics).  The variance uses
// a per-channel PIVOT (the tensor's first row) instead of raw second moments: sum (x - p)^2 - (sum (x - p))^2 / n has no
// cancellation problem when |mean| >> std (the pivot is a sample of the same distribution).
// HBM-bound: forward 3 passes over [M, C] (read, read + write), backward 3 reads + (3 reads + 1 write).
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_C = 256;
constexpr int BN_MAX_WG = 512;

struct BnGeom {
    int c4;    // float4 columns per row
    int rpi;   // rows per workgroup iteration
    int nwg;   // workgroups
};

inline BnGeom bn_geom(int M, int C) {
    BnGeom g;
    g.c4 = C / 4;
    g.rpi = BN_THREADS / g.c4;
    long long iters = ((long long)M + g.rpi - 1) / g.rpi;
    // a workgroup should stream at least ~8 iterations; at most BN_MAX_WG partial rows for the last block to reduce
    long long nwg = (iters + 7) / 8;  // (the loops keep four / two rows in flight per trip)
    if (nwg > BN_MAX_WG) nwg = BN_MAX_WG;
    if (nwg < 1) nwg = 1;
    g.nwg = (int)nwg;
    return g;
}

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// combine the per-thread float4 pairs (a, b) of the threads that share a column (same t % c4) through LDS; the first
// `c4` threads end up with the column totals
__device__ __forceinline__ void column_reduce(float4& a, float4& b, int c4, int rpi, float4* s_a, float4* s_b) {
    const int t = threadIdx.x;
    s_a[t] = a;
    s_b[t] = b;
    __syncthreads();
    if (t < c4) {
        float4 ra = s_a[t], rb = s_b[t];
        for (int r = 1; r < rpi; r++) {  // fixed order
            ra = f4_add(ra, s_a[r * c4 + t]);
            rb = f4_add(rb, s_b[r * c4 + t]);
        }
        a = ra;
        b = rb;
    }
}

// The last workgroup's reduction of the per-workgroup partials [nwg][2C]: all 256 threads take part (a thread per
// channel walking 1000 partials alone took longer than the whole statistics pass), columns are tiled over the threads,
// the rows of a column are split over `nrg` row groups and combined in a fixed order (deterministic), double
// accumulators.  Result: s_out[0..2C) in LDS.
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partials, int nwg, int C, double* s_tmp,
                                                double* s_out) {
    const int t = threadIdx.x, ncol = 2 * C;
    for (int col0 = 0; col0 < ncol; col0 += BN_THREADS) {
        const int tile = min(BN_THREADS, ncol - col0);
        const int nrg = BN_THREADS / tile;
        const int j = t % tile, rg = t / tile;
        double acc = 0.0;
        if (rg < nrg) {
            // eight partials in flight per trip (one at a time this kernel took 22 us for 1024 partials, longer than the
            // statistics pass it follows); the order of the additions stays fixed
            int w = rg;
            for (; w + 7 * nrg < nwg; w += 8 * nrg) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = partials[(size_t)(w + u * nrg) * ncol + col0 + j];
#pragma unroll
                for (int u = 0; u < 8; u++) acc += (double)v[u];
            }
            for (; w < nwg; w += nrg) acc += (double)partials[(size_t)w * ncol + col0 + j];
        }
        s_tmp[t] = acc;
        __syncthreads();
        if (t < tile) {
            double a = 0.0;
            for (int r = 0; r < nrg; r++) a += s_tmp[r * tile + t];
            s_out[col0 + t] = a;
        }
        __syncthreads();
    }
}

// One workgroup: the partials [P][2C] in a fixed order, then one thread per channel.
//   forward (means == nullptr): sums are sum(x - pivot), sum((x - pivot)^2) over n elements per channel; pivot of channel c
//   = x[c * pivot_stride]; writes save_mean / save_invstd and updates the running statistics.
//   backward: sums are sum(g), sum(g * xhat); writes dbeta / dgamma and means[2][C] = the sums / n.
__global__ __launch_bounds__(BN_THREADS) void k_bn_finalize(const float* __restrict__ partials, int P, int C, double n,
                                                           const float* __restrict__ x, long long pivot_stride, float eps,
                                                           float momentum, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ means) {
    __shared__ double s_tmp[BN_THREADS], s_out[2 * BN_MAX_C];
    reduce_partials(partials, P, C, s_tmp, s_out);
    for (int c = threadIdx.x; c < C; c += BN_THREADS) {
        const double s = s_out[c], q = s_out[C + c];
        if (means) {
            if (dbeta) dbeta[c] = (float)s;
            if (dgamma) dgamma[c] = (float)q;
            means[c] = (float)(s / n);
            means[C + c] = (float)(q / n);
            continue;
        }
        const double dm = s / n;  // mean - pivot
        double var = q / n - dm * dm;
        if (var < 0.0) var = 0.0;
        const float mean = (float)((double)x[(size_t)c * pivot_stride] + dm);
        save_mean[c] = mean;
        save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// ---- forward statistics -----------------------------------------------------------------------------------------------
// partials layout: [nwg][2][C] floats.
__global__ __launch_bounds__(BN_THREADS) void k_bn_stats(const float* __restrict__ x, int M, int C, int c4, int rpi,
                                                        float* __restrict__ partials) {
    __shared__ float4 s_a[BN_THREADS], s_b[BN_THREADS];
    const int t = threadIdx.x;
    const int col = t % c4, r0 = t / c4;
    const bool active = r0 < rpi;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), sq = sum;
    const float4 piv = x4[col];  // row 0
    if (active) {
        // four rows per trip, all four loads issued before the first use (one load per trip is a chain of dependent
        // memory round trips: 41 us for 523k x 16 against ~10 us with four in flight)
        const long long stride = (long long)gridDim.x * rpi;
        long long row = (long long)blockIdx.x * rpi + r0;
        auto acc = [&](const float4 v) {
            const float dx = v.x - piv.x, dy = v.y - piv.y, dz = v.z - piv.z, dw = v.w - piv.w;
            sum.x += dx; sum.y += dy; sum.z += dz; sum.w += dw;
            sq.x = fmaf(dx, dx, sq.x); sq.y = fmaf(dy, dy, sq.y); sq.z = fmaf(dz, dz, sq.z); sq.w = fmaf(dw, dw, sq.w);
        };
        for (; row + 3 * stride < M; row += 4 * stride) {
            const float4 v0 = x4[row * c4 + col], v1 = x4[(row + stride) * c4 + col];
            const float4 v2 = x4[(row + 2 * stride) * c4 + col], v3 = x4[(row + 3 * stride) * c4 + col];
            acc(v0); acc(v1); acc(v2); acc(v3);
        }
        for (; row < M; row += stride) acc(x4[row * c4 + col]);
    }
    column_reduce(sum, sq, c4, rpi, s_a, s_b);
    float* mine = partials + (size_t)blockIdx.x * 2 * C;
    if (t < c4) {
        reinterpret_cast<float4*>(mine)[t] = sum;
        reinterpret_cast<float4*>(mine + C)[t] = sq;
    }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_apply(const float* __restrict__ x, long long total4, int c4,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ save_mean,
                                                        const float* __restrict__ save_invstd, int relu,
                                                        float* __restrict__ y) {
    // a thread keeps its column for the whole grid-stride walk when the stride is a multiple of c4
    const long long stride = (long long)gridDim.x * BN_THREADS / c4 * c4;
    long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x;
    if (i >= stride) return;
    const int col = (int)(i % c4);
    const float4 g = reinterpret_cast<const float4*>(gamma)[col], b = reinterpret_cast<const float4*>(beta)[col];
    const float4 m = reinterpret_cast<const float4*>(save_mean)[col], is = reinterpret_cast<const float4*>(save_invstd)[col];
    const float4 sc = make_float4(g.x * is.x, g.y * is.y, g.z * is.z, g.w * is.w);
    const float4 sh = make_float4(b.x - m.x * sc.x, b.y - m.y * sc.y, b.z - m.z * sc.z, b.w - m.w * sc.w);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (; i < total4; i += stride) {
        const float4 v = x4[i];
        float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
        if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        y4[i] = o;
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------------
// means layout written by the last block: [2][C] = mean(g), mean(g * xhat)
__global__ __launch_bounds__(BN_THREADS) void k_bn_bwd_reduce(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy, int M, int C, int c4, int rpi,
                                                             const float* __restrict__ save_mean,
                                                             const float* __restrict__ save_invstd, int relu,
                                                             float* __restrict__ partials) {
    __shared__ float4 s_a[BN_THREADS], s_b[BN_THREADS];
    const int t = threadIdx.x;
    const int col = t % c4, r0 = t / c4;
    const bool active = r0 < rpi;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const float4* d4 = reinterpret_cast<const float4*>(dy);
    const float4 m = reinterpret_cast<const float4*>(save_mean)[col], is = reinterpret_cast<const float4*>(save_invstd)[col];
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgx = sg;
    if (active) {
        const long long stride = (long long)gridDim.x * rpi;
        long long row = (long long)blockIdx.x * rpi + r0;
        auto acc = [&](float4 g, const float4 v, const float4 o) {
            if (relu) {
                g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
            }
            sg = f4_add(sg, g);
            sgx.x = fmaf(g.x, (v.x - m.x) * is.x, sgx.x); sgx.y = fmaf(g.y, (v.y - m.y) * is.y, sgx.y);
            sgx.z = fmaf(g.z, (v.z - m.z) * is.z, sgx.z); sgx.w = fmaf(g.w, (v.w - m.w) * is.w, sgx.w);
        };
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (; row + stride < M; row += 2 * stride) {  // two rows (six loads) in flight per trip
            const long long i0 = row * c4 + col, i1 = (row + stride) * c4 + col;
            const float4 g0 = d4[i0], v0 = x4[i0], g1 = d4[i1], v1 = x4[i1];
            const float4 o0 = relu ? y4[i0] : zero, o1 = relu ? y4[i1] : zero;
            acc(g0, v0, o0); acc(g1, v1, o1);
        }
        for (; row < M; row += stride) {
            const long long i = row * c4 + col;
            acc(d4[i], x4[i], relu ? y4[i] : zero);
        }
    }
    column_reduce(sg, sgx, c4, rpi, s_a, s_b);
    float* mine = partials + (size_t)blockIdx.x * 2 * C;
    if (t < c4) {
        reinterpret_cast<float4*>(mine)[t] = sg;
        reinterpret_cast<float4*>(mine + C)[t] = sgx;
    }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ dy, long long total4, int C, int c4,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_invstd,
                                                            const float* __restrict__ means, int relu,
                                                            const float* addend, float* dx) {
    // (addend may be dx itself: the element is read and written by the same thread)
    const long long stride = (long long)gridDim.x * BN_THREADS / c4 * c4;
    long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x;
    if (i >= stride) return;
    const int col = (int)(i % c4);
    const float4 ga = reinterpret_cast<const float4*>(gamma)[col];
    const float4 m = reinterpret_cast<const float4*>(save_mean)[col], is = reinterpret_cast<const float4*>(save_invstd)[col];
    const float4 c1 = reinterpret_cast<const float4*>(means)[col], c2 = reinterpret_cast<const float4*>(means + C)[col];
    const float4 k = make_float4(ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const float4* d4 = reinterpret_cast<const float4*>(dy);
    float4* o4 = reinterpret_cast<float4*>(dx);
    for (; i < total4; i += stride) {
        float4 g = d4[i];
        const float4 v = x4[i];
        if (relu) {
            const float4 o = y4[i];
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        float4 r;
        r.x = (g.x - c1.x - (v.x - m.x) * is.x * c2.x) * k.x;
        r.y = (g.y - c1.y - (v.y - m.y) * is.y * c2.y) * k.y;
        r.z = (g.z - c1.z - (v.z - m.z) * is.z * c2.z) * k.z;
        r.w = (g.w - c1.w - (v.w - m.w) * is.w * c2.w) * k.w;
        if (addend) r = f4_add(r, reinterpret_cast<const float4*>(addend)[i]);
        o4[i] = r;
    }
}

// =====================================================================================================================
// Channel-major form: x[B, C, L] (nn.BatchNorm1d over [B, C, L], nn.BatchNorm2d over [B, C, H, W] with L = H * W) -- the
// semantic head and mask tower over all points ([1, 16, N]), the set-abstraction MLP ([B, 32, 2048, 64]).  The
// framework route was var_mean + five element-wise passes forward and a dozen kernels backward per layer.  Same
// launch structure as above; a workgroup owns one chunk of one (batch, channel) row, loads are plain dwords (L is a
// point count: rows are not 16-byte aligned), eight in flight per thread.
// partials layout: [P = B * nchunk][2][C] so that reduce_partials() above serves both forms.
// =====================================================================================================================
struct BnClGeom {
    int nchunk;      // chunks per row
    long long chunk; // elements per chunk
};
inline BnClGeom bncl_geom(int R, long long L) {
    BnClGeom g;
    long long nchunk = 1024 / (R > 0 ? R : 1);
    const long long max_by_len = (L + 2047) / 2048;  // at least ~2048 elements per workgroup
    if (nchunk > max_by_len) nchunk = max_by_len;
    if (nchunk > 256) nchunk = 256;
    if (nchunk < 1) nchunk = 1;
    g.nchunk = (int)nchunk;
    g.chunk = (L + nchunk - 1) / nchunk;
    return g;
}

__device__ __forceinline__ float2 block_sum2(float a, float b, float2* s_w) {
    a = gf_wave_sum(a);  // (the __shfl_xor butterfly without the LDS crossbar: common.h)
    b = gf_wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = make_float2(a, b);
    __syncthreads();
    float2 r = s_w[0];
    for (int i = 1; i < BN_THREADS / 64; i++) {
        r.x += s_w[i].x;
        r.y += s_w[i].y;
    }
    return r;
}

__global__ __launch_bounds__(BN_THREADS) void k_bncl_stats(const float* __restrict__ x, int B, int C, long long L,
                                                          long long chunk, float* __restrict__ partials) {
    __shared__ float2 s_w[BN_THREADS / 64];
    const int r = blockIdx.y, c = r % C, b = r / C, k = blockIdx.x, t = threadIdx.x;
    const float piv = x[(size_t)c * L];  // channel c, batch 0, element 0
    const float* row = x + (size_t)r * L;
    const long long lo = (long long)k * chunk, hi = min(L, lo + chunk);
    float s = 0.f, q = 0.f;
    long long i = lo + t;
    for (; i + 7 * BN_THREADS < hi; i += 8 * BN_THREADS) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = row[i + j * BN_THREADS];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float d = v[j] - piv;
            s += d;
            q = fmaf(d, d, q);
        }
    }
    for (; i < hi; i += BN_THREADS) {
        const float d = row[i] - piv;
        s += d;
        q = fmaf(d, d, q);
    }
    const float2 tot = block_sum2(s, q, s_w);
    const int p = b * (int)gridDim.x + k;
    if (t == 0) {
        partials[(size_t)p * 2 * C + c] = tot.x;
        partials[(size_t)p * 2 * C + C + c] = tot.y;
    }
}

__global__ __launch_bounds__(BN_THREADS) void k_bncl_apply(const float* __restrict__ x, int C, long long L, long long chunk,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, int relu,
                                                          float* __restrict__ y) {
    const int r = blockIdx.y, c = r % C;
    const float sc = gamma[c] * save_invstd[c], sh = beta[c] - save_mean[c] * sc;
    const float* row = x + (size_t)r * L;
    float* orow = y + (size_t)r * L;
    const long long lo = (long long)blockIdx.x * chunk, hi = min(L, lo + chunk);
    long long i = lo + threadIdx.x;
    for (; i + 7 * BN_THREADS < hi; i += 8 * BN_THREADS) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = row[i + j * BN_THREADS];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float o = fmaf(v[j], sc, sh);
            orow[i + j * BN_THREADS] = relu ? fmaxf(o, 0.f) : o;
        }
    }
    for (; i < hi; i += BN_THREADS) {
        const float o = fmaf(row[i], sc, sh);
        orow[i] = relu ? fmaxf(o, 0.f) : o;
    }
}

__global__ __launch_bounds__(BN_THREADS) void k_bncl_bwd_reduce(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ dy, int B, int C, long long L,
                                                               long long chunk, const float* __restrict__ save_mean,
                                                               const float* __restrict__ save_invstd, int relu,
                                                               float* __restrict__ partials) {
    __shared__ float2 s_w[BN_THREADS / 64];
    const int r = blockIdx.y, c = r % C, b = r / C, k = blockIdx.x, t = threadIdx.x;
    const float m = save_mean[c], is = save_invstd[c];
    const size_t base = (size_t)r * L;
    const long long lo = (long long)k * chunk, hi = min(L, lo + chunk);
    float sg = 0.f, sgx = 0.f;
    long long i = lo + t;
    for (; i + 3 * BN_THREADS < hi; i += 4 * BN_THREADS) {
        float g[4], v[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            g[j] = dy[base + i + j * BN_THREADS];
            v[j] = x[base + i + j * BN_THREADS];
            o[j] = relu ? y[base + i + j * BN_THREADS] : 1.f;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float gg = o[j] > 0.f ? g[j] : 0.f;
            sg += gg;
            sgx = fmaf(gg, (v[j] - m) * is, sgx);
        }
    }
    for (; i < hi; i += BN_THREADS) {
        const float o = relu ? y[base + i] : 1.f;
        const float gg = o > 0.f ? dy[base + i] : 0.f;
        sg += gg;
        sgx = fmaf(gg, (x[base + i] - m) * is, sgx);
    }
    const float2 tot = block_sum2(sg, sgx, s_w);
    const int p = b * (int)gridDim.x + k;
    if (t == 0) {
        partials[(size_t)p * 2 * C + c] = tot.x;
        partials[(size_t)p * 2 * C + C + c] = tot.y;
    }
}

__global__ __launch_bounds__(BN_THREADS) void k_bncl_bwd_apply(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ dy, int C, long long L,
                                                              long long chunk, const float* __restrict__ gamma,
                                                              const float* __restrict__ save_mean,
                                                              const float* __restrict__ save_invstd,
                                                              const float* __restrict__ means, int relu,
                                                              float* __restrict__ dx) {
    const int r = blockIdx.y, c = r % C;
    const float m = save_mean[c], is = save_invstd[c], kk = gamma[c] * is, c1 = means[c], c2 = means[C + c];
    const size_t base = (size_t)r * L;
    const long long lo = (long long)blockIdx.x * chunk, hi = min(L, lo + chunk);
    long long i = lo + threadIdx.x;
    for (; i + 3 * BN_THREADS < hi; i += 4 * BN_THREADS) {
        float g[4], v[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            g[j] = dy[base + i + j * BN_THREADS];
            v[j] = x[base + i + j * BN_THREADS];
            o[j] = relu ? y[base + i + j * BN_THREADS] : 1.f;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float gg = o[j] > 0.f ? g[j] : 0.f;
            dx[base + i + j * BN_THREADS] = (gg - c1 - (v[j] - m) * is * c2) * kk;
        }
    }
    for (; i < hi; i += BN_THREADS) {
        const float o = relu ? y[base + i] : 1.f;
        const float gg = o > 0.f ? dy[base + i] : 0.f;
        dx[base + i] = (gg - c1 - (x[base + i] - m) * is * c2) * kk;
    }
}

int apply_grid(long long total4) {
    long long wg = (total4 + (long long)BN_THREADS * 4 - 1) / ((long long)BN_THREADS * 4);  // ~4 float4 per thread
    if (wg > 2048) wg = 2048;
    if (wg < 1) wg = 1;
    return (int)wg;
}

}  // namespace

extern "C" size_t gf_bn_train_scratch_floats(int M, int C) {
    if (M <= 0 || C <= 0 || (C % 4) != 0) return 0;
    return (size_t)bn_geom(M, C).nwg * 2 * C + 2 * (size_t)C;
}

extern "C" int gf_bn_relu_train_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps,
                                    float momentum, int relu, float* running_mean, float* running_var, float* y,
                                    float* save_mean, float* save_invstd, float* scratch, void* stream) {
    GF_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && scratch, "gf_bn_relu_train_fwd: null argument");
    GF_CHECK_ARG(M >= 2, "gf_bn_relu_train_fwd: batch statistics need at least two rows (M=%d)", M);
    GF_CHECK_ARG(C >= 4 && C <= BN_MAX_C && (C % 4) == 0, "gf_bn_relu_train_fwd: C=%d (multiples of 4 up to %d)", C, BN_MAX_C);
    GF_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "gf_bn_relu_train_fwd: running_mean / running_var come together");
    GF_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)save_mean) |
                   ((uintptr_t)save_invstd) | ((uintptr_t)scratch)) % 16) == 0, "gf_bn_relu_train_fwd: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const BnGeom g = bn_geom(M, C);
    hipLaunchKernelGGL(k_bn_stats, dim3(g.nwg), dim3(BN_THREADS), 0, st, x, M, C, g.c4, g.rpi, scratch);
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(BN_THREADS), 0, st, scratch, g.nwg, C, (double)M, x, 1LL, eps, momentum,
                       running_mean, running_var, save_mean, save_invstd, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    const long long total4 = (long long)M * g.c4;
    hipLaunchKernelGGL(k_bn_apply, dim3(apply_grid(total4)), dim3(BN_THREADS), 0, st, x, total4, g.c4, gamma, beta, save_mean,
                       save_invstd, relu, y);
    GF_CHECK_LAUNCH("gf_bn_relu_train_fwd");
    return GF_OK;
}

extern "C" int gf_bn_relu_train_bwd(const float* x, const float* y, const float* dy, int M, int C, const float* gamma,
                                    const float* save_mean, const float* save_invstd, int relu, float* dx, float* dgamma,
                                    float* dbeta, float* scratch, void* stream) {
    return gf_bn_relu_train_bwd_add(x, y, dy, M, C, gamma, save_mean, save_invstd, relu, nullptr, dx, dgamma, dbeta, scratch,
                                    stream);
}

extern "C" int gf_bn_relu_train_bwd_add(const float* x, const float* y, const float* dy, int M, int C, const float* gamma,
                                        const float* save_mean, const float* save_invstd, int relu, const float* addend,
                                        float* dx, float* dgamma, float* dbeta, float* scratch, void* stream) {
    GF_CHECK_ARG(x && dy && gamma && save_mean && save_invstd && scratch && (y || !relu), "gf_bn_relu_train_bwd: null argument");
    GF_CHECK_ARG(M >= 2, "gf_bn_relu_train_bwd: M=%d", M);
    GF_CHECK_ARG(C >= 4 && C <= BN_MAX_C && (C % 4) == 0, "gf_bn_relu_train_bwd: C=%d (multiples of 4 up to %d)", C, BN_MAX_C);
    GF_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)gamma) |
                   ((uintptr_t)save_mean) | ((uintptr_t)save_invstd) | ((uintptr_t)scratch) | ((uintptr_t)addend)) % 16) == 0,
                 "gf_bn_relu_train_bwd: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const BnGeom g = bn_geom(M, C);
    float* means = scratch + (size_t)g.nwg * 2 * C;
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3(g.nwg), dim3(BN_THREADS), 0, st, x, y, dy, M, C, g.c4, g.rpi, save_mean, save_invstd,
                       relu, scratch);
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(BN_THREADS), 0, st, scratch, g.nwg, C, (double)M, (const float*)nullptr, 0LL,
                       0.f, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, dgamma, dbeta, means);
    if (dx) {
        const long long total4 = (long long)M * g.c4;
        hipLaunchKernelGGL(k_bn_bwd_apply, dim3(apply_grid(total4)), dim3(BN_THREADS), 0, st, x, y, dy, total4, C, g.c4, gamma,
                           save_mean, save_invstd, means, relu, addend, dx);
    }
    GF_CHECK_LAUNCH("gf_bn_relu_train_bwd");
    return GF_OK;
}

// ---- channel-major entry points ------------------------------------------------------------------------------------------
extern "C" size_t gf_bn_train_cl_scratch_floats(int B, int C, long long L) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    return (size_t)B * bncl_geom(B * C, L).nchunk * 2 * C + 2 * (size_t)C;
}

extern "C" int gf_bn_relu_train_cl_fwd(const float* x, int B, int C, long long L, const float* gamma, const float* beta,
                                       float eps, float momentum, int relu, float* running_mean, float* running_var,
                                       float* y, float* save_mean, float* save_invstd, float* scratch, void* stream) {
    GF_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && scratch, "gf_bn_relu_train_cl_fwd: null argument");
    GF_CHECK_ARG(B >= 1 && C >= 1 && C <= BN_MAX_C && L >= 1 && (long long)B * L >= 2 && (long long)B * C < 65536,
                 "gf_bn_relu_train_cl_fwd: B=%d C=%d L=%lld", B, C, L);
    GF_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "gf_bn_relu_train_cl_fwd: running_mean / running_var come together");
    hipStream_t st = (hipStream_t)stream;
    const BnClGeom g = bncl_geom(B * C, L);
    dim3 grid(g.nchunk, B * C);
    hipLaunchKernelGGL(k_bncl_stats, grid, dim3(BN_THREADS), 0, st, x, B, C, L, g.chunk, scratch);
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(BN_THREADS), 0, st, scratch, B * g.nchunk, C, (double)B * (double)L, x, L, eps,
                       momentum, running_mean, running_var, save_mean, save_invstd, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr);
    hipLaunchKernelGGL(k_bncl_apply, grid, dim3(BN_THREADS), 0, st, x, C, L, g.chunk, gamma, beta, save_mean, save_invstd, relu, y);
    GF_CHECK_LAUNCH("gf_bn_relu_train_cl_fwd");
    return GF_OK;
}

extern "C" int gf_bn_relu_train_cl_bwd(const float* x, const float* y, const float* dy, int B, int C, long long L,
                                       const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                                       float* dx, float* dgamma, float* dbeta, float* scratch, void* stream) {
    GF_CHECK_ARG(x && dy && gamma && save_mean && save_invstd && scratch && (y || !relu), "gf_bn_relu_train_cl_bwd: null argument");
    GF_CHECK_ARG(B >= 1 && C >= 1 && C <= BN_MAX_C && L >= 1 && (long long)B * C < 65536, "gf_bn_relu_train_cl_bwd: B=%d C=%d L=%lld", B, C, L);
    hipStream_t st = (hipStream_t)stream;
    const BnClGeom g = bncl_geom(B * C, L);
    dim3 grid(g.nchunk, B * C);
    float* means = scratch + (size_t)B * g.nchunk * 2 * C;
    hipLaunchKernelGGL(k_bncl_bwd_reduce, grid, dim3(BN_THREADS), 0, st, x, y, dy, B, C, L, g.chunk, save_mean, save_invstd, relu,
                       scratch);
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(BN_THREADS), 0, st, scratch, B * g.nchunk, C, (double)B * (double)L,
                       (const float*)nullptr, 0LL, 0.f, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       dgamma, dbeta, means);
    if (dx)
        hipLaunchKernelGGL(k_bncl_bwd_apply, grid, dim3(BN_THREADS), 0, st, x, y, dy, C, L, g.chunk, gamma, save_mean, save_invstd,
                           means, relu, dx);
    GF_CHECK_LAUNCH("gf_bn_relu_train_cl_bwd");
    return GF_OK;
}
