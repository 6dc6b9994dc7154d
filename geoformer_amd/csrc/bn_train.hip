// Training-mode BatchNorm1d (+ ReLU) over voxel rows x[M, C], forward and backward (gfx950).
//
// The reference's sparse U-Net is pre-activation: every sparse convolution is fed by BatchNorm1d(eps 1e-4, momentum
// 0.1) + ReLU over the [M_L, C] rows of its level (geoformer_modules.py:10-35,52-129; geoformer.py:39) -- 67 such
// pairs per training forward.  Through the framework each pair is 4 launches forward (statistics, normalise, counter,
// ReLU) and 3-4 backward, and its two reduction kernels take 33 us each on a 523k x 16 batch (4 TB/s would be 8 us).
// Here a pair is TWO launches per direction:
//   forward   k_bn_stats   per-channel sum / sum of squares of (x - pivot) over row slabs -> per-workgroup partials; the
//                          LAST workgroup to finish (arrival counter) reduces the partials in a fixed order, writes
//                          mean / invstd and updates the running statistics
//             k_bn_apply   y = max(0, (x - mean) * invstd * gamma + beta)
//   backward  k_bn_bwd_reduce   g = dy * [y > 0]; per-channel sum g, sum g * xhat -> partials -> last workgroup:
//                               dbeta, dgamma and the two means the input gradient needs
//             k_bn_bwd_apply    dx = (g - mean(g) - xhat * mean(g * xhat)) * gamma * invstd
// Rows are read as float4 (C is a multiple of 4: the U-Net's widths are multiples of 16); a thread owns one float4
// column and walks rows; partial sums are combined in a fixed order (deterministic, unlike atomics).  The variance uses
// a per-channel PIVOT (the tensor's first row) instead of raw second moments: sum (x - p)^2 - (sum (x - p))^2 / n has no
// cancellation problem when |mean| >> std (the pivot is a sample of the same distribution).
// HBM-bound: forward 3 passes over [M, C] (read, read + write), backward 3 reads + (3 reads + 1 write).
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_C = 256;
constexpr int BN_MAX_WG = 1024;

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
    long long nwg = (iters + 7) / 8;
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

// ---- forward statistics -----------------------------------------------------------------------------------------------
// partials layout: [nwg][2][C] floats.  counter: one int, zero on entry, zero again on exit (reset by the last block).
__global__ __launch_bounds__(BN_THREADS) void k_bn_stats(const float* __restrict__ x, int M, int C, int c4, int rpi,
                                                        float eps, float momentum, float* __restrict__ partials,
                                                        int* __restrict__ counter, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, float* __restrict__ save_mean,
                                                        float* __restrict__ save_invstd) {
    __shared__ float4 s_a[BN_THREADS], s_b[BN_THREADS];
    __shared__ int s_last;
    const int t = threadIdx.x;
    const int col = t % c4, r0 = t / c4;
    const bool active = r0 < rpi;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), sq = sum;
    const float4 piv = x4[col];  // row 0
    if (active) {
        for (long long row = (long long)blockIdx.x * rpi + r0; row < M; row += (long long)gridDim.x * rpi) {
            const float4 v = x4[row * c4 + col];
            const float dx = v.x - piv.x, dy = v.y - piv.y, dz = v.z - piv.z, dw = v.w - piv.w;
            sum.x += dx; sum.y += dy; sum.z += dz; sum.w += dw;
            sq.x = fmaf(dx, dx, sq.x); sq.y = fmaf(dy, dy, sq.y); sq.z = fmaf(dz, dz, sq.z); sq.w = fmaf(dw, dw, sq.w);
        }
    }
    column_reduce(sum, sq, c4, rpi, s_a, s_b);
    float* mine = partials + (size_t)blockIdx.x * 2 * C;
    if (t < c4) {
        reinterpret_cast<float4*>(mine)[t] = sum;
        reinterpret_cast<float4*>(mine + C)[t] = sq;
    }
    __threadfence();
    __syncthreads();
    if (t == 0) s_last = (atomicAdd(counter, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // last workgroup: channel c is reduced by thread c over the workgroups in index order (double accumulators: up to
    // 1024 partials of up to ~10^3 rows each)
    for (int c = t; c < C; c += BN_THREADS) {
        double s = 0.0, q = 0.0;
        for (int w = 0; w < (int)gridDim.x; w++) {
            s += (double)partials[(size_t)w * 2 * C + c];
            q += (double)partials[(size_t)w * 2 * C + C + c];
        }
        const double n = (double)M;
        const double dm = s / n;  // mean - pivot
        double var = q / n - dm * dm;
        if (var < 0.0) var = 0.0;
        const float mean = (float)((double)x[c] + dm);
        save_mean[c] = mean;
        save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = M > 1 ? var * n / (n - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    if (t == 0) *counter = 0;
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
                                                             float* __restrict__ partials, int* __restrict__ counter,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ means) {
    __shared__ float4 s_a[BN_THREADS], s_b[BN_THREADS];
    __shared__ int s_last;
    const int t = threadIdx.x;
    const int col = t % c4, r0 = t / c4;
    const bool active = r0 < rpi;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const float4* d4 = reinterpret_cast<const float4*>(dy);
    const float4 m = reinterpret_cast<const float4*>(save_mean)[col], is = reinterpret_cast<const float4*>(save_invstd)[col];
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgx = sg;
    if (active) {
        for (long long row = (long long)blockIdx.x * rpi + r0; row < M; row += (long long)gridDim.x * rpi) {
            const long long i = row * c4 + col;
            float4 g = d4[i];
            const float4 v = x4[i];
            if (relu) {
                const float4 o = y4[i];
                g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
            }
            sg = f4_add(sg, g);
            sgx.x = fmaf(g.x, (v.x - m.x) * is.x, sgx.x); sgx.y = fmaf(g.y, (v.y - m.y) * is.y, sgx.y);
            sgx.z = fmaf(g.z, (v.z - m.z) * is.z, sgx.z); sgx.w = fmaf(g.w, (v.w - m.w) * is.w, sgx.w);
        }
    }
    column_reduce(sg, sgx, c4, rpi, s_a, s_b);
    float* mine = partials + (size_t)blockIdx.x * 2 * C;
    if (t < c4) {
        reinterpret_cast<float4*>(mine)[t] = sg;
        reinterpret_cast<float4*>(mine + C)[t] = sgx;
    }
    __threadfence();
    __syncthreads();
    if (t == 0) s_last = (atomicAdd(counter, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int c = t; c < C; c += BN_THREADS) {
        double s = 0.0, q = 0.0;
        for (int w = 0; w < (int)gridDim.x; w++) {
            s += (double)partials[(size_t)w * 2 * C + c];
            q += (double)partials[(size_t)w * 2 * C + C + c];
        }
        if (dbeta) dbeta[c] = (float)s;
        if (dgamma) dgamma[c] = (float)q;
        means[c] = (float)(s / (double)M);
        means[C + c] = (float)(q / (double)M);
    }
    if (t == 0) *counter = 0;
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ dy, long long total4, int C, int c4,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_invstd,
                                                            const float* __restrict__ means, int relu,
                                                            float* __restrict__ dx) {
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
        o4[i] = r;
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
                                    float* save_mean, float* save_invstd, float* scratch, int32_t* counter, void* stream) {
    GF_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && scratch && counter, "gf_bn_relu_train_fwd: null argument");
    GF_CHECK_ARG(M >= 2, "gf_bn_relu_train_fwd: batch statistics need at least two rows (M=%d)", M);
    GF_CHECK_ARG(C >= 4 && C <= BN_MAX_C && (C % 4) == 0, "gf_bn_relu_train_fwd: C=%d (multiples of 4 up to %d)", C, BN_MAX_C);
    GF_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "gf_bn_relu_train_fwd: running_mean / running_var come together");
    GF_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)save_mean) |
                   ((uintptr_t)save_invstd) | ((uintptr_t)scratch)) % 16) == 0, "gf_bn_relu_train_fwd: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const BnGeom g = bn_geom(M, C);
    hipLaunchKernelGGL(k_bn_stats, dim3(g.nwg), dim3(BN_THREADS), 0, st, x, M, C, g.c4, g.rpi, eps, momentum, scratch, counter,
                       running_mean, running_var, save_mean, save_invstd);
    const long long total4 = (long long)M * g.c4;
    hipLaunchKernelGGL(k_bn_apply, dim3(apply_grid(total4)), dim3(BN_THREADS), 0, st, x, total4, g.c4, gamma, beta, save_mean,
                       save_invstd, relu, y);
    GF_CHECK_LAUNCH("gf_bn_relu_train_fwd");
    return GF_OK;
}

extern "C" int gf_bn_relu_train_bwd(const float* x, const float* y, const float* dy, int M, int C, const float* gamma,
                                    const float* save_mean, const float* save_invstd, int relu, float* dx, float* dgamma,
                                    float* dbeta, float* scratch, int32_t* counter, void* stream) {
    GF_CHECK_ARG(x && dy && gamma && save_mean && save_invstd && scratch && counter && (y || !relu),
                 "gf_bn_relu_train_bwd: null argument");
    GF_CHECK_ARG(M >= 2, "gf_bn_relu_train_bwd: M=%d", M);
    GF_CHECK_ARG(C >= 4 && C <= BN_MAX_C && (C % 4) == 0, "gf_bn_relu_train_bwd: C=%d (multiples of 4 up to %d)", C, BN_MAX_C);
    GF_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)gamma) |
                   ((uintptr_t)save_mean) | ((uintptr_t)save_invstd) | ((uintptr_t)scratch)) % 16) == 0,
                 "gf_bn_relu_train_bwd: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const BnGeom g = bn_geom(M, C);
    float* means = scratch + (size_t)g.nwg * 2 * C;
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3(g.nwg), dim3(BN_THREADS), 0, st, x, y, dy, M, C, g.c4, g.rpi, save_mean, save_invstd,
                       relu, scratch, counter, dgamma, dbeta, means);
    if (dx) {
        const long long total4 = (long long)M * g.c4;
        hipLaunchKernelGGL(k_bn_bwd_apply, dim3(apply_grid(total4)), dim3(BN_THREADS), 0, st, x, y, dy, total4, C, g.c4, gamma,
                           save_mean, save_invstd, means, relu, dx);
    }
    GF_CHECK_LAUNCH("gf_bn_relu_train_bwd");
    return GF_OK;
}
