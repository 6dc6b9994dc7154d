// Fused per-scene voxel transformer of the two deepest U-Net levels (inference).
//
// Replaces, for one level, the ~75 small launches of
//   before_transformer_linear -> TransformerEncoder(d_model=128, N=2, heads=4, d_ff=64) -> after_transformer_linear
// (reference: model/geoformer/geoformer_modules.py:64-68,120-127 and model/transformer.py:62-188) by ONE
// launch: a level holds 10..10^3 voxels per scene, so the stage is pure launch latency on the host.
//
// One 1024-thread workgroup per scene walks the phases below with workgroup barriers in between; the
// token matrices live in a global scratch that never leaves L2 (T x 640 floats).  Every product runs on
// v_mfma_f32_16x16x4_f32 with operands loaded straight from the row-major activations and from the
// row-major nn.Linear weights [out, in] (lane (j = lane&15, g = lane>>4) loads 16 bytes at column 4g of
// row j; the MFMA k index is only a summation index, so both operands use the same 4g+kk permutation and
// no packing is needed).  Attention keeps everything transposed like the decoder kernel: S^T = K Q^T puts
// the query on the column, so the soft-max statistics are per-lane-column values and the accumulator of
// S^T is directly the B operand of O^T = V^T P^T (online soft-max over 16-key tiles).
//
//   x   = before(f) + pos(mean_j(p_i - p_j))
//   per layer:  x2 = Norm1(x); q,k,v = Linear(x2); x += out(softmax(q k^T / sqrt(32)) v)
//               x2 = Norm2(x); x += ff2(relu(ff1(x2)))
//   y   = after(Norm(x))
// Norm = alpha * (x - mean) / (std_unbiased + eps) + bias  (transformer.py:62-76).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BT_D 128
#define BT_H 4
#define BT_DK 32
#define BT_FF 64
#define BT_MAXL 4
#define BT_THREADS 1024
#define BT_SCRATCH_PER_TOKEN (BT_D * 5)  // x, x2, qkv

struct BtLayer {
    const float *n1a, *n1b, *qw, *qb, *kw, *kb, *vw, *vb, *ow, *ob, *n2a, *n2b, *f1w, *f1b, *f2w, *f2b;
};
struct BtParams {
    const float *bw, *bb, *pw, *pb;
    BtLayer L[BT_MAXL];
    const float *na, *nb, *aw, *ab;
    int nl;
};

__device__ __forceinline__ f32x4 mfma4(float4 a, float4 b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// Y[t][n] (op)= sum_k X[t][k] W[n][k] + b[n];  MODE 0: store, 1: ReLU then store, 2: add into Y (residual),
// 3: store + position term  pw[n][:] . rel(t) + pb[n]  (the first product of the stage).
template <int MODE>
__device__ __forceinline__ void bt_gemm(const float* __restrict__ X, int ldx, int T, int K,
                                        const float* __restrict__ W, const float* __restrict__ bias, int N,
                                        float* __restrict__ Y, int ldy, int wave, int nwaves, int lane,
                                        const int* __restrict__ coords = nullptr, const int* psum = nullptr,
                                        const float* __restrict__ pw = nullptr,
                                        const float* __restrict__ pb = nullptr) {
    const int j = lane & 15, g = lane >> 4;
    const int RT = (T + 15) >> 4, CT = N >> 4, KC = K >> 4;
    for (int task = wave; task < RT * CT; task += nwaves) {
        const int rt = task / CT, ct = task - rt * CT;
        const int row = rt * 16 + j;
        const float* xa = X + (size_t)row * ldx + 4 * g;
        const float* wb = W + (size_t)(ct * 16 + j) * K + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool live = row < T;
#pragma unroll 4
        for (int kc = 0; kc < KC; kc++) {
            float4 a = live ? *reinterpret_cast<const float4*>(xa + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b = *reinterpret_cast<const float4*>(wb + kc * 16);
            acc = mfma4(a, b, acc);
        }
        const int col = ct * 16 + j;
        const float bs = bias[col];
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, pbv = 0.f;
        if (MODE == 3) {
            p0 = pw[col * 3 + 0];
            p1 = pw[col * 3 + 1];
            p2 = pw[col * 3 + 2];
            pbv = pb[col];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = rt * 16 + 4 * g + i;
            if (r >= T) continue;
            float v = acc[i] + bs;
            float* y = Y + (size_t)r * ldy + col;
            if (MODE == 1) v = fmaxf(v, 0.f);
            if (MODE == 2) v = *y + v;
            if (MODE == 3) {
                // mean_j (p_r - p_j): exact integer sum, one division (transformer.py:175-177)
                const float ft = (float)T;
                float r0 = (float)(T * coords[r * 4 + 1] - psum[0]) / ft;
                float r1 = (float)(T * coords[r * 4 + 2] - psum[1]) / ft;
                float r2 = (float)(T * coords[r * 4 + 3] - psum[2]) / ft;
                v = v + (fmaf(p2, r2, fmaf(p1, r1, p0 * r0)) + pbv);
            }
            *y = v;
        }
    }
}

// q, k, v in one sweep: column tiles 0..7 -> q, 8..15 -> k, 16..23 -> v of QKV[T][384]
__device__ __forceinline__ void bt_qkv(const float* __restrict__ X2, int T, const BtLayer& L, float* __restrict__ QKV,
                                       int wave, int nwaves, int lane) {
    const int j = lane & 15, g = lane >> 4;
    const int RT = (T + 15) >> 4;
    for (int task = wave; task < RT * 24; task += nwaves) {
        const int rt = task / 24, ct = task - rt * 24;
        const int which = ct >> 3, c8 = ct & 7;
        const float* W = which == 0 ? L.qw : (which == 1 ? L.kw : L.vw);
        const float* B = which == 0 ? L.qb : (which == 1 ? L.kb : L.vb);
        const int row = rt * 16 + j;
        const bool live = row < T;
        const float* xa = X2 + (size_t)row * BT_D + 4 * g;
        const float* wb = W + (size_t)(c8 * 16 + j) * BT_D + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < BT_D / 16; kc++) {
            float4 a = live ? *reinterpret_cast<const float4*>(xa + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b = *reinterpret_cast<const float4*>(wb + kc * 16);
            acc = mfma4(a, b, acc);
        }
        const float bs = B[c8 * 16 + j];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = rt * 16 + 4 * g + i;
            if (r < T) QKV[(size_t)r * (3 * BT_D) + ct * 16 + j] = acc[i] + bs;
        }
    }
}

// Norm (transformer.py:62-76): one wave per token, two channels per lane
__device__ __forceinline__ void bt_norm(const float* __restrict__ X, int T, const float* __restrict__ alpha,
                                        const float* __restrict__ beta, float* __restrict__ Y, int wave, int nwaves,
                                        int lane) {
    const float a0 = alpha[2 * lane], a1 = alpha[2 * lane + 1], b0 = beta[2 * lane], b1 = beta[2 * lane + 1];
    for (int t = wave; t < T; t += nwaves) {
        float2 v = *reinterpret_cast<const float2*>(X + (size_t)t * BT_D + 2 * lane);
        float s = v.x + v.y;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        const float mu = s / (float)BT_D;
        const float dx = v.x - mu, dy = v.y - mu;
        float q = dx * dx + dy * dy;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) q += __shfl_xor(q, d, 64);
        const float den = sqrtf(q / (float)(BT_D - 1)) + 1e-6f;
        float2 o;
        o.x = a0 * dx / den + b0;
        o.y = a1 * dy / den + b1;
        *reinterpret_cast<float2*>(Y + (size_t)t * BT_D + 2 * lane) = o;
    }
}

// multi-head attention, one wave per (16-query tile, head); output into O[T][128]
__device__ __forceinline__ void bt_attention(const float* __restrict__ QKV, int T, float* __restrict__ O, int wave,
                                             int nwaves, int lane) {
    const int j = lane & 15, g = lane >> 4;
    const int QT = (T + 15) >> 4;
    const float scale = 0.17677669529663687f;  // 1 / sqrt(32)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int task = wave; task < QT * BT_H; task += nwaves) {
        const int qt = task >> 2, h = task & 3;
        const int qrow = qt * 16 + j;
        float4 bq0 = z4, bq1 = z4;
        if (qrow < T) {
            const float* qp = QKV + (size_t)qrow * (3 * BT_D) + h * BT_DK + 4 * g;
            bq0 = *reinterpret_cast<const float4*>(qp);
            bq1 = *reinterpret_cast<const float4*>(qp + 16);
        }
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, l = 0.f;
        for (int kt = 0; kt < QT; kt++) {
            const int krow = kt * 16 + j;
            float4 a0 = z4, a1 = z4;
            if (krow < T) {
                const float* kp = QKV + (size_t)krow * (3 * BT_D) + BT_D + h * BT_DK + 4 * g;
                a0 = *reinterpret_cast<const float4*>(kp);
                a1 = *reinterpret_cast<const float4*>(kp + 16);
            }
            // V^T operands of this key tile (issued early, used after the soft-max update)
            float v0[4], v1[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kt * 16 + 4 * g + i;
                const float* vp = QKV + (size_t)key * (3 * BT_D) + 2 * BT_D + h * BT_DK + j;
                v0[i] = key < T ? vp[0] : 0.f;
                v1[i] = key < T ? vp[16] : 0.f;
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            s = mfma4(a0, bq0, s);
            s = mfma4(a1, bq1, s);
            float sc[4];
#pragma unroll
            for (int i = 0; i < 4; i++) sc[i] = (kt * 16 + 4 * g + i) < T ? s[i] * scale : -INFINITY;
            float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);  // finite: key 0 of tile 0 always exists
            const float corr = expf(m - mnew);
            float p[4];
#pragma unroll
            for (int i = 0; i < 4; i++) p[i] = expf(sc[i] - mnew);
            l = l * corr + ((p[0] + p[1]) + (p[2] + p[3]));
            o0 *= corr;
            o1 *= corr;
            m = mnew;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[i], p[i], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[i], p[i], o1, 0, 0, 0);
            }
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (qrow < T) {
            float* op = O + (size_t)qrow * BT_D + h * BT_DK + 4 * g;
            *reinterpret_cast<float4*>(op) = make_float4(o0[0] / l, o0[1] / l, o0[2] / l, o0[3] / l);
            *reinterpret_cast<float4*>(op + 16) = make_float4(o1[0] / l, o1[1] / l, o1[2] / l, o1[3] / l);
        }
    }
}

__global__ __launch_bounds__(BT_THREADS) void k_backbone_transformer(const float* __restrict__ feats,
                                                                     const int* __restrict__ coords,
                                                                     const int* __restrict__ scene_offsets, int c,
                                                                     BtParams P, float* __restrict__ scratch,
                                                                     float* __restrict__ out) {
    __shared__ int psum[3];
    const int s0 = scene_offsets[blockIdx.x], T = scene_offsets[blockIdx.x + 1] - s0;
    if (T <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = BT_THREADS / 64;
    float* X = scratch + (size_t)s0 * BT_SCRATCH_PER_TOKEN;
    float* X2 = X + (size_t)T * BT_D;
    float* QKV = X2 + (size_t)T * BT_D;
    const float* f = feats + (size_t)s0 * c;
    const int* xyz = coords + (size_t)s0 * 4;
    float* y = out + (size_t)s0 * c;

    if (threadIdx.x < 3) psum[threadIdx.x] = 0;
    __syncthreads();
    {
        int a0 = 0, a1 = 0, a2 = 0;
        for (int t = threadIdx.x; t < T; t += BT_THREADS) {
            a0 += xyz[t * 4 + 1];
            a1 += xyz[t * 4 + 2];
            a2 += xyz[t * 4 + 3];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            a0 += __shfl_xor(a0, d, 64);
            a1 += __shfl_xor(a1, d, 64);
            a2 += __shfl_xor(a2, d, 64);
        }
        if (lane == 0 && wave * 64 < T) {
            atomicAdd(&psum[0], a0);
            atomicAdd(&psum[1], a1);
            atomicAdd(&psum[2], a2);
        }
    }
    __syncthreads();
    bt_gemm<3>(f, c, T, c, P.bw, P.bb, BT_D, X, BT_D, wave, nw, lane, xyz, psum, P.pw, P.pb);
    __syncthreads();
    for (int li = 0; li < P.nl; li++) {
        const BtLayer& L = P.L[li];
        bt_norm(X, T, L.n1a, L.n1b, X2, wave, nw, lane);
        __syncthreads();
        bt_qkv(X2, T, L, QKV, wave, nw, lane);
        __syncthreads();
        bt_attention(QKV, T, X2, wave, nw, lane);
        __syncthreads();
        bt_gemm<2>(X2, BT_D, T, BT_D, L.ow, L.ob, BT_D, X, BT_D, wave, nw, lane);
        __syncthreads();
        bt_norm(X, T, L.n2a, L.n2b, X2, wave, nw, lane);
        __syncthreads();
        bt_gemm<1>(X2, BT_D, T, BT_D, L.f1w, L.f1b, BT_FF, QKV, BT_FF, wave, nw, lane);
        __syncthreads();
        bt_gemm<2>(QKV, BT_FF, T, BT_FF, L.f2w, L.f2b, BT_D, X, BT_D, wave, nw, lane);
        __syncthreads();
    }
    bt_norm(X, T, P.na, P.nb, X2, wave, nw, lane);
    __syncthreads();
    bt_gemm<0>(X2, BT_D, T, BT_D, P.aw, P.ab, c, y, c, wave, nw, lane);
}

extern "C" size_t gf_backbone_transformer_scratch_bytes(int M) {
    return (size_t)(M > 0 ? M : 0) * BT_SCRATCH_PER_TOKEN * sizeof(float);
}

extern "C" int gf_backbone_transformer_num_params(int n_layers) { return 8 + 16 * n_layers; }

extern "C" int gf_backbone_transformer(const float* feats, const int* coords, const int* scene_offsets, int n_scenes,
                                       int M, int c, int n_layers, const float* const* params, void* scratch,
                                       float* out, void* stream) {
    GF_CHECK_ARG(c > 0 && c % 16 == 0, "gf_backbone_transformer: channel width %d must be a multiple of 16", c);
    GF_CHECK_ARG(n_layers >= 1 && n_layers <= BT_MAXL, "gf_backbone_transformer: 1..%d layers, got %d", BT_MAXL,
                 n_layers);
    GF_CHECK_ARG(n_scenes >= 0 && M >= 0, "gf_backbone_transformer: bad sizes");
    GF_CHECK_ARG(params != nullptr, "gf_backbone_transformer: params is null");
    if (n_scenes == 0 || M == 0) return GF_OK;
    const int np = 8 + 16 * n_layers;
    for (int i = 0; i < np; i++)
        GF_CHECK_ARG(params[i] != nullptr, "gf_backbone_transformer: params[%d] is null", i);
    BtParams P;
    int k = 0;
    P.bw = params[k++];
    P.bb = params[k++];
    P.pw = params[k++];
    P.pb = params[k++];
    for (int l = 0; l < n_layers; l++) {
        BtLayer& L = P.L[l];
        L.n1a = params[k++];
        L.n1b = params[k++];
        L.qw = params[k++];
        L.qb = params[k++];
        L.kw = params[k++];
        L.kb = params[k++];
        L.vw = params[k++];
        L.vb = params[k++];
        L.ow = params[k++];
        L.ob = params[k++];
        L.n2a = params[k++];
        L.n2b = params[k++];
        L.f1w = params[k++];
        L.f1b = params[k++];
        L.f2w = params[k++];
        L.f2b = params[k++];
    }
    for (int l = n_layers; l < BT_MAXL; l++) P.L[l] = P.L[0];
    P.na = params[k++];
    P.nb = params[k++];
    P.aw = params[k++];
    P.ab = params[k++];
    P.nl = n_layers;
    hipLaunchKernelGGL(k_backbone_transformer, dim3(n_scenes), dim3(BT_THREADS), 0, (hipStream_t)stream, feats,
                       coords, scene_offsets, c, P, (float*)scratch, out);
    GF_CHECK_LAUNCH("gf_backbone_transformer");
    return GF_OK;
}
