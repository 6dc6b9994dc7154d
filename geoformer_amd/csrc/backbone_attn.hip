// Fused per-scene voxel transformer of the two deepest U-Net levels (inference).
//
// Replaces, for one level, the ~75 small launches of
//   before_transformer_linear -> TransformerEncoder(d_model=128, N, heads=4, d_ff=64) -> after_transformer_linear
// (reference: model/geoformer/geoformer_modules.py:64-68,120-127 and model/transformer.py:62-188) by N+1
// launches: a level holds 10..10^3 voxels per scene, so the stage is pure launch latency on the host, and a
// single-workgroup formulation is pure dependent latency on the device.  Everything except the attention itself
// is local to a token, so the work is cut at the only grid-wide dependency (every token's K and V):
//
//   k_bt_pre   (a workgroup per 16 tokens)  x = before(f) + pos(mean_j(p_i - p_j));  qkv = Linear(Norm1_0(x))
//   k_bt_layer (a workgroup per 16 queries) x += out(softmax(q k^T / sqrt(32)) v)      <- all tokens' K, V
//                                           x += ff2(relu(ff1(Norm2(x))))
//                                           next layer: qkv = Linear(Norm1_{l+1}(x));  last: y = after(Norm(x))
//
// Token tiles live in LDS; x and qkv of the scene sit in a global scratch between launches (T x 512 floats, L2).
// Every product runs on v_mfma_f32_16x16x4_f32 with operands loaded straight from the row-major activations and
// from the row-major nn.Linear weights [out, in] (lane (j = lane&15, g = lane>>4) loads 16 bytes at column 4g of
// row j; the MFMA k index is only a summation index, so both operands use the same 4g+kk permutation and no
// packing is needed).  Attention keeps everything transposed like the decoder kernel: S^T = K Q^T puts the
// query on the column, so the soft-max statistics are per-lane-column values and the accumulator of S^T is
// directly the B operand of O^T = V^T P^T (online soft-max over 16-key tiles, one tile of look-ahead).
// Norm = alpha * (x - mean) / (std_unbiased + eps) + bias  (transformer.py:62-76).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BT_D 128
#define BT_H 4
#define BT_DK 32
#define BT_FF 64
#define BT_MAXL 4
#define BT_THREADS 512
#define BT_LD 132  // padded LDS row (floats)
#define BT_SCRATCH_PER_TOKEN (BT_D * 4)  // x, qkv

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

// one 16-row tile: epi(r, col, act(sum_k A[r][k] W[col][k] + b[col])) for the column tiles ct = wave, wave+nw, ...
// A in LDS or global memory (row stride lda); rows >= nvalid read as zero and are not emitted
template <bool RELU, typename Epi>
__device__ __forceinline__ void bt_tile_gemm(const float* A, int lda, int nvalid, int K, const float* __restrict__ W,
                                             const float* __restrict__ bias, int N, int wave, int nwaves, int lane,
                                             Epi epi) {
    const int j = lane & 15, g = lane >> 4;
    const int KC = K >> 4;
    for (int ct = wave; ct < (N >> 4); ct += nwaves) {
        const float* xa = A + (size_t)j * lda + 4 * g;
        const float* wb = W + (size_t)(ct * 16 + j) * K + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int kc = 0; kc < KC; kc++) {
            float4 a = j < nvalid ? *reinterpret_cast<const float4*>(xa + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b = *reinterpret_cast<const float4*>(wb + kc * 16);
            acc = mfma4(a, b, acc);
        }
        const int col = ct * 16 + j;
        const float bs = bias[col];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * g + i;
            if (r >= nvalid) continue;
            float v = acc[i] + bs;
            if (RELU) v = fmaxf(v, 0.f);
            epi(r, col, v);
        }
    }
}

// Norm (transformer.py:62-76) of the rows of an LDS tile: one wave per row, two channels per lane
template <typename Out>
__device__ __forceinline__ void bt_tile_norm(const float (*S)[BT_LD], int nvalid, const float* __restrict__ alpha,
                                             const float* __restrict__ beta, int wave, int nwaves, int lane, Out out) {
    const float a0 = alpha[2 * lane], a1 = alpha[2 * lane + 1], b0 = beta[2 * lane], b1 = beta[2 * lane + 1];
    for (int r = wave; r < nvalid; r += nwaves) {
        const float2 v = *reinterpret_cast<const float2*>(&S[r][2 * lane]);
        float s = v.x + v.y;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        const float mu = s / (float)BT_D;
        const float dx = v.x - mu, dy = v.y - mu;
        float q = dx * dx + dy * dy;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) q += __shfl_xor(q, d, 64);
        const float den = sqrtf(q / (float)(BT_D - 1)) + 1e-6f;
        out(r, 2 * lane, a0 * dx / den + b0, a1 * dy / den + b1);
    }
}

// q, k, v of the tile from its normed rows (LDS) into the scene's QKV[T][384]
__device__ __forceinline__ void bt_tile_qkv(const float (*S)[BT_LD], int nvalid, const BtLayer& L, float* qkv_rows,
                                            int wave, int nwaves, int lane) {
    bt_tile_gemm<false>(&S[0][0], BT_LD, nvalid, BT_D, L.qw, L.qb, BT_D, wave, nwaves, lane,
                        [&](int r, int c, float v) { qkv_rows[(size_t)r * (3 * BT_D) + c] = v; });
    bt_tile_gemm<false>(&S[0][0], BT_LD, nvalid, BT_D, L.kw, L.kb, BT_D, wave, nwaves, lane,
                        [&](int r, int c, float v) { qkv_rows[(size_t)r * (3 * BT_D) + BT_D + c] = v; });
    bt_tile_gemm<false>(&S[0][0], BT_LD, nvalid, BT_D, L.vw, L.vb, BT_D, wave, nwaves, lane,
                        [&](int r, int c, float v) { qkv_rows[(size_t)r * (3 * BT_D) + 2 * BT_D + c] = v; });
}

__global__ __launch_bounds__(BT_THREADS) void k_bt_pre(const float* __restrict__ feats, const int* __restrict__ coords,
                                                       const int* __restrict__ scene_offsets,
                                                       const int* __restrict__ tile_scene,
                                                       const int* __restrict__ tile_first, int c, BtParams P,
                                                       float* __restrict__ scratch) {
    __shared__ float sX[16][BT_LD], sT[16][BT_LD];
    __shared__ int psum[3];
    const int sc = tile_scene[blockIdx.x];
    if (sc < 0) return;  // padding tile (the grid is sized by an upper bound)
    const int s0 = scene_offsets[sc], T = scene_offsets[sc + 1] - s0;
    const int t0 = (blockIdx.x - tile_first[sc]) * 16;
    const int nvalid = min(16, T - t0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = BT_THREADS / 64;
    float* X = scratch + (size_t)s0 * BT_SCRATCH_PER_TOKEN;
    float* QKV = X + (size_t)T * BT_D;
    const int* xyz = coords + (size_t)s0 * 4;
    if (threadIdx.x < 3) psum[threadIdx.x] = 0;
    __syncthreads();
    {
        // sum of the scene's voxel coordinates (exact integers): mean_j (p_i - p_j) = (T p_i - sum) / T
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
    const float ft = (float)T;
    const int* tz = xyz + (size_t)t0 * 4;
    bt_tile_gemm<false>(feats + ((size_t)s0 + t0) * c, c, nvalid, c, P.bw, P.bb, BT_D, wave, nw, lane,
                        [&](int r, int col, float v) {
                            const float r0 = (float)(T * tz[r * 4 + 1] - psum[0]) / ft;
                            const float r1 = (float)(T * tz[r * 4 + 2] - psum[1]) / ft;
                            const float r2 = (float)(T * tz[r * 4 + 3] - psum[2]) / ft;
                            const float pe = fmaf(P.pw[col * 3 + 2], r2, fmaf(P.pw[col * 3 + 1], r1, P.pw[col * 3] * r0));
                            sX[r][col] = v + (pe + P.pb[col]);
                        });
    __syncthreads();
    for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS)
        X[(size_t)(t0 + (i >> 7)) * BT_D + (i & 127)] = sX[i >> 7][i & 127];
    bt_tile_norm(sX, nvalid, P.L[0].n1a, P.L[0].n1b, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
        sT[r][c2] = v0;
        sT[r][c2 + 1] = v1;
    });
    __syncthreads();
    bt_tile_qkv(sT, nvalid, P.L[0], QKV + (size_t)t0 * (3 * BT_D), wave, nw, lane);
}

__global__ __launch_bounds__(BT_THREADS) void k_bt_layer(const int* __restrict__ scene_offsets,
                                                         const int* __restrict__ tile_scene,
                                                         const int* __restrict__ tile_first, int c, int li, BtParams P,
                                                         float* __restrict__ scratch_in, float* __restrict__ scratch_out,
                                                         float* __restrict__ out) {
    __shared__ float sX[16][BT_LD], sT[16][BT_LD], sO[16][BT_LD];
    const int sc = tile_scene[blockIdx.x];
    if (sc < 0) return;
    const int s0 = scene_offsets[sc], T = scene_offsets[sc + 1] - s0;
    const int t0 = (blockIdx.x - tile_first[sc]) * 16;
    const int nvalid = min(16, T - t0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = BT_THREADS / 64;
    const int j = lane & 15, g = lane >> 4;
    const float* X = scratch_in + (size_t)s0 * BT_SCRATCH_PER_TOKEN;
    const float* QKV = X + (size_t)T * BT_D;
    float* Xo = scratch_out + (size_t)s0 * BT_SCRATCH_PER_TOKEN;
    float* QKVo = Xo + (size_t)T * BT_D;
    const BtLayer& L = P.L[li];
    if (wave < BT_H) {
        // one wave per head: O[16 queries][32] over all keys of the scene
        const int h = wave;
        const int QT = (T + 15) >> 4;
        const float scale = 0.17677669529663687f;  // 1 / sqrt(32)
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int qrow = t0 + j;
        float4 bq0 = z4, bq1 = z4;
        if (qrow < T) {
            const float* qp = QKV + (size_t)qrow * (3 * BT_D) + h * BT_DK + 4 * g;
            bq0 = *reinterpret_cast<const float4*>(qp);
            bq1 = *reinterpret_cast<const float4*>(qp + 16);
        }
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, l = 0.f;
        float4 a0 = z4, a1 = z4;
        float v0[4], v1[4];
        auto fetch = [&](int kt, float4& k0, float4& k1, float (&w0)[4], float (&w1)[4]) {
            const int krow = kt * 16 + j;
            k0 = z4;
            k1 = z4;
            if (krow < T) {
                const float* kp = QKV + (size_t)krow * (3 * BT_D) + BT_D + h * BT_DK + 4 * g;
                k0 = *reinterpret_cast<const float4*>(kp);
                k1 = *reinterpret_cast<const float4*>(kp + 16);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kt * 16 + 4 * g + i;
                const float* vp = QKV + (size_t)key * (3 * BT_D) + 2 * BT_D + h * BT_DK + j;
                w0[i] = key < T ? vp[0] : 0.f;
                w1[i] = key < T ? vp[16] : 0.f;
            }
        };
        fetch(0, a0, a1, v0, v1);
        for (int kt = 0; kt < QT; kt++) {
            float4 n0 = z4, n1 = z4;
            float u0[4] = {0.f, 0.f, 0.f, 0.f}, u1[4] = {0.f, 0.f, 0.f, 0.f};
            if (kt + 1 < QT) fetch(kt + 1, n0, n1, u0, u1);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            s = mfma4(a0, bq0, s);
            s = mfma4(a1, bq1, s);
            float scv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) scv[i] = (kt * 16 + 4 * g + i) < T ? s[i] * scale : -INFINITY;
            float mx = fmaxf(fmaxf(scv[0], scv[1]), fmaxf(scv[2], scv[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);  // finite: key 0 of tile 0 always exists
            const float corr = expf(m - mnew);
            float p[4];
#pragma unroll
            for (int i = 0; i < 4; i++) p[i] = expf(scv[i] - mnew);
            l = l * corr + ((p[0] + p[1]) + (p[2] + p[3]));
            o0 *= corr;
            o1 *= corr;
            m = mnew;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[i], p[i], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[i], p[i], o1, 0, 0, 0);
            }
            a0 = n0;
            a1 = n1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                v0[i] = u0[i];
                v1[i] = u1[i];
            }
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            sO[j][h * BT_DK + 4 * g + i] = o0[i] / l;
            sO[j][h * BT_DK + 16 + 4 * g + i] = o1[i] / l;
        }
    }
    __syncthreads();
    // x += out(O)
    const float* xg = X + (size_t)t0 * BT_D;
    bt_tile_gemm<false>(&sO[0][0], BT_LD, nvalid, BT_D, L.ow, L.ob, BT_D, wave, nw, lane,
                        [&](int r, int col, float v) { sX[r][col] = xg[r * BT_D + col] + v; });
    __syncthreads();
    bt_tile_norm(sX, nvalid, L.n2a, L.n2b, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
        sT[r][c2] = v0;
        sT[r][c2 + 1] = v1;
    });
    __syncthreads();
    // x += ff2(relu(ff1(.)));  the hidden tile reuses sO
    bt_tile_gemm<true>(&sT[0][0], BT_LD, nvalid, BT_D, L.f1w, L.f1b, BT_FF, wave, nw, lane,
                       [&](int r, int col, float v) { sO[r][col] = v; });
    __syncthreads();
    bt_tile_gemm<false>(&sO[0][0], BT_LD, nvalid, BT_FF, L.f2w, L.f2b, BT_D, wave, nw, lane,
                        [&](int r, int col, float v) { sX[r][col] += v; });
    __syncthreads();
    if (li + 1 < P.nl) {
        for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS)
            Xo[(size_t)(t0 + (i >> 7)) * BT_D + (i & 127)] = sX[i >> 7][i & 127];
        bt_tile_norm(sX, nvalid, P.L[li + 1].n1a, P.L[li + 1].n1b, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
            sT[r][c2] = v0;
            sT[r][c2 + 1] = v1;
        });
        __syncthreads();
        bt_tile_qkv(sT, nvalid, P.L[li + 1], QKVo + (size_t)t0 * (3 * BT_D), wave, nw, lane);
    } else {
        bt_tile_norm(sX, nvalid, P.na, P.nb, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
            sT[r][c2] = v0;
            sT[r][c2 + 1] = v1;
        });
        __syncthreads();
        float* y = out + ((size_t)s0 + t0) * c;
        bt_tile_gemm<false>(&sT[0][0], BT_LD, nvalid, BT_D, P.aw, P.ab, c, wave, nw, lane,
                            [&](int r, int col, float v) { y[(size_t)r * c + col] = v; });
    }
}

// tile tables: tile_scene[i], tile_first[scene] from the scene offsets (one tiny launch, no host round trip)
__global__ void k_bt_tiles(const int* __restrict__ scene_offsets, int n_scenes, int max_tiles, int* __restrict__ tile_scene,
                           int* __restrict__ tile_first) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int t = 0;
    for (int s = 0; s < n_scenes; s++) {
        tile_first[s] = t;
        const int nt = (scene_offsets[s + 1] - scene_offsets[s] + 15) >> 4;
        for (int i = 0; i < nt && t < max_tiles; i++) tile_scene[t++] = s;
    }
    for (; t < max_tiles; t++) tile_scene[t] = -1;
}

extern "C" size_t gf_backbone_transformer_scratch_bytes(int M) {
    // two copies of (x, qkv) -- a layer reads one and writes the other -- and the tile tables
    const size_t m = (size_t)(M > 0 ? M : 0);
    return 2 * m * BT_SCRATCH_PER_TOKEN * sizeof(float) + (m / 16 + 2 * 4096 + 64) * sizeof(int);
}

extern "C" int gf_backbone_transformer_num_params(int n_layers) { return 8 + 16 * n_layers; }

extern "C" int gf_backbone_transformer(const float* feats, const int* coords, const int* scene_offsets, int n_scenes,
                                       int M, int c, int n_layers, const float* const* params, void* scratch,
                                       float* out, void* stream) {
    GF_CHECK_ARG(c > 0 && c % 16 == 0, "gf_backbone_transformer: channel width %d must be a multiple of 16", c);
    GF_CHECK_ARG(n_layers >= 1 && n_layers <= BT_MAXL, "gf_backbone_transformer: 1..%d layers, got %d", BT_MAXL,
                 n_layers);
    GF_CHECK_ARG(n_scenes >= 0 && n_scenes <= 4096 && M >= 0, "gf_backbone_transformer: bad sizes");
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
    hipStream_t st = (hipStream_t)stream;
    float* sA = (float*)scratch;
    float* sB = sA + (size_t)M * BT_SCRATCH_PER_TOKEN;
    int* tile_scene = (int*)(sB + (size_t)M * BT_SCRATCH_PER_TOKEN);
    const int max_tiles = M / 16 + n_scenes;  // sum of ceil(T_s / 16) never exceeds this
    int* tile_first = tile_scene + (M / 16 + 4096 + 8);
    hipLaunchKernelGGL(k_bt_tiles, dim3(1), dim3(64), 0, st, scene_offsets, n_scenes, max_tiles, tile_scene, tile_first);
    hipLaunchKernelGGL(k_bt_pre, dim3(max_tiles), dim3(BT_THREADS), 0, st, feats, coords, scene_offsets, tile_scene,
                       tile_first, c, P, sA);
    for (int l = 0; l < n_layers; l++) {
        float* in = (l & 1) ? sB : sA;
        float* ou = (l & 1) ? sA : sB;
        hipLaunchKernelGGL(k_bt_layer, dim3(max_tiles), dim3(BT_THREADS), 0, st, scene_offsets, tile_scene, tile_first, c,
                           l, P, in, ou, out);
    }
    GF_CHECK_LAUNCH("gf_backbone_transformer");
    return GF_OK;
}
