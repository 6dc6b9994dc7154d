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
#include "train_common.h"

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

// one 16-row tile: epi(r, col, act(sum_k A[r][k] W[col][k] + b[col]) (+ addend[r][col])) for the column tiles ct = wave,
// wave+nw, ...  A in LDS or global memory (row stride lda); rows >= nvalid read as zero and are not emitted.  The bias and the
// epilogue's global operand (`addend`, row stride add_ld) are requested before the products (decoder_layer.h: why).
template <bool RELU, typename Epi>
__device__ __forceinline__ void bt_tile_gemm(const float* A, int lda, int nvalid, int K, const float* __restrict__ W,
                                             const float* __restrict__ bias, int N, int wave, int nwaves, int lane,
                                             Epi epi, const float* __restrict__ addend = nullptr, int add_ld = 0) {
    const int j = lane & 15, g = lane >> 4;
    const int KC = K >> 4;
    for (int ct = wave; ct < (N >> 4); ct += nwaves) {
        const float* xa = A + (size_t)j * lda + 4 * g;
        const float* wb = W + (size_t)(ct * 16 + j) * K + 4 * g;
        const int col = ct * 16 + j;
        const float bs = bias[col];
        float ad[4] = {0.f, 0.f, 0.f, 0.f};
        if (addend) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (4 * g + i < nvalid) ad[i] = addend[(size_t)(4 * g + i) * add_ld + col];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int kc = 0; kc < KC; kc++) {
            float4 a = j < nvalid ? *reinterpret_cast<const float4*>(xa + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b = *reinterpret_cast<const float4*>(wb + kc * 16);
            acc = mfma4(a, b, acc);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * g + i;
            if (r >= nvalid) continue;
            float v = acc[i] + bs;
            if (RELU) v = fmaxf(v, 0.f);
            if (addend) v += ad[i];
            epi(r, col, v);
        }
    }
}

// The same product with its weight operands already in registers (decoder_layer.h, DlW: why): bt_w_load requests NT column
// tiles per wave x KC 16-channel steps + bias ahead of the phase, bt_tile_gemm_w consumes them.  Same arithmetic and order.
template <int NT, int KC>
struct BtW {
    float4 b[NT][KC];
    float bias[NT];
};
template <int NT, int KC>
__device__ __forceinline__ void bt_w_load(BtW<NT, KC>& w, const float* __restrict__ W, const float* __restrict__ bias, int N,
                                          int K, int wave, int nwaves, int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int ct = wave + t * nwaves;
        const bool on = ct < (N >> 4);
        const float* wb = W + (size_t)((on ? ct : 0) * 16 + j) * K + 4 * g;
        w.bias[t] = on ? bias[ct * 16 + j] : 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; kc++)
            w.b[t][kc] = (on && kc < (K >> 4)) ? *reinterpret_cast<const float4*>(wb + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <bool RELU, int NT, int KC, typename Epi>
__device__ __forceinline__ void bt_tile_gemm_w(const float* A, int lda, int nvalid, int K, int N, int wave, int nwaves, int lane,
                                               const BtW<NT, KC>& w, Epi epi, const float* __restrict__ addend = nullptr,
                                               int add_ld = 0) {
    const int j = lane & 15, g = lane >> 4;
    const float* xa = A + (size_t)j * lda + 4 * g;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int ct = wave + t * nwaves;
        if (ct >= (N >> 4)) break;
        const int col = ct * 16 + j;
        float ad[4] = {0.f, 0.f, 0.f, 0.f};
        if (addend) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (4 * g + i < nvalid) ad[i] = addend[(size_t)(4 * g + i) * add_ld + col];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            if (kc < (K >> 4)) {
                float4 a = j < nvalid ? *reinterpret_cast<const float4*>(xa + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = mfma4(a, w.b[t][kc], acc);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * g + i;
            if (r >= nvalid) continue;
            float v = acc[i] + w.bias[t];
            if (RELU) v = fmaxf(v, 0.f);
            if (addend) v += ad[i];
            epi(r, col, v);
        }
    }
}

// Norm (transformer.py:62-76) of the rows of an LDS tile: one wave per row, two channels per lane
struct BtNormP {  // a lane's two channels of a norm's alpha / bias (loaded ahead of the phase)
    float a0, a1, b0, b1;
};
__device__ __forceinline__ BtNormP bt_norm_load(const float* __restrict__ alpha, const float* __restrict__ beta, int lane) {
    return BtNormP{alpha[2 * lane], alpha[2 * lane + 1], beta[2 * lane], beta[2 * lane + 1]};
}
template <typename Out>
__device__ __forceinline__ void bt_tile_norm_p(const float (*S)[BT_LD], int nvalid, BtNormP np, int wave, int nwaves, int lane,
                                               Out out);
template <typename Out>
__device__ __forceinline__ void bt_tile_norm(const float (*S)[BT_LD], int nvalid, const float* __restrict__ alpha,
                                             const float* __restrict__ beta, int wave, int nwaves, int lane, Out out) {
    bt_tile_norm_p(S, nvalid, bt_norm_load(alpha, beta, lane), wave, nwaves, lane, out);
}
template <typename Out>
__device__ __forceinline__ void bt_tile_norm_p(const float (*S)[BT_LD], int nvalid, BtNormP np, int wave, int nwaves, int lane,
                                               Out out) {
    const float a0 = np.a0, a1 = np.a1, b0 = np.b0, b1 = np.b1;
    for (int r = wave; r < nvalid; r += nwaves) {
        const float2 v = *reinterpret_cast<const float2*>(&S[r][2 * lane]);
        const float s = gf_wave_sum(v.x + v.y);  // (the __shfl_xor butterfly without the LDS crossbar: common.h)
        const float mu = s / (float)BT_D;
        const float dx = v.x - mu, dy = v.y - mu;
        const float q = gf_wave_sum(dx * dx + dy * dy);
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
    // every phase's weights, biases and norm parameters requested now (bt_w_load: why)
    constexpr int KCI = 8;  // input widths up to 128 channels
    BtW<1, KCI> w_b;
    BtW<1, BT_D / 16> w_q, w_k, w_v;
    bt_w_load(w_b, P.bw, P.bb, BT_D, c, wave, nw, lane);
    const BtNormP np1 = bt_norm_load(P.L[0].n1a, P.L[0].n1b, lane);
    bt_w_load(w_q, P.L[0].qw, P.L[0].qb, BT_D, BT_D, wave, nw, lane);
    bt_w_load(w_k, P.L[0].kw, P.L[0].kb, BT_D, BT_D, wave, nw, lane);
    bt_w_load(w_v, P.L[0].vw, P.L[0].vb, BT_D, BT_D, wave, nw, lane);
    // (... and what the first product's epilogue reads: the positional layer's column of this lane, its rows' coordinates)
    const int pcol = wave * 16 + (lane & 15);  // the one column tile of this wave (BT_D / 16 = nw tiles)
    const float pw0 = P.pw[pcol * 3], pw1 = P.pw[pcol * 3 + 1], pw2 = P.pw[pcol * 3 + 2], pbc = P.pb[pcol];
    int tzr[4][3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int rr = min(t0 + 4 * (lane >> 4) + i, T - 1);
#pragma unroll
        for (int a = 0; a < 3; a++) tzr[i][a] = xyz[(size_t)rr * 4 + 1 + a];
    }
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
        a0 = gf_wave_sum_i(a0);
        a1 = gf_wave_sum_i(a1);
        a2 = gf_wave_sum_i(a2);
        if (lane == 0 && wave * 64 < T) {
            atomicAdd(&psum[0], a0);
            atomicAdd(&psum[1], a1);
            atomicAdd(&psum[2], a2);
        }
    }
    __syncthreads();
    const float ft = (float)T;
    static_assert(BT_D / 16 == BT_THREADS / 64, "k_bt_pre: one column tile of the first product per wave");
    if (c <= KCI * 16) {
        bt_tile_gemm_w<false>(feats + ((size_t)s0 + t0) * c, c, nvalid, c, BT_D, wave, nw, lane, w_b,
                              [&](int r, int col, float v) {
                                  const int i = r & 3;  // (r = 4 (lane >> 4) + i)
                                  const float r0 = (float)(T * tzr[i][0] - psum[0]) / ft;
                                  const float r1 = (float)(T * tzr[i][1] - psum[1]) / ft;
                                  const float r2 = (float)(T * tzr[i][2] - psum[2]) / ft;
                                  const float pe = fmaf(pw2, r2, fmaf(pw1, r1, pw0 * r0));
                                  sX[r][col] = v + (pe + pbc);
                              });
    } else {
        const int* tz = xyz + (size_t)t0 * 4;
        bt_tile_gemm<false>(feats + ((size_t)s0 + t0) * c, c, nvalid, c, P.bw, P.bb, BT_D, wave, nw, lane,
                            [&](int r, int col, float v) {
                                const float r0 = (float)(T * tz[r * 4 + 1] - psum[0]) / ft;
                                const float r1 = (float)(T * tz[r * 4 + 2] - psum[1]) / ft;
                                const float r2 = (float)(T * tz[r * 4 + 3] - psum[2]) / ft;
                                const float pe = fmaf(P.pw[col * 3 + 2], r2, fmaf(P.pw[col * 3 + 1], r1, P.pw[col * 3] * r0));
                                sX[r][col] = v + (pe + P.pb[col]);
                            });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS)
        X[(size_t)(t0 + (i >> 7)) * BT_D + (i & 127)] = sX[i >> 7][i & 127];
    bt_tile_norm_p(sX, nvalid, np1, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
        sT[r][c2] = v0;
        sT[r][c2 + 1] = v1;
    });
    __syncthreads();
    float* qkv_rows = QKV + (size_t)t0 * (3 * BT_D);
    bt_tile_gemm_w<false>(&sT[0][0], BT_LD, nvalid, BT_D, BT_D, wave, nw, lane, w_q,
                          [&](int r, int cc, float v) { qkv_rows[(size_t)r * (3 * BT_D) + cc] = v; });
    bt_tile_gemm_w<false>(&sT[0][0], BT_LD, nvalid, BT_D, BT_D, wave, nw, lane, w_k,
                          [&](int r, int cc, float v) { qkv_rows[(size_t)r * (3 * BT_D) + BT_D + cc] = v; });
    bt_tile_gemm_w<false>(&sT[0][0], BT_LD, nvalid, BT_D, BT_D, wave, nw, lane, w_v,
                          [&](int r, int cc, float v) { qkv_rows[(size_t)r * (3 * BT_D) + 2 * BT_D + cc] = v; });
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
    // the weights of the phases behind the self-attention, requested now; those of the last phase (next layer's q / k / v or
    // the output layer: 96 registers) once the attention's own registers are free
    BtW<1, BT_D / 16> w_o, w_f1;
    BtW<1, BT_FF / 16> w_f2;
    bt_w_load(w_o, L.ow, L.ob, BT_D, BT_D, wave, nw, lane);
    const BtNormP np2 = bt_norm_load(L.n2a, L.n2b, lane);
    bt_w_load(w_f1, L.f1w, L.f1b, BT_FF, BT_D, wave, nw, lane);
    bt_w_load(w_f2, L.f2w, L.f2b, BT_D, BT_FF, wave, nw, lane);
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
            mx = fmaxf(mx, gf_shfl_xor<16>(mx));
            mx = fmaxf(mx, gf_shfl_xor<32>(mx));
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
        l += gf_shfl_xor<16>(l);
        l += gf_shfl_xor<32>(l);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            sO[j][h * BT_DK + 4 * g + i] = o0[i] / l;
            sO[j][h * BT_DK + 16 + 4 * g + i] = o1[i] / l;
        }
    }
    // (the last phase's operands: requested here, four phases ahead of their use)
    const bool more = li + 1 < P.nl;
    BtW<1, BT_D / 16> w_q, w_k, w_v;
    const BtLayer& Ln = P.L[more ? li + 1 : li];
    const BtNormP np3 = more ? bt_norm_load(Ln.n1a, Ln.n1b, lane) : bt_norm_load(P.na, P.nb, lane);
    bt_w_load(w_q, more ? Ln.qw : P.aw, more ? Ln.qb : P.ab, more ? BT_D : c, BT_D, wave, nw, lane);
    if (more) {
        bt_w_load(w_k, Ln.kw, Ln.kb, BT_D, BT_D, wave, nw, lane);
        bt_w_load(w_v, Ln.vw, Ln.vb, BT_D, BT_D, wave, nw, lane);
    }
    __syncthreads();
    // x += out(O)
    const float* xg = X + (size_t)t0 * BT_D;
    bt_tile_gemm_w<false>(&sO[0][0], BT_LD, nvalid, BT_D, BT_D, wave, nw, lane, w_o,
                          [&](int r, int col, float v) { sX[r][col] = v; }, xg, BT_D);
    __syncthreads();
    bt_tile_norm_p(sX, nvalid, np2, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
        sT[r][c2] = v0;
        sT[r][c2 + 1] = v1;
    });
    __syncthreads();
    // x += ff2(relu(ff1(.)));  the hidden tile reuses sO
    bt_tile_gemm_w<true>(&sT[0][0], BT_LD, nvalid, BT_D, BT_FF, wave, nw, lane, w_f1,
                         [&](int r, int col, float v) { sO[r][col] = v; });
    __syncthreads();
    bt_tile_gemm_w<false>(&sO[0][0], BT_LD, nvalid, BT_FF, BT_D, wave, nw, lane, w_f2,
                          [&](int r, int col, float v) { sX[r][col] += v; });
    __syncthreads();
    if (more) {
        for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS)
            Xo[(size_t)(t0 + (i >> 7)) * BT_D + (i & 127)] = sX[i >> 7][i & 127];
        bt_tile_norm_p(sX, nvalid, np3, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
            sT[r][c2] = v0;
            sT[r][c2 + 1] = v1;
        });
        __syncthreads();
        float* qkv_rows = QKVo + (size_t)t0 * (3 * BT_D);
        bt_tile_gemm_w<false>(&sT[0][0], BT_LD, nvalid, BT_D, BT_D, wave, nw, lane, w_q,
                              [&](int r, int cc, float v) { qkv_rows[(size_t)r * (3 * BT_D) + cc] = v; });
        bt_tile_gemm_w<false>(&sT[0][0], BT_LD, nvalid, BT_D, BT_D, wave, nw, lane, w_k,
                              [&](int r, int cc, float v) { qkv_rows[(size_t)r * (3 * BT_D) + BT_D + cc] = v; });
        bt_tile_gemm_w<false>(&sT[0][0], BT_LD, nvalid, BT_D, BT_D, wave, nw, lane, w_v,
                              [&](int r, int cc, float v) { qkv_rows[(size_t)r * (3 * BT_D) + 2 * BT_D + cc] = v; });
    } else {
        bt_tile_norm_p(sX, nvalid, np3, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
            sT[r][c2] = v0;
            sT[r][c2 + 1] = v1;
        });
        __syncthreads();
        float* y = out + ((size_t)s0 + t0) * c;
        bt_tile_gemm_w<false>(&sT[0][0], BT_LD, nvalid, BT_D, c, wave, nw, lane, w_q,
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

// the tile tables alone (they need the scene offsets and nothing else): gf_unet_fwd builds them on its side stream with the
// rulebooks, so that two small launches per transformer level are not in front of the transformer on the main stream
int gf_backbone_transformer_tables(const int* scene_offsets, int n_scenes, int M, void* scratch, void* stream) {
    GF_CHECK_ARG(scene_offsets && scratch && n_scenes >= 1 && n_scenes <= 4096 && M >= 1, "gf_backbone_transformer_tables: bad arguments");
    float* sA = (float*)scratch;
    float* sB = sA + (size_t)M * BT_SCRATCH_PER_TOKEN;
    int* tile_scene = (int*)(sB + (size_t)M * BT_SCRATCH_PER_TOKEN);
    const int max_tiles = M / 16 + n_scenes;
    int* tile_first = tile_scene + (M / 16 + 4096 + 8);
    hipLaunchKernelGGL(k_bt_tiles, dim3(1), dim3(64), 0, (hipStream_t)stream, scene_offsets, n_scenes, max_tiles, tile_scene,
                       tile_first);
    GF_CHECK_LAUNCH("gf_backbone_transformer_tables");
    return GF_OK;
}

static int bt_forward(const float* feats, const int* coords, const int* scene_offsets, int n_scenes, int M, int c, int n_layers,
                      const float* const* params, void* scratch, float* out, void* stream, bool tables_ready);
extern "C" int gf_backbone_transformer(const float* feats, const int* coords, const int* scene_offsets, int n_scenes,
                                       int M, int c, int n_layers, const float* const* params, void* scratch,
                                       float* out, void* stream) {
    return bt_forward(feats, coords, scene_offsets, n_scenes, M, c, n_layers, params, scratch, out, stream, false);
}
int gf_backbone_transformer_prepared(const float* feats, const int* coords, const int* scene_offsets, int n_scenes, int M, int c,
                                     int n_layers, const float* const* params, void* scratch, float* out, void* stream) {
    return bt_forward(feats, coords, scene_offsets, n_scenes, M, c, n_layers, params, scratch, out, stream, true);
}
static int bt_forward(const float* feats, const int* coords, const int* scene_offsets, int n_scenes, int M, int c, int n_layers,
                      const float* const* params, void* scratch, float* out, void* stream, bool tables_ready) {
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
    if (!tables_ready)
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

// =====================================================================================================================
// Training: the same forward keeping what the backward needs, and the backward -- (n_layers + 2) + (2 n_layers + 2)
// launches for what the framework modules run as ~250 small ones per level and step (forward and backward of
// before_transformer_linear -> TransformerEncoder -> after_transformer_linear over a few hundred voxels per scene:
// the host's launch rate was all that took time, 3.3 ms per level of the batch-4 training step).
//
// Dropout (transformer.py:97,116,128,159-160: attention weights, hidden layer, the two residual branches; p = 0.1):
// the keep decision of an element is a hash of (seed, site, row, column) -- bt_keep below -- so the backward
// recomputes the masks instead of storing them, and a test can build the very same masks on the host.  The seed is
// drawn by the caller per call.  p = 0 (modules in eval mode with gradients enabled): every element is kept.
//
// Backward, per layer from the last: a token kernel (k_bt_bwd_tok: everything local to a token -- the residual /
// dropout / FFN / Norm / projection chain between two attentions, on 16-token tiles in LDS), an attention kernel
// (k_bt_bwd_attn: dq per query tile and dk, dv per key tile, probabilities recomputed from the saved log-sum-exp,
// sum_j dP_ij P_ij = dO_i . O_i also under dropout) and at the end ONE launch for every weight, bias and Norm gradient
// (k_bt_wgrad: dW = A^T B over all tokens on the fp32 matrix pipe, column sums; fixed summation order).
// =====================================================================================================================
typedef GfDrop BtDrop;
// site = 4 * layer + {0 attention weights, 1 attention branch, 2 hidden layer, 3 feed-forward branch}; row = token index
// in the batch; column = channel, or 4 * key + head
#define bt_keep gf_drop_keep

struct BtSave {  // what the forward keeps: [M, .] arrays indexed by the token's row in the batch
    float *X[BT_MAXL + 1], *QKV[BT_MAXL], *O[BT_MAXL], *XMID[BT_MAXL], *H[BT_MAXL], *LSE[BT_MAXL], *REL;
    int *offs, *tile_scene, *tile_first;
};
#define BT_REL_LD 16
#define BT_INTS_HEAD (4096 + 8)

static size_t bt_save_floats(int M, int nl) {
    const size_t m = (size_t)M;
    return m * BT_D * (nl + 1) + (size_t)nl * m * (3 * BT_D + BT_D + BT_D + BT_FF + BT_H) + m * BT_REL_LD;
}
static size_t bt_save_ints(int M) { return (size_t)(M / 16) + 3 * BT_INTS_HEAD; }

static BtSave bt_save_layout(void* save, int M, int nl) {
    BtSave S;
    float* p = (float*)save;
    const size_t m = (size_t)M;
    for (int l = 0; l <= BT_MAXL; l++) S.X[l] = nullptr;
    for (int l = 0; l <= nl; l++) { S.X[l] = p; p += m * BT_D; }
    for (int l = 0; l < BT_MAXL; l++) S.QKV[l] = S.O[l] = S.XMID[l] = S.H[l] = S.LSE[l] = nullptr;
    for (int l = 0; l < nl; l++) {
        S.QKV[l] = p; p += m * 3 * BT_D;
        S.O[l] = p; p += m * BT_D;
        S.XMID[l] = p; p += m * BT_D;
        S.H[l] = p; p += m * BT_FF;
        S.LSE[l] = p; p += m * BT_H;
    }
    S.REL = p; p += m * BT_REL_LD;
    int* q = (int*)p;
    S.offs = q; q += BT_INTS_HEAD;
    S.tile_first = q; q += BT_INTS_HEAD;
    S.tile_scene = q;
    return S;
}

// scene offsets from the sorted batch column of the coordinates, then the tile tables (one launch, no host round trip)
__global__ void k_bt_offsets_tiles(const int* __restrict__ coords, int M, int n_scenes, int max_tiles,
                                   int* __restrict__ offs, int* __restrict__ tile_scene, int* __restrict__ tile_first) {
    if (blockIdx.x != 0) return;
    for (int s = threadIdx.x; s <= n_scenes; s += blockDim.x) {
        int lo = 0, hi = M;  // first row whose scene id is >= s
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (coords[(size_t)mid * 4] < s) lo = mid + 1; else hi = mid;
        }
        offs[s] = s == n_scenes ? M : lo;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int t = 0;
    for (int s = 0; s < n_scenes; s++) {
        tile_first[s] = t;
        const int nt = (offs[s + 1] - offs[s] + 15) >> 4;
        for (int i = 0; i < nt && t < max_tiles; i++) tile_scene[t++] = s;
    }
    for (; t < max_tiles; t++) tile_scene[t] = -1;
}

// out[r][col] = sum_o A[r][o] W[o][col]: the product with a row-major nn.Linear weight [out, in] summed over its OUT
// index (gradient towards a layer's input).  A: LDS tile of 16 rows (rows that do not exist hold zeros), K = number of
// summed rows of W, N = columns produced (multiple of 16)
template <typename Epi>
__device__ __forceinline__ void bt_tile_gemm_t(const float* A, int lda, int K, const float* __restrict__ W, int ldw,
                                               int N, int wave, int nwaves, int lane, Epi epi) {
    const int j = lane & 15, g = lane >> 4;
    const int KC = K >> 4;
    for (int ct = wave; ct < (N >> 4); ct += nwaves) {
        const float* xa = A + (size_t)j * lda + 4 * g;
        const float* wb = W + (size_t)(4 * g) * ldw + ct * 16 + j;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int kc = 0; kc < KC; kc++) {
            const float4 a = *reinterpret_cast<const float4*>(xa + kc * 16);
            const float* w = wb + (size_t)kc * 16 * ldw;
            const float4 b = make_float4(w[0], w[ldw], w[2 * (size_t)ldw], w[3 * (size_t)ldw]);
            acc = mfma4(a, b, acc);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) epi(4 * g + i, ct * 16 + j, acc[i]);
    }
}

// backward of Norm over the rows of a tile: x, dy in LDS; out(r, c, dx_c, dx_c+1); the normalised rows go to XH
// (zeros for rows that do not exist) for the alpha gradient
template <typename Out>
__device__ __forceinline__ void bt_tile_norm_bwd(const float (*X)[BT_LD], const float (*G)[BT_LD], float (*XH)[BT_LD],
                                                 int nvalid, const float* __restrict__ alpha, int wave, int nwaves,
                                                 int lane, Out out) {
    const float a0 = alpha[2 * lane], a1 = alpha[2 * lane + 1];
    for (int r = wave; r < 16; r += nwaves) {
        if (r >= nvalid) {
            XH[r][2 * lane] = 0.f;
            XH[r][2 * lane + 1] = 0.f;
            continue;
        }
        const float2 v = *reinterpret_cast<const float2*>(&X[r][2 * lane]);
        const float s = gf_wave_sum(v.x + v.y);  // (the __shfl_xor butterfly without the LDS crossbar: common.h)
        const float mu = s / (float)BT_D;
        const float dx = v.x - mu, dy = v.y - mu;
        const float q = gf_wave_sum(dx * dx + dy * dy);
        const float sd = sqrtf(q / (float)(BT_D - 1));
        const float den = sd + 1e-6f;
        const float2 gy = *reinterpret_cast<const float2*>(&G[r][2 * lane]);
        const float g0 = gy.x * a0, g1 = gy.y * a1;
        float sg = g0 + g1, sgx = g0 * dx + g1 * dy;
        sg = gf_wave_sum(sg);
        sgx = gf_wave_sum(sgx);
        const float mg = sg / (float)BT_D;
        const float k = sd > 0.f ? sgx / (den * den * sd * (float)(BT_D - 1)) : 0.f;
        out(r, 2 * lane, (g0 - mg) / den - dx * k, (g1 - mg) / den - dy * k);
        XH[r][2 * lane] = dx / den;
        XH[r][2 * lane + 1] = dy / den;
    }
}

// per-tile partial sums of the Norm gradients: dalpha[c] = sum_r dy[r][c] xhat[r][c], dbeta[c] = sum_r dy[r][c]
__device__ __forceinline__ void bt_tile_norm_partials(const float (*G)[BT_LD], const float (*XH)[BT_LD], int nvalid,
                                                      float* __restrict__ dst) {
    if (threadIdx.x < 2 * BT_D) {
        const int c = threadIdx.x & (BT_D - 1);
        const bool is_alpha = threadIdx.x < BT_D;
        float s = 0.f;
        for (int r = 0; r < nvalid; r++) s += is_alpha ? G[r][c] * XH[r][c] : G[r][c];
        dst[threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(BT_THREADS) void k_bt_train_pre(const float* __restrict__ feats,
                                                             const int* __restrict__ coords, int c, BtParams P,
                                                             BtSave S) {
    __shared__ float sX[16][BT_LD], sT[16][BT_LD];
    __shared__ int psum[3];
    const int sc = S.tile_scene[blockIdx.x];
    if (sc < 0) return;
    const int s0 = S.offs[sc], T = S.offs[sc + 1] - s0;
    const int t0 = (blockIdx.x - S.tile_first[sc]) * 16;
    const int nvalid = min(16, T - t0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = BT_THREADS / 64;
    const int* xyz = coords + (size_t)s0 * 4;
    if (threadIdx.x < 3) psum[threadIdx.x] = 0;
    __syncthreads();
    {
        int a0 = 0, a1 = 0, a2 = 0;
        for (int t = threadIdx.x; t < T; t += BT_THREADS) {
            a0 += xyz[t * 4 + 1];
            a1 += xyz[t * 4 + 2];
            a2 += xyz[t * 4 + 3];
        }
        a0 = gf_wave_sum_i(a0);
        a1 = gf_wave_sum_i(a1);
        a2 = gf_wave_sum_i(a2);
        if (lane == 0 && wave * 64 < T) {
            atomicAdd(&psum[0], a0);
            atomicAdd(&psum[1], a1);
            atomicAdd(&psum[2], a2);
        }
    }
    __syncthreads();
    const float ft = (float)T;
    const int* tz = xyz + (size_t)t0 * 4;
    if (threadIdx.x < nvalid * BT_REL_LD) {
        const int r = threadIdx.x >> 4, a = threadIdx.x & 15;
        S.REL[((size_t)s0 + t0 + r) * BT_REL_LD + a] = a < 3 ? (float)(T * tz[r * 4 + 1 + a] - psum[a]) / ft : 0.f;
    }
    bt_tile_gemm<false>(feats + ((size_t)s0 + t0) * c, c, nvalid, c, P.bw, P.bb, BT_D, wave, nw, lane,
                        [&](int r, int col, float v) {
                            const float r0 = (float)(T * tz[r * 4 + 1] - psum[0]) / ft;
                            const float r1 = (float)(T * tz[r * 4 + 2] - psum[1]) / ft;
                            const float r2 = (float)(T * tz[r * 4 + 3] - psum[2]) / ft;
                            const float pe = fmaf(P.pw[col * 3 + 2], r2, fmaf(P.pw[col * 3 + 1], r1, P.pw[col * 3] * r0));
                            sX[r][col] = v + (pe + P.pb[col]);
                        });
    __syncthreads();
    float* X = S.X[0] + ((size_t)s0 + t0) * BT_D;
    for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS) X[i] = sX[i >> 7][i & 127];
    bt_tile_norm(sX, nvalid, P.L[0].n1a, P.L[0].n1b, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
        sT[r][c2] = v0;
        sT[r][c2 + 1] = v1;
    });
    __syncthreads();
    bt_tile_qkv(sT, nvalid, P.L[0], S.QKV[0] + ((size_t)s0 + t0) * (3 * BT_D), wave, nw, lane);
}

__global__ __launch_bounds__(BT_THREADS) void k_bt_train_layer(int c, int li, BtParams P, BtSave S, BtDrop dr,
                                                               float* __restrict__ out) {
    __shared__ float sX[16][BT_LD], sT[16][BT_LD], sO[16][BT_LD];
    const int sc = S.tile_scene[blockIdx.x];
    if (sc < 0) return;
    const int s0 = S.offs[sc], T = S.offs[sc + 1] - s0;
    const int t0 = (blockIdx.x - S.tile_first[sc]) * 16;
    const int nvalid = min(16, T - t0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = BT_THREADS / 64;
    const int j = lane & 15, g = lane >> 4;
    const float* QKV = S.QKV[li] + (size_t)s0 * (3 * BT_D);
    const BtLayer& L = P.L[li];
    const uint32_t site = 4u * li, grow = (uint32_t)(s0 + t0);
    if (wave < BT_H) {
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
        for (int kt = 0; kt < QT; kt++) {
            float4 a0 = z4, a1 = z4;
            float v0[4], v1[4];
            const int krow = kt * 16 + j;
            if (krow < T) {
                const float* kp = QKV + (size_t)krow * (3 * BT_D) + BT_D + h * BT_DK + 4 * g;
                a0 = *reinterpret_cast<const float4*>(kp);
                a1 = *reinterpret_cast<const float4*>(kp + 16);
            }
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
            float scv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) scv[i] = (kt * 16 + 4 * g + i) < T ? s[i] * scale : -INFINITY;
            float mx = fmaxf(fmaxf(scv[0], scv[1]), fmaxf(scv[2], scv[3]));
            mx = fmaxf(mx, gf_shfl_xor<16>(mx));
            mx = fmaxf(mx, gf_shfl_xor<32>(mx));
            const float mnew = fmaxf(m, mx);
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
                const float pd = p[i] * bt_keep(dr, site, grow + j, (uint32_t)(kt * 16 + 4 * g + i) * 4u + h);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[i], pd, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[i], pd, o1, 0, 0, 0);
            }
        }
        l += gf_shfl_xor<16>(l);
        l += gf_shfl_xor<32>(l);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            sO[j][h * BT_DK + 4 * g + i] = o0[i] / l;
            sO[j][h * BT_DK + 16 + 4 * g + i] = o1[i] / l;
        }
        if (g == 0 && qrow < T) S.LSE[li][((size_t)s0 + qrow) * BT_H + h] = m + logf(l);
    }
    __syncthreads();
    {
        float* Og = S.O[li] + ((size_t)s0 + t0) * BT_D;
        for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS) Og[i] = sO[i >> 7][i & 127];
    }
    // x += dropout(out(O))
    const float* xg = S.X[li] + ((size_t)s0 + t0) * BT_D;
    bt_tile_gemm<false>(&sO[0][0], BT_LD, nvalid, BT_D, L.ow, L.ob, BT_D, wave, nw, lane, [&](int r, int col, float v) {
        sX[r][col] = xg[r * BT_D + col] + v * bt_keep(dr, site + 1, grow + r, col);
    });
    __syncthreads();
    {
        float* Xm = S.XMID[li] + ((size_t)s0 + t0) * BT_D;
        for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS) Xm[i] = sX[i >> 7][i & 127];
    }
    bt_tile_norm(sX, nvalid, L.n2a, L.n2b, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
        sT[r][c2] = v0;
        sT[r][c2 + 1] = v1;
    });
    __syncthreads();
    // x += dropout(ff2(dropout(relu(ff1(.)))));  the hidden tile reuses sO; H keeps the hidden layer before its dropout
    float* Hg = S.H[li] + ((size_t)s0 + t0) * BT_FF;
    bt_tile_gemm<true>(&sT[0][0], BT_LD, nvalid, BT_D, L.f1w, L.f1b, BT_FF, wave, nw, lane,
                       [&](int r, int col, float v) {
                           Hg[r * BT_FF + col] = v;
                           sO[r][col] = v * bt_keep(dr, site + 2, grow + r, col);
                       });
    __syncthreads();
    bt_tile_gemm<false>(&sO[0][0], BT_LD, nvalid, BT_FF, L.f2w, L.f2b, BT_D, wave, nw, lane,
                        [&](int r, int col, float v) { sX[r][col] += v * bt_keep(dr, site + 3, grow + r, col); });
    __syncthreads();
    {
        float* Xo = S.X[li + 1] + ((size_t)s0 + t0) * BT_D;
        for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS) Xo[i] = sX[i >> 7][i & 127];
    }
    if (li + 1 < P.nl) {
        bt_tile_norm(sX, nvalid, P.L[li + 1].n1a, P.L[li + 1].n1b, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
            sT[r][c2] = v0;
            sT[r][c2 + 1] = v1;
        });
        __syncthreads();
        bt_tile_qkv(sT, nvalid, P.L[li + 1], S.QKV[li + 1] + ((size_t)s0 + t0) * (3 * BT_D), wave, nw, lane);
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

// ---- backward --------------------------------------------------------------------------------------------------------
struct BtWork {  // [M, .] arrays of the backward, per layer: operands of the weight gradients and what the kernels hand on
    float *DF[BT_MAXL], *HD[BT_MAXL], *DHDN[BT_MAXL], *X2B[BT_MAXL], *DA[BT_MAXL], *DO[BT_MAXL], *DD[BT_MAXL],
        *DQKV[BT_MAXL], *X2A[BT_MAXL], *PN[BT_MAXL];
    float *YN, *PNF, *DX0, *DXMID;
};
#define BT_QLD (3 * BT_D + 4)

static size_t bt_work_floats(int M, int nl, int max_tiles) {
    const size_t m = (size_t)M, t = (size_t)max_tiles;
    return (size_t)nl * (m * (BT_D + BT_FF + BT_FF + BT_D + BT_D + BT_D + BT_H + 3 * BT_D + BT_D) + t * 4 * BT_D) +
           m * BT_D * 3 + t * 2 * BT_D;
}
static BtWork bt_work_layout(void* ws, int M, int nl, int max_tiles) {
    BtWork W;
    float* p = (float*)ws;
    const size_t m = (size_t)M, t = (size_t)max_tiles;
    for (int l = 0; l < BT_MAXL; l++)
        W.DF[l] = W.HD[l] = W.DHDN[l] = W.X2B[l] = W.DA[l] = W.DO[l] = W.DD[l] = W.DQKV[l] = W.X2A[l] = W.PN[l] = nullptr;
    for (int l = 0; l < nl; l++) {
        W.DF[l] = p; p += m * BT_D;
        W.HD[l] = p; p += m * BT_FF;
        W.DHDN[l] = p; p += m * BT_FF;
        W.X2B[l] = p; p += m * BT_D;
        W.DA[l] = p; p += m * BT_D;
        W.DO[l] = p; p += m * BT_D;
        W.DD[l] = p; p += m * BT_H;
        W.DQKV[l] = p; p += m * 3 * BT_D;
        W.X2A[l] = p; p += m * BT_D;
        W.PN[l] = p; p += t * 4 * BT_D;
    }
    W.YN = p; p += m * BT_D;
    W.DX0 = p; p += m * BT_D;
    W.DXMID = p; p += m * BT_D;
    W.PNF = p;
    return W;
}

// stage n_layers: after-linear and final Norm backward, then the feed-forward / attention-output half of the last layer;
// stage l in 1 .. n_layers-1: the q/k/v projections and Norm 1 of layer l, then that half of layer l-1;
// stage 0: the q/k/v projections and Norm 1 of layer 0, then the position term and before-linear
__global__ __launch_bounds__(BT_THREADS) void k_bt_bwd_tok(const float* __restrict__ dY, int c, int stage, BtParams P,
                                                           BtSave S, BtWork Wk, BtDrop dr, float* __restrict__ dfeats) {
    __shared__ float sD[16][BT_LD], sA[16][BT_LD], sB[16][BT_LD], sC[16][BT_LD];
    __shared__ float sQ[16][BT_QLD];
    const int sc = S.tile_scene[blockIdx.x];
    const int nl = P.nl;
    if (sc < 0) {  // a tile that does not exist: its partial Norm sums must read as zero
        if (threadIdx.x < 2 * BT_D) {
            if (stage == nl) Wk.PNF[(size_t)blockIdx.x * 2 * BT_D + threadIdx.x] = 0.f;
            if (stage < nl) Wk.PN[stage][((size_t)blockIdx.x * 4 + 2) * BT_D + threadIdx.x] = 0.f;
            if (stage > 0) Wk.PN[stage - 1][(size_t)blockIdx.x * 4 * BT_D + threadIdx.x] = 0.f;
        }
        return;
    }
    const int s0 = S.offs[sc], T = S.offs[sc + 1] - s0;
    const int t0 = (blockIdx.x - S.tile_first[sc]) * 16;
    const int nvalid = min(16, T - t0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = BT_THREADS / 64;
    const size_t row0 = (size_t)s0 + t0;
    auto load_tile = [&](float (*dst)[BT_LD], const float* src) {
        for (int i = threadIdx.x; i < 16 * BT_D; i += BT_THREADS) {
            const int r = i >> 7, col = i & 127;
            dst[r][col] = r < nvalid ? src[(row0 + r) * BT_D + col] : 0.f;
        }
    };
    if (stage == nl) {
        for (int i = threadIdx.x; i < 16 * c; i += BT_THREADS) {
            const int r = i / c, col = i - r * c;
            sQ[r][col] = r < nvalid ? dY[(row0 + r) * c + col] : 0.f;
        }
        for (int i = threadIdx.x; i < 16 * BT_D; i += BT_THREADS) sD[i >> 7][i & 127] = 0.f;
        load_tile(sA, S.X[nl]);
        __syncthreads();
        bt_tile_norm(sA, nvalid, P.na, P.nb, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
            *reinterpret_cast<float2*>(&Wk.YN[(row0 + r) * BT_D + c2]) = make_float2(v0, v1);
        });
        bt_tile_gemm_t(&sQ[0][0], BT_QLD, c, P.aw, BT_D, BT_D, wave, nw, lane, [&](int r, int col, float v) { sC[r][col] = v; });
        __syncthreads();
        bt_tile_norm_bwd(sA, sC, sB, nvalid, P.na, wave, nw, lane, [&](int r, int c2, float d0, float d1) {
            sD[r][c2] = d0;
            sD[r][c2 + 1] = d1;
        });
        __syncthreads();
        bt_tile_norm_partials(sC, sB, nvalid, Wk.PNF + (size_t)blockIdx.x * 2 * BT_D);
        __syncthreads();
    } else {
        const int l = stage;
        const BtLayer& L = P.L[l];
        for (int i = threadIdx.x; i < 16 * 3 * BT_D; i += BT_THREADS) {
            const int r = i / (3 * BT_D), col = i - r * (3 * BT_D);
            sQ[r][col] = r < nvalid ? Wk.DQKV[l][(row0 + r) * (3 * BT_D) + col] : 0.f;
        }
        load_tile(sA, S.X[l]);
        load_tile(sD, Wk.DXMID);
        __syncthreads();
        bt_tile_norm(sA, nvalid, L.n1a, L.n1b, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
            *reinterpret_cast<float2*>(&Wk.X2A[l][(row0 + r) * BT_D + c2]) = make_float2(v0, v1);
        });
        bt_tile_gemm_t(&sQ[0][0], BT_QLD, BT_D, L.qw, BT_D, BT_D, wave, nw, lane, [&](int r, int col, float v) { sC[r][col] = v; });
        bt_tile_gemm_t(&sQ[0][BT_D], BT_QLD, BT_D, L.kw, BT_D, BT_D, wave, nw, lane, [&](int r, int col, float v) { sC[r][col] += v; });
        bt_tile_gemm_t(&sQ[0][2 * BT_D], BT_QLD, BT_D, L.vw, BT_D, BT_D, wave, nw, lane, [&](int r, int col, float v) { sC[r][col] += v; });
        __syncthreads();
        bt_tile_norm_bwd(sA, sC, sB, nvalid, L.n1a, wave, nw, lane, [&](int r, int c2, float d0, float d1) {
            sD[r][c2] += d0;
            sD[r][c2 + 1] += d1;
        });
        __syncthreads();
        bt_tile_norm_partials(sC, sB, nvalid, Wk.PN[l] + ((size_t)blockIdx.x * 4 + 2) * BT_D);
        __syncthreads();
    }
    if (stage > 0) {
        const int l = stage - 1;
        const BtLayer& L = P.L[l];
        const uint32_t site = 4u * l, grow = (uint32_t)row0;
        // feed-forward branch: x_out = x_mid + keep3 * ff2(keep2 * relu(ff1(Norm2(x_mid))))
        for (int i = threadIdx.x; i < 16 * BT_D; i += BT_THREADS) {
            const int r = i >> 7, col = i & 127;
            const float v = r < nvalid ? sD[r][col] * bt_keep(dr, site + 3, grow + r, col) : 0.f;
            sA[r][col] = v;
            if (r < nvalid) Wk.DF[l][(row0 + r) * BT_D + col] = v;
        }
        __syncthreads();
        bt_tile_gemm_t(&sA[0][0], BT_LD, BT_D, L.f2w, BT_FF, BT_FF, wave, nw, lane, [&](int r, int col, float v) { sB[r][col] = v; });
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * BT_FF; i += BT_THREADS) {
            const int r = i >> 6, col = i & 63;
            float d = 0.f;
            if (r < nvalid) {
                const float h = S.H[l][(row0 + r) * BT_FF + col];
                const float k = bt_keep(dr, site + 2, grow + r, col);
                Wk.HD[l][(row0 + r) * BT_FF + col] = h * k;
                d = h > 0.f ? sB[r][col] * k : 0.f;
                Wk.DHDN[l][(row0 + r) * BT_FF + col] = d;
            }
            sB[r][col] = d;
        }
        load_tile(sA, S.XMID[l]);
        __syncthreads();
        bt_tile_gemm_t(&sB[0][0], BT_LD, BT_FF, L.f1w, BT_D, BT_D, wave, nw, lane, [&](int r, int col, float v) { sC[r][col] = v; });
        bt_tile_norm(sA, nvalid, L.n2a, L.n2b, wave, nw, lane, [&](int r, int c2, float v0, float v1) {
            *reinterpret_cast<float2*>(&Wk.X2B[l][(row0 + r) * BT_D + c2]) = make_float2(v0, v1);
        });
        __syncthreads();
        bt_tile_norm_bwd(sA, sC, sB, nvalid, L.n2a, wave, nw, lane, [&](int r, int c2, float d0, float d1) {
            sD[r][c2] += d0;
            sD[r][c2 + 1] += d1;
        });
        __syncthreads();
        bt_tile_norm_partials(sC, sB, nvalid, Wk.PN[l] + (size_t)blockIdx.x * 4 * BT_D);
        // attention branch: x_mid = x + keep1 * out(O)
        for (int i = threadIdx.x; i < 16 * BT_D; i += BT_THREADS) {
            const int r = i >> 7, col = i & 127;
            const float d = sD[r][col];
            const float v = r < nvalid ? d * bt_keep(dr, site + 1, grow + r, col) : 0.f;
            sA[r][col] = v;
            if (r < nvalid) {
                Wk.DXMID[(row0 + r) * BT_D + col] = d;
                Wk.DA[l][(row0 + r) * BT_D + col] = v;
            }
        }
        __syncthreads();
        bt_tile_gemm_t(&sA[0][0], BT_LD, BT_D, L.ow, BT_D, BT_D, wave, nw, lane, [&](int r, int col, float v) { sC[r][col] = v; });
        __syncthreads();
        for (int r = wave; r < nvalid; r += nw) {
            const float2 d = *reinterpret_cast<const float2*>(&sC[r][2 * lane]);
            const float2 o = *reinterpret_cast<const float2*>(&S.O[l][(row0 + r) * BT_D + 2 * lane]);
            *reinterpret_cast<float2*>(&Wk.DO[l][(row0 + r) * BT_D + 2 * lane]) = d;
            float pr = d.x * o.x + d.y * o.y;
#pragma unroll
            for (int q = 8; q >= 1; q >>= 1) pr += __shfl_xor(pr, q, 64);
            if ((lane & 15) == 0) Wk.DD[l][(row0 + r) * BT_H + (lane >> 4)] = pr;
        }
    } else {
        for (int i = threadIdx.x; i < nvalid * BT_D; i += BT_THREADS) Wk.DX0[row0 * BT_D + i] = sD[i >> 7][i & 127];
        bt_tile_gemm_t(&sD[0][0], BT_LD, BT_D, P.bw, c, c, wave, nw, lane, [&](int r, int col, float v) {
            if (r < nvalid) dfeats[(row0 + r) * c + col] = v;
        });
    }
}

// dq for the tile's tokens as queries (waves 0..3, one head each), dk and dv for them as keys (waves 4..7)
__global__ __launch_bounds__(BT_THREADS) void k_bt_bwd_attn(int li, BtSave S, BtWork Wk, BtDrop dr) {
    const int sc = S.tile_scene[blockIdx.x];
    if (sc < 0) return;
    const int s0 = S.offs[sc], T = S.offs[sc + 1] - s0;
    const int t0 = (blockIdx.x - S.tile_first[sc]) * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int h = wave & 3;
    const float* QKV = S.QKV[li] + (size_t)s0 * (3 * BT_D);
    const float* dO = Wk.DO[li] + (size_t)s0 * BT_D;
    const float* LSE = S.LSE[li] + (size_t)s0 * BT_H;
    const float* DD = Wk.DD[li] + (size_t)s0 * BT_H;
    float* dQKV = Wk.DQKV[li] + (size_t)s0 * (3 * BT_D);
    const int NT = (T + 15) >> 4;
    const float scale = 0.17677669529663687f;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t site = 4u * li;
    const int own = t0 + j;  // this lane's column: a query (waves 0..3) or a key (waves 4..7)
    auto row4 = [&](const float* base, int ld, int row, int off, float4& lo, float4& hi) {
        lo = z4;
        hi = z4;
        if (row < T) {
            const float* p = base + (size_t)row * ld + off + h * BT_DK + 4 * g;
            lo = *reinterpret_cast<const float4*>(p);
            hi = *reinterpret_cast<const float4*>(p + 16);
        }
    };
    if (wave < BT_H) {
        float4 bq0, bq1, bd0, bd1;
        row4(QKV, 3 * BT_D, own, 0, bq0, bq1);
        row4(dO, BT_D, own, 0, bd0, bd1);
        const float lse_q = own < T ? LSE[(size_t)own * BT_H + h] : 0.f;
        const float dd_q = own < T ? DD[(size_t)own * BT_H + h] : 0.f;
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < NT; kt++) {
            float4 a0, a1, va0, va1;
            row4(QKV, 3 * BT_D, kt * 16 + j, BT_D, a0, a1);
            row4(QKV, 3 * BT_D, kt * 16 + j, 2 * BT_D, va0, va1);
            float k0[4], k1[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kt * 16 + 4 * g + i;
                const float* kp = QKV + (size_t)key * (3 * BT_D) + BT_D + h * BT_DK + j;
                k0[i] = key < T ? kp[0] : 0.f;
                k1[i] = key < T ? kp[16] : 0.f;
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            s = mfma4(a0, bq0, s);
            s = mfma4(a1, bq1, s);
            dp = mfma4(va0, bd0, dp);
            dp = mfma4(va1, bd1, dp);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kt * 16 + 4 * g + i;
                const float p = key < T ? expf(s[i] * scale - lse_q) : 0.f;
                const float keep = bt_keep(dr, site, (uint32_t)(s0 + own), (uint32_t)key * 4u + h);
                const float ds = p * (dp[i] * keep - dd_q);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[i], ds, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[i], ds, o1, 0, 0, 0);
            }
        }
        if (own < T) {
            float* dq = dQKV + (size_t)own * (3 * BT_D) + h * BT_DK + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                dq[i] = o0[i] * scale;
                dq[16 + i] = o1[i] * scale;
            }
        }
    } else {
        float4 bk0, bk1, bv0, bv1;
        row4(QKV, 3 * BT_D, own, BT_D, bk0, bk1);
        row4(QKV, 3 * BT_D, own, 2 * BT_D, bv0, bv1);
        f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = dk0, dv0 = dk0, dv1 = dk0;
        for (int qt = 0; qt < NT; qt++) {
            float4 aq0, aq1, ad0, ad1;
            row4(QKV, 3 * BT_D, qt * 16 + j, 0, aq0, aq1);
            row4(dO, BT_D, qt * 16 + j, 0, ad0, ad1);
            float q0[4], q1[4], e0[4], e1[4], lse[4], dd[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qr = qt * 16 + 4 * g + i;
                const bool ok = qr < T;
                const float* qp = QKV + (size_t)qr * (3 * BT_D) + h * BT_DK + j;
                const float* ep = dO + (size_t)qr * BT_D + h * BT_DK + j;
                q0[i] = ok ? qp[0] : 0.f;
                q1[i] = ok ? qp[16] : 0.f;
                e0[i] = ok ? ep[0] : 0.f;
                e1[i] = ok ? ep[16] : 0.f;
                lse[i] = ok ? LSE[(size_t)qr * BT_H + h] : 0.f;
                dd[i] = ok ? DD[(size_t)qr * BT_H + h] : 0.f;
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            s = mfma4(aq0, bk0, s);
            s = mfma4(aq1, bk1, s);
            dp = mfma4(ad0, bv0, dp);
            dp = mfma4(ad1, bv1, dp);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qr = qt * 16 + 4 * g + i;
                const float p = qr < T ? expf(s[i] * scale - lse[i]) : 0.f;
                const float keep = bt_keep(dr, site, (uint32_t)(s0 + qr), (uint32_t)own * 4u + h);
                const float pd = p * keep;
                const float ds = p * (dp[i] * keep - dd[i]);
                dv0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[i], pd, dv0, 0, 0, 0);
                dv1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[i], pd, dv1, 0, 0, 0);
                dk0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q0[i], ds, dk0, 0, 0, 0);
                dk1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q1[i], ds, dk1, 0, 0, 0);
            }
        }
        if (own < T) {
            float* dk = dQKV + (size_t)own * (3 * BT_D) + BT_D + h * BT_DK + 4 * g;
            float* dv = dk + BT_D;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                dk[i] = dk0[i] * scale;
                dk[16 + i] = dk1[i] * scale;
                dv[i] = dv0[i];
                dv[16 + i] = dv1[i];
            }
        }
    }
}

// every weight / bias / Norm gradient of a call in one launch (train_common.h)
__global__ __launch_bounds__(256) void k_bt_wgrad(GfWJobs J, int M) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x;
    if (b < J.gstart[J.ng]) {
        int k = 0;
        while (b >= J.gstart[k + 1]) k++;
        const GfGemmJob& G = J.g[k];
        const int tile = b - J.gstart[k], nti = (G.I + 15) >> 4;
        const int to = tile / nti, ti = tile - to * nti;
        const int j = lane & 15, g = lane >> 4;
        const float* ap = G.A + to * 16 + j;
        const float* bp = G.B + ti * 16 + j;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int tb = w * 4; tb < M; tb += 16) {  // uniform trip count: every lane takes part in the matrix instruction
            const int t = tb + g;
            const float a = t < M ? ap[(size_t)t * G.lda] : 0.f;
            const float bb = t < M ? bp[(size_t)t * G.ldb] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc, 0, 0, 0);
        }
        *reinterpret_cast<f32x4*>(&red[w][lane * 4]) = acc;
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float v = ((red[0][lane * 4 + i] + red[1][lane * 4 + i]) + red[2][lane * 4 + i]) + red[3][lane * 4 + i];
                const int o = to * 16 + 4 * g + i, ii = ti * 16 + j;
                if (o < G.O && ii < G.ivalid) G.dst[(size_t)o * G.ldd + ii] = v;
            }
        }
    } else {
        const int cb = b - J.gstart[J.ng];
        int k = 0;
        while (cb >= J.cstart[k + 1]) k++;
        const GfColJob& C = J.c[k];
        const int col = (cb - J.cstart[k]) * 64 + lane;
        float s = 0.f;
        if (col < C.n)
            for (int r = w; r < C.rows; r += 4) s += C.src[(size_t)r * C.ld + col];
        red[w][lane] = s;
        __syncthreads();
        if (w == 0 && col < C.n) C.dst[col] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    }
}

static int bt_unpack(const float* const* params, int n_layers, BtParams& P, const char* who) {
    GF_CHECK_ARG(params != nullptr, "%s: params is null", who);
    const int np = 8 + 16 * n_layers;
    for (int i = 0; i < np; i++) GF_CHECK_ARG(params[i] != nullptr, "%s: params[%d] is null", who, i);
    int k = 0;
    P.bw = params[k++];
    P.bb = params[k++];
    P.pw = params[k++];
    P.pb = params[k++];
    for (int l = 0; l < n_layers; l++) {
        const float** f = reinterpret_cast<const float**>(&P.L[l]);
        for (int i = 0; i < 16; i++) f[i] = params[k++];
    }
    for (int l = n_layers; l < BT_MAXL; l++) P.L[l] = P.L[0];
    P.na = params[k++];
    P.nb = params[k++];
    P.aw = params[k++];
    P.ab = params[k++];
    P.nl = n_layers;
    return GF_OK;
}

static int bt_train_check(const char* who, int n_scenes, int M, int c, int n_layers, float p) {
    GF_CHECK_ARG(c > 0 && c % 16 == 0 && c <= 3 * BT_D, "%s: channel width %d must be a multiple of 16, at most %d", who,
                 c, 3 * BT_D);
    GF_CHECK_ARG(n_layers >= 1 && n_layers <= BT_MAXL, "%s: 1..%d layers, got %d", who, BT_MAXL, n_layers);
    GF_CHECK_ARG(n_scenes >= 1 && n_scenes <= 4096 && M >= 1 && M < (1 << 26), "%s: bad sizes", who);
    GF_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability %g", who, (double)p);
    return GF_OK;
}

extern "C" size_t gf_backbone_transformer_train_save_bytes(int M, int n_layers) {
    return bt_save_floats(M > 0 ? M : 0, n_layers) * sizeof(float) + bt_save_ints(M > 0 ? M : 0) * sizeof(int);
}
extern "C" size_t gf_backbone_transformer_train_work_bytes(int M, int n_layers, int n_scenes) {
    const int m = M > 0 ? M : 0;
    return bt_work_floats(m, n_layers, m / 16 + n_scenes) * sizeof(float);
}
extern "C" long long gf_backbone_transformer_grad_floats(int c, int n_layers) {
    return (long long)BT_D * c + BT_D + 3 * BT_D + BT_D +
           (long long)n_layers * (2 * BT_D + 4 * (BT_D * BT_D + BT_D) + 2 * BT_D + BT_FF * BT_D + BT_FF + BT_D * BT_FF + BT_D) +
           2 * BT_D + (long long)c * BT_D + c;
}

extern "C" int gf_backbone_transformer_train_fwd(const float* feats, const int* coords, int n_scenes, int M, int c,
                                                 int n_layers, const float* const* params, float p, unsigned seed,
                                                 void* save, float* out, void* stream) {
    int rc = bt_train_check("gf_backbone_transformer_train_fwd", n_scenes, M, c, n_layers, p);
    if (rc != GF_OK) return rc;
    BtParams P;
    rc = bt_unpack(params, n_layers, P, "gf_backbone_transformer_train_fwd");
    if (rc != GF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const BtSave S = bt_save_layout(save, M, n_layers);
    const BtDrop dr = gf_drop_make(p, seed);
    const int max_tiles = M / 16 + n_scenes;
    hipLaunchKernelGGL(k_bt_offsets_tiles, dim3(1), dim3(256), 0, st, coords, M, n_scenes, max_tiles, S.offs, S.tile_scene,
                       S.tile_first);
    hipLaunchKernelGGL(k_bt_train_pre, dim3(max_tiles), dim3(BT_THREADS), 0, st, feats, coords, c, P, S);
    for (int l = 0; l < n_layers; l++)
        hipLaunchKernelGGL(k_bt_train_layer, dim3(max_tiles), dim3(BT_THREADS), 0, st, c, l, P, S, dr, out);
    GF_CHECK_LAUNCH("gf_backbone_transformer_train_fwd");
    return GF_OK;
}

extern "C" int gf_backbone_transformer_train_bwd(const float* feats, const float* dout, int n_scenes, int M, int c,
                                                 int n_layers, const float* const* params, float p, unsigned seed,
                                                 void* save, void* work, float* dfeats, float* grads, void* stream) {
    int rc = bt_train_check("gf_backbone_transformer_train_bwd", n_scenes, M, c, n_layers, p);
    if (rc != GF_OK) return rc;
    BtParams P;
    rc = bt_unpack(params, n_layers, P, "gf_backbone_transformer_train_bwd");
    if (rc != GF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const BtSave S = bt_save_layout(save, M, n_layers);
    const int max_tiles = M / 16 + n_scenes;
    const BtWork W = bt_work_layout(work, M, n_layers, max_tiles);
    const BtDrop dr = gf_drop_make(p, seed);
    for (int stage = n_layers; stage >= 0; stage--) {
        hipLaunchKernelGGL(k_bt_bwd_tok, dim3(max_tiles), dim3(BT_THREADS), 0, st, dout, c, stage, P, S, W, dr, dfeats);
        if (stage > 0) hipLaunchKernelGGL(k_bt_bwd_attn, dim3(max_tiles), dim3(BT_THREADS), 0, st, stage - 1, S, W, dr);
    }
    // gradients in the order of the parameter table
    GfWJobBuilder jb;
    auto gemm = [&](const float* A, int lda, const float* B, int ldb, int O, int I, int ivalid, float* dst, int ldd) {
        jb.gemm(A, lda, B, ldb, O, I, ivalid, dst, ldd);
    };
    auto cols = [&](const float* src, int ld, int n, int rows, float* dst) { jb.cols(src, ld, n, rows, dst); };
    float* g = grads;
    gemm(W.DX0, BT_D, feats, c, BT_D, c, c, g, c); g += (size_t)BT_D * c;                       // before.W
    cols(W.DX0, BT_D, BT_D, M, g); g += BT_D;                                                   // before.b
    gemm(W.DX0, BT_D, S.REL, BT_REL_LD, BT_D, BT_REL_LD, 3, g, 3); g += 3 * BT_D;               // position.W
    cols(W.DX0, BT_D, BT_D, M, g); g += BT_D;                                                   // position.b
    for (int l = 0; l < n_layers; l++) {
        cols(W.PN[l] + 2 * BT_D, 4 * BT_D, BT_D, max_tiles, g); g += BT_D;                      // norm1.alpha
        cols(W.PN[l] + 3 * BT_D, 4 * BT_D, BT_D, max_tiles, g); g += BT_D;                      // norm1.bias
        for (int k = 0; k < 3; k++) {                                                           // q, k, v
            gemm(W.DQKV[l] + k * BT_D, 3 * BT_D, W.X2A[l], BT_D, BT_D, BT_D, BT_D, g, BT_D); g += BT_D * BT_D;
            cols(W.DQKV[l] + k * BT_D, 3 * BT_D, BT_D, M, g); g += BT_D;
        }
        gemm(W.DA[l], BT_D, S.O[l], BT_D, BT_D, BT_D, BT_D, g, BT_D); g += BT_D * BT_D;         // out.W
        cols(W.DA[l], BT_D, BT_D, M, g); g += BT_D;
        cols(W.PN[l], 4 * BT_D, BT_D, max_tiles, g); g += BT_D;                                 // norm2.alpha
        cols(W.PN[l] + BT_D, 4 * BT_D, BT_D, max_tiles, g); g += BT_D;                          // norm2.bias
        gemm(W.DHDN[l], BT_FF, W.X2B[l], BT_D, BT_FF, BT_D, BT_D, g, BT_D); g += BT_FF * BT_D;  // ff1.W [64,128]
        cols(W.DHDN[l], BT_FF, BT_FF, M, g); g += BT_FF;
        gemm(W.DF[l], BT_D, W.HD[l], BT_FF, BT_D, BT_FF, BT_FF, g, BT_FF); g += BT_D * BT_FF;   // ff2.W [128,64]
        cols(W.DF[l], BT_D, BT_D, M, g); g += BT_D;
    }
    cols(W.PNF, 2 * BT_D, BT_D, max_tiles, g); g += BT_D;                                       // norm.alpha
    cols(W.PNF + BT_D, 2 * BT_D, BT_D, max_tiles, g); g += BT_D;                                // norm.bias
    gemm(dout, c, W.YN, BT_D, c, BT_D, BT_D, g, BT_D); g += (size_t)c * BT_D;                   // after.W [c,128]
    cols(dout, c, c, M, g); g += c;
    GF_CHECK_LAUNCH("gf_backbone_transformer_train_bwd");
    return jb.launch(M, st);
}

int gf_wgrad_launch(const GfWJobs& J, int M, hipStream_t st) {
    const int blocks = J.gstart[J.ng] + J.cstart[J.nc];
    if (blocks > 0) hipLaunchKernelGGL(k_bt_wgrad, dim3(blocks), dim3(256), 0, st, J, M);
    GF_CHECK_LAUNCH("gf_wgrad_launch");
    return GF_OK;
}
