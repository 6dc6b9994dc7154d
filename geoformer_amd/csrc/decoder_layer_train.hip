// Token-side stages of the DETR-style decoder, training: forward with the layer's dropouts and a native backward.
//
// Around the pairwise cross-attention (decoder_attn.hip, which has its own fused backward) a decoder layer
// (TransformerDecoderLayer.forward_pre_rel, model/transformer_detr.py:425-463) is ~35 framework launches forward and
// ~60 backward over B x nq x 64 tokens -- LayerNorms, the nn.MultiheadAttention self-attention, dropouts, residual
// adds, out_mlp, the FFN: 9 ms of host time per batch-4 training step for 3.5 ms of device work.  Here a layer is
//
//   pre  (x, query_pos) -> (t2n, q1):  t2 = norm1(x); q = k = t2 + query_pos; x1 = x + drop1(out_proj(MHA(q, k, t2)));
//                                      t2n = norm2(x1);  q1 = W1 t2n + b1   (query half of attn_mlp[0])
//   [cross-attention: q1, K1, Kv -> ca]
//   post (ca, t2n) -> (x3, inter):     x2 = relu(out_mlp(ca)) + drop2(t2n);
//                                      x3 = x2 + drop3(linear2(drop(relu(linear1(norm3(x2))))));  inter = norm(x3)
//
// forward: 2 + 1 launches, backward: 4 + 2 (token kernels on 16-token tiles in LDS, the self-attention backward with
// probabilities recomputed from the saved log-sum-exp, one launch for all weight / bias / LayerNorm gradients).  All
// tensors are [B, T, 64] row-major (row = b * T + t).  Dropout masks: gf_drop_keep (train_common.h) with
// site = 8 * layer + {0 attention weights, 1 drop1, 2 drop2, 3 hidden layer, 4 drop3}, row = the token's row,
// column = channel, or 4 * key + head for the attention weights.  Products on v_mfma_f32_16x16x4_f32.
#include "decoder_layer.h"
#include "train_common.h"

#define DT_THREADS 256
#define DT_QLD (3 * DL_D + 4)

struct DtPre {
    const float *n1w, *n1b, *ipw, *ipb, *opw, *opb, *n2w, *n2b, *w1w, *w1b;
};
struct DtPost {
    const float *omw, *omb, *n3w, *n3b, *l1w, *l1b, *l2w, *l2b, *fnw, *fnb;
};

// out[r][col] = sum_o A[r][o] W[o][col]: product with a row-major nn.Linear weight [out,in] summed over its OUT index
template <typename Epi>
__device__ __forceinline__ void dl_tile_gemm_t(const float* A, int lda, int K, const float* __restrict__ W, int ldw, int N,
                                               int wave, int nwaves, int lane, Epi epi) {
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
            acc = dl_mfma4(a, b, acc);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) epi(4 * g + i, ct * 16 + j, acc[i]);
    }
}

// backward of torch.nn.LayerNorm over the rows of a tile (x, dy in LDS): out(r, c, dx); the normalised rows go to XH
template <typename Out>
__device__ __forceinline__ void dl_tile_layernorm_bwd(const float (*X)[DL_LD], const float (*G)[DL_LD], float (*XH)[DL_LD],
                                                      int nvalid, const float* __restrict__ w, int wave, int nwaves,
                                                      int lane, Out out) {
    const float wl = w[lane];
    for (int r = wave; r < 16; r += nwaves) {
        if (r >= nvalid) {
            XH[r][lane] = 0.f;
            continue;
        }
        const float v = X[r][lane];
        const float s = gf_wave_sum(v);  // (the __shfl_xor butterflies without the LDS crossbar: common.h)
        const float dv = v - s / (float)DL_D;
        const float q = gf_wave_sum(dv * dv);
        const float rstd = 1.0f / sqrtf(q / (float)DL_D + 1e-5f);
        const float xh = dv * rstd;
        const float g = G[r][lane] * wl;
        float sg = g, sgx = g * xh;
        sg = gf_wave_sum(sg);
        sgx = gf_wave_sum(sgx);
        out(r, lane, rstd * (g - sg / (float)DL_D - xh * (sgx / (float)DL_D)));
        XH[r][lane] = xh;
    }
}

// per-tile partial sums of a LayerNorm's gradients: dst[0..63] = sum_r dy xhat, dst[64..127] = sum_r dy
__device__ __forceinline__ void dl_tile_ln_partials(const float (*G)[DL_LD], const float (*XH)[DL_LD], int nvalid,
                                                    float* __restrict__ dst) {
    if (threadIdx.x < 2 * DL_D) {
        const int c = threadIdx.x & (DL_D - 1);
        const bool is_w = threadIdx.x < DL_D;
        float s = 0.f;
        for (int r = 0; r < nvalid; r++) s += is_w ? G[r][c] * XH[r][c] : G[r][c];
        dst[threadIdx.x] = s;
    }
}

#define DT_TILE_SETUP                                                                   \
    const int b = blockIdx.y, t0 = blockIdx.x * 16;                                    \
    const int nvalid = min(16, T - t0);                                                \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = DT_THREADS / 64;  \
    const size_t row0 = (size_t)b * T + t0;                                            \
    const uint32_t grow = (uint32_t)row0;                                              \
    (void)lane; (void)wave; (void)nw; (void)grow;

__device__ __forceinline__ void dt_load_tile(float (*dst)[DL_LD], const float* src, size_t row0, int nvalid) {
    for (int i = threadIdx.x; i < 16 * DL_D; i += DT_THREADS) {
        const int r = i >> 6, c = i & 63;
        dst[r][c] = (src != nullptr && r < nvalid) ? src[(row0 + r) * DL_D + c] : 0.f;
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------
// pre, part a: t2 = norm1(x); q, k from t2 + query_pos, v from t2
__global__ __launch_bounds__(DT_THREADS) void k_dt_pre_a(const float* __restrict__ x, const float* __restrict__ qpos, int T,
                                                         DtPre pr, float* __restrict__ QKV) {
    __shared__ float sX[16][DL_LD], sT[16][DL_LD], sQ[16][DL_LD];
    DT_TILE_SETUP
    dt_load_tile(sX, x, row0, nvalid);
    __syncthreads();
    const float* qp = qpos + row0 * DL_D;
    dl_tile_layernorm(sX, nvalid, pr.n1w, pr.n1b, wave, nw, lane, [&](int r, int c, float v) {
        sT[r][c] = v;
        sQ[r][c] = v + qp[(size_t)r * DL_D + c];
    });
    __syncthreads();
    float* qkv = QKV + row0 * (3 * DL_D);
    dl_tile_gemm<false>(&sQ[0][0], DL_LD, nvalid, DL_D, pr.ipw, pr.ipb, 2 * DL_D, wave, nw, lane,
                        [&](int r, int c, float v) { qkv[(size_t)r * (3 * DL_D) + c] = v; });
    dl_tile_gemm<false>(&sT[0][0], DL_LD, nvalid, DL_D, pr.ipw + 2 * DL_D * DL_D, pr.ipb + 2 * DL_D, DL_D, wave, nw, lane,
                        [&](int r, int c, float v) { qkv[(size_t)r * (3 * DL_D) + 2 * DL_D + c] = v; });
}

// pre, part b: self-attention (a wave per head) with dropout on the weights, out_proj, residual, norm2, q1
__global__ __launch_bounds__(DT_THREADS) void k_dt_pre_b(const float* __restrict__ x, int T, DtPre pr, GfDrop dr, int layer,
                                                         const float* __restrict__ QKVall, float* __restrict__ Oall,
                                                         float* __restrict__ X1, float* __restrict__ LSE,
                                                         float* __restrict__ t2n, float* __restrict__ q1) {
    __shared__ float sO[16][DL_LD], sX[16][DL_LD], sT[16][DL_LD];
    DT_TILE_SETUP
    const int j = lane & 15, g = lane >> 4;
    const float* QKV = QKVall + (size_t)b * T * (3 * DL_D);
    const uint32_t site = 8u * layer;
    {
        const int h = wave;
        const int QT = (T + 15) >> 4;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int qrow = t0 + j;
        float4 bq = z4;
        if (qrow < T) bq = *reinterpret_cast<const float4*>(QKV + (size_t)qrow * (3 * DL_D) + h * DL_DK + 4 * g);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, l = 0.f;
        for (int kt = 0; kt < QT; kt++) {
            float4 ak = z4;
            float v[4];
            const int krow = kt * 16 + j;
            if (krow < T) ak = *reinterpret_cast<const float4*>(QKV + (size_t)krow * (3 * DL_D) + DL_D + h * DL_DK + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kt * 16 + 4 * g + i;
                v[i] = key < T ? QKV[(size_t)key * (3 * DL_D) + 2 * DL_D + h * DL_DK + j] : 0.f;
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            s = dl_mfma4(ak, bq, s);
            float sc[4];
#pragma unroll
            for (int i = 0; i < 4; i++) sc[i] = (kt * 16 + 4 * g + i) < T ? s[i] * 0.25f : -INFINITY;  // 1/sqrt(16)
            float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            mx = fmaxf(mx, gf_shfl_xor<16>(mx));
            mx = fmaxf(mx, gf_shfl_xor<32>(mx));
            const float mnew = fmaxf(m, mx);
            const float corr = expf(m - mnew);
            float p[4];
#pragma unroll
            for (int i = 0; i < 4; i++) p[i] = expf(sc[i] - mnew);
            l = l * corr + ((p[0] + p[1]) + (p[2] + p[3]));
            o *= corr;
            m = mnew;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float pd = p[i] * gf_drop_keep(dr, site, grow + j, (uint32_t)(kt * 16 + 4 * g + i) * 4u + h);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i], pd, o, 0, 0, 0);
            }
        }
        l += gf_shfl_xor<16>(l);
        l += gf_shfl_xor<32>(l);
#pragma unroll
        for (int i = 0; i < 4; i++) sO[j][h * DL_DK + 4 * g + i] = o[i] / l;
        if (g == 0 && qrow < T) LSE[(row0 + j) * DL_H + h] = m + logf(l);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nvalid * DL_D; i += DT_THREADS) Oall[row0 * DL_D + i] = sO[i >> 6][i & 63];
    const float* xg = x + row0 * DL_D;
    dl_tile_gemm<false>(&sO[0][0], DL_LD, nvalid, DL_D, pr.opw, pr.opb, DL_D, wave, nw, lane, [&](int r, int c, float v) {
        sX[r][c] = xg[r * DL_D + c] + v * gf_drop_keep(dr, site + 1, grow + r, c);
    });
    __syncthreads();
    for (int i = threadIdx.x; i < nvalid * DL_D; i += DT_THREADS) X1[row0 * DL_D + i] = sX[i >> 6][i & 63];
    float* tg = t2n + row0 * DL_D;
    dl_tile_layernorm(sX, nvalid, pr.n2w, pr.n2b, wave, nw, lane, [&](int r, int c, float v) {
        sT[r][c] = v;
        tg[r * DL_D + c] = v;
    });
    __syncthreads();
    float* qo = q1 + row0 * DL_D;
    dl_tile_gemm<false>(&sT[0][0], DL_LD, nvalid, DL_D, pr.w1w, pr.w1b, DL_D, wave, nw, lane,
                        [&](int r, int c, float v) { qo[r * DL_D + c] = v; });
}

__global__ __launch_bounds__(DT_THREADS) void k_dt_post(const float* __restrict__ ca, const float* __restrict__ t2n, int T,
                                                        int ff, DtPost po, GfDrop dr, int layer, float* __restrict__ Y,
                                                        float* __restrict__ X2, float* __restrict__ H,
                                                        float* __restrict__ x3, float* __restrict__ inter) {
    __shared__ float sX[16][DL_LD], sT[16][DL_LD], sH[16][DL_LDH];
    DT_TILE_SETUP
    const uint32_t site = 8u * layer;
    const float* tg = t2n + row0 * DL_D;
    float* yg = Y + row0 * DL_D;
    dl_tile_gemm<true>(ca + row0 * DL_D, DL_D, nvalid, DL_D, po.omw, po.omb, DL_D, wave, nw, lane, [&](int r, int c, float v) {
        yg[r * DL_D + c] = v;
        sX[r][c] = v + tg[r * DL_D + c] * gf_drop_keep(dr, site + 2, grow + r, c);
    });
    __syncthreads();
    for (int i = threadIdx.x; i < nvalid * DL_D; i += DT_THREADS) X2[row0 * DL_D + i] = sX[i >> 6][i & 63];
    dl_tile_layernorm(sX, nvalid, po.n3w, po.n3b, wave, nw, lane, [&](int r, int c, float v) { sT[r][c] = v; });
    __syncthreads();
    float* hg = H + row0 * ff;
    dl_tile_gemm<true>(&sT[0][0], DL_LD, nvalid, DL_D, po.l1w, po.l1b, ff, wave, nw, lane, [&](int r, int c, float v) {
        hg[(size_t)r * ff + c] = v;
        sH[r][c] = v * gf_drop_keep(dr, site + 3, grow + r, c);
    });
    __syncthreads();
    dl_tile_gemm<false>(&sH[0][0], DL_LDH, nvalid, ff, po.l2w, po.l2b, DL_D, wave, nw, lane,
                        [&](int r, int c, float v) { sX[r][c] += v * gf_drop_keep(dr, site + 4, grow + r, c); });
    __syncthreads();
    for (int i = threadIdx.x; i < nvalid * DL_D; i += DT_THREADS) x3[row0 * DL_D + i] = sX[i >> 6][i & 63];
    float* io = inter + row0 * DL_D;
    dl_tile_layernorm(sX, nvalid, po.fnw, po.fnb, wave, nw, lane, [&](int r, int c, float v) { io[r * DL_D + c] = v; });
}

// ---- backward --------------------------------------------------------------------------------------------------------
struct DtPostWork {  // [B*T, .] operands of the weight gradients; PN: per-tile LayerNorm partials [tiles][2][128]
    float *DF, *HD, *DH, *T3, *DY, *PN;
};

__global__ __launch_bounds__(DT_THREADS) void k_dt_post_bwd(const float* __restrict__ t2n_unused, const float* __restrict__ x3,
                                                            const float* __restrict__ d_x3,
                                                            const float* __restrict__ d_inter, int T, int ff, DtPost po,
                                                            GfDrop dr, int layer, const float* __restrict__ Y,
                                                            const float* __restrict__ X2, const float* __restrict__ H,
                                                            DtPostWork W, float* __restrict__ d_ca,
                                                            float* __restrict__ d_t2n) {
    __shared__ float sD[16][DL_LD], sA[16][DL_LD], sB[16][DL_LD], sC[16][DL_LD], sH[16][DL_LDH];
    DT_TILE_SETUP
    const uint32_t site = 8u * layer;
    const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    // inter = norm(x3): dx3 = d_x3 + LayerNorm backward of d_inter
    dt_load_tile(sA, x3, row0, nvalid);
    dt_load_tile(sC, d_inter, row0, nvalid);
    dt_load_tile(sD, d_x3, row0, nvalid);
    __syncthreads();
    dl_tile_layernorm_bwd(sA, sC, sB, nvalid, po.fnw, wave, nw, lane, [&](int r, int c, float d) { sD[r][c] += d; });
    __syncthreads();
    dl_tile_ln_partials(sC, sB, nvalid, W.PN + (tile * 2 + 0) * (2 * DL_D));
    // x3 = x2 + keep4 * linear2(keep3 * relu(linear1(norm3(x2))))
    for (int i = threadIdx.x; i < 16 * DL_D; i += DT_THREADS) {
        const int r = i >> 6, c = i & 63;
        const float v = r < nvalid ? sD[r][c] * gf_drop_keep(dr, site + 4, grow + r, c) : 0.f;
        sA[r][c] = v;
        if (r < nvalid) W.DF[(row0 + r) * DL_D + c] = v;
    }
    __syncthreads();
    dl_tile_gemm_t(&sA[0][0], DL_LD, DL_D, po.l2w, ff, ff, wave, nw, lane, [&](int r, int c, float v) { sH[r][c] = v; });
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * ff; i += DT_THREADS) {
        const int r = i / ff, c = i - r * ff;
        float d = 0.f;
        if (r < nvalid) {
            const float h = H[(row0 + r) * ff + c];
            const float k = gf_drop_keep(dr, site + 3, grow + r, c);
            W.HD[(row0 + r) * ff + c] = h * k;
            d = h > 0.f ? sH[r][c] * k : 0.f;
            W.DH[(row0 + r) * ff + c] = d;
        }
        sH[r][c] = d;
    }
    dt_load_tile(sA, X2, row0, nvalid);
    __syncthreads();
    dl_tile_gemm_t(&sH[0][0], DL_LDH, ff, po.l1w, DL_D, DL_D, wave, nw, lane, [&](int r, int c, float v) { sC[r][c] = v; });
    dl_tile_layernorm(sA, nvalid, po.n3w, po.n3b, wave, nw, lane,
                      [&](int r, int c, float v) { W.T3[(row0 + r) * DL_D + c] = v; });
    __syncthreads();
    dl_tile_layernorm_bwd(sA, sC, sB, nvalid, po.n3w, wave, nw, lane, [&](int r, int c, float d) { sD[r][c] += d; });
    __syncthreads();
    dl_tile_ln_partials(sC, sB, nvalid, W.PN + (tile * 2 + 1) * (2 * DL_D));
    // x2 = relu(out_mlp(ca)) + keep2 * t2n
    for (int i = threadIdx.x; i < 16 * DL_D; i += DT_THREADS) {
        const int r = i >> 6, c = i & 63;
        float dy = 0.f;
        if (r < nvalid) {
            const float d = sD[r][c];
            d_t2n[(row0 + r) * DL_D + c] = d * gf_drop_keep(dr, site + 2, grow + r, c);
            dy = Y[(row0 + r) * DL_D + c] > 0.f ? d : 0.f;
            W.DY[(row0 + r) * DL_D + c] = dy;
        }
        sA[r][c] = dy;
    }
    __syncthreads();
    dl_tile_gemm_t(&sA[0][0], DL_LD, DL_D, po.omw, DL_D, DL_D, wave, nw, lane, [&](int r, int c, float v) {
        if (r < nvalid) d_ca[(row0 + r) * DL_D + c] = v;
    });
}

struct DtPreWork {
    float *DX1, *DSA, *DO, *DD, *DQKV, *T2, *QKIN, *PN;  // PN: [tiles][2][128] (norm2, norm1)
};

// norm2 / W1 / out_proj backward up to dO and the per-head sums dO . O
__global__ __launch_bounds__(DT_THREADS) void k_dt_pre_bwd1(const float* __restrict__ d_t2n, const float* __restrict__ d_q1,
                                                            int T, DtPre pr, GfDrop dr, int layer,
                                                            const float* __restrict__ X1, const float* __restrict__ O,
                                                            DtPreWork W) {
    __shared__ float sD[16][DL_LD], sA[16][DL_LD], sB[16][DL_LD], sC[16][DL_LD];
    DT_TILE_SETUP
    const uint32_t site = 8u * layer;
    const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    dt_load_tile(sA, d_q1, row0, nvalid);
    dt_load_tile(sC, d_t2n, row0, nvalid);
    dt_load_tile(sD, X1, row0, nvalid);
    __syncthreads();
    dl_tile_gemm_t(&sA[0][0], DL_LD, DL_D, pr.w1w, DL_D, DL_D, wave, nw, lane, [&](int r, int c, float v) { sC[r][c] += v; });
    __syncthreads();
    dl_tile_layernorm_bwd(sD, sC, sB, nvalid, pr.n2w, wave, nw, lane, [&](int r, int c, float d) { sA[r][c] = d; });
    __syncthreads();
    dl_tile_ln_partials(sC, sB, nvalid, W.PN + (tile * 2 + 0) * (2 * DL_D));
    for (int i = threadIdx.x; i < 16 * DL_D; i += DT_THREADS) {
        const int r = i >> 6, c = i & 63;
        float v = 0.f;
        if (r < nvalid) {
            const float d = sA[r][c];
            W.DX1[(row0 + r) * DL_D + c] = d;
            v = d * gf_drop_keep(dr, site + 1, grow + r, c);
            W.DSA[(row0 + r) * DL_D + c] = v;
        }
        sD[r][c] = v;
    }
    __syncthreads();
    dl_tile_gemm_t(&sD[0][0], DL_LD, DL_D, pr.opw, DL_D, DL_D, wave, nw, lane, [&](int r, int c, float v) { sC[r][c] = v; });
    __syncthreads();
    for (int r = wave; r < nvalid; r += nw) {
        const float d = sC[r][lane];
        W.DO[(row0 + r) * DL_D + lane] = d;
        float pr2 = d * O[(row0 + r) * DL_D + lane];
#pragma unroll
        for (int q = 8; q >= 1; q >>= 1) pr2 += __shfl_xor(pr2, q, 64);
        if ((lane & 15) == 0) W.DD[(row0 + r) * DL_H + (lane >> 4)] = pr2;
    }
}

// self-attention backward: dq for the tile's tokens as queries (waves 0..3, one head each), dk and dv as keys (4..7)
__global__ __launch_bounds__(512) void k_dt_attn_bwd(int T, GfDrop dr, int layer, const float* __restrict__ QKVall,
                                                     const float* __restrict__ LSEall, DtPreWork W) {
    const int b = blockIdx.y, t0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4, h = wave & 3;
    const size_t base = (size_t)b * T;
    const float* QKV = QKVall + base * (3 * DL_D);
    const float* dO = W.DO + base * DL_D;
    const float* LSE = LSEall + base * DL_H;
    const float* DD = W.DD + base * DL_H;
    float* dQKV = W.DQKV + base * (3 * DL_D);
    const int NT = (T + 15) >> 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t site = 8u * layer;
    const int own = t0 + j;
    auto row4 = [&](const float* p, int ld, int row, int off) {
        return row < T ? *reinterpret_cast<const float4*>(p + (size_t)row * ld + off + h * DL_DK + 4 * g) : z4;
    };
    if (wave < DL_H) {
        const float4 bq = row4(QKV, 3 * DL_D, own, 0), bd = row4(dO, DL_D, own, 0);
        const float lse_q = own < T ? LSE[(size_t)own * DL_H + h] : 0.f;
        const float dd_q = own < T ? DD[(size_t)own * DL_H + h] : 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < NT; kt++) {
            const float4 ak = row4(QKV, 3 * DL_D, kt * 16 + j, DL_D), av = row4(QKV, 3 * DL_D, kt * 16 + j, 2 * DL_D);
            float kk[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kt * 16 + 4 * g + i;
                kk[i] = key < T ? QKV[(size_t)key * (3 * DL_D) + DL_D + h * DL_DK + j] : 0.f;
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            s = dl_mfma4(ak, bq, s);
            dp = dl_mfma4(av, bd, dp);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kt * 16 + 4 * g + i;
                const float p = key < T ? expf(s[i] * 0.25f - lse_q) : 0.f;
                const float keep = gf_drop_keep(dr, site, (uint32_t)(base + own), (uint32_t)key * 4u + h);
                const float ds = p * (dp[i] * keep - dd_q);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[i], ds, o, 0, 0, 0);
            }
        }
        if (own < T) {
            float* dq = dQKV + (size_t)own * (3 * DL_D) + h * DL_DK + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; i++) dq[i] = o[i] * 0.25f;
        }
    } else {
        const float4 bk = row4(QKV, 3 * DL_D, own, DL_D), bv = row4(QKV, 3 * DL_D, own, 2 * DL_D);
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
        for (int qt = 0; qt < NT; qt++) {
            const float4 aq = row4(QKV, 3 * DL_D, qt * 16 + j, 0), ad = row4(dO, DL_D, qt * 16 + j, 0);
            float qq[4], ee[4], lse[4], dd[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qr = qt * 16 + 4 * g + i;
                const bool ok = qr < T;
                qq[i] = ok ? QKV[(size_t)qr * (3 * DL_D) + h * DL_DK + j] : 0.f;
                ee[i] = ok ? dO[(size_t)qr * DL_D + h * DL_DK + j] : 0.f;
                lse[i] = ok ? LSE[(size_t)qr * DL_H + h] : 0.f;
                dd[i] = ok ? DD[(size_t)qr * DL_H + h] : 0.f;
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            s = dl_mfma4(aq, bk, s);
            dp = dl_mfma4(ad, bv, dp);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qr = qt * 16 + 4 * g + i;
                const float p = qr < T ? expf(s[i] * 0.25f - lse[i]) : 0.f;
                const float keep = gf_drop_keep(dr, site, (uint32_t)(base + qr), (uint32_t)own * 4u + h);
                dv = __builtin_amdgcn_mfma_f32_16x16x4f32(ee[i], p * keep, dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[i], p * (dp[i] * keep - dd[i]), dk, 0, 0, 0);
            }
        }
        if (own < T) {
            float* pk = dQKV + (size_t)own * (3 * DL_D) + DL_D + h * DL_DK + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                pk[i] = dk[i] * 0.25f;
                pk[DL_D + i] = dv[i];
            }
        }
    }
}

// in_proj and norm1 backward: dx = dx1 + LayerNorm backward of (dq Wq + dk Wk + dv Wv), d query_pos = dq Wq + dk Wk
__global__ __launch_bounds__(DT_THREADS) void k_dt_pre_bwd2(const float* __restrict__ x, const float* __restrict__ qpos, int T,
                                                            DtPre pr, DtPreWork W, float* __restrict__ dx,
                                                            float* __restrict__ dqpos) {
    __shared__ float sD[16][DL_LD], sA[16][DL_LD], sB[16][DL_LD], sC[16][DL_LD];
    __shared__ float sQ[16][DT_QLD];
    DT_TILE_SETUP
    const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 16 * 3 * DL_D; i += DT_THREADS) {
        const int r = i / (3 * DL_D), c = i - r * (3 * DL_D);
        sQ[r][c] = r < nvalid ? W.DQKV[(row0 + r) * (3 * DL_D) + c] : 0.f;
    }
    dt_load_tile(sA, x, row0, nvalid);
    dt_load_tile(sD, W.DX1, row0, nvalid);
    __syncthreads();
    dl_tile_gemm_t(&sQ[0][0], DT_QLD, DL_D, pr.ipw, DL_D, DL_D, wave, nw, lane, [&](int r, int c, float v) { sC[r][c] = v; });
    dl_tile_gemm_t(&sQ[0][DL_D], DT_QLD, DL_D, pr.ipw + DL_D * DL_D, DL_D, DL_D, wave, nw, lane, [&](int r, int c, float v) {
        const float d = sC[r][c] + v;
        sC[r][c] = d;
        if (r < nvalid) dqpos[(row0 + r) * DL_D + c] = d;
    });
    dl_tile_gemm_t(&sQ[0][2 * DL_D], DT_QLD, DL_D, pr.ipw + 2 * DL_D * DL_D, DL_D, DL_D, wave, nw, lane,
                   [&](int r, int c, float v) { sC[r][c] += v; });
    const float* qp = qpos + row0 * DL_D;
    dl_tile_layernorm(sA, nvalid, pr.n1w, pr.n1b, wave, nw, lane, [&](int r, int c, float v) {
        W.T2[(row0 + r) * DL_D + c] = v;
        W.QKIN[(row0 + r) * DL_D + c] = v + qp[(size_t)r * DL_D + c];
    });
    __syncthreads();
    dl_tile_layernorm_bwd(sA, sC, sB, nvalid, pr.n1w, wave, nw, lane, [&](int r, int c, float d) { sD[r][c] += d; });
    __syncthreads();
    dl_tile_ln_partials(sC, sB, nvalid, W.PN + (tile * 2 + 1) * (2 * DL_D));
    for (int i = threadIdx.x; i < nvalid * DL_D; i += DT_THREADS) dx[row0 * DL_D + i] = sD[i >> 6][i & 63];
}

// ---- host side -------------------------------------------------------------------------------------------------------
static int dt_check(const char* who, int T, int B, int ff, float p, int layer) {
    GF_CHECK_ARG(T >= 1 && B >= 1 && (long long)T * B < (1 << 26), "%s: bad sizes", who);
    GF_CHECK_ARG(ff > 0 && ff % 16 == 0 && ff <= DL_MAXFF, "%s: dim_feedforward %d not in 16..%d", who, ff, DL_MAXFF);
    GF_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability %g", who, (double)p);
    GF_CHECK_ARG(layer >= 0 && layer < 8, "%s: layer index %d not in 0..7", who, layer);
    return GF_OK;
}
template <typename S>
static int dt_unpack(const float* const* params, S& s, const char* who) {
    GF_CHECK_ARG(params != nullptr, "%s: params is null", who);
    const float** f = reinterpret_cast<const float**>(&s);
    for (int i = 0; i < 10; i++) {
        GF_CHECK_ARG(params[i] != nullptr, "%s: params[%d] is null", who, i);
        f[i] = params[i];
    }
    return GF_OK;
}

extern "C" size_t gf_decoder_pre_train_save_bytes(int T, int B) {
    return (size_t)T * B * (3 * DL_D + DL_D + DL_D + DL_H) * sizeof(float);  // QKV, O, X1, LSE
}
extern "C" size_t gf_decoder_pre_train_work_bytes(int T, int B) {
    const size_t tiles = (size_t)((T + 15) / 16) * B;
    return ((size_t)T * B * (DL_D * 3 + DL_H + 3 * DL_D + 2 * DL_D) + tiles * 2 * 2 * DL_D) * sizeof(float);
}
extern "C" long long gf_decoder_pre_grad_floats() {
    return 2 * DL_D + 3 * DL_D * DL_D + 3 * DL_D + DL_D * DL_D + DL_D + 2 * DL_D + DL_D * DL_D + DL_D;
}
extern "C" size_t gf_decoder_post_train_save_bytes(int T, int B, int ff) {
    return (size_t)T * B * (2 * DL_D + ff) * sizeof(float);  // Y, X2, H
}
extern "C" size_t gf_decoder_post_train_work_bytes(int T, int B, int ff) {
    const size_t tiles = (size_t)((T + 15) / 16) * B;
    return ((size_t)T * B * (DL_D + ff + ff + DL_D + DL_D) + tiles * 2 * 2 * DL_D) * sizeof(float);
}
extern "C" long long gf_decoder_post_grad_floats(int ff) {
    return DL_D * DL_D + DL_D + 2 * DL_D + (long long)ff * DL_D + ff + (long long)DL_D * ff + DL_D + 2 * DL_D;
}

extern "C" int gf_decoder_pre_train_fwd(const float* x, const float* qpos, int T, int B, const float* const* params, float p,
                                        unsigned seed, int layer, void* save, float* t2n, float* q1, void* stream) {
    int rc = dt_check("gf_decoder_pre_train_fwd", T, B, 16, p, layer);
    if (rc != GF_OK) return rc;
    DtPre pr;
    rc = dt_unpack(params, pr, "gf_decoder_pre_train_fwd");
    if (rc != GF_OK) return rc;
    const size_t n = (size_t)T * B;
    float* QKV = (float*)save;
    float* O = QKV + n * 3 * DL_D;
    float* X1 = O + n * DL_D;
    float* LSE = X1 + n * DL_D;
    const dim3 grid((T + 15) / 16, B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dt_pre_a, grid, dim3(DT_THREADS), 0, st, x, qpos, T, pr, QKV);
    hipLaunchKernelGGL(k_dt_pre_b, grid, dim3(DT_THREADS), 0, st, x, T, pr, gf_drop_make(p, seed), layer, QKV, O, X1, LSE, t2n,
                       q1);
    GF_CHECK_LAUNCH("gf_decoder_pre_train_fwd");
    return GF_OK;
}

extern "C" int gf_decoder_pre_train_bwd(const float* x, const float* qpos, const float* t2n, const float* d_t2n,
                                        const float* d_q1, int T, int B, const float* const* params, float p, unsigned seed,
                                        int layer, void* save, void* work, float* dx, float* dqpos, float* grads,
                                        void* stream) {
    int rc = dt_check("gf_decoder_pre_train_bwd", T, B, 16, p, layer);
    if (rc != GF_OK) return rc;
    DtPre pr;
    rc = dt_unpack(params, pr, "gf_decoder_pre_train_bwd");
    if (rc != GF_OK) return rc;
    GF_CHECK_ARG(d_t2n || d_q1, "gf_decoder_pre_train_bwd: no incoming gradient");
    const size_t n = (size_t)T * B;
    const int M = (int)n, tiles = ((T + 15) / 16) * B;
    float* QKV = (float*)save;
    float* O = QKV + n * 3 * DL_D;
    float* X1 = O + n * DL_D;
    float* LSE = X1 + n * DL_D;
    DtPreWork W;
    float* w = (float*)work;
    W.DX1 = w; w += n * DL_D;
    W.DSA = w; w += n * DL_D;
    W.DO = w; w += n * DL_D;
    W.DD = w; w += n * DL_H;
    W.DQKV = w; w += n * 3 * DL_D;
    W.T2 = w; w += n * DL_D;
    W.QKIN = w; w += n * DL_D;
    W.PN = w;
    const dim3 grid((T + 15) / 16, B);
    hipStream_t st = (hipStream_t)stream;
    const GfDrop dr = gf_drop_make(p, seed);
    hipLaunchKernelGGL(k_dt_pre_bwd1, grid, dim3(DT_THREADS), 0, st, d_t2n, d_q1, T, pr, dr, layer, X1, O, W);
    hipLaunchKernelGGL(k_dt_attn_bwd, grid, dim3(512), 0, st, T, dr, layer, QKV, LSE, W);
    hipLaunchKernelGGL(k_dt_pre_bwd2, grid, dim3(DT_THREADS), 0, st, x, qpos, T, pr, W, dx, dqpos);
    GF_CHECK_LAUNCH("gf_decoder_pre_train_bwd");
    GfWJobBuilder jb;
    float* g = grads;
    jb.cols(W.PN + 2 * DL_D, 4 * DL_D, DL_D, tiles, g); g += DL_D;                       // norm1.weight
    jb.cols(W.PN + 3 * DL_D, 4 * DL_D, DL_D, tiles, g); g += DL_D;                       // norm1.bias
    jb.gemm(W.DQKV, 3 * DL_D, W.QKIN, DL_D, 2 * DL_D, DL_D, DL_D, g, DL_D);              // in_proj rows q, k
    jb.gemm(W.DQKV + 2 * DL_D, 3 * DL_D, W.T2, DL_D, DL_D, DL_D, DL_D, g + 2 * DL_D * DL_D, DL_D);  // rows v
    g += 3 * DL_D * DL_D;
    jb.cols(W.DQKV, 3 * DL_D, 3 * DL_D, M, g); g += 3 * DL_D;                            // in_proj bias
    jb.gemm(W.DSA, DL_D, O, DL_D, DL_D, DL_D, DL_D, g, DL_D); g += DL_D * DL_D;          // out_proj.weight
    jb.cols(W.DSA, DL_D, DL_D, M, g); g += DL_D;
    jb.cols(W.PN, 4 * DL_D, DL_D, tiles, g); g += DL_D;                                  // norm2.weight
    jb.cols(W.PN + DL_D, 4 * DL_D, DL_D, tiles, g); g += DL_D;                           // norm2.bias
    if (d_q1) {
        jb.gemm(d_q1, DL_D, t2n, DL_D, DL_D, DL_D, DL_D, g, DL_D);                       // attn_mlp[0].weight (query half)
        jb.cols(d_q1, DL_D, DL_D, M, g + DL_D * DL_D);
    } else {
        GF_TRY(hipMemsetAsync(g, 0, (DL_D * DL_D + DL_D) * sizeof(float), st));
    }
    return jb.launch(M, st);
}

extern "C" int gf_decoder_post_train_fwd(const float* ca, const float* t2n, int T, int B, int ff,
                                         const float* const* params, float p, unsigned seed, int layer, void* save,
                                         float* x3, float* inter, void* stream) {
    int rc = dt_check("gf_decoder_post_train_fwd", T, B, ff, p, layer);
    if (rc != GF_OK) return rc;
    DtPost po;
    rc = dt_unpack(params, po, "gf_decoder_post_train_fwd");
    if (rc != GF_OK) return rc;
    const size_t n = (size_t)T * B;
    float* Y = (float*)save;
    float* X2 = Y + n * DL_D;
    float* H = X2 + n * DL_D;
    hipLaunchKernelGGL(k_dt_post, dim3((T + 15) / 16, B), dim3(DT_THREADS), 0, (hipStream_t)stream, ca, t2n, T, ff, po,
                       gf_drop_make(p, seed), layer, Y, X2, H, x3, inter);
    GF_CHECK_LAUNCH("gf_decoder_post_train_fwd");
    return GF_OK;
}

extern "C" int gf_decoder_post_train_bwd(const float* ca, const float* x3, const float* d_x3, const float* d_inter, int T,
                                         int B, int ff, const float* const* params, float p, unsigned seed, int layer,
                                         void* save, void* work, float* d_ca, float* d_t2n, float* grads, void* stream) {
    int rc = dt_check("gf_decoder_post_train_bwd", T, B, ff, p, layer);
    if (rc != GF_OK) return rc;
    DtPost po;
    rc = dt_unpack(params, po, "gf_decoder_post_train_bwd");
    if (rc != GF_OK) return rc;
    GF_CHECK_ARG(d_x3 || d_inter, "gf_decoder_post_train_bwd: no incoming gradient");
    const size_t n = (size_t)T * B;
    const int M = (int)n, tiles = ((T + 15) / 16) * B;
    float* Y = (float*)save;
    float* X2 = Y + n * DL_D;
    float* H = X2 + n * DL_D;
    DtPostWork W;
    float* w = (float*)work;
    W.DF = w; w += n * DL_D;
    W.HD = w; w += n * ff;
    W.DH = w; w += n * ff;
    W.T3 = w; w += n * DL_D;
    W.DY = w; w += n * DL_D;
    W.PN = w;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dt_post_bwd, dim3((T + 15) / 16, B), dim3(DT_THREADS), 0, st, (const float*)nullptr, x3, d_x3,
                       d_inter, T, ff, po, gf_drop_make(p, seed), layer, Y, X2, H, W, d_ca, d_t2n);
    GF_CHECK_LAUNCH("gf_decoder_post_train_bwd");
    GfWJobBuilder jb;
    float* g = grads;
    jb.gemm(W.DY, DL_D, ca, DL_D, DL_D, DL_D, DL_D, g, DL_D); g += DL_D * DL_D;          // out_mlp[0].weight
    jb.cols(W.DY, DL_D, DL_D, M, g); g += DL_D;
    jb.cols(W.PN + 2 * DL_D, 4 * DL_D, DL_D, tiles, g); g += DL_D;                       // norm3.weight
    jb.cols(W.PN + 3 * DL_D, 4 * DL_D, DL_D, tiles, g); g += DL_D;                       // norm3.bias
    jb.gemm(W.DH, ff, W.T3, DL_D, ff, DL_D, DL_D, g, DL_D); g += (size_t)ff * DL_D;      // linear1.weight [ff,64]
    jb.cols(W.DH, ff, ff, M, g); g += ff;
    jb.gemm(W.DF, DL_D, W.HD, ff, DL_D, ff, ff, g, ff); g += (size_t)DL_D * ff;          // linear2.weight [64,ff]
    jb.cols(W.DF, DL_D, DL_D, M, g); g += DL_D;
    jb.cols(W.PN, 4 * DL_D, DL_D, tiles, g); g += DL_D;                                  // decoder.norm.weight (this layer's share)
    jb.cols(W.PN + DL_D, 4 * DL_D, DL_D, tiles, g); g += DL_D;
    return jb.launch(M, st);
}
