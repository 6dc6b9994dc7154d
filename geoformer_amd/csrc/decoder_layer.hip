// Token-side stages of the DETR-style decoder, fused (inference).
//
// Around the pairwise cross-attention (decoder_attn.hip) a decoder layer (TransformerDecoderLayer.forward_pre_rel,
// model/transformer_detr.py:425-463) is ~30 small PyTorch launches over nq x 64 tokens: LayerNorm, the
// nn.MultiheadAttention self-attention, residual adds, out_mlp, the FFN, and the hoisted W1 q + b1 product of the
// next cross-attention.  One launch of this kernel runs everything BETWEEN two cross-attentions:
//
//   post (layer l):   tgt = relu(out_mlp(attn_out)) + tgt2;  tgt += linear2(relu(linear1(norm3(tgt))));
//                     inter[l] = norm(tgt)                                    (TransformerDecoder.forward:155-164)
//   pre  (layer l+1): t2 = norm1(tgt); q = k = t2 + query_pos; tgt += out_proj(MHA(q, k, t2));
//                     tgt2 = norm2(tgt);  Q1 = W1 tgt2 + b1   (first half of attn_mlp[0] for the next cross-attention)
//
// as two launches: stage A is token-parallel (a workgroup per 16 tokens, intermediates in LDS) and ends with the
// q/k/v projections; stage B is query-tile-parallel (a wave per head) and needs every token's K and V, which is why
// the launch boundary sits there.  A 4-layer decoder is 4 cross-attention launches + 9 of these small ones
// (~20 us each) instead of ~130.  All products are v_mfma_f32_16x16x4_f32 with operands loaded straight from
// row-major activations (LDS tiles or global rows) and nn.Linear weights [out,in] (operand scheme of
// backbone_attn.hip).  Self-attention: 4 heads x 16 channels, S^T = K Q^T keeps the query on the MFMA column so
// the online soft-max state is per lane column and P^T feeds V^T P^T directly; one key tile of look-ahead.
#include "decoder_layer.h"

struct DlPost {
    const float *omw, *omb, *n3w, *n3b, *l1w, *l1b, *l2w, *l2b, *fnw, *fnb;
};
struct DlPre {
    const float *n1w, *n1b, *ipw, *ipb, *opw, *opb, *n2w, *n2b, *w1w, *w1b;
};

#define DL_TILE_THREADS 256

#ifdef DL_TRACE
// dev build (tools/trace_decoder_stage.py; -DDL_TRACE via tools/build_variant.sh): cycle stamps of thread 0 of every workgroup
// at the phase boundaries -- stage A in words 0..8, stage B in words 16..22 of the workgroup's 32-word slot
__device__ unsigned long long* g_dl_trace = nullptr;
extern "C" int gf_dev_dl_trace(void* p) {
    GF_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dl_trace), &p, sizeof(p)));
    return GF_OK;
}
#define DL_STAMP(k) do { if (g_dl_trace && threadIdx.x == 0) g_dl_trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DL_STAMP(k) do { } while (0)
#endif

// Stage A, token-parallel (one workgroup per 16 tokens): the post part of a layer and the pre part of the next up
// to the q/k/v projections of the self-attention.  State in global memory: X (running target), TGT2, QKV.
__global__ __launch_bounds__(DL_TILE_THREADS) void k_decoder_stage_a(const float* __restrict__ attn_out,
                                                                     const float* __restrict__ tgt_in,
                                                                     const float* __restrict__ query_pos, int T,
                                                                     int B, int ff, int has_post, int has_pre,
                                                                     DlPost po, DlPre pr, float* __restrict__ state,
                                                                     float* __restrict__ inter_out) {
    __shared__ float sX[16][DL_LD], sT[16][DL_LD], sQ[16][DL_LD], sH[16][DL_LDH];
    const int b = blockIdx.y, t0 = blockIdx.x * 16;
    const int nvalid = min(16, T - t0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = DL_TILE_THREADS / 64;
    float* X = state + (size_t)b * T * (5 * DL_D);
    float* TGT2 = X + (size_t)T * DL_D;
    float* QKV = TGT2 + (size_t)T * DL_D;
    const int lds = B * DL_D;  // row stride of the [T,B,64] tensors
    DL_STAMP(0);
    // every phase's weights, biases and norm parameters requested NOW (one wave per SIMD: 512 registers each, and no phase
    // then opens with a fetch of its own -- decoder_layer.h)
    constexpr int NTF = DL_MAXFF / 16 / (DL_TILE_THREADS / 64);  // linear1's column tiles per wave at the widest FFN
    DlW<1, DL_D / 16> w_om;
    DlW<NTF, DL_D / 16> w_l1;
    DlW<1, DL_MAXFF / 16> w_l2;
    DlW<2, DL_D / 16> w_qk;
    DlW<1, DL_D / 16> w_v;
    float n3w = 0.f, n3b = 0.f, fnw = 0.f, fnb = 0.f, n1w = 0.f, n1b = 0.f;
    if (has_post) {
        dl_w_load(w_om, po.omw, po.omb, DL_D, DL_D, wave, nw, lane);
        n3w = po.n3w[lane]; n3b = po.n3b[lane];
        dl_w_load(w_l1, po.l1w, po.l1b, ff, DL_D, wave, nw, lane);
        dl_w_load(w_l2, po.l2w, po.l2b, DL_D, ff, wave, nw, lane);
        fnw = po.fnw[lane]; fnb = po.fnb[lane];
    }
    if (has_pre) {
        n1w = pr.n1w[lane]; n1b = pr.n1b[lane];
        dl_w_load(w_qk, pr.ipw, pr.ipb, 2 * DL_D, DL_D, wave, nw, lane);
        dl_w_load(w_v, pr.ipw + 2 * DL_D * DL_D, pr.ipb + 2 * DL_D, DL_D, DL_D, wave, nw, lane);
    }
    if (has_post) {
        // tgt = relu(out_mlp(attn)) + tgt2
        const float* tg = TGT2 + (size_t)t0 * DL_D;
        dl_tile_gemm_w<true>(attn_out + ((size_t)b * T + t0) * DL_D, DL_D, nvalid, DL_D, DL_D, wave, nw, lane, w_om,
                             [&](int r, int c, float v) { sX[r][c] = v; }, tg, DL_D);
        __syncthreads();
        DL_STAMP(1);
        dl_tile_layernorm_p(sX, nvalid, n3w, n3b, wave, nw, lane, [&](int r, int c, float v) { sT[r][c] = v; });
        __syncthreads();
        DL_STAMP(2);
        dl_tile_gemm_w<true>(&sT[0][0], DL_LD, nvalid, DL_D, ff, wave, nw, lane, w_l1,
                             [&](int r, int c, float v) { sH[r][c] = v; });
        __syncthreads();
        DL_STAMP(3);
        dl_tile_gemm_w<false>(&sH[0][0], DL_LDH, nvalid, ff, DL_D, wave, nw, lane, w_l2,
                              [&](int r, int c, float v) { sX[r][c] += v; });
        __syncthreads();
        DL_STAMP(4);
        float* io = inter_out + (size_t)t0 * lds + b * DL_D;
        dl_tile_layernorm_p(sX, nvalid, fnw, fnb, wave, nw, lane,
                            [&](int r, int c, float v) { io[(size_t)r * lds + c] = v; });
    } else {
        for (int i = threadIdx.x; i < nvalid * DL_D; i += DL_TILE_THREADS)
            sX[i >> 6][i & 63] = tgt_in[(size_t)(t0 + (i >> 6)) * lds + b * DL_D + (i & 63)];
        __syncthreads();
    }
    DL_STAMP(5);
    if (!has_pre) return;
    // t2 = norm1(tgt);  q = k = t2 + query_pos
    const float* qp = query_pos + (size_t)t0 * lds + b * DL_D;
    dl_tile_layernorm_p(sX, nvalid, n1w, n1b, wave, nw, lane, [&](int r, int c, float v) {
        sT[r][c] = v;
        sQ[r][c] = v + qp[(size_t)r * lds + c];
    });
    for (int i = threadIdx.x; i < nvalid * DL_D; i += DL_TILE_THREADS)
        X[(size_t)(t0 + (i >> 6)) * DL_D + (i & 63)] = sX[i >> 6][i & 63];
    __syncthreads();
    DL_STAMP(6);
    // in_proj: rows 0..127 of the packed weight -> q, k (from t2 + pos), rows 128..191 -> v (from t2)
    float* qkv = QKV + (size_t)t0 * (3 * DL_D);
    dl_tile_gemm_w<false>(&sQ[0][0], DL_LD, nvalid, DL_D, 2 * DL_D, wave, nw, lane, w_qk,
                          [&](int r, int c, float v) { qkv[(size_t)r * (3 * DL_D) + c] = v; });
    DL_STAMP(7);
    dl_tile_gemm_w<false>(&sT[0][0], DL_LD, nvalid, DL_D, DL_D, wave, nw, lane, w_v,
                          [&](int r, int c, float v) { qkv[(size_t)r * (3 * DL_D) + 2 * DL_D + c] = v; });
    DL_STAMP(8);
}

// self-attention of one head over at most QTN x 16 tokens for a wave's 16 queries, two passes (see the call): adds to o
// (channels x queries, this lane's four) and l (this lane's part of the row sums)
template <int QTN>
__device__ __forceinline__ void dl_attn_two_pass(const float* __restrict__ QKV, int T, int h, int j, int g, float4 bq,
                                                 f32x4& o, float& l) {
    float4 akt[QTN];
    float vt[QTN][4];
    const int last = T - 1;
#pragma unroll
    for (int kt = 0; kt < QTN; kt++) {
        const int krow = min(kt * 16 + j, last);
        akt[kt] = *reinterpret_cast<const float4*>(QKV + (size_t)krow * (3 * DL_D) + DL_D + h * DL_DK + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int key = min(kt * 16 + 4 * g + i, last);
            vt[kt][i] = QKV[(size_t)key * (3 * DL_D) + 2 * DL_D + h * DL_DK + j];
        }
    }
    float sc[QTN][4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < QTN; kt++) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        s = dl_mfma4(akt[kt], bq, s);  // (a clamped row beyond T: a finite product, masked below)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            sc[kt][i] = (kt * 16 + 4 * g + i) < T ? s[i] * 0.25f : -INFINITY;  // 1/sqrt(16)
            mx = fmaxf(mx, sc[kt][i]);
        }
    }
    mx = fmaxf(mx, gf_shfl_xor<16>(mx));
    mx = fmaxf(mx, gf_shfl_xor<32>(mx));  // finite: key 0 exists
#pragma unroll
    for (int kt = 0; kt < QTN; kt++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // (v_exp_f32 on the base-2 argument: the library expf is ~30 instructions, 64 of them per lane; a masked key: 0)
            const float pi = __builtin_amdgcn_exp2f((sc[kt][i] - mx) * 1.4426950408889634f);
            l += pi;
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[kt][i], pi, o, 0, 0, 0);
        }
    }
}

// Stage B, query-tile-parallel (one workgroup per 16 queries, one wave per head): self-attention over all
// tokens, out_proj + residual, norm2, and the query half of the next cross-attention's first linear.
__global__ __launch_bounds__(DL_TILE_THREADS) void k_decoder_stage_b(int T, int B, DlPre pr,
                                                                     float* __restrict__ state,
                                                                     float* __restrict__ q1_out) {
    __shared__ float sO[16][DL_LD], sX[16][DL_LD], sT[16][DL_LD];
    const int b = blockIdx.y, t0 = blockIdx.x * 16;
    const int nvalid = min(16, T - t0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = DL_TILE_THREADS / 64;
    const int j = lane & 15, g = lane >> 4;
    float* X = state + (size_t)b * T * (5 * DL_D);
    float* TGT2 = X + (size_t)T * DL_D;
    const float* QKV = TGT2 + (size_t)T * DL_D;
    DL_STAMP(16);
    // (the two products behind the self-attention: operands requested now)
    DlW<1, DL_D / 16> w_op, w_w1;
    dl_w_load(w_op, pr.opw, pr.opb, DL_D, DL_D, wave, nw, lane);
    dl_w_load(w_w1, pr.w1w, pr.w1b, DL_D, DL_D, wave, nw, lane);
    const float n2w = pr.n2w[lane], n2b = pr.n2b[lane];
    {
        const int h = wave;  // 4 waves = 4 heads
        const int QT = (T + 15) >> 4;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int qrow = t0 + j;
        float4 bq = z4;
        if (qrow < T) bq = *reinterpret_cast<const float4*>(QKV + (size_t)qrow * (3 * DL_D) + h * DL_DK + 4 * g);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, l = 0.f;
        if (QT <= 16) {
            // Two passes instead of the online soft-max below: with one wave per SIMD nothing hides the online form's chain
            // -- product, maximum, exponentials, product, 16 times in a row: 33 000 cycles of this kernel's 45 000 (cycle
            // stamps, round 6).  dl_attn_two_pass: the K rows and V columns of ALL tiles requested at once (clamped
            // addresses, masked afterwards: no branch per load), independent score products, an exact maximum.
            if (QT <= 8) dl_attn_two_pass<8>(QKV, T, h, j, g, bq, o, l);
            else dl_attn_two_pass<16>(QKV, T, h, j, g, bq, o, l);
        } else {
        // operands of key tile 0, then one tile of look-ahead
        float4 ak = z4;
        float v[4];
        {
            if (j < T) ak = *reinterpret_cast<const float4*>(QKV + (size_t)j * (3 * DL_D) + DL_D + h * DL_DK + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = 4 * g + i;
                v[i] = key < T ? QKV[(size_t)key * (3 * DL_D) + 2 * DL_D + h * DL_DK + j] : 0.f;
            }
        }
        for (int kt = 0; kt < QT; kt++) {
            float4 ak_n = z4;
            float v_n[4] = {0.f, 0.f, 0.f, 0.f};
            if (kt + 1 < QT) {
                const int krow = (kt + 1) * 16 + j;
                if (krow < T)
                    ak_n = *reinterpret_cast<const float4*>(QKV + (size_t)krow * (3 * DL_D) + DL_D + h * DL_DK + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int key = (kt + 1) * 16 + 4 * g + i;
                    v_n[i] = key < T ? QKV[(size_t)key * (3 * DL_D) + 2 * DL_D + h * DL_DK + j] : 0.f;
                }
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
            for (int i = 0; i < 4; i++) o = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i], p[i], o, 0, 0, 0);
            ak = ak_n;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = v_n[i];
        }
        }
        l += gf_shfl_xor<16>(l);
        l += gf_shfl_xor<32>(l);
#pragma unroll
        for (int i = 0; i < 4; i++) sO[j][h * DL_DK + 4 * g + i] = o[i] / l;
    }
    __syncthreads();
    DL_STAMP(17);
    // tgt += out_proj(O)
    const float* xg = X + (size_t)t0 * DL_D;
    dl_tile_gemm_w<false>(&sO[0][0], DL_LD, nvalid, DL_D, DL_D, wave, nw, lane, w_op,
                          [&](int r, int c, float v) { sX[r][c] = v; }, xg, DL_D);
    __syncthreads();
    DL_STAMP(18);
    float* xo = X + (size_t)t0 * DL_D;
    for (int i = threadIdx.x; i < nvalid * DL_D; i += DL_TILE_THREADS) xo[i] = sX[i >> 6][i & 63];
    float* tg = TGT2 + (size_t)t0 * DL_D;
    dl_tile_layernorm_p(sX, nvalid, n2w, n2b, wave, nw, lane, [&](int r, int c, float v) {
        sT[r][c] = v;
        tg[r * DL_D + c] = v;
    });
    __syncthreads();
    DL_STAMP(19);
    float* q1 = q1_out + ((size_t)b * T + t0) * DL_D;
    dl_tile_gemm_w<false>(&sT[0][0], DL_LD, nvalid, DL_D, DL_D, wave, nw, lane, w_w1,
                          [&](int r, int c, float v) { q1[r * DL_D + c] = v; });
    DL_STAMP(20);
}

extern "C" size_t gf_decoder_token_state_bytes(int nq, int B) {
    return (size_t)(nq > 0 ? nq : 0) * (B > 0 ? B : 0) * (5 * DL_D) * sizeof(float);  // X, TGT2, QKV
}

extern "C" int gf_decoder_token_stage(const float* attn_out, const float* tgt_in, const float* query_pos, int nq, int B,
                                      int d, int nhead, int ff, const float* const* post_params,
                                      const float* const* pre_params, void* state, float* inter_out, float* q1_out,
                                      void* stream) {
    GF_CHECK_ARG(d == DL_D && nhead == DL_H, "gf_decoder_token_stage: built for d_model=64, 4 heads (got %d, %d)", d,
                 nhead);
    GF_CHECK_ARG(ff > 0 && ff % 16 == 0 && ff <= DL_MAXFF, "gf_decoder_token_stage: dim_feedforward %d not in 16..%d",
                 ff, DL_MAXFF);
    GF_CHECK_ARG(nq >= 0 && B >= 0, "gf_decoder_token_stage: bad sizes");
    GF_CHECK_ARG(post_params || pre_params, "gf_decoder_token_stage: nothing to do");
    GF_CHECK_ARG(post_params ? (attn_out && inter_out) : (tgt_in != nullptr),
                 "gf_decoder_token_stage: post needs attn_out and inter_out, a first stage needs tgt_in");
    GF_CHECK_ARG(!pre_params || (query_pos && q1_out), "gf_decoder_token_stage: pre needs query_pos and q1_out");
    if (nq == 0 || B == 0) return GF_OK;
    DlPost po = {};
    DlPre pr = {};
    if (post_params) {
        for (int i = 0; i < 10; i++) GF_CHECK_ARG(post_params[i], "gf_decoder_token_stage: post_params[%d] is null", i);
        po = {post_params[0], post_params[1], post_params[2], post_params[3], post_params[4],
              post_params[5], post_params[6], post_params[7], post_params[8], post_params[9]};
    }
    if (pre_params) {
        for (int i = 0; i < 10; i++) GF_CHECK_ARG(pre_params[i], "gf_decoder_token_stage: pre_params[%d] is null", i);
        pr = {pre_params[0], pre_params[1], pre_params[2], pre_params[3], pre_params[4],
              pre_params[5], pre_params[6], pre_params[7], pre_params[8], pre_params[9]};
    }
    const dim3 grid((nq + 15) / 16, B);
    hipLaunchKernelGGL(k_decoder_stage_a, grid, dim3(DL_TILE_THREADS), 0, (hipStream_t)stream, attn_out, tgt_in,
                       query_pos, nq, B, ff, post_params ? 1 : 0, pre_params ? 1 : 0, po, pr, (float*)state, inter_out);
    if (pre_params)
        hipLaunchKernelGGL(k_decoder_stage_b, grid, dim3(DL_TILE_THREADS), 0, (hipStream_t)stream, nq, B, pr,
                           (float*)state, q1_out);
    GF_CHECK_LAUNCH("gf_decoder_token_stage");
    return GF_OK;
}
