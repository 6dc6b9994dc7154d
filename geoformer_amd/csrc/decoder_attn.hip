// Geodesic-guided vector cross-attention of the decoder, fused (gfx950).
//
//   reference: TransformerDecoderLayer.forward_pre_rel, model/transformer_detr.py:443-454, with the
//   relative embedding built in GeoFormer.forward_decoder, model/geoformer/geoformer.py:619-651:
//       r_ij   = fourier(g_ij)                      g_ij = geodesic distance query i -> context j
//                                                    (unreachable: max_geo_i + |xyz_i - xyz_j| per axis)
//       sim_ij = W2 relu(W1 (q_i - k_j + r_ij) + b1) + b2
//       attn   = softmax_j(sim / sqrt(64))           PER CHANNEL
//       out_i  = sum_j attn_ij * (Wv (k_j + r_ij) + bv)
//   The reference materialises six [nq, nc, B, 64] tensors per layer (134 MB each at 256 x 2048); here
//   r_ij is recomputed on the fly and nothing of size nq*nc*64 ever exists.  Linear parts that do not
//   depend on the pair are hoisted by the caller: Q1 = W1 q + b1, K1 = W1 k, Kv = Wv k + bv.
//
// Mapping.  One 1024-thread workgroup per (query, batch); its 16 waves split the contexts in tiles of
// 16.  Everything is kept TRANSPOSED -- channels on the MFMA row index, contexts on the column
// (lane&15) -- so the accumulator of one product is directly the B operand of the next
// (v_mfma_f32_16x16x4_f32: lane (g = lane>>4, j = lane&15) holds rows 4g..4g+3 of a 16-row block, and
// since the MFMA k index is only a summation index, step s simply consumes row 4g+s):
//       H^T   = W1 . R^T   (64 MFMAs)  -> + Q1_i - K1_j, ReLU
//       sim^T = W2 . H^T   (64 MFMAs)  -> + b2, / 8
//       v^T   = Wv . R^T   (64 MFMAs)  -> + Kv_j
// The weights sit in LDS pre-packed in A-operand lane order (one conflict-free ds_read_b128 per four
// MFMAs).  Each lane computes the 16 embedding channels it feeds (8 projections -> sin and cos), and
// carries an online-softmax state (max, sum, weighted sum) for its 16 channels over the contexts it
// sees; the 16 x 16 partial states per channel are merged once at the end (channel maximum, rescale, row sums).  MFMA-bound: 3*2*64*64 flop
// per pair = 12.9 GFLOP per layer at 256 x 2048, against ~1 MB of input.
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DA_WAVES 16
#define DA_D 64

// max / sum over the 16 lanes that share lane>>4 (one DPP row): quad xor 1, quad xor 2, half mirror, mirror
__device__ __forceinline__ float da_row_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
    return v;
}
__device__ __forceinline__ float da_row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

// 16 waves per (query, batch) = 4 per SIMD: the kernel has to fit 128 VGPRs, so only the online soft-max state
// (48 values) stays in registers across tiles; the per-query constants (Q1 row, the lane's Fourier projections)
// sit in LDS next to the weights.  b2 is not needed at all: the soft-max runs over the contexts separately for
// every channel, and a per-channel constant cancels in it.
// WAVES = 8 (gf_decoder_cross_attn_cfg): half the contexts' tiles per wave more, but a 512-thread workgroup at 120
// registers takes 240 of a SIMD's 512 -- it fits on a compute unit BESIDE a 512-thread BFS workgroup (144 registers, 75 KiB
// of LDS), which the 16-wave shape does not: a serving loop runs the previous scene's decoder under the current
// scene's sampling / BFS stretch that way (GeoFormer.forward_split; alone the 8-wave launch is 6 % slower).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_decoder_cross_attn(
    const float* __restrict__ geo_ctx, const float* __restrict__ max_geo, const float* __restrict__ qloc,
    const float* __restrict__ cloc, const float* __restrict__ lo, const float* __restrict__ hi,
    const float* __restrict__ gaussB, const float* __restrict__ Q1, const float* __restrict__ K1,
    const float* __restrict__ Kv, const float4* __restrict__ Wpack, int nq, int nc, float* __restrict__ out,
    float* __restrict__ stat_m, float* __restrict__ stat_l) {
    __shared__ float4 sW[3 * 16 * 64];       // [3][4 rb][4 kb][64 lanes]
    __shared__ float4 sQ1[16];               // Q1 row of this query: channel rb*16 + 4g + r at [rb*4 + g]
    __shared__ float4 sB[3][2][4];           // sB[axis][half][g] = gaussB[axis][half*16 + 4g .. +3]
    __shared__ float sRed[3][WAVES][DA_D];
    const int qi = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    for (int t = tid; t < 3 * 16 * 64; t += WAVES * 64) sW[t] = Wpack[t];
    if (tid < 16) sQ1[tid] = *reinterpret_cast<const float4*>(Q1 + ((size_t)b * nq + qi) * DA_D + tid * 4);
    if (tid >= 64 && tid < 64 + 24) {
        const int e = tid - 64, axis = e >> 3, half = (e >> 2) & 1, gg = e & 3;
        sB[axis][half][gg] = *reinterpret_cast<const float4*>(gaussB + axis * 32 + half * 16 + 4 * gg);
    }
    geo_ctx += ((size_t)b * nq + qi) * nc;
    cloc += (size_t)b * nc * 3;
    K1 += (size_t)b * nc * DA_D;
    Kv += (size_t)b * nc * DA_D;
    const float mg = max_geo[(size_t)b * nq + qi];
    const float qx = qloc[((size_t)b * nq + qi) * 3 + 0], qy = qloc[((size_t)b * nq + qi) * 3 + 1],
                qz = qloc[((size_t)b * nq + qi) * 3 + 2];
    const float lx = lo[b * 3 + 0], ly = lo[b * 3 + 1], lz = lo[b * 3 + 2];
    const float sx = hi[b * 3 + 0] - lx, sy = hi[b * 3 + 1] - ly, sz = hi[b * 3 + 2] - lz;
    float sm[4][4], sl[4][4], sa[4][4];  // online softmax: running max, sum, weighted sum
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            sm[rb][r] = -3.0e38f;
            sl[rb][r] = 0.f;
            sa[rb][r] = 0.f;
        }
    __syncthreads();

    const int ntiles = (nc + 15) >> 4;
    for (int t = w; t < ntiles; t += WAVES) {
        const int ctx = t * 16 + j;
        const bool valid = ctx < nc;
        const int cc = valid ? ctx : nc - 1;
        // --- relative embedding channels of this lane ---
        const float gd = geo_ctx[cc];
        float g0 = gd, g1 = gd, g2 = gd;
        if (gd < 0.f) {
            g0 = mg + fabsf(qx - cloc[cc * 3 + 0]);
            g1 = mg + fabsf(qy - cloc[cc * 3 + 1]);
            g2 = mg + fabsf(qz - cloc[cc * 3 + 2]);
        }
        const float t0 = ((g0 - lx) / sx) * 6.2831855f, t1 = ((g1 - ly) / sy) * 6.2831855f,
                    t2 = ((g2 - lz) / sz) * 6.2831855f;
        float R[4][4];  // R[kb][s]: channel kb*16 + 4g + s
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const float4 b0 = sB[0][half][g], b1 = sB[1][half][g], b2v = sB[2][half][g];
            const float p[4] = {fmaf(t2, b2v.x, fmaf(t1, b1.x, t0 * b0.x)), fmaf(t2, b2v.y, fmaf(t1, b1.y, t0 * b0.y)),
                                fmaf(t2, b2v.z, fmaf(t1, b1.z, t0 * b0.z)), fmaf(t2, b2v.w, fmaf(t1, b1.w, t0 * b0.w))};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float sn, cs;
                __sincosf(p[e], &sn, &cs);
                R[half][e] = sn;
                R[2 + half][e] = cs;
            }
        }
        // --- H^T = W1 . R^T ---
        f32x4 H[4];
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const float4 a = sW[((0 * 4 + rb) * 4 + kb) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, R[kb][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, R[kb][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, R[kb][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, R[kb][3], acc, 0, 0, 0);
            }
            const float4 k1 = *reinterpret_cast<const float4*>(K1 + (size_t)cc * DA_D + rb * 16 + 4 * g);
            const float4 q1 = sQ1[rb * 4 + g];
            acc[0] = fmaxf(acc[0] + q1.x - k1.x, 0.f);
            acc[1] = fmaxf(acc[1] + q1.y - k1.y, 0.f);
            acc[2] = fmaxf(acc[2] + q1.z - k1.z, 0.f);
            acc[3] = fmaxf(acc[3] + q1.w - k1.w, 0.f);
            H[rb] = acc;
            __builtin_amdgcn_sched_barrier(0);  // keep the next block's 4 weight reads from being hoisted (VGPRs)
        }
        // --- sim^T = W2 . H^T ,  v^T = Wv . R^T , online softmax ---
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const float4 a2 = sW[((1 * 4 + rb) * 4 + kb) * 64 + lane];
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, H[kb][0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, H[kb][1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, H[kb][2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, H[kb][3], s, 0, 0, 0);
                const float4 av = sW[((2 * 4 + rb) * 4 + kb) * 64 + lane];
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, R[kb][0], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, R[kb][1], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, R[kb][2], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, R[kb][3], v, 0, 0, 0);
            }
            const float4 kv = *reinterpret_cast<const float4*>(Kv + (size_t)cc * DA_D + rb * 16 + 4 * g);
            const float kvv[4] = {kv.x, kv.y, kv.z, kv.w};
            if (valid) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float x = s[r] * 0.125f;
                    const float val = v[r] + kvv[r];
                    const float mn = fmaxf(sm[rb][r], x);
                    const float corr = __expf(sm[rb][r] - mn), p = __expf(x - mn);
                    sl[rb][r] = sl[rb][r] * corr + p;
                    sa[rb][r] = sa[rb][r] * corr + p * val;
                    sm[rb][r] = mn;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // --- merge: channel maximum over all lanes, rescale, then plain sums ---
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float m = da_row_max(sm[rb][r]);
            if (j == 0) sRed[0][w][rb * 16 + 4 * g + r] = m;
        }
    __syncthreads();
    if (tid < DA_D) {
        float m = sRed[0][0][tid];
#pragma unroll
        for (int u = 1; u < WAVES; u++) m = fmaxf(m, sRed[0][u][tid]);
        sRed[0][0][tid] = m;
    }
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int c = rb * 16 + 4 * g + r;
            const float f = __expf(sm[rb][r] - sRed[0][0][c]);
            const float l = da_row_sum(sl[rb][r] * f), a = da_row_sum(sa[rb][r] * f);
            if (j == 0) {
                sRed[1][w][c] = l;
                sRed[2][w][c] = a;
            }
        }
    __syncthreads();
    if (tid < DA_D) {
        float L = 0.f, A = 0.f;
#pragma unroll
        for (int u = 0; u < WAVES; u++) {
            L += sRed[1][u][tid];
            A += sRed[2][u][tid];
        }
        out[((size_t)b * nq + qi) * DA_D + tid] = A / L;
        if (stat_m) {  // soft-max statistics of the scaled logits per channel: the backward recomputes the rest
            stat_m[((size_t)b * nq + qi) * DA_D + tid] = sRed[0][0][tid];
            stat_l[((size_t)b * nq + qi) * DA_D + tid] = L;
        }
    }
}

// ------------------------------------------------------------------------------------
// Round 4: the same layer with its three 64 x 64 products on the bf16 matrix pipe at fp32 accuracy
// (k_decoder_cross_attn_bf3).  An fp32 number is exactly the sum of three bf16 numbers (x = hi + mid + lo by
// truncation, 8 + 8 + 8 significant bits), a product of two fp32 numbers therefore the sum of nine bf16 products, of
// which the six largest carry everything above 3 * 2^-24 of it -- one fp32 rounding -- and each of which the MFMA adds
// exactly into its fp32 accumulator.  v_mfma_f32_16x16x32_bf16 takes K = 32 channels in 16 cycles where
// v_mfma_f32_16x16x4_f32 takes K = 4 in 32: a 16-row block of one product is 2 K-steps x 6 piece pairs = 12
// instructions (192 cycles) instead of 16 (512).  The weights' pieces are packed once (k_decoder_pack_bf3) and sit in LDS
// in A-operand lane order; an activation's pieces are cut from the accumulator registers that hold it (the K index is
// only a summation index: a lane's 8 values of a K-step are rows 4g..4g+3 of two 16-row blocks), ~5.5 vector
// instructions per value.  Same mapping, soft-max and merge as k_decoder_cross_attn<16>.
// ------------------------------------------------------------------------------------
#define DA_WPACK_F32 (3 * 16 * 64 * 4)
#define DA_WPACK_BF3_U4 (3 * 4 * 2 * 3 * 64)
__device__ __forceinline__ void da_split3(float x, unsigned& h, unsigned& m, unsigned& l) {  // bf16 patterns in the HIGH halves
    h = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);  // exact
    m = __float_as_uint(r1) & 0xffff0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));  // exact, at most 8 significant bits: its low half is zero
}
typedef __bf16 da_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned da_u32x4 __attribute__((ext_vector_type(4)));
struct DaPieces {
    da_u32x4 p[2][3];  // [K-step][hi, mid, lo]
};
// the three-piece split of a lane's 16 values x[kb][s] (kb = 16-row block, s = row 4g+s) as B operands of the two K-steps
__device__ __forceinline__ void da_split_pack(const float (&x)[4][4], DaPieces& P) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        unsigned pc[3][8];
#pragma unroll
        for (int e = 0; e < 8; e++) da_split3(x[2 * h + (e >> 2)][e & 3], pc[0][e], pc[1][e], pc[2][e]);
#pragma unroll
        for (int piece = 0; piece < 3; piece++)
#pragma unroll
            for (int i = 0; i < 4; i++)  // bytes {hi half of element 2i+1, hi half of element 2i}
                P.p[h][piece][i] = __builtin_amdgcn_perm(pc[piece][2 * i + 1], pc[piece][2 * i], 0x07060302u);
    }
}
// acc += W_m[rb-block] . X over both K-steps, six piece pairs each, smallest first (Wp3: the block's 2 x 3 operands in LDS)
__device__ __forceinline__ f32x4 da_gemm_bf3(const uint4* __restrict__ sW3, int m, int rb, int lane, const DaPieces& X, f32x4 acc) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint4* w = sW3 + (size_t)(((m * 4 + rb) * 2 + h) * 3) * 64 + lane;
        const da_bf16x8 ah = __builtin_bit_cast(da_bf16x8, w[0]), am = __builtin_bit_cast(da_bf16x8, w[64]),
                        al = __builtin_bit_cast(da_bf16x8, w[128]);
        const da_bf16x8 xh = __builtin_bit_cast(da_bf16x8, X.p[h][0]), xm = __builtin_bit_cast(da_bf16x8, X.p[h][1]),
                        xl = __builtin_bit_cast(da_bf16x8, X.p[h][2]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc, 0, 0, 0);
    }
    return acc;
}

#ifndef DA_BF3_WAVES
#define DA_BF3_WAVES 16
#endif
// NT tiles of 16 contexts per trip of a wave: every weight operand read from LDS serves NT products (a wave used to
// re-read all 73 KB of packed weights for every tile: 2.9 GB of ds_read_b128 per launch, the LDS 43 % busy), and the NT
// accumulator chains are independent, so consecutive MFMAs do not wait for each other.  <16, 1> is the round-4 shape
// (128 registers, four waves per SIMD); <8, 2> holds two tiles' pieces (two waves per SIMD).
template <int NT>
__device__ __forceinline__ void da_gemm_bf3_n(const uint4* __restrict__ sW3, int m, int rb, int lane, const DaPieces (&X)[NT],
                                              f32x4 (&acc)[NT]) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint4* w = sW3 + (size_t)(((m * 4 + rb) * 2 + h) * 3) * 64 + lane;
        const da_bf16x8 ah = __builtin_bit_cast(da_bf16x8, w[0]), am = __builtin_bit_cast(da_bf16x8, w[64]),
                        al = __builtin_bit_cast(da_bf16x8, w[128]);
        // six piece pairs, smallest first; the tiles alternate inside a pair
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(da_bf16x8, X[u].p[h][0]), acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(da_bf16x8, X[u].p[h][2]), acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, __builtin_bit_cast(da_bf16x8, X[u].p[h][1]), acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, __builtin_bit_cast(da_bf16x8, X[u].p[h][0]), acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(da_bf16x8, X[u].p[h][1]), acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(da_bf16x8, X[u].p[h][0]), acc[u], 0, 0, 0);
    }
}

template <int WAVES, int NT>
__global__ __launch_bounds__(WAVES * 64) void k_decoder_cross_attn_bf3(
    const float* __restrict__ geo_ctx, const float* __restrict__ max_geo, const float* __restrict__ qloc,
    const float* __restrict__ cloc, const float* __restrict__ lo, const float* __restrict__ hi,
    const float* __restrict__ gaussB, const float* __restrict__ Q1, const float* __restrict__ K1,
    const float* __restrict__ Kv, const uint4* __restrict__ Wp3, int nq, int nc, float* __restrict__ out,
    float* __restrict__ stat_m, float* __restrict__ stat_l) {
    extern __shared__ __attribute__((aligned(16))) unsigned char da_smem[];
    uint4* sW3 = reinterpret_cast<uint4*>(da_smem);  // [3][4 rb][2 h][3 pieces][64 lanes]
    __shared__ float4 sQ1[16];
    __shared__ float4 sB[3][2][4];
    __shared__ float sRed[3][WAVES][DA_D];
    const int qi = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    for (int t = tid; t < DA_WPACK_BF3_U4; t += WAVES * 64) sW3[t] = Wp3[t];
    if (tid < 16) sQ1[tid] = *reinterpret_cast<const float4*>(Q1 + ((size_t)b * nq + qi) * DA_D + tid * 4);
    if (tid >= 64 && tid < 64 + 24) {
        const int e = tid - 64, axis = e >> 3, half = (e >> 2) & 1, gg = e & 3;
        sB[axis][half][gg] = *reinterpret_cast<const float4*>(gaussB + axis * 32 + half * 16 + 4 * gg);
    }
    geo_ctx += ((size_t)b * nq + qi) * nc;
    cloc += (size_t)b * nc * 3;
    K1 += (size_t)b * nc * DA_D;
    Kv += (size_t)b * nc * DA_D;
    const float mg = max_geo[(size_t)b * nq + qi];
    const float qx = qloc[((size_t)b * nq + qi) * 3 + 0], qy = qloc[((size_t)b * nq + qi) * 3 + 1],
                qz = qloc[((size_t)b * nq + qi) * 3 + 2];
    const float lx = lo[b * 3 + 0], ly = lo[b * 3 + 1], lz = lo[b * 3 + 2];
    // Round 6: the normalisation as one multiplication by a reciprocal taken once (three IEEE divisions per tile and lane
    // were ~30 instructions), the phases kept in REVOLUTIONS -- v_sin_f32 / v_cos_f32 take revolutions, so the reference's
    // "* 2 pi" and the intrinsic's "* 1 / (2 pi)" cancel instead of being executed --, and the soft-max in base 2 with
    // log2(e) / 8 folded into one scale: ~10 % fewer vector instructions per tile and NO change in the launch's time
    // (80.7 against 81.0 us; without the K1 / Kv loads 77.6: DESIGN.md 7) -- the kernel is not bound by its vector issue.
    const float ix = 1.0f / (hi[b * 3 + 0] - lx), iy = 1.0f / (hi[b * 3 + 1] - ly), iz = 1.0f / (hi[b * 3 + 2] - lz);
    constexpr float kScale2 = 0.125f * 1.44269504088896341f;  // sim / sqrt(64), in units of log2
    float sm[4][4], sl[4][4], sa[4][4];  // online softmax: running max (log2 units), sum, weighted sum
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            sm[rb][r] = -3.0e38f;
            sl[rb][r] = 0.f;
            sa[rb][r] = 0.f;
        }
    __syncthreads();

    const int ntiles = (nc + 15) >> 4;
    for (int t0 = w * NT; t0 < ntiles; t0 += WAVES * NT) {
        bool valid[NT];
        int cc[NT];
        DaPieces RP[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) {
            const int ctx = (t0 + u) * 16 + j;
            valid[u] = ctx < nc;
            cc[u] = valid[u] ? ctx : nc - 1;
            // --- relative embedding channels of this lane (as k_decoder_cross_attn) ---
            const float gd = geo_ctx[cc[u]];
            float g0 = gd, g1 = gd, g2 = gd;
            if (gd < 0.f) {
                g0 = mg + fabsf(qx - cloc[cc[u] * 3 + 0]);
                g1 = mg + fabsf(qy - cloc[cc[u] * 3 + 1]);
                g2 = mg + fabsf(qz - cloc[cc[u] * 3 + 2]);
            }
            const float t0f = (g0 - lx) * ix, t1f = (g1 - ly) * iy, t2f = (g2 - lz) * iz;  // normalised, in revolutions
            float R[4][4];  // R[kb][s]: channel kb*16 + 4g + s
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const float4 b0 = sB[0][half][g], b1 = sB[1][half][g], b2v = sB[2][half][g];
                const float p[4] = {fmaf(t2f, b2v.x, fmaf(t1f, b1.x, t0f * b0.x)), fmaf(t2f, b2v.y, fmaf(t1f, b1.y, t0f * b0.y)),
                                    fmaf(t2f, b2v.z, fmaf(t1f, b1.z, t0f * b0.z)), fmaf(t2f, b2v.w, fmaf(t1f, b1.w, t0f * b0.w))};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    R[half][e] = __builtin_amdgcn_sinf(p[e]);      // sin(2 pi p)
                    R[2 + half][e] = __builtin_amdgcn_cosf(p[e]);  // cos(2 pi p)
                }
            }
            da_split_pack(R, RP[u]);
        }
        // --- H^T = W1 . R^T, + Q1_i - K1_j, ReLU;  v^T = Wv . R^T ---
        float H[NT][4][4];
        f32x4 V[NT][4];
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            f32x4 acc[NT], vv[NT];
#pragma unroll
            for (int u = 0; u < NT; u++) {
                acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                vv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            da_gemm_bf3_n<NT>(sW3, 0, rb, lane, RP, acc);
            const float4 q1 = sQ1[rb * 4 + g];
#pragma unroll
            for (int u = 0; u < NT; u++) {
                const float4 k1 = *reinterpret_cast<const float4*>(K1 + (size_t)cc[u] * DA_D + rb * 16 + 4 * g);
                H[u][rb][0] = fmaxf(acc[u][0] + q1.x - k1.x, 0.f);
                H[u][rb][1] = fmaxf(acc[u][1] + q1.y - k1.y, 0.f);
                H[u][rb][2] = fmaxf(acc[u][2] + q1.z - k1.z, 0.f);
                H[u][rb][3] = fmaxf(acc[u][3] + q1.w - k1.w, 0.f);
            }
            da_gemm_bf3_n<NT>(sW3, 2, rb, lane, RP, vv);
#pragma unroll
            for (int u = 0; u < NT; u++) V[u][rb] = vv[u];
            __builtin_amdgcn_sched_barrier(0);
        }
        DaPieces HP[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) da_split_pack(H[u], HP[u]);
        // --- sim^T = W2 . H^T, online softmax ---
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            f32x4 s[NT];
#pragma unroll
            for (int u = 0; u < NT; u++) s[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            da_gemm_bf3_n<NT>(sW3, 1, rb, lane, HP, s);
#pragma unroll
            for (int u = 0; u < NT; u++) {
                const float4 kv = *reinterpret_cast<const float4*>(Kv + (size_t)cc[u] * DA_D + rb * 16 + 4 * g);
                const float kvv[4] = {kv.x, kv.y, kv.z, kv.w};
                if (valid[u]) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float x = s[u][r] * kScale2;
                        const float val = V[u][rb][r] + kvv[r];
                        const float mn = fmaxf(sm[rb][r], x);
                        const float corr = __builtin_amdgcn_exp2f(sm[rb][r] - mn), p = __builtin_amdgcn_exp2f(x - mn);
                        sl[rb][r] = sl[rb][r] * corr + p;
                        sa[rb][r] = sa[rb][r] * corr + p * val;
                        sm[rb][r] = mn;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // --- merge: channel maximum over all lanes, rescale, then plain sums (as k_decoder_cross_attn) ---
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float m = da_row_max(sm[rb][r]);
            if (j == 0) sRed[0][w][rb * 16 + 4 * g + r] = m;
        }
    __syncthreads();
    if (tid < DA_D) {
        float m = sRed[0][0][tid];
#pragma unroll
        for (int u = 1; u < WAVES; u++) m = fmaxf(m, sRed[0][u][tid]);
        sRed[0][0][tid] = m;
    }
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int c = rb * 16 + 4 * g + r;
            const float f = __builtin_amdgcn_exp2f(sm[rb][r] - sRed[0][0][c]);
            const float l = da_row_sum(sl[rb][r] * f), a = da_row_sum(sa[rb][r] * f);
            if (j == 0) {
                sRed[1][w][c] = l;
                sRed[2][w][c] = a;
            }
        }
    __syncthreads();
    if (tid < DA_D) {
        float L = 0.f, A = 0.f;
#pragma unroll
        for (int u = 0; u < WAVES; u++) {
            L += sRed[1][u][tid];
            A += sRed[2][u][tid];
        }
        out[((size_t)b * nq + qi) * DA_D + tid] = A / L;
        if (stat_m) {
            stat_m[((size_t)b * nq + qi) * DA_D + tid] = sRed[0][0][tid] * 0.69314718055994531f;  // (natural units: the backward's)
            stat_l[((size_t)b * nq + qi) * DA_D + tid] = L;
        }
    }
}

// Wpack = [fp32 pack: 3 x 16 x 64 float4 | bf16 three-piece pack for k_decoder_cross_attn_bf3: 3 x 4 x 2 x 3 x 64 uint4]
extern "C" size_t gf_decoder_wpack_floats(void) { return (size_t)DA_WPACK_F32 + (size_t)DA_WPACK_BF3_U4 * 4; }

// Wpack[m][rb][kb][lane][s] = W_m[rb*16 + (lane&15)][kb*16 + 4*(lane>>4) + s]   (m: 0 = W1, 1 = W2, 2 = Wv; W is [out,in])
__global__ void k_decoder_pack(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ Wv,
                               float* __restrict__ Wp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * 16 * 64 * 4) return;
    const int s = t & 3, lane = (t >> 2) & 63, kb = (t >> 8) & 3, rb = (t >> 10) & 3, m = t >> 12;
    const float* W = m == 0 ? W1 : (m == 1 ? W2 : Wv);
    Wp[t] = W[(rb * 16 + (lane & 15)) * DA_D + kb * 16 + 4 * (lane >> 4) + s];
}

// The bf16 pack: Wp3[(((m*4 + rb)*2 + h)*3 + piece)*64 + lane] = 8 bf16 = the piece (0 hi, 1 mid, 2 lo of the exact
// three-piece split, da_split3) of W_m[rb*16 + (lane&15)][kb*16 + 4*(lane>>4) + s] for e = 0..7, kb = 2h + (e >> 2), s = e & 3:
// the A operand of v_mfma_f32_16x16x32_bf16 for output rows rb*16.., channels of K-step h.
__global__ void k_decoder_pack_bf3(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ Wv,
                                   uint4* __restrict__ Wp3) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;  // one (m, rb, h, lane): three uint4
    if (t >= 3 * 4 * 2 * 64) return;
    const int lane = t & 63, h = (t >> 6) & 1, rb = (t >> 7) & 3, m = t >> 9;
    const float* W = m == 0 ? W1 : (m == 1 ? W2 : Wv);
    unsigned pc[3][8];
    for (int e = 0; e < 8; e++) {
        const int kb = 2 * h + (e >> 2), sidx = e & 3;
        da_split3(W[(rb * 16 + (lane & 15)) * DA_D + kb * 16 + 4 * (lane >> 4) + sidx], pc[0][e], pc[1][e], pc[2][e]);
    }
    for (int piece = 0; piece < 3; piece++) {
        uint4 v;
        v.x = (pc[piece][1] & 0xffff0000u) | (pc[piece][0] >> 16);
        v.y = (pc[piece][3] & 0xffff0000u) | (pc[piece][2] >> 16);
        v.z = (pc[piece][5] & 0xffff0000u) | (pc[piece][4] >> 16);
        v.w = (pc[piece][7] & 0xffff0000u) | (pc[piece][6] >> 16);
        Wp3[((size_t)(((m * 4 + rb) * 2 + h) * 3 + piece)) * 64 + lane] = v;
    }
}

extern "C" int gf_decoder_pack_weights(const float* W1, const float* W2, const float* Wv, float* Wpack, void* stream) {
    hipLaunchKernelGGL(k_decoder_pack, dim3(48), dim3(256), 0, (hipStream_t)stream, W1, W2, Wv, Wpack);
    hipLaunchKernelGGL(k_decoder_pack_bf3, dim3(6), dim3(256), 0, (hipStream_t)stream, W1, W2, Wv,
                       reinterpret_cast<uint4*>(Wpack + DA_WPACK_F32));
    GF_CHECK_LAUNCH("gf_decoder_pack_weights");
    return GF_OK;
}

static int g_da_bf3 = -1;
extern "C" int gf_dev_cross_attn_bf3(int on) {  // 1 / 0 / -1 (default, GF_CROSS_ATTN_BF3)
    g_da_bf3 = on < 0 ? -1 : (on != 0);
    return GF_OK;
}

extern "C" int gf_decoder_cross_attn_cfg(const float* geo_ctx, const float* max_geo, const float* qloc, const float* cloc,
                                         const float* lo, const float* hi, const float* gaussB, const float* Q1,
                                         const float* K1, const float* Kv, const float* Wpack, const float* b2, int B,
                                         int nq, int nc, int d, float* out, float* stat_m, float* stat_l, int wg_waves,
                                         void* stream) {
    GF_CHECK_ARG(d == DA_D, "gf_decoder_cross_attn: implemented for dec_dim = 64 (got %d)", d);
    GF_CHECK_ARG(B >= 0 && nq >= 0 && nc >= 1, "gf_decoder_cross_attn: bad sizes");
    GF_CHECK_ARG((stat_m == nullptr) == (stat_l == nullptr), "gf_decoder_cross_attn: stat_m and stat_l come together");
    GF_CHECK_ARG(wg_waves == 16 || wg_waves == 8, "gf_decoder_cross_attn_cfg: %d waves per workgroup (16 or 8)", wg_waves);
    if (B == 0 || nq == 0) return GF_OK;
    (void)b2;  // a per-channel constant cancels in the per-channel soft-max over the contexts
    // (gf_dev_cross_attn_bf3(0): the fp32-MFMA kernel for the 16-wave shape too -- the split-vs-fp32 test's dev knob.  An
    //  eight-wave, two-tiles-per-trip instance of the bf16 kernel measured 85-86 against 84 us: HISTORY.md 7; removed.)
    if (wg_waves == 16 && g_da_bf3 != 0) {
        const size_t lds = (size_t)DA_WPACK_BF3_U4 * sizeof(uint4);
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)k_decoder_cross_attn_bf3<DA_BF3_WAVES, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        GF_LAUNCH_OP(GF_OP_CROSS_ATTN, (k_decoder_cross_attn_bf3<DA_BF3_WAVES, 1>), dim3(nq, B), dim3(DA_BF3_WAVES * 64), lds,
                     (hipStream_t)stream, geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv,
                     reinterpret_cast<const uint4*>(Wpack + DA_WPACK_F32), nq, nc, out, stat_m, stat_l);
        GF_CHECK_LAUNCH("gf_decoder_cross_attn");
        return GF_OK;
    }
    if (wg_waves == 16)
        GF_LAUNCH_OP(GF_OP_CROSS_ATTN, k_decoder_cross_attn<16>, dim3(nq, B), dim3(16 * 64), 0, (hipStream_t)stream, geo_ctx,
                     max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, reinterpret_cast<const float4*>(Wpack), nq, nc, out,
                     stat_m, stat_l);
    else
        GF_LAUNCH_OP(GF_OP_CROSS_ATTN, k_decoder_cross_attn<8>, dim3(nq, B), dim3(8 * 64), 0, (hipStream_t)stream, geo_ctx,
                     max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, reinterpret_cast<const float4*>(Wpack), nq, nc, out,
                     stat_m, stat_l);
    GF_CHECK_LAUNCH("gf_decoder_cross_attn");
    return GF_OK;
}

extern "C" int gf_decoder_cross_attn(const float* geo_ctx, const float* max_geo, const float* qloc, const float* cloc,
                                     const float* lo, const float* hi, const float* gaussB, const float* Q1,
                                     const float* K1, const float* Kv, const float* Wpack, const float* b2, int B,
                                     int nq, int nc, int d, float* out, float* stat_m, float* stat_l, void* stream) {
    return gf_decoder_cross_attn_cfg(geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, Wpack, b2, B, nq, nc, d, out,
                                     stat_m, stat_l, 16, stream);
}


// ------------------------------------------------------------------------------------
// Backward of the fused cross-attention (training).
//
//   with a = softmax_j(sim / 8) per channel and out_i = sum_j a_ij * v_ij:
//       dv_ij   = a_ij * gout_i                      dsim_ij = a_ij * gout_i * (v_ij - out_i) / 8
//       dH_ij   = (W2^T dsim_ij) * (H_ij > 0)        dQ1_i = sum_j dH_ij      dK1_j = -sum_i dH_ij      dKv_j = sum_i dv_ij
//       dW1 += dH_ij r_ij^T      dW2 += dsim_ij H_ij^T      dWv += dv_ij r_ij^T      (pair parts; the hoisted parts
//       W1 q, W1 k, Wv k are plain GEMMs whose gradients autograd adds)
//   r_ij carries no gradient (geodesic distances are data, gauss_B is frozen), and b2 cancels in the soft-max.
//
// Nothing of size nq*nc*64 is stored by the forward (only the soft-max maximum and sum per (query, channel)): a
// tile's embedding, H, sim and v are recomputed with the forward's MFMAs (orientation A: channels on the accumulator
// rows, the 16 contexts of the tile on the columns).  The products that follow need two facts about
// v_mfma_f32_16x16x4_f32 operands (the k index is only a summation label):
//   * an orientation-A accumulator IS a valid A operand of a product whose output has the contexts on the ROWS
//     (orientation B): dH_B = dsim_A . W2T   -- no data movement;
//   * an orientation-B register block (context 4g+r on the rows, channel on lane&15) is a valid A AND B operand of
//     the weight-gradient products, whose k index runs over the contexts: dW[co][ci] += X_B[co]^T . Y_B[ci].
// R, H, dsim and dv change orientation through a 16 x 64 tile in LDS (four 16-byte writes, sixteen 4-byte reads per
// lane and quantity).  A wave owns ONE context tile of one scene and walks a slice of the queries: the three 64 x 64
// weight gradients (192 registers) and the tile's dK1 / dKv rows stay in registers for the whole walk; per query it
// emits the column sums of dH (dQ1) into a [tiles, B*nq, 64] buffer.  One wave per SIMD (about 330 registers).
// 448 MFMAs per (query, tile) against the forward's 192.
// ------------------------------------------------------------------------------------
#define DAB_TS 68  // row stride (floats) of the transposition tile: 16 contexts x 64 channels

// W2T[cb][kb][lane][s] = W2[kb*16 + 4*(lane>>4) + s][cb*16 + (lane&15)]   (B operand of dH_B = dsim_A . W2)
__global__ void k_decoder_pack_w2t(const float* __restrict__ W2, float* __restrict__ Wp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 16 * 64 * 4) return;
    const int s = t & 3, lane = (t >> 2) & 63, kb = (t >> 8) & 3, cb = (t >> 10) & 3;
    Wp[t] = W2[(kb * 16 + 4 * (lane >> 4) + s) * DA_D + cb * 16 + (lane & 15)];
}

__global__ __launch_bounds__(256, 1) void k_decoder_cross_attn_bwd(
    const float* __restrict__ geo_ctx, const float* __restrict__ max_geo, const float* __restrict__ qloc,
    const float* __restrict__ cloc, const float* __restrict__ lo, const float* __restrict__ hi,
    const float* __restrict__ gaussB, const float* __restrict__ Q1, const float* __restrict__ K1,
    const float* __restrict__ Kv, const float4* __restrict__ Wpack, const float4* __restrict__ W2Tpack,
    const float* __restrict__ outv, const float* __restrict__ stat_m, const float* __restrict__ stat_l,
    const float* __restrict__ gout, int B, int nq, int nc, int qsplit, float* __restrict__ dQ1p,
    float* __restrict__ dK1, float* __restrict__ dKv, float* __restrict__ dWp) {
    __shared__ float4 sW[3 * 16 * 64];
    __shared__ float4 sW2T[16 * 64];
    __shared__ float4 sB[3][2][4];
    __shared__ __attribute__((aligned(16))) float sT[4][16 * DAB_TS];  // per wave: transposition tile
    __shared__ __attribute__((aligned(16))) float sV[4][4][DA_D];      // per wave: Q1_i, m_i, gout_i / L_i, out_i
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    for (int t = tid; t < 3 * 16 * 64; t += 256) sW[t] = Wpack[t];
    for (int t = tid; t < 16 * 64; t += 256) sW2T[t] = W2Tpack[t];
    if (tid < 24) {
        const int axis = tid >> 3, half = (tid >> 2) & 1, gg = tid & 3;
        sB[axis][half][gg] = *reinterpret_cast<const float4*>(gaussB + axis * 32 + half * 16 + 4 * gg);
    }
    __syncthreads();
    const int ntiles = (nc + 15) >> 4;
    const int wave = blockIdx.x * 4 + w;
    const int items = B * ntiles * qsplit;
    if (wave >= items) return;  // (no barrier below this point)
    const int part = wave % qsplit, bt = wave / qsplit;
    const int tile = bt % ntiles, b = bt / ntiles;
    const int qper = (nq + qsplit - 1) / qsplit;
    const int q0 = part * qper, q1 = min(nq, q0 + qper);
    float* T = sT[w];
    float* V = &sV[w][0][0];

    // ---- context side of the tile (constant over the queries) ----
    const int ctx = tile * 16 + j;
    const bool valid = ctx < nc;
    const int cc = valid ? ctx : nc - 1;
    const float* K1b = K1 + ((size_t)b * nc + cc) * DA_D;
    const float* Kvb = Kv + ((size_t)b * nc + cc) * DA_D;
    float4 k1[4], kv[4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
        k1[rb] = *reinterpret_cast<const float4*>(K1b + rb * 16 + 4 * g);
        kv[rb] = *reinterpret_cast<const float4*>(Kvb + rb * 16 + 4 * g);
    }
    const float cx = cloc[((size_t)b * nc + cc) * 3 + 0], cy = cloc[((size_t)b * nc + cc) * 3 + 1],
                cz = cloc[((size_t)b * nc + cc) * 3 + 2];
    const float lx = lo[b * 3 + 0], ly = lo[b * 3 + 1], lz = lo[b * 3 + 2];
    const float sx = hi[b * 3 + 0] - lx, sy = hi[b * 3 + 1] - ly, sz = hi[b * 3 + 2] - lz;

    f32x4 aW1[4][4], aW2[4][4], aWv[4][4];  // [out block][in block]: rows co = ob*16+4g+r, col ci = ib*16+j
    f32x4 aK1[4], aKv[4];                   // orientation B: rows = contexts 4g+r, col = channel cb*16+j
#pragma unroll
    for (int a = 0; a < 4; a++) {
        aK1[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        aKv[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            aW1[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            aW2[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            aWv[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // orientation A -> B through the wave's LDS tile: X_A[rb] (rows ch = rb*16+4g+r, col context j) is written as
    // T[context j][ch], read back as X_B[cb][r] = T[context 4g+r][ch = cb*16+j]
    auto transpose = [&](const f32x4 (&XA)[4], f32x4 (&XB)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // earlier reads of T are done
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
            *reinterpret_cast<float4*>(&T[j * DAB_TS + rb * 16 + 4 * g]) = make_float4(XA[rb][0], XA[rb][1], XA[rb][2], XA[rb][3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int cb = 0; cb < 4; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) XB[cb][r] = T[(4 * g + r) * DAB_TS + cb * 16 + j];
    };

    for (int qi = q0; qi < q1; qi++) {
        const size_t qrow = (size_t)b * nq + qi;
        // per-query vectors -> LDS (lanes 0..15 each move a float4 of each of the four vectors)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16) {
            const float4 q1v = *reinterpret_cast<const float4*>(Q1 + qrow * DA_D + lane * 4);
            const float4 mv = *reinterpret_cast<const float4*>(stat_m + qrow * DA_D + lane * 4);
            const float4 lv = *reinterpret_cast<const float4*>(stat_l + qrow * DA_D + lane * 4);
            const float4 gv = *reinterpret_cast<const float4*>(gout + qrow * DA_D + lane * 4);
            const float4 ov = *reinterpret_cast<const float4*>(outv + qrow * DA_D + lane * 4);
            *reinterpret_cast<float4*>(V + 0 * DA_D + lane * 4) = q1v;
            *reinterpret_cast<float4*>(V + 1 * DA_D + lane * 4) = mv;
            *reinterpret_cast<float4*>(V + 2 * DA_D + lane * 4) = make_float4(gv.x / lv.x, gv.y / lv.y, gv.z / lv.z, gv.w / lv.w);
            *reinterpret_cast<float4*>(V + 3 * DA_D + lane * 4) = ov;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- relative embedding of the pair (query qi, context j): the forward's expressions ----
        const float mg = max_geo[qrow];
        const float gd = geo_ctx[qrow * nc + cc];
        float g0 = gd, g1 = gd, g2 = gd;
        if (gd < 0.f) {
            g0 = mg + fabsf(qloc[qrow * 3 + 0] - cx);
            g1 = mg + fabsf(qloc[qrow * 3 + 1] - cy);
            g2 = mg + fabsf(qloc[qrow * 3 + 2] - cz);
        }
        const float t0 = ((g0 - lx) / sx) * 6.2831855f, t1 = ((g1 - ly) / sy) * 6.2831855f,
                    t2 = ((g2 - lz) / sz) * 6.2831855f;
        f32x4 RA[4];  // RA[kb][s]: channel kb*16 + 4g + s of context j
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const float4 b0 = sB[0][half][g], b1 = sB[1][half][g], b2v = sB[2][half][g];
            const float p[4] = {fmaf(t2, b2v.x, fmaf(t1, b1.x, t0 * b0.x)), fmaf(t2, b2v.y, fmaf(t1, b1.y, t0 * b0.y)),
                                fmaf(t2, b2v.z, fmaf(t1, b1.z, t0 * b0.z)), fmaf(t2, b2v.w, fmaf(t1, b1.w, t0 * b0.w))};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float sn, cs;
                __sincosf(p[e], &sn, &cs);
                RA[half][e] = sn;
                RA[2 + half][e] = cs;
            }
        }
        // ---- H_A = relu(W1 R + Q1_i - K1_j) ----
        f32x4 HA[4];
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const float4 a = sW[((0 * 4 + rb) * 4 + kb) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, RA[kb][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, RA[kb][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, RA[kb][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, RA[kb][3], acc, 0, 0, 0);
            }
            const float4 q1v = *reinterpret_cast<const float4*>(V + 0 * DA_D + rb * 16 + 4 * g);
            acc[0] = fmaxf(acc[0] + q1v.x - k1[rb].x, 0.f);
            acc[1] = fmaxf(acc[1] + q1v.y - k1[rb].y, 0.f);
            acc[2] = fmaxf(acc[2] + q1v.z - k1[rb].z, 0.f);
            acc[3] = fmaxf(acc[3] + q1v.w - k1[rb].w, 0.f);
            HA[rb] = acc;
        }
        // ---- sim_A, v_A -> dsim_A, dv_A ----
        f32x4 dsA[4], dvA[4];
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const float4 a2 = sW[((1 * 4 + rb) * 4 + kb) * 64 + lane];
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, HA[kb][0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, HA[kb][1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, HA[kb][2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, HA[kb][3], s, 0, 0, 0);
                const float4 av = sW[((2 * 4 + rb) * 4 + kb) * 64 + lane];
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, RA[kb][0], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, RA[kb][1], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, RA[kb][2], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, RA[kb][3], v, 0, 0, 0);
            }
            const float4 mv = *reinterpret_cast<const float4*>(V + 1 * DA_D + rb * 16 + 4 * g);
            const float4 gl = *reinterpret_cast<const float4*>(V + 2 * DA_D + rb * 16 + 4 * g);
            const float4 ov = *reinterpret_cast<const float4*>(V + 3 * DA_D + rb * 16 + 4 * g);
            const float mm[4] = {mv.x, mv.y, mv.z, mv.w}, gg[4] = {gl.x, gl.y, gl.z, gl.w}, oo[4] = {ov.x, ov.y, ov.z, ov.w};
            const float kk[4] = {kv[rb].x, kv[rb].y, kv[rb].z, kv[rb].w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float P = valid ? __expf(s[r] * 0.125f - mm[r]) * gg[r] : 0.f;  // a_ij * gout_i
                dvA[rb][r] = P;
                dsA[rb][r] = P * (v[r] + kk[r] - oo[r]) * 0.125f;
            }
        }
        // ---- dH_B = (dsim_A . W2) masked by H > 0 : rows contexts 4g+r, col ci = cb*16+j ----
        f32x4 HB[4], dHB[4];
        transpose(HA, HB);
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const float4 bw = sW2T[(cb * 4 + kb) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dsA[kb][0], bw.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dsA[kb][1], bw.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dsA[kb][2], bw.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dsA[kb][3], bw.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] = HB[cb][r] > 0.f ? acc[r] : 0.f;
            dHB[cb] = acc;
            aK1[cb][0] -= acc[0]; aK1[cb][1] -= acc[1]; aK1[cb][2] -= acc[2]; aK1[cb][3] -= acc[3];
            // dQ1_i[ci] partial: sum over the tile's contexts (rows): registers, then the four lane groups
            float cs = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            cs += __shfl_xor(cs, 16, 64);
            cs += __shfl_xor(cs, 32, 64);
            if (g == 0) dQ1p[((size_t)tile * B * nq + qrow) * DA_D + cb * 16 + j] = cs;
        }
        // ---- weight gradients: k index = context (lane group kq, step s <-> context 4kq+s) ----
        f32x4 XB[4], RB[4];
        transpose(RA, RB);
#pragma unroll
        for (int ob = 0; ob < 4; ob++)
#pragma unroll
            for (int ib = 0; ib < 4; ib++)
#pragma unroll
                for (int s = 0; s < 4; s++)
                    aW1[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(dHB[ob][s], RB[ib][s], aW1[ob][ib], 0, 0, 0);
        transpose(dvA, XB);
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            aKv[cb][0] += XB[cb][0]; aKv[cb][1] += XB[cb][1]; aKv[cb][2] += XB[cb][2]; aKv[cb][3] += XB[cb][3];
        }
#pragma unroll
        for (int ob = 0; ob < 4; ob++)
#pragma unroll
            for (int ib = 0; ib < 4; ib++)
#pragma unroll
                for (int s = 0; s < 4; s++)
                    aWv[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(XB[ob][s], RB[ib][s], aWv[ob][ib], 0, 0, 0);
        transpose(dsA, XB);
#pragma unroll
        for (int ob = 0; ob < 4; ob++)
#pragma unroll
            for (int ib = 0; ib < 4; ib++)
#pragma unroll
                for (int s = 0; s < 4; s++)
                    aW2[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(XB[ob][s], HB[ib][s], aW2[ob][ib], 0, 0, 0);
    }
    // ---- the tile's dK1 / dKv rows (contexts 4g+r, channel cb*16+j) ----
#pragma unroll
    for (int cb = 0; cb < 4; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int c2 = tile * 16 + 4 * g + r;
            if (c2 < nc) {
                float* d1 = dK1 + ((size_t)b * nc + c2) * DA_D + cb * 16 + j;
                float* d2 = dKv + ((size_t)b * nc + c2) * DA_D + cb * 16 + j;
                if (qsplit == 1) {
                    *d1 = aK1[cb][r];
                    *d2 = aKv[cb][r];
                } else {
                    atomicAdd(d1, aK1[cb][r]);
                    atomicAdd(d2, aKv[cb][r]);
                }
            }
        }
    // ---- this wave's weight-gradient partials: dWp[wave][m][co][ci] ----
    float* P = dWp + (size_t)wave * 3 * DA_D * DA_D;
#pragma unroll
    for (int ob = 0; ob < 4; ob++)
#pragma unroll
        for (int ib = 0; ib < 4; ib++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int co = ob * 16 + 4 * g + r, ci = ib * 16 + j;
                P[0 * DA_D * DA_D + co * DA_D + ci] = aW1[ob][ib][r];
                P[1 * DA_D * DA_D + co * DA_D + ci] = aW2[ob][ib][r];
                P[2 * DA_D * DA_D + co * DA_D + ci] = aWv[ob][ib][r];
            }
}

// out[i] = sum over `parts` of in[part * n + i]
// (n is a multiple of 4 for both callers: 3 * 64 * 64 weight words, B * nq * 64 query words.)  A thread owns four
// consecutive words and keeps eight parts in flight; the additions stay in part order (deterministic).  One load per
// trip took 134 us for the 1024 x 12288 weight partials of a training batch.
__global__ void k_sum_parts(const float* __restrict__ in, int parts, size_t n, float* __restrict__ out) {
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = n >> 2;
    if (i4 >= n4) return;
    const float4* in4 = reinterpret_cast<const float4*>(in);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = 0;
    for (; p + 8 <= parts; p += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = in4[(size_t)(p + u) * n4 + i4];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
        }
    }
    for (; p < parts; p++) {
        const float4 v = in4[(size_t)p * n4 + i4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i4] = s;
}

static void dab_plan(int B, int nq, int nc, int* qsplit, int* waves) {
    const int ntiles = (nc + 15) / 16;
    int qs = (256 * 4 + B * ntiles - 1) / (B * ntiles);  // one wave per SIMD
    if (qs > nq) qs = nq;
    if (qs > 8) qs = 8;
    if (qs < 1) qs = 1;
    *qsplit = qs;
    *waves = B * ntiles * qs;
}

extern "C" size_t gf_decoder_cross_attn_bwd_scratch_floats(int B, int nq, int nc) {
    int qs, waves;
    dab_plan(B, nq, nc, &qs, &waves);
    const size_t ntiles = (size_t)(nc + 15) / 16;
    return (size_t)waves * 3 * DA_D * DA_D + ntiles * (size_t)B * nq * DA_D + (size_t)16 * 64 * 4;
}

// dQ1 [B,nq,64], dK1 / dKv [B,nc,64] (zero on entry), dW [3,64,64] = pair parts of dW1, dW2, dWv (overwritten)
extern "C" int gf_decoder_cross_attn_bwd(const float* geo_ctx, const float* max_geo, const float* qloc, const float* cloc,
                                         const float* lo, const float* hi, const float* gaussB, const float* Q1,
                                         const float* K1, const float* Kv, const float* Wpack, const float* W2,
                                         const float* out, const float* stat_m, const float* stat_l, const float* gout,
                                         int B, int nq, int nc, int d, float* dQ1, float* dK1, float* dKv, float* dW,
                                         float* scratch, void* stream) {
    GF_CHECK_ARG(d == DA_D, "gf_decoder_cross_attn_bwd: implemented for dec_dim = 64 (got %d)", d);
    GF_CHECK_ARG(B >= 0 && nq >= 0 && nc >= 1, "gf_decoder_cross_attn_bwd: bad sizes");
    if (B == 0 || nq == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    int qs, waves;
    dab_plan(B, nq, nc, &qs, &waves);
    const int ntiles = (nc + 15) / 16;
    float* dWp = scratch;
    float* dQ1p = dWp + (size_t)waves * 3 * DA_D * DA_D;
    float* w2t = dQ1p + (size_t)ntiles * B * nq * DA_D;
    hipLaunchKernelGGL(k_decoder_pack_w2t, dim3(16), dim3(256), 0, st, W2, w2t);
    GF_LAUNCH_OP(GF_OP_CROSS_ATTN_BWD, k_decoder_cross_attn_bwd, dim3((waves + 3) / 4), dim3(256), 0, st, geo_ctx, max_geo,
                 qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, reinterpret_cast<const float4*>(Wpack),
                 reinterpret_cast<const float4*>(w2t), out, stat_m, stat_l, gout, B, nq, nc, qs, dQ1p, dK1, dKv, dWp);
    const size_t nW = (size_t)3 * DA_D * DA_D, nQ = (size_t)B * nq * DA_D;
    hipLaunchKernelGGL(k_sum_parts, dim3(gf_div_up((long long)nW / 4, 64)), dim3(64), 0, st, dWp, waves, nW, dW);
    hipLaunchKernelGGL(k_sum_parts, dim3(gf_div_up((long long)nQ / 4, 64)), dim3(64), 0, st, dQ1p, ntiles, nQ, dQ1);
    GF_CHECK_LAUNCH("gf_decoder_cross_attn_bwd");
    return GF_OK;
}
