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
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DA_WAVES 16
#define DA_D 64

// max / sum over the 16 lanes that share lane>>4 (one DPP row): quad xor 1, quad xor 2, half mirror, mirror
__device__ __forceinline__ float da_row_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false)));
    return v;
}
__device__ __forceinline__ float da_row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false));
    return v;
}

// 16 waves per (query, batch) = 4 per SIMD: the kernel has to fit 128 VGPRs, so only the online soft-max state
// (48 values) stays in registers across tiles; the per-query constants (Q1 row, the lane's Fourier projections)
// sit in LDS next to the weights.  b2 is not needed at all: the soft-max runs over the contexts separately for
// every channel, and a per-channel constant cancels in it.
__global__ __launch_bounds__(DA_WAVES * 64) void k_decoder_cross_attn(
    const float* __restrict__ geo_ctx, const float* __restrict__ max_geo, const float* __restrict__ qloc,
    const float* __restrict__ cloc, const float* __restrict__ lo, const float* __restrict__ hi,
    const float* __restrict__ gaussB, const float* __restrict__ Q1, const float* __restrict__ K1,
    const float* __restrict__ Kv, const float4* __restrict__ Wpack, int nq, int nc, float* __restrict__ out) {
    __shared__ float4 sW[3 * 16 * 64];       // [3][4 rb][4 kb][64 lanes]
    __shared__ float4 sQ1[16];               // Q1 row of this query: channel rb*16 + 4g + r at [rb*4 + g]
    __shared__ float4 sB[3][2][4];           // sB[axis][half][g] = gaussB[axis][half*16 + 4g .. +3]
    __shared__ float sRed[3][DA_WAVES][DA_D];
    const int qi = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    for (int t = tid; t < 3 * 16 * 64; t += DA_WAVES * 64) sW[t] = Wpack[t];
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
    for (int t = w; t < ntiles; t += DA_WAVES) {
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
        for (int u = 1; u < DA_WAVES; u++) m = fmaxf(m, sRed[0][u][tid]);
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
        for (int u = 0; u < DA_WAVES; u++) {
            L += sRed[1][u][tid];
            A += sRed[2][u][tid];
        }
        out[((size_t)b * nq + qi) * DA_D + tid] = A / L;
    }
}

extern "C" size_t gf_decoder_wpack_floats(void) { return (size_t)3 * 16 * 64 * 4; }

// Wpack[m][rb][kb][lane][s] = W_m[rb*16 + (lane&15)][kb*16 + 4*(lane>>4) + s]   (m: 0 = W1, 1 = W2, 2 = Wv; W is [out,in])
__global__ void k_decoder_pack(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ Wv,
                               float* __restrict__ Wp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * 16 * 64 * 4) return;
    const int s = t & 3, lane = (t >> 2) & 63, kb = (t >> 8) & 3, rb = (t >> 10) & 3, m = t >> 12;
    const float* W = m == 0 ? W1 : (m == 1 ? W2 : Wv);
    Wp[t] = W[(rb * 16 + (lane & 15)) * DA_D + kb * 16 + 4 * (lane >> 4) + s];
}

extern "C" int gf_decoder_pack_weights(const float* W1, const float* W2, const float* Wv, float* Wpack, void* stream) {
    hipLaunchKernelGGL(k_decoder_pack, dim3(48), dim3(256), 0, (hipStream_t)stream, W1, W2, Wv, Wpack);
    GF_CHECK_LAUNCH("gf_decoder_pack_weights");
    return GF_OK;
}

extern "C" int gf_decoder_cross_attn(const float* geo_ctx, const float* max_geo, const float* qloc, const float* cloc,
                                     const float* lo, const float* hi, const float* gaussB, const float* Q1,
                                     const float* K1, const float* Kv, const float* Wpack, const float* b2, int B,
                                     int nq, int nc, int d, float* out, void* stream) {
    GF_CHECK_ARG(d == DA_D, "gf_decoder_cross_attn: implemented for dec_dim = 64 (got %d)", d);
    GF_CHECK_ARG(B >= 0 && nq >= 0 && nc >= 1, "gf_decoder_cross_attn: bad sizes");
    if (B == 0 || nq == 0) return GF_OK;
    (void)b2;  // a per-channel constant cancels in the per-channel soft-max over the contexts
    hipLaunchKernelGGL(k_decoder_cross_attn, dim3(nq, B), dim3(DA_WAVES * 64), 0, (hipStream_t)stream, geo_ctx, max_geo,
                       qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, reinterpret_cast<const float4*>(Wpack), nq, nc, out);
    GF_CHECK_LAUNCH("gf_decoder_cross_attn");
    return GF_OK;
}
