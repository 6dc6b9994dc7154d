// Dynamic-convolution mask head, fused (gfx950).
//
//   reference: GeoFormer.mask_heads_forward (model/geoformer/geoformer.py:286-324) -- per query q and
//   foreground point p:   rel = q_xyz - p_xyz;  if geo[q,p] < 0: rel += sqrt(max_geo_q) * sign(rel)
//                         h   = relu(W1_q [rel ; f_p] + b1_q)        (W1_q: 16 x 19, generated per query)
//                         out = W2_q h + b2_q                        (W2_q: 1 x 16)
//   The reference materialises x.repeat(nq,1,1) (nq*19*N floats, 1.17 GB at 256 x 60k) and runs a
//   grouped conv1d; here nothing but geo (read once) and the logits (written once) touches HBM.
//
// Mapping.  One wave owns MH_Q queries (their generated parameters live in registers) and streams
// blocks of 64 points.  For a 16-point tile the feature part W1f_q[16x16] . F^T[16x16] is four
// v_mfma_f32_16x16x4_f32 with the point on the column (lane&15): lane (g = lane>>4, j = lane&15) supplies
// f_p[4g..4g+3] -- one 16-byte load per lane, reused by every query -- and the matching weight columns
// 4g+s (the MFMA k index is only a summation index).  The 3 coordinate taps and the bias are a fifth MFMA over
// k = (rel_x, rel_y, rel_z, 1); ReLU and the 16->1 contraction run on the accumulator layout (rows 4g..4g+3 per lane); the four row groups of the four
// tiles are combined with a two-step butterfly so lane (g,j) ends up with the logit of point 16g+j and
// a wave stores 256 contiguous bytes per query.  Roofline: 4*(2*nq*N) bytes of geo/logit traffic
// against 2*nq*N*(19*16+16) flops -> 80 flop/B: MFMA/VALU-bound, ~10 GFLOP per 150k-point scene.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef MH_Q
#define MH_Q 2   // queries per wave: 2 at 3 workgroups per CU measured 209 us vs 262 us for 4 at 2 (S150k, nq=256)
#endif
#ifndef MH_MINB
#define MH_MINB 3
#endif

// value of lane (i ^ 32) / (i ^ 16): one v_permlane{32,16}_swap (VALU) instead of a trip through the LDS crossbar
__device__ __forceinline__ float mh_xor32(float v, int lane) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(lane < 32 ? r[1] : r[0]);
}
__device__ __forceinline__ float mh_xor16(float v, int lane) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane & 16) ? r[0] : r[1]);
}

template <bool USE_GEO>
__global__ __launch_bounds__(256, MH_MINB) void k_mask_head(const float* __restrict__ feat, const float* __restrict__ coords,
                                                      const float* __restrict__ geo, const float* __restrict__ qxyz,
                                                      const float* __restrict__ mx, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, int ldp, int N, int nq, int chunks,
                                                      float* __restrict__ out) {
    // ldp: 0 = four dense arrays (w1 [nq,16,19], b1 [nq,16], w2 [nq,16], b2 [nq]); else the row stride of ONE packed
    // parameter matrix the four pointers point into (the controller's output, geoformer.py:264-284)
    const int ld_w1 = ldp ? ldp : 16 * 19, ld_v = ldp ? ldp : 16, ld_s = ldp ? ldp : 1;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int qgroups = (nq + MH_Q - 1) / MH_Q;
    const int qg = wave % qgroups, chunk = wave / qgroups;
    if (chunk >= chunks) return;
    const int nblocks = (N + 63) >> 6;
    const int per = (nblocks + chunks - 1) / chunks;
    const int blk0 = chunk * per, blk1 = min(nblocks, blk0 + per);

    // per-query parameters in registers.  The three coordinate taps and the bias ride in a FIFTH MFMA whose k index
    // runs over (rel_x, rel_y, rel_z, 1): lane group g supplies component g of the relative position of point j (its
    // own coordinate of the point only -- one load, one subtract, one select) and the matching weight column
    // W1[q][j][g] (g = 3: the bias b1[q][j]).  The first version applied the taps to the accumulator rows on the VALU
    // (twelve fma per lane and tile): 42 VALU instructions per tile and query against 20 MFMA passes.
    float wf[MH_Q][4];  // A operand of the feature part: W1[q][c=j][3 + 4g + s]
    float w5[MH_Q];     // A operand of the coordinate part: W1[q][j][g] (g < 3), b1[q][j] (g = 3)
    float ww[MH_Q][4], b2q[MH_Q], qc[MH_Q], mq[MH_Q];
#pragma unroll
    for (int t = 0; t < MH_Q; t++) {
        const int q = min(qg * MH_Q + t, nq - 1);
        const float* W = w1 + (size_t)q * ld_w1;
#pragma unroll
        for (int s = 0; s < 4; s++) wf[t][s] = W[j * 19 + 3 + 4 * g + s];
        w5[t] = g < 3 ? W[j * 19 + g] : b1[(size_t)q * ld_v + j];
#pragma unroll
        for (int r = 0; r < 4; r++) ww[t][r] = w2[(size_t)q * ld_v + 4 * g + r];
        b2q[t] = b2[(size_t)q * ld_s];
        qc[t] = g < 3 ? qxyz[q * 3 + g] : 0.f;
        mq[t] = USE_GEO ? mx[q] : 0.f;
    }

    // Operands of one 64-point block: features (B operand) and this lane group's coordinate of column j for the four
    // 16-point tiles, and the block's geodesic distances for this wave's queries.  Everything a block needs is
    // requested in one go, one block ahead of the arithmetic (a load inside the per-tile code would expose one HBM
    // round trip per tile and query: an earlier version of this loop spent most of its time there).
    struct Block {
        float4 f[4];
        float pc[4];
        float gd[MH_Q][4];
    };
    auto fetch = [&](int blk, Block& B) {
        const int p0 = blk * 64;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int pc = min(p0 + 16 * t + j, N - 1);
            B.f[t] = *reinterpret_cast<const float4*>(feat + (size_t)pc * 16 + 4 * g);
            B.pc[t] = coords[(size_t)pc * 3 + min(g, 2)];
#pragma unroll
            for (int u = 0; u < MH_Q; u++)
                B.gd[u][t] = USE_GEO ? geo[(size_t)min(qg * MH_Q + u, nq - 1) * N + pc] : 0.f;
        }
    };
    Block cur, nxt;
    if (blk0 < blk1) fetch(blk0, cur);
    for (int blk = blk0; blk < blk1; blk++) {
        const int p0 = blk * 64;
        if (blk + 1 < blk1) fetch(blk + 1, nxt);
#pragma unroll
        for (int t = 0; t < MH_Q; t++) {
            const int q = qg * MH_Q + t;
            float part[4];
#pragma unroll
            for (int tl = 0; tl < 4; tl++) {
                float rel = qc[t] - cur.pc[tl];
                if (USE_GEO) {
                    // unreachable point: rel += sqrt(max_geo_q) * sign(rel), sign(0) = 0  (branch-free)
                    const float m = cur.gd[t][tl] < 0.f ? mq[t] : 0.f;
                    rel = rel + m * (rel > 0.f ? 1.f : (rel < 0.f ? -1.f : 0.f));
                }
                if (g == 3) rel = 1.0f;  // the bias row
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w5[t], rel, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][0], cur.f[tl].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][1], cur.f[tl].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][2], cur.f[tl].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][3], cur.f[tl].w, acc, 0, 0, 0);
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) s = fmaf(ww[t][r], fmaxf(acc[r], 0.f), s);
                part[tl] = s;
            }
            // butterfly over the four row groups: afterwards lane (g,j) holds the full sum of tile g
            {
                // step 1 (xor 32): groups {0,1} keep tiles {0,1}, groups {2,3} keep tiles {2,3}
                const bool hi = g >= 2;
                const float send0 = hi ? part[0] : part[2], send1 = hi ? part[1] : part[3];
                const float r0 = mh_xor32(send0, lane), r1 = mh_xor32(send1, lane);
                const float k0 = (hi ? part[2] : part[0]) + r0, k1 = (hi ? part[3] : part[1]) + r1;
                // step 2 (xor 16): even group keeps the first of its pair, odd group the second
                const bool odd = g & 1;
                const float send = odd ? k0 : k1;
                const float r = mh_xor16(send, lane);
                const float tot = (odd ? k1 : k0) + r + b2q[t];
                const int p = p0 + 16 * g + j;
                if (p < N && q < nq) out[(size_t)q * N + p] = tot;
            }
        }
        cur = nxt;
    }
}

extern "C" int gf_mask_head_packed(const float* feat, const float* coords, const float* geo, const float* qxyz,
                                   const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                                   const float* b2, int ldp, int N, int nq, int C, float* out, void* stream);
extern "C" int gf_mask_head(const float* feat, const float* coords, const float* geo, const float* qxyz,
                            const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                            const float* b2, int N, int nq, int C, float* out, void* stream) {
    return gf_mask_head_packed(feat, coords, geo, qxyz, sqrt_max_geo, w1, b1, w2, b2, 0, N, nq, C, out, stream);
}

// the four per-query parameter blocks as pointers INTO one matrix with row stride ldp (floats): the controller's
// [nq, 16*19 + 16 + 16 + 1] output read in place, no split / reshape / contiguous copies in front of the launch
extern "C" int gf_mask_head_packed(const float* feat, const float* coords, const float* geo, const float* qxyz,
                                   const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                                   const float* b2, int ldp, int N, int nq, int C, float* out, void* stream) {
    GF_CHECK_ARG(ldp >= 0, "gf_mask_head: negative parameter stride");
    GF_CHECK_ARG(C == 16, "gf_mask_head: only the 16-channel mask head (m=16) is implemented, got C=%d", C);
    GF_CHECK_ARG(N >= 0 && nq >= 0, "gf_mask_head: bad sizes");
    GF_CHECK_ARG((geo == nullptr) == (sqrt_max_geo == nullptr), "gf_mask_head: geo and sqrt_max_geo come together");
    if (N == 0 || nq == 0) return GF_OK;
    const int qgroups = (nq + MH_Q - 1) / MH_Q;
    const int nblocks = (N + 63) / 64;
    // enough waves for ~8 per SIMD; every wave sweeps a contiguous range of 64-point blocks
    int chunks = (256 * 4 * 8 + qgroups - 1) / qgroups;
    if (chunks > nblocks) chunks = nblocks;
    if (chunks < 1) chunks = 1;
    const long long waves = (long long)qgroups * chunks;
    dim3 grid((unsigned)((waves + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (geo)
        hipLaunchKernelGGL((k_mask_head<true>), grid, dim3(256), 0, st, feat, coords, geo, qxyz, sqrt_max_geo, w1, b1,
                           w2, b2, ldp, N, nq, chunks, out);
    else
        hipLaunchKernelGGL((k_mask_head<false>), grid, dim3(256), 0, st, feat, coords, geo, qxyz, sqrt_max_geo, w1, b1,
                           w2, b2, ldp, N, nq, chunks, out);
    GF_CHECK_LAUNCH("gf_mask_head");
    return GF_OK;
}
