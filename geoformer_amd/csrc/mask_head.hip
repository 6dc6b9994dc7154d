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

// max(x, 0) as exactly one instruction.  fmaxf(x, 0.f) compiles to v_max(x, x) + v_max(0, .): IEEE maxnum must quiet a
// signalling NaN and the compiler cannot see that an MFMA result never is one (fmed3 is folded back to that pair; inline
// assembly would read the MFMA's result without the wait states the compiler inserts for instructions it knows).  The
// INTEGER maximum of the bit pattern with 0 is the same function for every non-NaN float: negative floats (and -0) are
// negative integers, positive floats keep their order.
__device__ __forceinline__ float mh_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// ---- fp32 products on the bf16 matrix pipe (round 4) ----
// The fp32 MFMA (v_mfma_f32_16x16x4_f32: 32 cycles for K = 4) is the slowest matrix instruction of the chip and, measured
// on this kernel, does not overlap with the vector instructions around it: the loop without its MFMAs takes 64 us, with
// them 135 (62 us of MFMA time at the peak rate).  An fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8
// significant bits, by truncation: x = hi + mid + lo), so a product of two fp32 numbers is the sum of nine bf16 products,
// of which the six largest carry everything above 2^-24 of the result (dropped: mid*lo, lo*mid, lo*lo <= 3 * 2^-24
// relative -- the size of ONE fp32 rounding); bf16 products are exact in the MFMA's fp32 accumulator.  One
// v_mfma_f32_16x16x16_bf16 (16 cycles, K = 16) does the whole 16-channel feature product of a tile per piece pair:
// 6 x 16 cycles instead of 4 x 32.  The features' three pieces are written once per scene by k_mh_split (every query
// shares them), the generated weights are split once per wave.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short mh_s8 __attribute__((ext_vector_type(8)));
typedef __bf16 mh_bf16x8 __attribute__((ext_vector_type(8)));
// Two 16-channel piece products in ONE v_mfma_f32_16x16x32_bf16 (16 cycles, the same as the K = 16 form): the K index
// is a summation index, so lane group g carries channels 4g..4g+3 of the first pair in elements 0-3 and of the second pair
// in elements 4-7 -- a1 . b1 + a2 . b2 of the lane's 4-channel slices.
__device__ __forceinline__ f32x4 mh_mfma2(bf16x4 a1, bf16x4 a2, bf16x4 b1, bf16x4 b2, f32x4 acc) {
    const mh_s8 a = __builtin_shufflevector(a1, a2, 0, 1, 2, 3, 4, 5, 6, 7), b = __builtin_shufflevector(b1, b2, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mh_bf16x8, a), __builtin_bit_cast(mh_bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mh_split3(float x, unsigned& h, unsigned& m, unsigned& l) {  // the pieces as bf16 bit patterns
    const unsigned hb = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hb);  // exact
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);  // exact, at most 8 significant bits left
    h = hb >> 16;
    m = mb >> 16;
    l = __float_as_uint(r2) >> 16;
}
__device__ __forceinline__ void mh_split3x4(const float (&x)[4], bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        unsigned a, b, c;
        mh_split3(x[s], a, b, c);
        h[s] = (short)a;
        m[s] = (short)b;
        l[s] = (short)c;
    }
}
// feat fp32 [N,16] -> fs ushort [3][N][16] (hi, mid, lo)
__global__ void k_mh_split(const float* __restrict__ feat, int n16, unsigned short* __restrict__ fs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    unsigned h, m, l;
    mh_split3(feat[i], h, m, l);
    fs[i] = (unsigned short)h;
    fs[(size_t)n16 + i] = (unsigned short)m;
    fs[2 * (size_t)n16 + i] = (unsigned short)l;
}

// TINY: fewer than 64 points (point indices clamped per lane).  Otherwise the last block of the scene is shifted back to
// end at the last point (p0 = N - 64: the overlap is computed and stored twice, same values), so that no index needs a
// clamp and a block's sixteen loads are constant offsets from six running addresses.
template <bool USE_GEO, bool SPLIT, bool TINY>
__global__ __launch_bounds__(256, MH_MINB) void k_mask_head(const float* __restrict__ feat, const float* __restrict__ coords,
                                                      const float* __restrict__ geo, const float* __restrict__ qxyz,
                                                      const float* __restrict__ mx, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, int ldp, int N, int nq, int qmod,
                                                      int chunks, const unsigned short* __restrict__ fs,
                                                      float* __restrict__ out) {
    // nq = E * qmod rows of generated parameters and of logits: E episodes (few-shot re-queries of one cached scene,
    // test_fs.py:157-174) over the SAME qmod queries -- row qq reads the geodesic row, position and maximum of query
    // qq % qmod.  qmod = nq: one episode.
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
        const int qs = q % qmod;  // the scene-side query of this parameter row
        // lane group 3 feeds the bias row of the fifth MFMA: its "relative coordinate" is the constant 1 = 1 - 0 (its
        // point coordinate is loaded as 0 and its fix-up magnitude is 0), no select per (query, tile)
        qc[t] = g < 3 ? qxyz[qs * 3 + g] : 1.f;
        mq[t] = (USE_GEO && g < 3) ? mx[qs] : 0.f;
    }
    bf16x4 wh[MH_Q], wm[MH_Q], wl[MH_Q];  // (SPLIT) the feature weights' three bf16 pieces
    if constexpr (SPLIT) {
#pragma unroll
        for (int t = 0; t < MH_Q; t++) mh_split3x4(wf[t], wh[t], wm[t], wl[t]);
    }
    int qrow[MH_Q];  // geodesic row of each of the wave's queries
#pragma unroll
    for (int u = 0; u < MH_Q; u++) qrow[u] = min(qg * MH_Q + u, nq - 1) % qmod;

    // Operands of one 64-point block: features (B operand) and this lane group's coordinate of column j for the four
    // 16-point tiles, and the block's geodesic distances for this wave's queries.  Everything a block needs is
    // requested in one go, one block ahead of the arithmetic (a load inside the per-tile code would expose one HBM
    // round trip per tile and query: an earlier version of this loop spent most of its time there).
    struct Block {
        float4 f[4];          // (!SPLIT) fp32 features of the lane's four channels
        bf16x4 fh[4], fm[4], fl[4];  // (SPLIT) their three bf16 pieces
        float pc[4];
        float gd[MH_Q][4];
    };
    const size_t n16 = (size_t)N * 16;
    auto fetch = [&](int blk, Block& B) {
        const int p0 = TINY ? blk * 64 : min(blk * 64, N - 64);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int pc = TINY ? min(p0 + 16 * t + j, N - 1) : p0 + 16 * t + j;
            if constexpr (SPLIT) {
                const size_t e = (size_t)pc * 16 + 4 * g;
                B.fh[t] = *reinterpret_cast<const bf16x4*>(fs + e);
                B.fm[t] = *reinterpret_cast<const bf16x4*>(fs + n16 + e);
                B.fl[t] = *reinterpret_cast<const bf16x4*>(fs + 2 * n16 + e);
            } else {
                B.f[t] = *reinterpret_cast<const float4*>(feat + (size_t)pc * 16 + 4 * g);
            }
            B.pc[t] = g < 3 ? coords[(size_t)pc * 3 + g] : 0.f;
#pragma unroll
            for (int u = 0; u < MH_Q; u++)
                B.gd[u][t] = USE_GEO ? geo[(size_t)qrow[u] * N + pc] : 0.f;
        }
    };
    // (two blocks per trip, the buffers alternating: `cur = nxt` at the end of a one-block loop was 42 register moves
    //  per block, a sixth of its vector instructions)
    auto compute = [&](const Block& cur, int blk) {
        const int p0 = TINY ? blk * 64 : min(blk * 64, N - 64);
#pragma unroll
        for (int t = 0; t < MH_Q; t++) {
            const int q = qg * MH_Q + t;
            float part[4];
#pragma unroll
            for (int tl = 0; tl < 4; tl++) {
                float rel = qc[t] - cur.pc[tl];
                if (USE_GEO) {
                    // unreachable point: rel += sqrt(max_geo_q) * sign(rel), sign(0) = 0.  m * sign(rel) is m with rel's
                    // sign bit (m >= 0) unless rel is zero; the two conditions meet in one scalar AND of the compare masks
                    const float ms = __uint_as_float((__float_as_uint(rel) & 0x80000000u) | __float_as_uint(mq[t]));
                    rel = rel + ((cur.gd[t][tl] < 0.f && rel != 0.f) ? ms : 0.f);
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#ifdef MH_EXP_NOMFMA  // dev experiment: the loop without its matrix instructions (results are garbage)
                acc[0] = w5[t] * rel; acc[1] = wf[t][0] * cur.pc[tl]; acc[2] = wf[t][1] * cur.pc[tl]; acc[3] = wf[t][2] * cur.pc[tl];
#else
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w5[t], rel, acc, 0, 0, 0);
                if constexpr (SPLIT) {  // the six piece pairs, smallest first, two per instruction
                    acc = mh_mfma2(wl[t], wh[t], cur.fh[tl], cur.fl[tl], acc);  // lo.hi + hi.lo
                    acc = mh_mfma2(wm[t], wm[t], cur.fm[tl], cur.fh[tl], acc);  // mid.mid + mid.hi
                    acc = mh_mfma2(wh[t], wh[t], cur.fm[tl], cur.fh[tl], acc);  // hi.mid + hi.hi
                } else {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][0], cur.f[tl].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][1], cur.f[tl].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][2], cur.f[tl].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][3], cur.f[tl].w, acc, 0, 0, 0);
                }
#endif
#ifdef MH_EXP_NOVALU  // dev experiment: the loop without the activation / contraction arithmetic
                part[tl] = acc[0];
                continue;
#endif
                float s = 0.f;
                // (mh_relu: ONE v_max; fmaxf costs a second instruction that canonicalises its operand first, 32 of the
                // loop's ~290 vector instructions)
#pragma unroll
                for (int r = 0; r < 4; r++) s = fmaf(ww[t][r], mh_relu(acc[r]), s);
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
    };
    Block bufA, bufB;
    if (blk0 < blk1) fetch(blk0, bufA);
    for (int blk = blk0; blk < blk1; blk += 2) {
        if (blk + 1 < blk1) fetch(blk + 1, bufB);
        compute(bufA, blk);
        if (blk + 1 < blk1) {
            if (blk + 2 < blk1) fetch(blk + 2, bufA);
            compute(bufB, blk + 1);
        }
    }
}

extern "C" int gf_mask_head_packed(const float* feat, const float* coords, const float* geo, const float* qxyz,
                                   const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                                   const float* b2, int ldp, int N, int nq, int C, float* out, void* stream);
extern "C" int gf_mask_head_episodes(const float* feat, const float* coords, const float* geo, const float* qxyz,
                                     const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                                     const float* b2, int ldp, int N, int nq, int E, int C, void* split_ws, float* out,
                                     void* stream);
extern "C" size_t gf_mask_head_split_bytes(int N) { return (size_t)(N > 0 ? N : 0) * 16 * 3 * sizeof(unsigned short); }
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
    return gf_mask_head_episodes(feat, coords, geo, qxyz, sqrt_max_geo, w1, b1, w2, b2, ldp, N, nq, 1, C, nullptr, out, stream);
}

// E episodes over one scene in ONE launch (test_fs.py:157-174 re-queries a cached scene once per
// (label, run)): parameters [E * nq, ...] and logits [E * nq, N], episode-major; geo [nq, N], qxyz [nq, 3],
// sqrt_max_geo [nq], feat and coords are the scene's and shared by all episodes.
// split_ws (optional, gf_mask_head_split_bytes(N), 8-byte aligned): with it the 16-channel feature product runs as six
// bf16 MFMAs over the exact three-piece split of both operands (fp32-accurate, see above); without it on the fp32 MFMA.
extern "C" int gf_mask_head_episodes(const float* feat, const float* coords, const float* geo, const float* qxyz,
                                     const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                                     const float* b2, int ldp, int N, int nq_scene, int E, int C, void* split_ws, float* out,
                                     void* stream) {
    GF_CHECK_ARG(ldp >= 0, "gf_mask_head: negative parameter stride");
    GF_CHECK_ARG(C == 16, "gf_mask_head: only the 16-channel mask head (m=16) is implemented, got C=%d", C);
    GF_CHECK_ARG(N >= 0 && nq_scene >= 0 && E >= 1, "gf_mask_head: bad sizes");
    GF_CHECK_ARG((long long)nq_scene * E < (1ll << 30), "gf_mask_head: %d episodes x %d queries", E, nq_scene);
    GF_CHECK_ARG((geo == nullptr) == (sqrt_max_geo == nullptr), "gf_mask_head: geo and sqrt_max_geo come together");
    if (N == 0 || nq_scene == 0) return GF_OK;
    const int qmod = nq_scene, nq = nq_scene * E;
    const int qgroups = (nq + MH_Q - 1) / MH_Q;
    const int nblocks = (N + 63) / 64;
    // enough waves for ~8 per SIMD; every wave sweeps a contiguous range of 64-point blocks
    int chunks = (256 * 4 * 8 + qgroups - 1) / qgroups;
    if (chunks > nblocks) chunks = nblocks;
    if (chunks < 1) chunks = 1;
    const long long waves = (long long)qgroups * chunks;
    dim3 grid((unsigned)((waves + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const unsigned short* fs = (const unsigned short*)split_ws;
    if (fs) {
        GF_CHECK_ARG(((uintptr_t)fs % 8) == 0, "gf_mask_head: split_ws must be 8-byte aligned");
        const int n16 = N * 16;
        hipLaunchKernelGGL(k_mh_split, dim3(gf_div_up(n16, 256)), dim3(256), 0, st, feat, n16, (unsigned short*)split_ws);
    }
#define MH_LAUNCH(GEO_, SPLIT_)                                                                                          \
    do {                                                                                                                 \
        if (N < 64)                                                                                                      \
            GF_LAUNCH_OP(GF_OP_MASK_HEAD, (k_mask_head<GEO_, SPLIT_, true>), grid, dim3(256), 0, st, feat, coords, geo, qxyz, \
                         sqrt_max_geo, w1, b1, w2, b2, ldp, N, nq, qmod, chunks, fs, out);                               \
        else                                                                                                             \
            GF_LAUNCH_OP(GF_OP_MASK_HEAD, (k_mask_head<GEO_, SPLIT_, false>), grid, dim3(256), 0, st, feat, coords, geo, qxyz, \
                         sqrt_max_geo, w1, b1, w2, b2, ldp, N, nq, qmod, chunks, fs, out);                               \
    } while (0)
    if (geo && fs) MH_LAUNCH(true, true);
    else if (geo) MH_LAUNCH(true, false);
    else if (fs) MH_LAUNCH(false, true);
    else MH_LAUNCH(false, false);
#undef MH_LAUNCH
    GF_CHECK_LAUNCH("gf_mask_head");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// Backward of the fused mask head (training: geoformer.py:286-324 under autograd).
//
//   given gout[q,p] = dL/dlogit:   dh[q,c,p] = gout * W2_q[c] * (h_pre > 0)
//       dFeat[p,k]  = sum_q sum_c W1f_q[c,k] dh[q,c,p]           dW2_q[c] = sum_p gout * relu(h_pre)
//       dW1_q[c,:]  = sum_p dh[q,c,p] * [rel(q,p) ; f_p]          db1_q[c] = sum_p dh[q,c,p]      db2_q = sum_p gout
//   (no gradient flows into the coordinates or the geodesic distances).
//
// Nothing of the forward is stored: h_pre is recomputed with the forward's own MFMAs, once per orientation,
// because the two reductions want the hidden activations on different operand sides:
//   k_mask_head_bwd_feat   (forward's orientation: channels on the accumulator rows, point on the column)
//       dFeat^T[k,p] += W1f^T . dh     A operand = weights, B operand = dh straight from the accumulator registers;
//       a wave owns 64 points, sums over (a slice of) the queries in registers, stores once
//   k_mask_head_bwd_param  (operands swapped: points on the rows, channel on the column)
//       dW1^T[k,c]   += X . dh^T       A operand = features / relative coordinates, B operand = dh^T from the registers;
//       a wave owns MHB_Q queries and one CHUNK of the points, sums over the chunk in registers and writes its
//       337 partial sums per query to a [chunks, nq, 337] buffer that a last small kernel adds up.
// The MFMA k index is only a summation index, so "register r of lane group g" stands for row 4g + r on either side
// and no value ever changes lanes.  (A first version did both in one point-stationary kernel and added the parameter
// gradients with fp32 atomics, 7 wave-wide atomics per (wave, query): every wave walks the queries in the same
// order, so ~500 waves hit the same 337 addresses together -- 1.1 ms per call against 45 us for the forward.)
// 9 + 13 MFMAs per (query, 16-point tile) against the forward's 5; deterministic (no atomics) when the feature
// kernel does not have to split the queries.
// ------------------------------------------------------------------------------------
#define MHB_Q 4  // queries per wave of the parameter kernel (tile data loaded once for all of them)

template <bool USE_GEO>
__global__ __launch_bounds__(256, 2) void k_mask_head_bwd_feat(const float* __restrict__ feat, const float* __restrict__ coords,
                                                               const float* __restrict__ geo, const float* __restrict__ qxyz,
                                                               const float* __restrict__ mx, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                               const float* __restrict__ gout, int ldp, int N, int nq,
                                                               int qmod, int qsplit, float* __restrict__ dfeat) {
    const int ld_w1 = ldp ? ldp : 16 * 19, ld_v = ldp ? ldp : 16;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nblocks = (N + 63) >> 6;
    const int blk = wave / qsplit, part = wave - blk * qsplit;
    if (blk >= nblocks) return;
    const int p0 = blk * 64;
    const int qper = (nq + qsplit - 1) / qsplit;
    const int q0 = part * qper, q1 = min(nq, q0 + qper);
    float4 fA[4];  // feat[p0+16t+j][4g..4g+3]
    float pcA[4];  // coords[p0+16t+j][g]
    f32x4 accF[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int pa = min(p0 + 16 * t + j, N - 1);
        fA[t] = *reinterpret_cast<const float4*>(feat + (size_t)pa * 16 + 4 * g);
        pcA[t] = coords[(size_t)pa * 3 + min(g, 2)];
        accF[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int q = q0; q < q1; q++) {
        const float* W = w1 + (size_t)q * ld_w1;
        float wf[4], wT[4], ww[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            wf[s] = W[j * 19 + 3 + 4 * g + s];    // W1[c=j][3+4g+s]
            wT[s] = W[(4 * g + s) * 19 + 3 + j];  // W1[c=4g+s][3+k=j]
            ww[s] = w2[(size_t)q * ld_v + 4 * g + s];
        }
        const float w5 = g < 3 ? W[j * 19 + g] : b1[(size_t)q * ld_v + j];
        const int qs = q % qmod;  // the scene-side query of this parameter row (episodes share geo / qxyz / mx)
        const float qcA = g < 3 ? qxyz[qs * 3 + g] : 0.f;
        const float mq = USE_GEO ? mx[qs] : 0.f;
        float goA[4], gdA[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int pa = p0 + 16 * t + j;
            goA[t] = pa < N ? gout[(size_t)q * N + pa] : 0.f;
            gdA[t] = USE_GEO ? geo[(size_t)qs * N + min(pa, N - 1)] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            float rel = qcA - pcA[t];
            if (USE_GEO) rel = rel + (gdA[t] < 0.f ? mq : 0.f) * (rel > 0.f ? 1.f : (rel < 0.f ? -1.f : 0.f));
            if (g == 3) rel = 1.0f;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};  // h_pre[c = 4g+r][p = j]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w5, rel, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], fA[t].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], fA[t].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], fA[t].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], fA[t].w, acc, 0, 0, 0);
            // dFeat^T[k][p] += sum_c W1f[c][k] dh[c][p]: step s covers c = 4*kq + s (kq = lane group of the operand lane)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const float dh = acc[s] > 0.f ? goA[t] * ww[s] : 0.f;
                accF[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wT[s], dh, accF[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {  // rows k = 4g+r, column p = j
        const int p = p0 + 16 * t + j;
        if (p < N) {
            float* dst = dfeat + (size_t)p * 16 + 4 * g;
            if (qsplit == 1) {
                *reinterpret_cast<float4*>(dst) = make_float4(accF[t][0], accF[t][1], accF[t][2], accF[t][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) atomicAdd(&dst[r], accF[t][r]);
            }
        }
    }
}

// partial[(chunk * nq + q) * 337 + ...] in the packed column order w1 (16x19) | w2 (16) | b1 (16) | b2
template <bool USE_GEO>
__global__ __launch_bounds__(256, 2) void k_mask_head_bwd_param(const float* __restrict__ feat, const float* __restrict__ coords,
                                                                const float* __restrict__ geo, const float* __restrict__ qxyz,
                                                                const float* __restrict__ mx, const float* __restrict__ w1,
                                                                const float* __restrict__ b1, const float* __restrict__ w2,
                                                                const float* __restrict__ gout, int ldp, int N, int nq,
                                                                int qmod, int chunks, float* __restrict__ partial) {
    const int ld_w1 = ldp ? ldp : 16 * 19, ld_v = ldp ? ldp : 16;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int qgroups = (nq + MHB_Q - 1) / MHB_Q;
    const int qg = wave % qgroups, chunk = wave / qgroups;
    if (chunk >= chunks) return;
    const int ntiles = (N + 15) >> 4;
    const int per = (ntiles + chunks - 1) / chunks;
    const int t0 = chunk * per, t1 = min(ntiles, t0 + per);
    float wf[MHB_Q][4], w5[MHB_Q], w2j[MHB_Q], qcA[MHB_Q], qcB[MHB_Q], mq[MHB_Q];
    f32x4 accW[MHB_Q], accWc[MHB_Q];
    float pw2[MHB_Q], pb2[MHB_Q];
#pragma unroll
    for (int u = 0; u < MHB_Q; u++) {
        const int q = min(qg * MHB_Q + u, nq - 1);
        const float* W = w1 + (size_t)q * ld_w1;
#pragma unroll
        for (int s = 0; s < 4; s++) wf[u][s] = W[j * 19 + 3 + 4 * g + s];
        w5[u] = g < 3 ? W[j * 19 + g] : b1[(size_t)q * ld_v + j];
        w2j[u] = w2[(size_t)q * ld_v + j];
        const int qs = q % qmod;
        qcA[u] = g < 3 ? qxyz[qs * 3 + g] : 0.f;
        qcB[u] = j < 3 ? qxyz[qs * 3 + j] : 0.f;
        mq[u] = USE_GEO ? mx[qs] : 0.f;
        accW[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accWc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pw2[u] = 0.f;
        pb2[u] = 0.f;
    }
    for (int tile = t0; tile < t1; tile++) {
        const int pt = tile * 16;
        const int pa = min(pt + j, N - 1);
        const float4 fA = *reinterpret_cast<const float4*>(feat + (size_t)pa * 16 + 4 * g);  // point j, features 4g..
        const float pcA = coords[(size_t)pa * 3 + min(g, 2)];
        float xT[4], pcB[4];  // feature j / coordinate j of point 4g+s
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int pb = min(pt + 4 * g + s, N - 1);
            xT[s] = feat[(size_t)pb * 16 + j];
            pcB[s] = coords[(size_t)pb * 3 + min(j, 2)];
        }
#pragma unroll
        for (int u = 0; u < MHB_Q; u++) {
            const int q = min(qg * MHB_Q + u, nq - 1);
            const int qs = q % qmod;
            const float gdA = USE_GEO ? geo[(size_t)qs * N + pa] : 0.f;
            float goB[4], gdB[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int pb = pt + 4 * g + s;
                goB[s] = pb < N ? gout[(size_t)q * N + pb] : 0.f;
                gdB[s] = USE_GEO ? geo[(size_t)qs * N + min(pb, N - 1)] : 0.f;
            }
            float rel = qcA[u] - pcA;
            if (USE_GEO) rel = rel + (gdA < 0.f ? mq[u] : 0.f) * (rel > 0.f ? 1.f : (rel < 0.f ? -1.f : 0.f));
            if (g == 3) rel = 1.0f;
            f32x4 accT = {0.f, 0.f, 0.f, 0.f};  // h_pre^T[p = 4g+r][c = j]
            accT = __builtin_amdgcn_mfma_f32_16x16x4f32(rel, w5[u], accT, 0, 0, 0);
            accT = __builtin_amdgcn_mfma_f32_16x16x4f32(fA.x, wf[u][0], accT, 0, 0, 0);
            accT = __builtin_amdgcn_mfma_f32_16x16x4f32(fA.y, wf[u][1], accT, 0, 0, 0);
            accT = __builtin_amdgcn_mfma_f32_16x16x4f32(fA.z, wf[u][2], accT, 0, 0, 0);
            accT = __builtin_amdgcn_mfma_f32_16x16x4f32(fA.w, wf[u][3], accT, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const float hT = fmaxf(accT[s], 0.f);
                pw2[u] = fmaf(goB[s], hT, pw2[u]);
                pb2[u] += goB[s];
                const float dhT = accT[s] > 0.f ? goB[s] * w2j[u] : 0.f;
                // feature block: dW1f^T[k][c] += X[k][p = 4kq+s] dh^T[p][c]
                accW[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xT[s], dhT, accW[u], 0, 0, 0);
                // coordinate / bias block: rows m = 0..2 relative coordinate, 3 the bias, rest zero
                float relB = 0.f;
                if (j < 3) {
                    relB = qcB[u] - pcB[s];
                    if (USE_GEO) relB = relB + (gdB[s] < 0.f ? mq[u] : 0.f) * (relB > 0.f ? 1.f : (relB < 0.f ? -1.f : 0.f));
                } else if (j == 3) {
                    relB = 1.0f;
                }
                accWc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(relB, dhT, accWc[u], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MHB_Q; u++) {
        const int q = qg * MHB_Q + u;
        if (q >= nq) continue;
        float* P = partial + ((size_t)chunk * nq + q) * 337;
        // rows 4g+r of the accumulators, column c = j
#pragma unroll
        for (int r = 0; r < 4; r++) P[j * 19 + 3 + 4 * g + r] = accW[u][r];
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < 3; r++) P[j * 19 + r] = accWc[u][r];
            P[320 + j] = accWc[u][3];
        }
        float t2 = pw2[u] + mh_xor32(pw2[u], lane);
        t2 = t2 + mh_xor16(t2, lane);
        if (g == 0) P[304 + j] = t2;
        float t3 = pb2[u] + mh_xor32(pb2[u], lane);
        t3 = t3 + mh_xor16(t3, lane);
        if (lane == 0) P[336] = t3;
    }
}

// dparams[q, col] (packed layout, row stride ldp >= 337) = sum over chunks of partial[chunk, q, col]
__global__ void k_mask_head_bwd_reduce(const float* __restrict__ partial, int chunks, int nq, int ldp,
                                       float* __restrict__ dparams) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * 337) return;
    const int q = t / 337, c = t - q * 337;
    float s = 0.f;
    for (int k = 0; k < chunks; k++) s += partial[((size_t)k * nq + q) * 337 + c];
    dparams[(size_t)q * ldp + c] = s;
}

extern "C" size_t gf_mask_head_bwd_scratch_floats(int N, int nq) {
    int chunks = 0;
    (void)N;
    chunks = 64;
    return (size_t)chunks * (size_t)(nq > 0 ? nq : 0) * 337;
}

// dparams fp32 [nq, ldp] in the packed column order (w1 | w2 | b1 | b2, ldp >= 337) is OVERWRITTEN; dfeat fp32 [N,16]
// must be zero on entry when the queries are split (always pass it zeroed); scratch: gf_mask_head_bwd_scratch_floats.
extern "C" int gf_mask_head_bwd_episodes(const float* feat, const float* coords, const float* geo, const float* qxyz,
                                         const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                                         const float* gout, int ldp, int N, int nq_scene, int E, int C, float* dparams,
                                         float* dfeat, float* scratch, void* stream);
extern "C" int gf_mask_head_bwd(const float* feat, const float* coords, const float* geo, const float* qxyz,
                                const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                                const float* gout, int ldp, int N, int nq, int C, float* dparams, float* dfeat,
                                float* scratch, void* stream) {
    return gf_mask_head_bwd_episodes(feat, coords, geo, qxyz, sqrt_max_geo, w1, b1, w2, gout, ldp, N, nq, 1, C, dparams,
                                     dfeat, scratch, stream);
}

// E episodes over one scene (the decoder layers of a training step: same features, coordinates, geodesic rows and
// query positions, E sets of generated parameters): parameters / gout / dparams have E * nq_scene rows, episode-major;
// dfeat is the sum over all of them.  scratch: gf_mask_head_bwd_scratch_floats(N, E * nq_scene).
extern "C" int gf_mask_head_bwd_episodes(const float* feat, const float* coords, const float* geo, const float* qxyz,
                                         const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                                         const float* gout, int ldp, int N, int nq_scene, int E, int C, float* dparams,
                                         float* dfeat, float* scratch, void* stream) {
    GF_CHECK_ARG(E >= 1 && nq_scene >= 0 && (long long)nq_scene * E < (1ll << 30), "gf_mask_head_bwd: %d episodes x %d queries",
                 E, nq_scene);
    const int nq = nq_scene * E, qmod = nq_scene > 0 ? nq_scene : 1;
    GF_CHECK_ARG(ldp >= 337, "gf_mask_head_bwd: packed parameters expected (row stride >= 337), got %d", ldp);
    GF_CHECK_ARG(C == 16, "gf_mask_head_bwd: only the 16-channel mask head (m=16) is implemented, got C=%d", C);
    GF_CHECK_ARG(N >= 0 && nq >= 0, "gf_mask_head_bwd: bad sizes");
    GF_CHECK_ARG((geo == nullptr) == (sqrt_max_geo == nullptr), "gf_mask_head_bwd: geo and sqrt_max_geo come together");
    GF_CHECK_ARG(dparams && dfeat && scratch, "gf_mask_head_bwd: null output");
    if (N == 0 || nq == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nblocks = (N + 63) / 64;
    // feature gradient: ~2 waves per SIMD; split the queries only when the points alone do not give them
    int qsplit = (2 * 256 * 4 + nblocks - 1) / nblocks;
    if (qsplit > nq) qsplit = nq;
    if (qsplit > 16) qsplit = 16;
    if (qsplit < 1) qsplit = 1;
    {
        const long long waves = (long long)nblocks * qsplit;
        dim3 grid((unsigned)((waves + 3) / 4));
        if (geo)
            GF_LAUNCH_OP(GF_OP_MASK_HEAD_BWD_FEAT, (k_mask_head_bwd_feat<true>), grid, dim3(256), 0, st, feat, coords, geo, qxyz,
                         sqrt_max_geo, w1, b1, w2, gout, ldp, N, nq, qmod, qsplit, dfeat);
        else
            GF_LAUNCH_OP(GF_OP_MASK_HEAD_BWD_FEAT, (k_mask_head_bwd_feat<false>), grid, dim3(256), 0, st, feat, coords, geo, qxyz,
                         sqrt_max_geo, w1, b1, w2, gout, ldp, N, nq, qmod, qsplit, dfeat);
    }
    // parameter gradients: (query group, point chunk) per wave
    const int qgroups = (nq + MHB_Q - 1) / MHB_Q;
    const int ntiles = (N + 15) / 16;
    int chunks = (2 * 256 * 4 + qgroups - 1) / qgroups;
    if (chunks > 64) chunks = 64;
    if (chunks > ntiles) chunks = ntiles;
    if (chunks < 1) chunks = 1;
    {
        const long long waves = (long long)qgroups * chunks;
        dim3 grid((unsigned)((waves + 3) / 4));
        if (geo)
            GF_LAUNCH_OP(GF_OP_MASK_HEAD_BWD_PARAM, (k_mask_head_bwd_param<true>), grid, dim3(256), 0, st, feat, coords, geo,
                         qxyz, sqrt_max_geo, w1, b1, w2, gout, ldp, N, nq, qmod, chunks, scratch);
        else
            GF_LAUNCH_OP(GF_OP_MASK_HEAD_BWD_PARAM, (k_mask_head_bwd_param<false>), grid, dim3(256), 0, st, feat, coords, geo,
                         qxyz, sqrt_max_geo, w1, b1, w2, gout, ldp, N, nq, qmod, chunks, scratch);
    }
    hipLaunchKernelGGL(k_mask_head_bwd_reduce, dim3(gf_div_up((long long)nq * 337, 256)), dim3(256), 0, st, scratch, chunks,
                       nq, ldp, dparams);
    GF_CHECK_LAUNCH("gf_mask_head_bwd");
    return GF_OK;
}
