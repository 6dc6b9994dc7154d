// Proposal extraction of the eval forward, fused (GeoFormer.generate_proposal,
// model/geoformer/geoformer.py:193-262).
//
// The reference runs ~40 PyTorch launches over the [nq, N] mask logits (sigmoid, compare, three reductions,
// an [nq,N]x[N,classes] product, nonzero over the selected masks, index_put).  Here:
//   k_proposal_stats   one workgroup per query: a single sweep over its logit row gives the point count,
//                      the mean mask probability and the mean semantic probability of the predicted class;
//                      the class soft-max/arg-max, the score product and the three acceptance tests finish in
//                      the same launch.  HBM: 4*nq*N bytes read once.
//   k_proposal_scatter the accepted rows are swept once more and written as 0/1 rows over the scene's points
//                      (proposals[i, fg_idxs[p]] = 1), no nonzero()/index_put round trip.
// mask membership is  sigmoid(logit) >= logit_thresh  with sigmoid = 1 / (1 + expf(-x)), evaluated by the same
// device function in both kernels.
#include "common.h"

#define PR_THREADS 1024

__device__ __forceinline__ float pr_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(PR_THREADS) void k_proposal_stats(const float* __restrict__ logits,
                                                               const float* __restrict__ cls_logits,
                                                               const float* __restrict__ sem_prob, int N, int ncls,
                                                               float logit_thresh, float score_thresh,
                                                               int npoint_thresh, int min_class,
                                                               int* __restrict__ cls_pred_out,
                                                               int* __restrict__ npoints_out,
                                                               float* __restrict__ scores_out,
                                                               int* __restrict__ final_out) {
    __shared__ int s_cls;
    __shared__ float s_cls_score;
    __shared__ int r_cnt[PR_THREADS / 64];
    __shared__ float r_prob[PR_THREADS / 64], r_sem[PR_THREADS / 64];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        // soft-max over the classes and its arg-max (first maximum), geoformer.py:215-216
        const float* c = cls_logits + (size_t)q * ncls;
        float mx = c[0];
        int arg = 0;
        for (int k = 1; k < ncls; k++)
            if (c[k] > mx) {
                mx = c[k];
                arg = k;
            }
        float den = 0.f;
        for (int k = 0; k < ncls; k++) den += expf(c[k] - mx);
        s_cls = arg;
        s_cls_score = 1.0f / den;  // exp(0) / sum
    }
    __syncthreads();
    const int cls = s_cls;
    const float* row = logits + (size_t)q * N;
    const float* sem_row = sem_prob + (size_t)cls * N;  // class-major: the predicted class is one contiguous row
    int cnt = 0;
    float sp = 0.f, ss = 0.f;
    // four independent points per thread and trip (loads of a trip go out together; the class probability is
    // fetched for every point, from a clamped index, and selected afterwards)
    for (int p0 = tid; p0 < N; p0 += 4 * PR_THREADS) {
        float x[4], sv[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int p = p0 + e * PR_THREADS;
            const int pc = p < N ? p : N - 1;
            x[e] = row[pc];
            sv[e] = sem_row[pc];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float pr = pr_sigmoid(x[e]);
            const bool in = (p0 + e * PR_THREADS) < N && pr >= logit_thresh;
            cnt += in ? 1 : 0;
            sp += in ? pr : 0.f;
            ss += in ? sv[e] : 0.f;
        }
    }
    cnt = gf_wave_sum_i(cnt);  // (the __shfl_xor butterflies without the LDS crossbar: common.h)
    sp = gf_wave_sum(sp);
    ss = gf_wave_sum(ss);
    if (lane == 0) {
        r_cnt[wave] = cnt;
        r_prob[wave] = sp;
        r_sem[wave] = ss;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        float a = 0.f, b = 0.f;
        for (int w = 0; w < PR_THREADS / 64; w++) {
            n += r_cnt[w];
            a += r_prob[w];
            b += r_sem[w];
        }
        const float den = (float)n + 1e-6f;
        const float mask_score = a / den, sem_score = b / den;
        cls_pred_out[q] = cls;
        npoints_out[q] = n;
        scores_out[q] = mask_score * sqrtf(s_cls_score) * sem_score;
        final_out[q] = (cls >= min_class) && (n >= npoint_thresh) && (mask_score >= score_thresh);
    }
}

// Few-shot variant (GeoFormerFS.generate_proposal, model/geoformer/geoformer_fs.py:205-238): no class head -- the
// query's cosine similarity to the support prototype takes the class score's place.
__global__ __launch_bounds__(PR_THREADS) void k_proposal_stats_fs(const float* __restrict__ logits,
                                                                  const float* __restrict__ sim, int N,
                                                                  float logit_thresh, float score_thresh,
                                                                  int npoint_thresh, float sim_thresh,
                                                                  int* __restrict__ npoints_out,
                                                                  float* __restrict__ scores_out,
                                                                  int* __restrict__ final_out) {
    __shared__ int r_cnt[PR_THREADS / 64];
    __shared__ float r_prob[PR_THREADS / 64];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (size_t)q * N;
    int cnt = 0;
    float sp = 0.f;
    for (int p0 = tid; p0 < N; p0 += 4 * PR_THREADS) {
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int p = p0 + e * PR_THREADS;
            x[e] = row[p < N ? p : N - 1];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float pr = pr_sigmoid(x[e]);
            const bool in = (p0 + e * PR_THREADS) < N && pr >= logit_thresh;
            cnt += in ? 1 : 0;
            sp += in ? pr : 0.f;
        }
    }
    cnt = gf_wave_sum_i(cnt);
    sp = gf_wave_sum(sp);
    if (lane == 0) {
        r_cnt[wave] = cnt;
        r_prob[wave] = sp;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        float a = 0.f;
        for (int w = 0; w < PR_THREADS / 64; w++) {
            n += r_cnt[w];
            a += r_prob[w];
        }
        const float mask_score = a / ((float)n + 1e-6f), s = sim[q];
        npoints_out[q] = n;
        scores_out[q] = mask_score * sqrtf(s);  // NaN for a negative similarity, like torch.pow(sim, 0.5); never accepted
        final_out[q] = (s >= sim_thresh) && (n >= npoint_thresh) && (mask_score >= score_thresh);
    }
}

__global__ __launch_bounds__(256) void k_proposal_scatter(const float* __restrict__ logits,
                                                          const int* __restrict__ sel, int N,
                                                          const long long* __restrict__ fg_idxs, float logit_thresh,
                                                          int num_points, int* __restrict__ proposals) {
    const int i = blockIdx.y;
    const float* row = logits + (size_t)sel[i] * N;
    int* out = proposals + (size_t)i * num_points;
    const int stride = gridDim.x * 256;
    for (int p0 = blockIdx.x * 256 + threadIdx.x; p0 < N; p0 += 4 * stride) {
        float x[4];
        long long dst[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int p = p0 + e * stride;
            const int pc = p < N ? p : N - 1;
            x[e] = row[pc];
            dst[e] = fg_idxs[pc];
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
            if ((p0 + e * stride) < N && pr_sigmoid(x[e]) >= logit_thresh) out[dst[e]] = 1;
    }
}

extern "C" int gf_proposal_stats(const float* mask_logits, const float* cls_logits, const float* sem_prob, int nq,
                                 int N, int ncls, float logit_thresh, float score_thresh, int npoint_thresh,
                                 int min_class, int* cls_pred, int* npoints, float* scores, int* final_mask,
                                 void* stream) {
    GF_CHECK_ARG(nq >= 0 && N >= 0 && ncls >= 1, "gf_proposal_stats: bad sizes nq=%d N=%d ncls=%d", nq, N, ncls);
    if (nq == 0) return GF_OK;
    hipLaunchKernelGGL(k_proposal_stats, dim3(nq), dim3(PR_THREADS), 0, (hipStream_t)stream, mask_logits, cls_logits,
                       sem_prob, N, ncls, logit_thresh, score_thresh, npoint_thresh, min_class, cls_pred, npoints,
                       scores, final_mask);
    GF_CHECK_LAUNCH("gf_proposal_stats");
    return GF_OK;
}

extern "C" int gf_proposal_stats_fs(const float* mask_logits, const float* sim, int nq, int N, float logit_thresh,
                                    float score_thresh, int npoint_thresh, float sim_thresh, int* npoints,
                                    float* scores, int* final_mask, void* stream) {
    GF_CHECK_ARG(nq >= 0 && N >= 0, "gf_proposal_stats_fs: bad sizes nq=%d N=%d", nq, N);
    if (nq == 0) return GF_OK;
    hipLaunchKernelGGL(k_proposal_stats_fs, dim3(nq), dim3(PR_THREADS), 0, (hipStream_t)stream, mask_logits, sim, N,
                       logit_thresh, score_thresh, npoint_thresh, sim_thresh, npoints, scores, final_mask);
    GF_CHECK_LAUNCH("gf_proposal_stats_fs");
    return GF_OK;
}

extern "C" int gf_proposal_scatter(const float* mask_logits, const int* sel, int n_sel, int N,
                                   const long long* fg_idxs, float logit_thresh, int num_points, int* proposals,
                                   void* stream) {
    GF_CHECK_ARG(n_sel >= 0 && N >= 0 && num_points >= 0, "gf_proposal_scatter: bad sizes");
    if (n_sel == 0 || N == 0) return GF_OK;
    int bx = gf_div_up(N, 256 * 4);
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_proposal_scatter, dim3(bx, n_sel), dim3(256), 0, (hipStream_t)stream, mask_logits, sel, N,
                       fg_idxs, logit_thresh, num_points, proposals);
    GF_CHECK_LAUNCH("gf_proposal_scatter");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// Pairwise intersections of the proposal masks for matrix NMS (util/utils_3d.py:95-141: the
// einsum("nc,mc->nm") over [n, N] float masks, 9.8 GFLOP at 256 x 150k).  The masks are 0/1, so the products are
// exact integers: pack every row into bits (one ballot per 64 points) and count AND-ed words.  Bit-exact with
// the fp32 einsum (counts < 2^24), ~0.1 GB of traffic instead of a dense GEMM over 150 MB of floats.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mask_pack_bits(const int32_t* __restrict__ masks, int n, int N, int W2,
                                                        unsigned long long* __restrict__ bits) {
    // one wave per (row, 64-point block)
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (long long)n * W2) return;
    const int row = (int)(wave / W2), blk = (int)(wave - (long long)row * W2);
    const int p = blk * 64 + lane;
    const bool on = p < N && masks[(size_t)row * N + p] != 0;
    const unsigned long long b = __ballot(on);
    if (lane == 0) bits[(size_t)row * W2 + blk] = b;
}

__global__ __launch_bounds__(256) void k_mask_intersections(const unsigned long long* __restrict__ bits, int n, int W2,
                                                            int32_t* __restrict__ inter) {
    // one wave per pair (i, j >= i); mirrored on store
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (long long)n * n) return;
    const int i = (int)(wave / n), j = (int)(wave - (long long)i * n);
    if (j < i) return;
    const unsigned long long *a = bits + (size_t)i * W2, *b = bits + (size_t)j * W2;
    int c = 0;
    for (int w = lane; w < W2; w += 64) c += __popcll(a[w] & b[w]);
    c = gf_wave_sum_i(c);
    if (lane == 0) {
        inter[(size_t)i * n + j] = c;
        inter[(size_t)j * n + i] = c;
    }
}

extern "C" size_t gf_mask_intersections_scratch_bytes(int n, int N) {
    return (size_t)(n > 0 ? n : 0) * ((size_t)(N > 0 ? N : 0) + 63) / 64 * sizeof(unsigned long long) + 64;
}

extern "C" int gf_mask_intersections(const int32_t* masks, int n, int N, void* scratch, int32_t* inter, void* stream) {
    GF_CHECK_ARG(n >= 0 && N >= 0, "gf_mask_intersections: bad sizes");
    if (n == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    const int W2 = (N + 63) / 64;
    unsigned long long* bits = (unsigned long long*)scratch;
    if (W2 > 0) {
        const long long waves = (long long)n * W2;
        hipLaunchKernelGGL(k_mask_pack_bits, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, masks, n, N, W2, bits);
    }
    const long long pairs = (long long)n * n;
    hipLaunchKernelGGL(k_mask_intersections, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, st, bits, n, W2, inter);
    GF_CHECK_LAUNCH("gf_mask_intersections");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// Ingredients of the decoder's relative position embedding (GeoFormer.forward_decoder, geoformer.py:619-651), for
// one scene:  geo_ctx[q, j] = geo[q, inds[j]]  (geodesic distance from query q to context point j) and
// max_geo[q] = max_j geo_ctx[q, j], replaced by the largest row maximum where a row is entirely unreachable (< 0).
// PyTorch: index, copy, max, max, compare, where, copy -- seven launches between the set abstraction and the first
// cross-attention; here one launch per step of the dependency (row maxima, then the fix-up over nq values).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_relpos_gather(const float* __restrict__ geo, const int32_t* __restrict__ inds,
                                                       int n, int nc, float* __restrict__ geo_ctx,
                                                       float* __restrict__ row_max) {
    __shared__ float s_m[4];
    const int q = blockIdx.x;
    const float* g = geo + (size_t)q * n;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < nc; j += 256) {
        const float v = g[inds[j]];
        geo_ctx[(size_t)q * nc + j] = v;
        m = fmaxf(m, v);
    }
    m = gf_wave_max(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) row_max[q] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

__global__ __launch_bounds__(1024) void k_relpos_fix(float* __restrict__ row_max, int nq) {
    __shared__ float s_m[16];
    float m = -INFINITY;
    for (int q = threadIdx.x; q < nq; q += 1024) m = fmaxf(m, row_max[q]);
    m = gf_wave_max(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    float all = s_m[0];
#pragma unroll
    for (int w = 1; w < 16; w++) all = fmaxf(all, s_m[w]);
    for (int q = threadIdx.x; q < nq; q += 1024)
        if (row_max[q] < 0.f) row_max[q] = all;
}

extern "C" int gf_relpos_prepare(const float* geo, const int32_t* inds, int nq, int n, int nc, float* geo_ctx,
                                 float* max_geo, void* stream) {
    GF_CHECK_ARG(geo && inds && geo_ctx && max_geo, "gf_relpos_prepare: null argument");
    GF_CHECK_ARG(nq >= 0 && n >= 1 && nc >= 1, "gf_relpos_prepare: bad sizes");
    if (nq == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_relpos_gather, dim3(nq), dim3(256), 0, st, geo, inds, n, nc, geo_ctx, max_geo);
    hipLaunchKernelGGL(k_relpos_fix, dim3(1), dim3(1024), 0, st, max_geo, nq);
    GF_CHECK_LAUNCH("gf_relpos_prepare");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// Accepted queries in ascending order, with their classes and scores, on the device (generate_proposal,
// geoformer.py:236-243: `final = ...; cls_final = cls_pred[final]; scores_final = scores[final]`): the host then
// reads back ONE integer (how many) instead of the flags, and the membership scatter takes the list as it is.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_proposal_select(const int32_t* __restrict__ final_, const int32_t* __restrict__ cls_pred,
                                                          const float* __restrict__ scores, int nq,
                                                          int32_t* __restrict__ sel, long long* __restrict__ cls_out,
                                                          float* __restrict__ scores_out, int32_t* __restrict__ count) {
    __shared__ int s_w[16];
    __shared__ int s_run;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nq; base += 1024) {
        const int q = base + threadIdx.x;
        const bool f = q < nq && final_[q] != 0;
        const unsigned long long bal = __ballot(f);
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int c = s_w[w];
            if (w < wave) before += c;
            total += c;
        }
        const int run = s_run;
        if (f) {
            const int pos = run + before + __popcll(bal & ((1ull << lane) - 1ull));
            sel[pos] = q;
            cls_out[pos] = cls_pred[q];
            scores_out[pos] = scores[q];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = s_run;
}

extern "C" int gf_proposal_select(const int32_t* final_, const int32_t* cls_pred, const float* scores, int nq,
                                  int32_t* sel, long long* cls_out, float* scores_out, int32_t* d_count, void* stream) {
    GF_CHECK_ARG(final_ && cls_pred && scores && sel && cls_out && scores_out && d_count && nq >= 0,
                 "gf_proposal_select: bad arguments");
    hipLaunchKernelGGL(k_proposal_select, dim3(1), dim3(1024), 0, (hipStream_t)stream, final_, cls_pred, scores, nq, sel,
                       cls_out, scores_out, d_count);
    GF_CHECK_LAUNCH("gf_proposal_select");
    return GF_OK;
}
