// Dice and focal loss of the matched (query, instance) pairs of one scene and one decoder layer, fused (gfx950).
//
//   reference: InstSetCriterion.single_layer_loss -> compute_dice_loss / compute_sigmoid_focal_loss on the matched rows
//   (criterion.py:137-190, 26-58): with x = mask_logits[match_q[k]], t = inst_masks[k] over the scene's n points,
//       p      = sigmoid(x)
//       dice_k = 1 - (2 sum(p t) + 1) / (sum(p) + sum(t) + 1)
//       ce     = max(x, 0) - x t + log1p(exp(-|x|))                 (binary_cross_entropy_with_logits)
//       p_t    = p t + (1 - p)(1 - t)
//       f      = (0.25 t + 0.75 (1 - t)) ce (1 - p_t)^2 ,  focal_k = mean_j f
//       dice   = sum_k dice_k / (n_match + 1e-6) ,  focal = sum_k focal_k / (n_match + 1e-6)      (matched k only)
//   As PyTorch operators this is ~25 launches forward and ~50 backward over [K, n] tensors per (scene, layer) --
//   24 times per batch-4 training step, 1800 launches of 3-5 us that the host cannot issue faster than the device
//   finishes them.  Here: one launch for the four row sums, one tiny one for the two scalars, one for the gradient.
#include "common.h"

namespace {

constexpr int PL_THREADS = 256;
constexpr int PL_SPLIT = 8;  // workgroups per row (a row is ~30 000 points, a scene has ~20 matched rows)

__device__ __forceinline__ float pl_block_sum(float v, float* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PL_THREADS / 64; i++) s += red[i];
    return s;
}

// part[k][s] = (sum p t, sum p, sum t, sum f) over the s-th segment of instance row k (zeros for an unmatched row)
__global__ __launch_bounds__(PL_THREADS) void k_pair_loss_sums(const float* __restrict__ logits,
                                                                const float* __restrict__ inst,
                                                                const int32_t* __restrict__ match_q, int n,
                                                                float4* __restrict__ part) {
    __shared__ float red[PL_THREADS / 64];
    const int k = blockIdx.x, sp = blockIdx.y;
    const int q = match_q[k];
    if (q < 0) {
        if (threadIdx.x == 0) part[k * PL_SPLIT + sp] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int seg = (n + PL_SPLIT - 1) / PL_SPLIT;
    const int j0 = sp * seg, j1 = min(n, j0 + seg);
    const float* x = logits + (size_t)q * n;
    const float* t = inst + (size_t)k * n;
    float a = 0.f, b = 0.f, c = 0.f, f = 0.f;
    for (int j = j0 + threadIdx.x; j < j1; j += PL_THREADS) {
        const float xv = x[j], tv = t[j];
        const float p = 1.f / (1.f + expf(-xv));
        const float ce = fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
        const float pt = p * tv + (1.f - p) * (1.f - tv);
        const float om = 1.f - pt;
        a += p * tv;
        b += p;
        c += tv;
        f += (0.25f * tv + 0.75f * (1.f - tv)) * ce * (om * om);
    }
    a = pl_block_sum(a, red);
    b = pl_block_sum(b, red);
    c = pl_block_sum(c, red);
    f = pl_block_sum(f, red);
    if (threadIdx.x == 0) part[k * PL_SPLIT + sp] = make_float4(a, b, c, f);
}

// sums[k] = the row sums (segments added in a fixed order), out[0] = dice, out[1] = focal
__global__ __launch_bounds__(PL_THREADS) void k_pair_loss_final(const float4* __restrict__ part,
                                                                 const int32_t* __restrict__ match_q, int K, int n,
                                                                 const int32_t* __restrict__ n_match,
                                                                 float4* __restrict__ sums, float* __restrict__ out) {
    __shared__ float red[PL_THREADS / 64];
    float dice = 0.f, focal = 0.f;
    for (int k = threadIdx.x; k < K; k += PL_THREADS) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < PL_SPLIT; i++) {
            const float4 v = part[k * PL_SPLIT + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        sums[k] = s;
        if (match_q[k] < 0) continue;
        dice += 1.f - (2.f * s.x + 1.f) / (s.y + s.z + 1.f);
        focal += s.w / (float)n;
    }
    dice = pl_block_sum(dice, red);
    focal = pl_block_sum(focal, red);
    if (threadIdx.x == 0) {
        const float nm = (float)n_match[0] + 1e-6f;
        out[0] = dice / nm;
        out[1] = focal / nm;
    }
}

// d_logits[q] = gout[0] d dice / d x + gout[1] d focal / d x for the query's matched instance, zeros for the others
__global__ __launch_bounds__(PL_THREADS) void k_pair_loss_bwd(const float* __restrict__ logits,
                                                               const float* __restrict__ inst,
                                                               const int32_t* __restrict__ match_of_q,
                                                               const float4* __restrict__ sums, int n,
                                                               const int32_t* __restrict__ n_match,
                                                               const float* __restrict__ gout,
                                                               float* __restrict__ d_logits) {
    const int q = blockIdx.x;
    const int k = match_of_q[q];
    float* dx = d_logits + (size_t)q * n;
    const int seg = (n + PL_SPLIT - 1) / PL_SPLIT;
    const int j0 = blockIdx.y * seg, j1 = min(n, j0 + seg);
    if (k < 0) {
        for (int j = j0 + threadIdx.x; j < j1; j += PL_THREADS) dx[j] = 0.f;
        return;
    }
    const float nm = (float)n_match[0] + 1e-6f;
    const float gd = gout[0] / nm, gf = gout[1] / (nm * (float)n);
    const float4 s = sums[k];
    const float S1 = s.y + s.z + 1.f, A2 = 2.f * s.x + 1.f;
    const float inv2 = 1.f / (S1 * S1);
    const float* x = logits + (size_t)q * n;
    const float* t = inst + (size_t)k * n;
    for (int j = j0 + threadIdx.x; j < j1; j += PL_THREADS) {
        const float xv = x[j], tv = t[j];
        const float p = 1.f / (1.f + expf(-xv));
        const float dp = p * (1.f - p);
        const float ce = fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
        const float pt = p * tv + (1.f - p) * (1.f - tv);
        const float om = 1.f - pt;
        const float al = 0.25f * tv + 0.75f * (1.f - tv);
        // dice_k = 1 - A2 / S1 : d/dp_j = -(2 t_j S1 - A2) / S1^2
        const float ddice = -(2.f * tv * S1 - A2) * inv2 * dp;
        // f = al ce om^2 : d ce / dx = p - t, d om / dx = -(2 t - 1) p (1 - p)
        const float dfoc = al * ((p - tv) * om * om - 2.f * ce * om * (2.f * tv - 1.f) * dp);
        dx[j] = gd * ddice + gf * dfoc;
    }
}

}  // namespace

extern "C" size_t gf_pair_losses_sums_floats(int K) { return (size_t)K * 4 * (1 + PL_SPLIT); }

extern "C" int gf_pair_losses_fwd(const float* mask_logits, const float* inst_masks, const int32_t* match_q, int nq, int K,
                                  int n, const int32_t* n_match, float* sums, float* out, void* stream) {
    GF_CHECK_ARG(mask_logits && inst_masks && match_q && n_match && sums && out, "gf_pair_losses_fwd: null argument");
    GF_CHECK_ARG(nq >= 1 && K >= 0 && n >= 1, "gf_pair_losses_fwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    // sums: [K, 4] row sums followed by the [K, PL_SPLIT, 4] partial sums of the segments (gf_pair_losses_sums_floats)
    float4* rows = reinterpret_cast<float4*>(sums);
    float4* part = rows + K;
    if (K > 0)
        hipLaunchKernelGGL(k_pair_loss_sums, dim3(K, PL_SPLIT), dim3(PL_THREADS), 0, st, mask_logits, inst_masks, match_q,
                           n, part);
    hipLaunchKernelGGL(k_pair_loss_final, dim3(1), dim3(PL_THREADS), 0, st, part, match_q, K, n, n_match, rows, out);
    GF_CHECK_LAUNCH("gf_pair_losses_fwd");
    return GF_OK;
}

extern "C" int gf_pair_losses_bwd(const float* mask_logits, const float* inst_masks, const int32_t* match_of_q,
                                  const float* sums, int nq, int K, int n, const int32_t* n_match, const float* grad_out,
                                  float* d_logits, void* stream) {
    GF_CHECK_ARG(mask_logits && inst_masks && match_of_q && n_match && sums && grad_out && d_logits,
                 "gf_pair_losses_bwd: null argument");
    GF_CHECK_ARG(nq >= 1 && K >= 0 && n >= 1, "gf_pair_losses_bwd: bad sizes");
    hipLaunchKernelGGL(k_pair_loss_bwd, dim3(nq, PL_SPLIT), dim3(PL_THREADS), 0, (hipStream_t)stream, mask_logits, inst_masks,
                       match_of_q, reinterpret_cast<const float4*>(sums), n, n_match, grad_out, d_logits);
    GF_CHECK_LAUNCH("gf_pair_losses_bwd");
    return GF_OK;
}
