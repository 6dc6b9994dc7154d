// Per-point MLP chains of the eval forward, fused: y = L_n(... relu(affine(L_1 x)) ...) over the rows of x[N,C].
//
// The reference runs them as Conv1d(k=1)/Linear + BatchNorm1d + ReLU launches over [1,C,N] / [N,C] tensors:
//   mask_tower        geoformer.py:64-71    3 x (conv1d 16->16, BN, ReLU) + conv1d 16->16      over the fg points
//   semantic head     geoformer.py:54-62    2 x (Linear 16->16, BN, ReLU) + Linear 16->classes  over all points
// (~10 launches and 2*4*N*C bytes of HBM traffic each; at C = 16 a layer is 64 B in, 64 B out per point).  Here a
// wave owns 16-point tiles and keeps the activations TRANSPOSED in MFMA accumulators (channel on the row, point on
// the column): H^T = W . X^T -- the accumulator of one layer is directly the B operand of the next, so a point's
// features are read once and its outputs written once.  Eval-mode BatchNorm and the bias are folded by the caller
// into one per-channel (scale, shift) pair per layer.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PM_MAXL 4
#define PM_MAXC 64  // channels per layer (4 tiles of 16)

struct PmLayer {
    const float *W, *scale, *shift;  // W row-major [cout, cin]
    int cin, cout, relu;
};
struct PmArgs {
    PmLayer L[PM_MAXL];
    int nl;
};

// The layers' weights are staged once per workgroup in LDS, zero-padded to multiples of 16 in both dimensions with a
// row stride of cin_padded + 4 floats: the A operand of four MFMAs is then ONE aligned ds_read_b128 without predicates.
// Reading them in place (row-major [cout, cin] with cin = 3 + C = 35 for the set-abstraction MLP: unaligned rows)
// took four predicated scalar loads per operand, 72 per 16-sample tile and lane -- the kernel spent its time there.
#define GM_WSTRIDE(cinp) ((cinp) + 4)
__device__ __forceinline__ void pm_stage_weights(const PmArgs& A, float* __restrict__ s_w, int* woff, int* soff) {
    int cur = 0;
#pragma unroll
    for (int l = 0; l < PM_MAXL; l++) {
        if (l >= A.nl) break;
        const int cinp = (A.L[l].cin + 15) & ~15, coutp = (A.L[l].cout + 15) & ~15;
        woff[l] = cur;
        cur += coutp * GM_WSTRIDE(cinp);
        soff[l] = cur;
        cur += 2 * coutp;
    }
#pragma unroll
    for (int l = 0; l < PM_MAXL; l++) {
        if (l >= A.nl) break;
        const PmLayer& L = A.L[l];
        const int cinp = (L.cin + 15) & ~15, coutp = (L.cout + 15) & ~15, ws = GM_WSTRIDE(cinp);
        for (int t = threadIdx.x; t < coutp * cinp; t += blockDim.x) {
            const int r = t / cinp, c = t - r * cinp;
            s_w[woff[l] + r * ws + c] = (r < L.cout && c < L.cin) ? L.W[(size_t)r * L.cin + c] : 0.f;
        }
        for (int t = threadIdx.x; t < coutp; t += blockDim.x) {
            s_w[soff[l] + t] = t < L.cout ? L.scale[t] : 0.f;
            s_w[soff[l] + coutp + t] = t < L.cout ? L.shift[t] : 0.f;
        }
    }
    __syncthreads();
}

// rows: optional int32 [N] row indirection of the input (out[p] = MLP(x[rows[p]])): the p2v gather of the semantic
// head rides in the first layer's operand load instead of materialising feats[p2v_map]
__global__ __launch_bounds__(256) void k_pointwise_mlp(const float* __restrict__ x, const int32_t* __restrict__ rows,
                                                       int N, PmArgs A, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int ntiles = (N + 15) >> 4;
    const int c0 = A.L[0].cin, cl = A.L[A.nl - 1].cout;
    int woff[PM_MAXL], soff[PM_MAXL];
    pm_stage_weights(A, s_w, woff, soff);
    for (int t = wave; t < ntiles; t += nwaves) {
        const int p = t * 16 + j;
        const bool live = p < N;
        const int src = (rows && live) ? rows[p] : p;
        // B operand of the first layer: channels kc*16 + 4g .. +3 of point j
        float4 h[PM_MAXC / 16];
#pragma unroll
        for (int kc = 0; kc < PM_MAXC / 16; kc++) {
            h[kc] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kc * 16 < c0 && live) h[kc] = *reinterpret_cast<const float4*>(x + (size_t)src * c0 + kc * 16 + 4 * g);
        }
#pragma unroll
        for (int l = 0; l < PM_MAXL; l++) {
            if (l >= A.nl) break;
            const PmLayer& L = A.L[l];
            const int cinp = (L.cin + 15) & ~15, coutp = (L.cout + 15) & ~15, ws = GM_WSTRIDE(cinp);
            const float* Wl = s_w + woff[l];
            const float* Sl = s_w + soff[l];
            float4 o[PM_MAXC / 16];
#pragma unroll
            for (int ct = 0; ct < PM_MAXC / 16; ct++) {
                o[ct] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ct * 16 >= L.cout) continue;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* wrow = Wl + (ct * 16 + j) * ws + 4 * g;  // output channel ct*16 + j is this lane's A row
#pragma unroll
                for (int kc = 0; kc < PM_MAXC / 16; kc++) {
                    if (kc * 16 >= L.cin) continue;
                    const float4 a = *reinterpret_cast<const float4*>(wrow + kc * 16);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, h[kc].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, h[kc].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, h[kc].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, h[kc].w, acc, 0, 0, 0);
                }
                // accumulator: channels ct*16 + 4g + i of point j; padded channels come out as 0 (scale = shift = 0)
                const float4 sc = *reinterpret_cast<const float4*>(Sl + ct * 16 + 4 * g);
                const float4 sh = *reinterpret_cast<const float4*>(Sl + coutp + ct * 16 + 4 * g);
                float r[4] = {fmaf(acc[0], sc.x, sh.x), fmaf(acc[1], sc.y, sh.y), fmaf(acc[2], sc.z, sh.z),
                              fmaf(acc[3], sc.w, sh.w)};
                if (L.relu) {
#pragma unroll
                    for (int i = 0; i < 4; i++) r[i] = fmaxf(r[i], 0.f);
                }
                o[ct] = make_float4(r[0], r[1], r[2], r[3]);
            }
#pragma unroll
            for (int ct = 0; ct < PM_MAXC / 16; ct++) h[ct] = o[ct];
        }
        if (live) {
#pragma unroll
            for (int ct = 0; ct < PM_MAXC / 16; ct++) {
                const int ch = ct * 16 + 4 * g;
                if (ch + 3 < cl && (cl & 3) == 0) {
                    *reinterpret_cast<float4*>(out + (size_t)p * cl + ch) = h[ct];
                } else {
                    const float v[4] = {h[ct].x, h[ct].y, h[ct].z, h[ct].w};
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (ch + i < cl) out[(size_t)p * cl + ch + i] = v[i];
                }
            }
        }
    }
}

static size_t gm_lds_bytes(const PmArgs& A) {
    size_t n = 0;
    for (int l = 0; l < A.nl; l++) {
        const size_t cinp = (A.L[l].cin + 15) & ~15, coutp = (A.L[l].cout + 15) & ~15;
        n += coutp * GM_WSTRIDE(cinp) + 2 * coutp;
    }
    return n * sizeof(float);  // <= 4 * (64 * 68 + 128) * 4 = 71 KB worst case, ~20 KB for the set-abstraction MLP
}

extern "C" int gf_pointwise_mlp_rows(const float* x, const int32_t* rows, int N, int n_layers, const float* const* W,
                                     const float* const* scale, const float* const* shift, const int* channels,
                                     const int* relu, float* out, void* stream);
extern "C" int gf_pointwise_mlp(const float* x, int N, int n_layers, const float* const* W, const float* const* scale,
                                const float* const* shift, const int* channels, const int* relu, float* out,
                                void* stream) {
    return gf_pointwise_mlp_rows(x, nullptr, N, n_layers, W, scale, shift, channels, relu, out, stream);
}

extern "C" int gf_pointwise_mlp_rows(const float* x, const int32_t* rows, int N, int n_layers, const float* const* W,
                                     const float* const* scale, const float* const* shift, const int* channels,
                                     const int* relu, float* out, void* stream) {
    GF_CHECK_ARG(n_layers >= 1 && n_layers <= PM_MAXL, "gf_pointwise_mlp: 1..%d layers, got %d", PM_MAXL, n_layers);
    GF_CHECK_ARG(N >= 0, "gf_pointwise_mlp: bad N");
    PmArgs A;
    for (int l = 0; l < n_layers; l++) {
        const int cin = channels[l], cout = channels[l + 1];
        GF_CHECK_ARG(cin >= 16 && cin <= PM_MAXC && cin % 16 == 0,
                     "gf_pointwise_mlp: input width %d of layer %d must be a multiple of 16 in 16..%d", cin, l, PM_MAXC);
        GF_CHECK_ARG(cout >= 1 && cout <= PM_MAXC && (l == n_layers - 1 || cout % 16 == 0),
                     "gf_pointwise_mlp: output width %d of layer %d (hidden widths: multiples of 16, all <= %d)",
                     cout, l, PM_MAXC);
        GF_CHECK_ARG(W[l] && scale[l] && shift[l], "gf_pointwise_mlp: null parameter of layer %d", l);
        A.L[l] = {W[l], scale[l], shift[l], cin, cout, relu[l]};
    }
    for (int l = n_layers; l < PM_MAXL; l++) A.L[l] = A.L[0];
    A.nl = n_layers;
    if (N == 0) return GF_OK;
    const int ntiles = (N + 15) / 16;
    int blocks = (ntiles + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    static bool lds_ok = false;
    if (!lds_ok) {
        (void)hipFuncSetAttribute((const void*)k_pointwise_mlp, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        lds_ok = true;
    }
    hipLaunchKernelGGL(k_pointwise_mlp, dim3(blocks), dim3(256), gm_lds_bytes(A), (hipStream_t)stream, x, rows, N, A, out);
    GF_CHECK_LAUNCH("gf_pointwise_mlp");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// Set-abstraction MLP + max-pool, fused (PointnetSAModuleVotes: SharedMLP over the grouped features and
// F.max_pool2d over the samples, lib/pointnet2/pointnet2_modules.py:335-349 with pytorch_utils.SharedMLP):
//   out[b, :, i] = max_s  L_n(... L_1(grouped[b, :, i, s]))        L_l = ReLU(BN(Conv2d 1x1))
// One wave per centre point: its nsample columns are the MFMA columns (16 per tile), activations stay transposed
// in accumulators from layer to layer (see k_pointwise_mlp), the maximum over the samples is a DPP row reduction
// plus a running maximum over the tiles.  The [B, C, npoint, nsample] intermediates of the reference never exist.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float pm_row_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
    return v;
}

// GATHER: the grouped tensor is never materialised -- sample `smp` of centre `pt` is point idx[pt][smp], its input
// channels are the centred (and radius-scaled) coordinates followed by the point's feature column
// (QueryAndGroup.forward, pointnet2_utils.py:326-356).
struct PmGather {
    const float *xyz, *feats, *new_xyz;  // [B,n,3], [B,C,n], [B,np,3]
    const int32_t* idx;                  // [B,np,ns]
    int n, C, use_xyz;
    float inv_radius;  // 1/radius with normalize_xyz (torch divides by a Python scalar as a * (1/b)), else 1
};
template <bool GATHER>
__global__ __launch_bounds__(256) void k_group_mlp_max(const float* __restrict__ grouped, int B, int np, int ns,
                                                       PmArgs A, PmGather Gx, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // per layer: W [coutp][cinp + 4], scale, shift [coutp]
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int c0 = A.L[0].cin, cl = A.L[A.nl - 1].cout;
    const size_t cstride = (size_t)np * ns;
    int woff[PM_MAXL], soff[PM_MAXL];
    pm_stage_weights(A, s_w, woff, soff);
    for (int item = wave; item < B * np; item += nwaves) {
        const int b = item / np, pt = item - b * np;
        const float* gp = GATHER ? nullptr : grouped + (size_t)b * c0 * cstride + (size_t)pt * ns;
        float best[PM_MAXC / 16][4];
#pragma unroll
        for (int ct = 0; ct < PM_MAXC / 16; ct++)
#pragma unroll
            for (int i = 0; i < 4; i++) best[ct][i] = -INFINITY;
        for (int s0 = 0; s0 < ns; s0 += 16) {
            const int smp = s0 + j;
            const bool live = smp < ns;
            float4 h[PM_MAXC / 16];
            int id = 0;
            if (GATHER && live) id = Gx.idx[((size_t)b * np + pt) * ns + smp];
#pragma unroll
            for (int kc = 0; kc < PM_MAXC / 16; kc++) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (kc * 16 < c0 && live) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int ch = kc * 16 + 4 * g + i;
                        if (ch < c0) {
                            if (!GATHER) {
                                v[i] = gp[(size_t)ch * cstride + smp];
                            } else if (Gx.use_xyz && ch < 3) {
                                v[i] = (Gx.xyz[((size_t)b * Gx.n + id) * 3 + ch] -
                                        Gx.new_xyz[((size_t)b * np + pt) * 3 + ch]) * Gx.inv_radius;
                            } else {
                                const int fc = ch - (Gx.use_xyz ? 3 : 0);
                                v[i] = Gx.feats[((size_t)b * Gx.C + fc) * Gx.n + id];
                            }
                        }
                    }
                }
                h[kc] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int l = 0; l < PM_MAXL; l++) {
                if (l >= A.nl) break;
                const PmLayer& L = A.L[l];
                const int cinp = (L.cin + 15) & ~15, coutp = (L.cout + 15) & ~15, ws = GM_WSTRIDE(cinp);
                const float* Wl = s_w + woff[l];
                const float* Sl = s_w + soff[l];
                float4 o[PM_MAXC / 16];
#pragma unroll
                for (int ct = 0; ct < PM_MAXC / 16; ct++) {
                    o[ct] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ct * 16 >= L.cout) continue;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const float* wrow = Wl + (ct * 16 + j) * ws + 4 * g;
#pragma unroll
                    for (int kc = 0; kc < PM_MAXC / 16; kc++) {
                        if (kc * 16 >= L.cin) continue;
                        const float4 a = *reinterpret_cast<const float4*>(wrow + kc * 16);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, h[kc].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, h[kc].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, h[kc].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, h[kc].w, acc, 0, 0, 0);
                    }
                    // padded channels: scale = shift = 0 -> 0 (also after the ReLU)
                    const float4 sc = *reinterpret_cast<const float4*>(Sl + ct * 16 + 4 * g);
                    const float4 sh = *reinterpret_cast<const float4*>(Sl + coutp + ct * 16 + 4 * g);
                    float r[4] = {fmaf(acc[0], sc.x, sh.x), fmaf(acc[1], sc.y, sh.y), fmaf(acc[2], sc.z, sh.z),
                                  fmaf(acc[3], sc.w, sh.w)};
                    if (L.relu) {
#pragma unroll
                        for (int i = 0; i < 4; i++) r[i] = fmaxf(r[i], 0.f);
                    }
                    o[ct] = make_float4(r[0], r[1], r[2], r[3]);
                }
#pragma unroll
                for (int ct = 0; ct < PM_MAXC / 16; ct++) h[ct] = o[ct];
            }
#pragma unroll
            for (int ct = 0; ct < PM_MAXC / 16; ct++) {
                if (ct * 16 >= cl) continue;
                const float v[4] = {h[ct].x, h[ct].y, h[ct].z, h[ct].w};
#pragma unroll
                for (int i = 0; i < 4; i++) best[ct][i] = fmaxf(best[ct][i], pm_row_max(live ? v[i] : -INFINITY));
            }
        }
        if (j == 0) {
#pragma unroll
            for (int ct = 0; ct < PM_MAXC / 16; ct++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int ch = ct * 16 + 4 * g + i;
                    if (ch < cl) out[((size_t)b * cl + ch) * np + pt] = best[ct][i];
                }
        }
    }
}

template <bool GATHER>
static void gm_allow_lds() {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)k_group_mlp_max<GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  96 * 1024);
        done = true;
    }
}

static int pm_fill_args(const char* who, int n_layers, const float* const* W, const float* const* scale,
                        const float* const* shift, const int* channels, const int* relu, PmArgs& A) {
    GF_CHECK_ARG(n_layers >= 1 && n_layers <= PM_MAXL, "%s: 1..%d layers, got %d", who, PM_MAXL, n_layers);
    for (int l = 0; l < n_layers; l++) {
        const int cin = channels[l], cout = channels[l + 1];
        GF_CHECK_ARG(cin >= 1 && cin <= PM_MAXC && (l == 0 || cin % 16 == 0) && cout >= 1 && cout <= PM_MAXC &&
                         (l == n_layers - 1 || cout % 16 == 0),
                     "%s: layer %d widths %d -> %d (hidden widths: multiples of 16, all <= %d)", who, l, cin, cout,
                     PM_MAXC);
        GF_CHECK_ARG(W[l] && scale[l] && shift[l], "%s: null parameter of layer %d", who, l);
        A.L[l] = {W[l], scale[l], shift[l], cin, cout, relu[l]};
    }
    for (int l = n_layers; l < PM_MAXL; l++) A.L[l] = A.L[0];
    A.nl = n_layers;
    return GF_OK;
}

extern "C" int gf_group_mlp_max(const float* grouped, int B, int npoint, int nsample, int n_layers,
                                const float* const* W, const float* const* scale, const float* const* shift,
                                const int* channels, const int* relu, float* out, void* stream) {
    GF_CHECK_ARG(B >= 0 && npoint >= 0 && nsample >= 1, "gf_group_mlp_max: bad sizes");
    PmArgs A;
    if (int rc = pm_fill_args("gf_group_mlp_max", n_layers, W, scale, shift, channels, relu, A)) return rc;
    if (B == 0 || npoint == 0) return GF_OK;
    long long items = (long long)B * npoint;
    int blocks = (int)((items + 3) / 4);
    if (blocks > 256 * 2) blocks = 256 * 2;  // the weight image is staged once per workgroup
    gm_allow_lds<false>();
    hipLaunchKernelGGL(k_group_mlp_max<false>, dim3(blocks), dim3(256), gm_lds_bytes(A), (hipStream_t)stream, grouped, B,
                       npoint, nsample, A, PmGather{}, out);
    GF_CHECK_LAUNCH("gf_group_mlp_max");
    return GF_OK;
}

// The set-abstraction stage of the eval forward in two launches (PointnetSAModuleVotes with given sample indices,
// pointnet2_modules.py:262-333): ball query around xyz[inds] (which also writes new_xyz), then the shared MLP + max
// pool reading the neighbours through idx.  Replaces gather, ball query, two groupings, centring, scaling, the
// concatenation and the [B, 3+C, np, ns] tensor they produce.
extern "C" int gf_ball_query_centres(const float* xyz, const int32_t* centre_idx, int b, int n, int m, float radius,
                                     int nsample, float* new_xyz, int32_t* idx, void* stream);
extern "C" int gf_ball_query_grid(const float* xyz, int n, const int32_t* centre_idx, const float* centres, int m,
                                  float radius, int nsample, void* scratch, int grid_ready, float* new_xyz,
                                  int32_t* idx, void* stream);
extern "C" int gf_sa_group_mlp_max(const float* xyz, const float* feats, const int32_t* inds, int B, int n, int C,
                                   int npoint, float radius, int nsample, int use_xyz, int normalize_xyz,
                                   int n_layers, const float* const* W, const float* const* scale,
                                   const float* const* shift, const int* channels, const int* relu, float* new_xyz,
                                   int32_t* idx, float* out, void* scratch, int grid_ready, void* stream) {
    GF_CHECK_ARG(xyz && inds && new_xyz && idx && out && (feats || C == 0), "gf_sa_group_mlp_max: null argument");
    GF_CHECK_ARG(B >= 0 && n >= 1 && C >= 0 && npoint >= 0 && nsample >= 1 && radius > 0.f,
                 "gf_sa_group_mlp_max: bad sizes");
    GF_CHECK_ARG(use_xyz || C > 0, "gf_sa_group_mlp_max: no input channels");
    PmArgs A;
    if (int rc = pm_fill_args("gf_sa_group_mlp_max", n_layers, W, scale, shift, channels, relu, A)) return rc;
    GF_CHECK_ARG(channels[0] == C + (use_xyz ? 3 : 0), "gf_sa_group_mlp_max: first layer expects %d channels, got %d",
                 channels[0], C + (use_xyz ? 3 : 0));
    if (B == 0 || npoint == 0) return GF_OK;
    // a grid pays once the linear scan is long: 27 buckets per centre against every point
    if (scratch && B == 1 && n >= 4096) {
        if (int rc = gf_ball_query_grid(xyz, n, inds, nullptr, npoint, radius, nsample, scratch, grid_ready, new_xyz, idx,
                                        stream))
            return rc;
    } else if (int rc = gf_ball_query_centres(xyz, inds, B, n, npoint, radius, nsample, new_xyz, idx, stream)) {
        return rc;
    }
    PmGather Gx{xyz, feats, new_xyz, idx, n, C, use_xyz, normalize_xyz ? 1.0f / radius : 1.0f};
    long long items = (long long)B * npoint;
    int blocks = (int)((items + 3) / 4);
    if (blocks > 256 * 2) blocks = 256 * 2;
    gm_allow_lds<true>();
    hipLaunchKernelGGL(k_group_mlp_max<true>, dim3(blocks), dim3(256), gm_lds_bytes(A), (hipStream_t)stream, nullptr, B,
                       npoint, nsample, A, Gx, out);
    GF_CHECK_LAUNCH("gf_sa_group_mlp_max");
    return GF_OK;
}
