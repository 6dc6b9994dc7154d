// The sparse U-Net in training mode as a layer program (include/geoformer_hip.h: gf_unet_train_fwd / _bwd).
//
// Reference structure: GeoFormer.input_conv -> UBlock x7 -> output_layer (model/geoformer/geoformer.py:39-53,398-401;
// ResidualBlock / UBlock: model/geoformer/geoformer_modules.py:10-35,52-129), BatchNorm1d in training mode, the
// backward of train.py:63-75.
//
// Why it exists: per batch-4 step the module tree costs the host ~10 ms forward (71 convolutions, 70 BatchNorm + ReLU
// pairs, residual adds, concatenations: ~600 framework / ctypes calls) and ~12 ms backward (135 Python autograd
// functions + ~250 framework nodes on the autograd thread) for ~21 ms of device work; the step was bound by the host.
// The same entry points issued from C++ in program order cost ~2 us per launch.  Nothing is computed differently:
// gf_conv_fwd (+ residual epilogue), gf_conv_pack_weights(_t), gf_conv_wgrad_masked, gf_bn_relu_train_fwd / _bwd_add.
#include "common.h"
#include "conv_pack.h"

namespace {

// out[r] = (a[r], b[r]) in 16-byte pieces
__global__ void k_tr_concat2(const float4* __restrict__ a, const float4* __restrict__ b, long long M, int ca4, int cb4,
                             float4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = ca4 + cb4;
    if (i >= M * c4) return;
    const long long r = i / c4;
    const int c = (int)(i - r * c4);
    out[i] = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
}

// (ga[r], gb[r]) (+)= g[r]: the concatenation's backward
__global__ void k_tr_split2(const float4* __restrict__ g, long long M, int ca4, int cb4, float4* ga, int acc_a, float4* gb,
                            int acc_b) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = ca4 + cb4;
    if (i >= M * c4) return;
    const long long r = i / c4;
    const int c = (int)(i - r * c4);
    const float4 v = g[i];
    float4* dst = c < ca4 ? ga + r * ca4 + c : gb + r * cb4 + (c - ca4);
    if (c < ca4 ? acc_a : acc_b) {
        const float4 o = *dst;
        *dst = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
    } else {
        *dst = v;
    }
}

__global__ void k_tr_add(float4* __restrict__ dst, const float4* __restrict__ src, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 a = dst[i], b = src[i];
    dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// All weight packs of a range in one launch (blockIdx.y = convolution): the per-convolution gf_conv_pack_weights(_t)
// launches are ~4 us each, 71 forward + 64 backward per step.  The table travels as a kernel argument.
constexpr int kPackMax = 80;
struct PackTable {
    const float* w[kPackMax];
    long long dst[kPackMax];  // floats into the destination area
    short K[kPackMax], Cin[kPackMax], Cout[kPackMax], flip[kPackMax];
};
template <bool T>
__global__ void k_tr_pack_batch(PackTable tb, float* __restrict__ base) {
    const int e = blockIdx.y;
    const int K = tb.K[e], Cin = tb.Cin[e], Cout = tb.Cout[e];
    // T: the packed operand is W' [K,Cout,Cin]
    const int NCH = ((T ? Cout : Cin) + 15) / 16, NCB = ((T ? Cin : Cout) + 15) / 16;
    const size_t total = (size_t)K * NCH * NCB * 64;
    float4* out = reinterpret_cast<float4*>(base + tb.dst[e]);
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x)
        out[t] = T ? gf_pack_weights_t_elem(tb.w[e], K, Cin, Cout, NCH, NCB, tb.flip[e], t)
                   : gf_pack_weights_elem(tb.w[e], Cin, Cout, NCH, NCB, t);
}

struct ConvGeom {  // one convolution op resolved against the step's levels
    const int32_t *tbl, *btbl, *steps, *bsteps, *flat;  // flat: step table of a symmetric (submanifold) relation, or null
    const uint32_t *gmask, *bgmask;
    int K, rows_in, rows_out, ld, bld, flip;
};

inline bool conv_geom(const GfTrainOp& op, const GfTrainLevel* lv, ConvGeom& g) {
    const GfTrainLevel& L = lv[op.level];
    g.steps = g.bsteps = nullptr;
    g.flat = nullptr;
    g.flip = 0;
    switch (op.table) {
    case 0:
        g.tbl = g.btbl = nullptr;
        g.gmask = g.bgmask = nullptr;
        g.K = 1;
        g.rows_in = g.rows_out = L.M;
        g.ld = g.bld = 0;
        return true;
    case 1:  // the submanifold relation is symmetric: the same table with weights W[26-k]^T gives the input gradient
        g.tbl = g.btbl = L.nbr;
        g.gmask = g.bgmask = L.gmask;
        g.steps = g.bsteps = L.steps;
        g.flat = L.flat;
        g.K = 27;
        g.rows_in = g.rows_out = L.M;
        g.ld = g.bld = L.ld;
        g.flip = 1;
        return true;
    case 2:
        g.tbl = L.child; g.gmask = L.gmask_down; g.ld = L.ld_down;
        g.btbl = L.up; g.bgmask = L.gmask_up; g.bld = L.ld_up;
        g.K = 8;
        g.rows_in = L.M;
        g.rows_out = L.M_coarse;
        return true;
    case 3:
        g.tbl = L.up; g.gmask = L.gmask_up; g.ld = L.ld_up;
        g.btbl = L.child; g.bgmask = L.gmask_down; g.bld = L.ld_down;
        g.K = 8;
        g.rows_in = L.M_coarse;
        g.rows_out = L.M;
        return true;
    default:
        return false;
    }
}

inline long long rows_of(const GfTrainOp& op, const GfTrainLevel* lv) { return lv[op.level].M; }

}  // namespace

extern "C" size_t gf_unet_train_scratch_floats(const GfTrainOp* ops, int nops, const GfTrainLevel* levels) {
    size_t bn = 0, wt = 0;
    for (int i = 0; i < nops; i++) {
        const GfTrainOp& op = ops[i];
        if (op.kind == 0) {
            const size_t f = gf_bn_train_scratch_floats(levels[op.level].M, op.Cin);
            if (f > bn) bn = f;
        } else if (op.kind == 1) {  // every convolution's transposed pack has its own slot (packed in one launch)
            const int K = op.table == 0 ? 1 : (op.table == 1 ? 27 : 8);
            wt += (gf_conv_packed_floats(K, op.Cout, op.Cin) + 63) & ~(size_t)63;
        }
    }
    return ((bn + 63) & ~(size_t)63) + wt + 64;
}

extern "C" int gf_unet_train_fwd(const GfTrainOp* ops, int op_begin, int op_end, const GfTrainLevel* levels,
                                 float* const* act, float* wp, float* stats, float* scratch, void* stream) {
    GF_CHECK_ARG(ops && levels && act && wp && stats && scratch && op_begin >= 0 && op_end >= op_begin,
                 "gf_unet_train_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    {  // the range's weights, packed for the MFMA B operand, in one launch per kPackMax convolutions
        PackTable tb;
        int n = 0;
        size_t most = 0;
        auto flush = [&]() {
            if (n) hipLaunchKernelGGL(k_tr_pack_batch<false>, dim3(gf_div_up((long long)most, 256 * 4), n), dim3(256), 0, st, tb, wp);
            n = 0;
            most = 0;
        };
        for (int i = op_begin; i < op_end; i++) {
            const GfTrainOp& op = ops[i];
            if (op.kind != 1) continue;
            ConvGeom g;
            GF_CHECK_ARG(conv_geom(op, levels, g), "gf_unet_train_fwd: op %d: table %d", i, op.table);
            tb.w[n] = op.w; tb.dst[n] = op.wp_off;
            tb.K[n] = (short)g.K; tb.Cin[n] = (short)op.Cin; tb.Cout[n] = (short)op.Cout; tb.flip[n] = 0;
            const size_t f4 = gf_conv_packed_floats(g.K, op.Cin, op.Cout) / 4;
            if (f4 > most) most = f4;
            if (++n == kPackMax) flush();
        }
        flush();
    }
    for (int i = op_begin; i < op_end; i++) {
        const GfTrainOp& op = ops[i];
        if (op.kind == 0) {
            const int M = (int)rows_of(op, levels);
            int rc = gf_bn_relu_train_fwd(act[op.src], M, op.Cin, op.gamma, op.beta, op.eps, op.momentum, 1, op.running_mean,
                                          op.running_var, act[op.dst], stats + op.stats_off, stats + op.stats_off + op.Cin,
                                          scratch, stream);
            if (rc != GF_OK) return rc;
        } else if (op.kind == 1) {
            ConvGeom g;
            GF_CHECK_ARG(conv_geom(op, levels, g), "gf_unet_train_fwd: op %d: table %d", i, op.table);
            float* wpk = wp + op.wp_off;
            int rc = gf_conv_fwd_flat(act[op.src], wpk, g.tbl, g.gmask, g.steps, g.flat, g.K, g.rows_in, g.rows_out, g.ld, op.Cin, op.Cout,
                                      nullptr, nullptr, op.aux >= 0 ? act[op.aux] : nullptr, nullptr, nullptr, act[op.dst], nullptr, stream);
            if (rc != GF_OK) return rc;
        } else if (op.kind == 2) {
            const long long M = rows_of(op, levels);
            GF_CHECK_ARG((op.Cin % 4) == 0 && (op.Cout % 4) == 0, "gf_unet_train_fwd: op %d: widths %d + %d", i, op.Cin, op.Cout);
            const int ca4 = op.Cin / 4, cb4 = op.Cout / 4;  // (Cin / Cout: widths of src / aux)
            const long long n = M * (ca4 + cb4);
            if (n > 0)
                hipLaunchKernelGGL(k_tr_concat2, dim3(gf_div_up(n, 256)), dim3(256), 0, st, (const float4*)act[op.src],
                                   (const float4*)act[op.aux], M, ca4, cb4, (float4*)act[op.dst]);
        } else {
            GF_CHECK_ARG(false, "gf_unet_train_fwd: op %d: kind %d", i, op.kind);
        }
    }
    GF_CHECK_LAUNCH("gf_unet_train_fwd");
    return GF_OK;
}

extern "C" int gf_unet_train_bwd(const GfTrainOp* ops, int op_begin, int op_end, const GfTrainLevel* levels,
                                 float* const* act, float** grad, unsigned char* ghas, const float* stats, float* pgrad,
                                 float* scratch, void* stream) {
    GF_CHECK_ARG(ops && levels && act && grad && ghas && stats && pgrad && scratch && op_begin >= 0 && op_end >= op_begin,
                 "gf_unet_train_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    // scratch: [BatchNorm partials | transposed weight pack]; the split is the one gf_unet_train_scratch_floats makes
    size_t bn = 0;
    for (int i = op_begin; i < op_end; i++)
        if (ops[i].kind == 0) {
            const size_t f = gf_bn_train_scratch_floats(levels[ops[i].level].M, ops[i].Cin);
            if (f > bn) bn = f;
        }
    float* wt = scratch + ((bn + 63) & ~(size_t)63);
    long long wt_off[kPackMax * 4];
    {
        // the transposed (input-gradient) packs of the range in one launch, and ONE zero fill for the range's weight
        // gradients (op order = pgrad order: the range is contiguous; gf_conv_wgrad_masked clears per convolution)
        GF_CHECK_ARG(op_end - op_begin <= kPackMax * 4, "gf_unet_train_bwd: %d ops in a range", op_end - op_begin);
        PackTable tb;
        int n = 0;
        size_t most = 0, off = 0;
        long long p_lo = -1, p_hi = -1;
        auto flush = [&]() {
            if (n) hipLaunchKernelGGL(k_tr_pack_batch<true>, dim3(gf_div_up((long long)most, 256 * 4), n), dim3(256), 0, st, tb, wt);
            n = 0;
            most = 0;
        };
        for (int i = op_begin; i < op_end; i++) {
            const GfTrainOp& op = ops[i];
            wt_off[i - op_begin] = -1;
            if (op.kind != 1) continue;
            ConvGeom g;
            GF_CHECK_ARG(conv_geom(op, levels, g), "gf_unet_train_bwd: op %d: table %d", i, op.table);
            const long long pe = op.pgrad_off + (long long)g.K * op.Cin * op.Cout;
            if (p_lo < 0 || op.pgrad_off < p_lo) p_lo = op.pgrad_off;
            if (pe > p_hi) p_hi = pe;
            if (op.no_dgrad) continue;
            const size_t f = gf_conv_packed_floats(g.K, op.Cout, op.Cin);
            wt_off[i - op_begin] = (long long)off;
            tb.w[n] = op.w; tb.dst[n] = (long long)off;
            tb.K[n] = (short)g.K; tb.Cin[n] = (short)op.Cin; tb.Cout[n] = (short)op.Cout; tb.flip[n] = (short)g.flip;
            off += (f + 63) & ~(size_t)63;
            if (f / 4 > most) most = f / 4;
            if (++n == kPackMax) flush();
        }
        flush();
        // (BatchNorm's dgamma / dbeta inside the span are written in full by their op afterwards)
        if (p_hi > p_lo) GF_TRY(hipMemsetAsync(pgrad + p_lo, 0, (size_t)(p_hi - p_lo) * sizeof(float), st));
    }
    for (int i = op_end - 1; i >= op_begin; i--) {
        const GfTrainOp& op = ops[i];
        GF_CHECK_ARG(ghas[op.dst], "gf_unet_train_bwd: op %d: no gradient for its output (buffer %d)", i, op.dst);
        if (op.kind == 0) {
            const int M = (int)rows_of(op, levels);
            float* dgb = pgrad + op.pgrad_off;
            int rc = gf_bn_relu_train_bwd_add(act[op.src], act[op.dst], grad[op.dst], M, op.Cin, op.gamma, stats + op.stats_off,
                                              stats + op.stats_off + op.Cin, 1, ghas[op.src] ? grad[op.src] : nullptr,
                                              grad[op.src], dgb, dgb + op.Cin, scratch, stream);
            if (rc != GF_OK) return rc;
            if (!ghas[op.src]) ghas[op.src] = 1;
        } else if (op.kind == 1) {
            ConvGeom g;
            GF_CHECK_ARG(conv_geom(op, levels, g), "gf_unet_train_bwd: op %d: table %d", i, op.table);
            const float* gy = grad[op.dst];
            if (op.aux >= 0) {  // residual operand: its gradient is gy
                const long long n4 = (long long)g.rows_out * op.Cout / 4;
                if (ghas[op.aux]) {
                    if (n4 > 0)
                        hipLaunchKernelGGL(k_tr_add, dim3(gf_div_up(n4, 256)), dim3(256), 0, st, (float4*)grad[op.aux],
                                           (const float4*)gy, n4);
                } else if (ghas[op.dst] == 2) {  // not ours to write into later: copy
                    GF_TRY(hipMemcpyAsync(grad[op.aux], gy, (size_t)n4 * 16, hipMemcpyDeviceToDevice, st));
                    ghas[op.aux] = 1;
                } else {
                    grad[op.aux] = grad[op.dst];  // (dead after this op: later in-place accumulation is safe)
                    ghas[op.aux] = 1;
                }
            }
            if (!op.no_dgrad) {
                // (rows of the gradient = the forward's output rows; the residual epilogue adds what the source has)
                // (a submanifold relation is its own transpose: the forward's flat step table serves the gradient too)
                int rc = gf_conv_fwd_flat(gy, wt + wt_off[i - op_begin], g.btbl, g.bgmask, g.bsteps, g.flat, g.K, g.rows_out, g.rows_in, g.bld, op.Cout, op.Cin, nullptr,
                                          nullptr, ghas[op.src] ? grad[op.src] : nullptr, nullptr, nullptr, grad[op.src], nullptr, stream);
                if (rc != GF_OK) return rc;
                if (!ghas[op.src]) ghas[op.src] = 1;
            }
            int rc = gf_conv_wgrad_masked_acc(act[op.src], gy, g.tbl, g.gmask, g.K, g.rows_out, g.ld, op.Cin, op.Cout,
                                          pgrad + op.pgrad_off, stream);
            if (rc != GF_OK) return rc;
        } else if (op.kind == 2) {
            const long long M = rows_of(op, levels);
            const int ca4 = op.Cin / 4, cb4 = op.Cout / 4;
            const long long n = M * (ca4 + cb4);
            if (n > 0)
                hipLaunchKernelGGL(k_tr_split2, dim3(gf_div_up(n, 256)), dim3(256), 0, st, (const float4*)grad[op.dst], M, ca4,
                                   cb4, (float4*)grad[op.src], ghas[op.src] ? 1 : 0, (float4*)grad[op.aux],
                                   ghas[op.aux] ? 1 : 0);
            if (!ghas[op.src]) ghas[op.src] = 1;
            if (!ghas[op.aux]) ghas[op.aux] = 1;
        } else {
            GF_CHECK_ARG(false, "gf_unet_train_bwd: op %d: kind %d", i, op.kind);
        }
    }
    GF_CHECK_LAUNCH("gf_unet_train_bwd");
    return GF_OK;
}
