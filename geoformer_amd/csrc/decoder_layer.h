// Tile helpers of the decoder's token-side kernels (decoder_layer.hip: inference, decoder_layer_train.hip: training):
// d_model 64, 4 heads x 16 channels; products on v_mfma_f32_16x16x4_f32 with operands loaded straight from row-major
// activations (LDS tiles or global rows) and nn.Linear weights [out,in] (operand scheme of backbone_attn.hip).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DL_D 64
#define DL_H 4
#define DL_DK 16
#define DL_THREADS 1024
#define DL_MAXFF 256

__device__ __forceinline__ f32x4 dl_mfma4(float4 a, float4 b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

#define DL_LD 68     // padded LDS row (floats): 16 rows x float4 reads without bank conflicts
#define DL_LDH 260

// one 16-row tile: out(r, col, act(sum_k A[r][k] W[col][k] + b[col]) (+ addend[r][col])) for the column tiles ct = wave,
// wave+nw, ...  A may live in LDS or global memory (row stride lda); rows >= nvalid read as zero and are not emitted.
// A phase of the token-side kernels is one of these between two barriers, and what it costs is its dependent memory round
// trips, not its 16-64 MFMAs (cycle stamps, round 6: ~5 000 cycles per phase whatever its size): the bias and the
// epilogue's global operand (`addend`, row stride add_ld: a residual row the caller would otherwise read inside `epi`) are
// therefore requested BEFORE the products, beside the weights, instead of one after the other behind them.
template <bool RELU, typename Epi>
__device__ __forceinline__ void dl_tile_gemm(const float* A, int lda, int nvalid, int K, const float* __restrict__ W,
                                             const float* __restrict__ bias, int N, int wave, int nwaves, int lane,
                                             Epi epi, const float* __restrict__ addend = nullptr, int add_ld = 0) {
    const int j = lane & 15, g = lane >> 4;
    const int KC = K >> 4;
    for (int ct = wave; ct < (N >> 4); ct += nwaves) {
        const float* xa = A + (size_t)j * lda + 4 * g;
        const float* wb = W + (size_t)(ct * 16 + j) * K + 4 * g;
        const int col = ct * 16 + j;
        const float bs = bias[col];
        float ad[4] = {0.f, 0.f, 0.f, 0.f};
        if (addend) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (4 * g + i < nvalid) ad[i] = addend[(size_t)(4 * g + i) * add_ld + col];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int kc = 0; kc < KC; kc++) {
            float4 a = j < nvalid ? *reinterpret_cast<const float4*>(xa + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b = *reinterpret_cast<const float4*>(wb + kc * 16);
            acc = dl_mfma4(a, b, acc);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * g + i;
            if (r >= nvalid) continue;
            float v = acc[i] + bs;
            if (RELU) v = fmaxf(v, 0.f);
            if (addend) v += ad[i];
            epi(r, col, v);
        }
    }
}

// The same product with its weight operands ALREADY in registers: a token-side kernel is a chain of small products between
// barriers, one wave per SIMD, and every phase used to open with its own weight fetch (~1 500-2 000 cycles of a ~4 500-cycle
// phase: cycle stamps, round 6).  dl_w_load requests a phase's operands -- NT column tiles per wave x KC 16-channel steps,
// + bias -- any number of phases ahead (the kernels do it at entry, all phases at once: one round trip for the lot);
// dl_tile_gemm_w consumes them.  Same arithmetic and order as dl_tile_gemm.
template <int NT, int KC>
struct DlW {
    float4 b[NT][KC];
    float bias[NT];
};
template <int NT, int KC>
__device__ __forceinline__ void dl_w_load(DlW<NT, KC>& w, const float* __restrict__ W, const float* __restrict__ bias, int N,
                                          int K, int wave, int nwaves, int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int ct = wave + t * nwaves;
        const bool on = ct < (N >> 4);
        const float* wb = W + (size_t)((on ? ct : 0) * 16 + j) * K + 4 * g;
        w.bias[t] = on ? bias[ct * 16 + j] : 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; kc++)
            w.b[t][kc] = (on && kc < (K >> 4)) ? *reinterpret_cast<const float4*>(wb + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <bool RELU, int NT, int KC, typename Epi>
__device__ __forceinline__ void dl_tile_gemm_w(const float* A, int lda, int nvalid, int K, int N, int wave, int nwaves, int lane,
                                               const DlW<NT, KC>& w, Epi epi, const float* __restrict__ addend = nullptr,
                                               int add_ld = 0) {
    const int j = lane & 15, g = lane >> 4;
    const float* xa = A + (size_t)j * lda + 4 * g;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int ct = wave + t * nwaves;
        if (ct >= (N >> 4)) break;
        const int col = ct * 16 + j;
        float ad[4] = {0.f, 0.f, 0.f, 0.f};
        if (addend) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (4 * g + i < nvalid) ad[i] = addend[(size_t)(4 * g + i) * add_ld + col];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            if (kc < (K >> 4)) {
                float4 a = j < nvalid ? *reinterpret_cast<const float4*>(xa + kc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = dl_mfma4(a, w.b[t][kc], acc);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * g + i;
            if (r >= nvalid) continue;
            float v = acc[i] + w.bias[t];
            if (RELU) v = fmaxf(v, 0.f);
            if (addend) v += ad[i];
            epi(r, col, v);
        }
    }
}

// torch.nn.LayerNorm over 64 channels (biased variance, eps inside the root) of the rows of an LDS tile;
// one wave per row, one channel per lane
template <typename Out>
__device__ __forceinline__ void dl_tile_layernorm_p(const float (*S)[DL_LD], int nvalid, float wl, float bl, int wave,
                                                    int nwaves, int lane, Out out);
template <typename Out>
__device__ __forceinline__ void dl_tile_layernorm(const float (*S)[DL_LD], int nvalid, const float* __restrict__ w,
                                                  const float* __restrict__ b, int wave, int nwaves, int lane,
                                                  Out out) {
    dl_tile_layernorm_p(S, nvalid, w[lane], b[lane], wave, nwaves, lane, out);
}
// (wl, bl: the lane's channel of the norm's weight and bias, loaded by the caller -- ahead of the phase, if it likes)
template <typename Out>
__device__ __forceinline__ void dl_tile_layernorm_p(const float (*S)[DL_LD], int nvalid, float wl, float bl, int wave,
                                                    int nwaves, int lane, Out out) {
    for (int r = wave; r < nvalid; r += nwaves) {
        const float v = S[r][lane];
        // (gf_wave_sum: the same butterfly as six __shfl_xor steps without the LDS crossbar -- a norm of 16 rows by four
        //  waves was 48 dependent ds_bpermute round trips, ~5 300 cycles)
        const float s = gf_wave_sum(v);
        const float mu = s / (float)DL_D;
        const float dv = v - mu;
        const float q = gf_wave_sum(dv * dv);
        const float rstd = 1.0f / sqrtf(q / (float)DL_D + 1e-5f);
        out(r, lane, dv * rstd * wl + bl);
    }
}

