// Output-stationary sparse convolution for gfx950: gather -> fp32 MFMA -> single store.
//
//   out[o,:] = sum_k act(in[nbr[k][o],:]) @ W[k]  (+ residual[o,:])
//
// One wave owns a GROUP of 16 consecutive output rows.  For every kernel offset k that is
// present anywhere in the group (gmask bit), each lane (r = lane&15, q = lane>>4) gathers
// 16 bytes of row nbr[k][16g+r] straight into the A-operand layout of
// v_mfma_f32_16x16x4_f32 -- no LDS staging and no scatter atomics: the 4 lanes that share a
// row cover one contiguous 64-byte segment, so a wave-instruction fetches 16 row segments.
// The MFMA k index is only a summation index, so lane q supplies channels
// c0 + 4q .. c0 + 4q + 3 of the current 16-channel chunk and the B operand uses the same
// permutation (channel c0 + 4q + kk in step kk).  Accumulation is exact fp32 (the f32 MFMA
// is a k-ordered fmaf chain), accumulators stay in registers over all offsets, and the
// eval-mode BatchNorm+ReLU that precedes every conv of the U-Net (geoformer_modules.py:19-26)
// and the residual add (geoformer_modules.py:33) can be fused as prologue / epilogue.
//
// Roofline: HBM-bound for C <= 80.  Algorithmic bytes per launch are
// 4*(R*Cin + M_out*Cout + K*Cin*Cout) + 8*R with R = number of non-negative table entries.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCB, bool VEC>
__global__ __launch_bounds__(256) void k_conv_os(const float* __restrict__ in, const float* __restrict__ W,
                                                 const int32_t* __restrict__ nbr, const uint32_t* __restrict__ gmask,
                                                 int K, int M_out, int ld, int Cin, int Cout,
                                                 const float* __restrict__ in_scale,
                                                 const float* __restrict__ in_shift,
                                                 const float* __restrict__ residual, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int ngroups = (M_out + 15) >> 4;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const bool act = in_scale != nullptr;

    for (int g = wave; g < ngroups; g += nwaves) {
        const int o = g * 16 + r;
        f32x4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        uint32_t mask = nbr ? (gmask ? gmask[g] : ((K >= 32) ? 0xffffffffu : ((1u << K) - 1u))) : 1u;
        mask = __builtin_amdgcn_readfirstlane(mask);
        while (mask) {
            const int k = __builtin_ctz(mask);
            mask &= mask - 1;
            int idx = -1;
            if (o < M_out) idx = nbr ? nbr[(size_t)k * ld + o] : o;
            const float* src = in + (size_t)(idx < 0 ? 0 : idx) * Cin;
            const float* wk = W + (size_t)k * Cin * Cout;
            for (int c0 = 0; c0 < Cin; c0 += 16) {
                const int ch = c0 + q * 4;
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                if (idx >= 0) {
                    if (VEC) {
                        const float4 v = *reinterpret_cast<const float4*>(src + ch);
                        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (ch + j < Cin) a[j] = src[ch + j];
                    }
                    if (act) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (VEC || ch + j < Cin) a[j] = fmaxf(fmaf(a[j], in_scale[ch + j], in_shift[ch + j]), 0.f);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int wc = ch + kk;  // weight row (input channel) this lane supplies in step kk
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) {
                        const int col = cb * 16 + r;
                        float b = 0.f;
                        if ((VEC || wc < Cin) && col < Cout) b = wk[(size_t)wc * Cout + col];
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b, acc[cb], 0, 0, 0);
                    }
                }
            }
        }
        // C/D layout: col = lane&15, row = (lane>>4)*4 + j
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            const int col = cb * 16 + r;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = g * 16 + q * 4 + j;
                if (row < M_out && col < Cout) {
                    float v = acc[cb][j];
                    if (residual) v += residual[(size_t)row * Cout + col];
                    out[(size_t)row * Cout + col] = v;
                }
            }
        }
    }
}

template <int NCB>
static void launch_conv(bool vec, dim3 grid, hipStream_t st, const float* in, const float* W, const int32_t* nbr,
                        const uint32_t* gmask, int K, int M_out, int ld, int Cin, int Cout, const float* sc,
                        const float* sh, const float* res, float* out) {
    if (vec)
        hipLaunchKernelGGL((k_conv_os<NCB, true>), grid, dim3(256), 0, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout,
                           sc, sh, res, out);
    else
        hipLaunchKernelGGL((k_conv_os<NCB, false>), grid, dim3(256), 0, st, in, W, nbr, gmask, K, M_out, ld, Cin,
                           Cout, sc, sh, res, out);
}

extern "C" int gf_conv_fwd(const float* in, const float* W, const int32_t* nbr, const uint32_t* gmask, int K,
                           int M_out, int ld, int Cin, int Cout, const float* in_scale, const float* in_shift,
                           const float* residual, float* out, void* stream) {
    GF_CHECK_ARG(K >= 1 && K <= 32, "gf_conv_fwd: K=%d out of range [1,32]", K);
    GF_CHECK_ARG(Cin >= 1 && Cout >= 1 && Cout <= 128, "gf_conv_fwd: Cin=%d Cout=%d unsupported (Cout<=128)", Cin,
                 Cout);
    GF_CHECK_ARG(nbr != nullptr || K == 1, "gf_conv_fwd: nbr==NULL requires K==1");
    GF_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "gf_conv_fwd: in_scale/in_shift must come together");
    if (M_out <= 0) return GF_OK;
    const int ngroups = (M_out + 15) / 16;
    const int ncb = (Cout + 15) / 16;
    const bool vec = (Cin % 16) == 0 && (((uintptr_t)in) % 16) == 0;
    int blocks = (ngroups + 3) / 4;  // 4 waves (groups) per 256-thread block
    if (blocks > 256 * 8) blocks = 256 * 8;
    dim3 grid(blocks);
    hipStream_t st = (hipStream_t)stream;
    switch (ncb) {
        case 1: launch_conv<1>(vec, grid, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out); break;
        case 2: launch_conv<2>(vec, grid, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out); break;
        case 3: launch_conv<3>(vec, grid, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out); break;
        case 4: launch_conv<4>(vec, grid, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out); break;
        case 5: launch_conv<5>(vec, grid, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out); break;
        case 6: launch_conv<6>(vec, grid, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out); break;
        case 7: launch_conv<7>(vec, grid, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out); break;
        default: launch_conv<8>(vec, grid, st, in, W, nbr, gmask, K, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out); break;
    }
    GF_CHECK_LAUNCH("gf_conv_fwd");
    return GF_OK;
}
