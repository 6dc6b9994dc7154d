// Output-stationary sparse convolution for gfx950: gather -> fp32 MFMA -> single store.
//
//   out[o,:] = sum_k act(in[nbr[k][o],:]) @ W[k]  (+ residual[o,:])
//
// One wave owns a GROUP of 16 consecutive output rows (and NCBW 16-column blocks of the
// output).  For every kernel offset k present anywhere in the group (gmask bit) each lane
// (r = lane&15, q = lane>>4) gathers 16 bytes of row nbr[k][16g+r] straight into the
// A-operand layout of v_mfma_f32_16x16x4_f32 -- no LDS staging, no scatter atomics: the 4
// lanes that share a row cover one contiguous 64-byte segment, so one wave-instruction
// fetches 16 row segments.  The MFMA k index is only a summation index, so lane q supplies
// channels c0+4q .. c0+4q+3 of the current 16-channel chunk; the weights are pre-packed
// (gf_conv_pack_weights) so that the matching B operands of a (k, chunk, column block) are
// ONE coalesced 16-byte load per lane.  Gathers, index loads and weight loads of the next
// step are issued before the MFMAs of the current one.  Accumulation is exact fp32 (the
// f32 MFMA is a k-ordered fmaf chain); accumulators stay in registers across all offsets;
// eval-mode BatchNorm+ReLU in front of every U-Net conv (geoformer_modules.py:19-26) and
// the residual add (geoformer_modules.py:33) fuse as prologue / epilogue.
//
// Roofline: HBM-bound for C <= 80.  Algorithmic bytes per launch are
// 4*(R*Cin + M_out*Cout + K*Cin*Cout) + 8*R with R = number of non-negative table entries.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

extern "C" size_t gf_conv_packed_floats(int K, int Cin, int Cout) {
    const size_t nch = (Cin + 15) / 16, ncb = (Cout + 15) / 16;
    return (size_t)K * nch * ncb * 64 * 4;
}

// Wp[(((k*NCH + ch)*NCB + cb)*64 + lane)*4 + kk] = W[k][ch*16 + 4*(lane>>4) + kk][cb*16 + (lane&15)]
__global__ void k_pack_weights(const float* __restrict__ W, int K, int Cin, int Cout, int NCH, int NCB,
                               float* __restrict__ Wp) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)K * NCH * NCB * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    size_t u = t >> 6;
    const int cb = (int)(u % NCB);
    u /= NCB;
    const int ch = (int)(u % NCH);
    const int k = (int)(u / NCH);
    const int r = lane & 15, q = lane >> 4;
    const int col = cb * 16 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int row = ch * 16 + 4 * q + kk;
        if (row < Cin && col < Cout) pv[kk] = W[((size_t)k * Cin + row) * Cout + col];
    }
    reinterpret_cast<float4*>(Wp)[t] = v;
}

extern "C" int gf_conv_pack_weights(const float* W, int K, int Cin, int Cout, float* Wp, void* stream) {
    GF_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1, "gf_conv_pack_weights: bad sizes");
    const int nch = (Cin + 15) / 16, ncb = (Cout + 15) / 16;
    size_t total = (size_t)K * nch * ncb * 64;
    hipLaunchKernelGGL(k_pack_weights, dim3(gf_div_up((long long)total, 256)), dim3(256), 0, (hipStream_t)stream, W, K,
                       Cin, Cout, nch, ncb, Wp);
    GF_CHECK_LAUNCH("gf_conv_pack_weights");
    return GF_OK;
}

template <bool VEC>
__device__ __forceinline__ float4 load_a(const float* __restrict__ in, int idx, int Cin, int ch,
                                         const float* __restrict__ sc, const float* __restrict__ sh) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx >= 0) {
        const float* src = in + (size_t)idx * Cin + ch;
        if (VEC) {
            a = *reinterpret_cast<const float4*>(src);
            if (sc) {
                const float4 s = *reinterpret_cast<const float4*>(sc + ch);
                const float4 t = *reinterpret_cast<const float4*>(sh + ch);
                a.x = fmaxf(fmaf(a.x, s.x, t.x), 0.f);
                a.y = fmaxf(fmaf(a.y, s.y, t.y), 0.f);
                a.z = fmaxf(fmaf(a.z, s.z, t.z), 0.f);
                a.w = fmaxf(fmaf(a.w, s.w, t.w), 0.f);
            }
        } else {
            float* pa = reinterpret_cast<float*>(&a);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ch + j < Cin) {
                    float v = src[j];
                    if (sc) v = fmaxf(fmaf(v, sc[ch + j], sh[ch + j]), 0.f);
                    pa[j] = v;
                }
        }
    }
    return a;
}

template <int NCBW, bool VEC>
__global__ __launch_bounds__(256) void k_conv_os(const float* __restrict__ in, const float4* __restrict__ Wp,
                                                 const int32_t* __restrict__ nbr, const uint32_t* __restrict__ gmask,
                                                 int K, int M_out, int ld, int Cin, int Cout, int NCH, int NCB,
                                                 int nsplit, const float* __restrict__ in_scale,
                                                 const float* __restrict__ in_shift,
                                                 const float* __restrict__ residual, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int ngroups = (M_out + 15) >> 4;
    const int nitems = ngroups * nsplit;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;

    for (int item = wave; item < nitems; item += nwaves) {
        const int g = item / nsplit;
        const int cb0 = (item - g * nsplit) * NCBW;
        const int o = g * 16 + r;
        const bool row_ok = o < M_out;
        f32x4 acc[NCBW];
#pragma unroll
        for (int cb = 0; cb < NCBW; cb++) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        uint32_t mask = nbr ? (gmask ? gmask[g] : ((K >= 32) ? 0xffffffffu : ((1u << K) - 1u))) : 1u;
        mask = __builtin_amdgcn_readfirstlane(mask);
        if (mask) {
            // (k, idx) of the current offset and of the next one (index loaded one offset ahead)
            int k = __builtin_ctz(mask);
            mask &= mask - 1;
            int idx = row_ok ? (nbr ? nbr[(size_t)k * ld + o] : o) : -1;
            int k2 = -1, idx2 = -1;
            if (mask) {
                k2 = __builtin_ctz(mask);
                mask &= mask - 1;
                idx2 = row_ok ? nbr[(size_t)k2 * ld + o] : -1;
            }
            int c = 0;
            float4 a = load_a<VEC>(in, idx, Cin, 4 * q, in_scale, in_shift);
            float4 b[NCBW];
#pragma unroll
            for (int cb = 0; cb < NCBW; cb++)
                b[cb] = (cb0 + cb < NCB) ? Wp[(((size_t)k * NCH + 0) * NCB + cb0 + cb) * 64 + lane]
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            while (true) {
                // ---- issue the loads of the next step ----
                int nc = c + 1;
                bool more = true;
                if (nc == NCH) {
                    nc = 0;
                    if (k2 >= 0) {
                        k = k2;
                        idx = idx2;
                        if (mask) {
                            k2 = __builtin_ctz(mask);
                            mask &= mask - 1;
                            idx2 = row_ok ? nbr[(size_t)k2 * ld + o] : -1;
                        } else {
                            k2 = -1;
                        }
                    } else {
                        more = false;
                    }
                }
                float4 an = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 bn[NCBW];
                if (more) {
                    an = load_a<VEC>(in, idx, Cin, nc * 16 + 4 * q, in_scale, in_shift);
#pragma unroll
                    for (int cb = 0; cb < NCBW; cb++)
                        bn[cb] = (cb0 + cb < NCB) ? Wp[(((size_t)k * NCH + nc) * NCB + cb0 + cb) * 64 + lane]
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                // ---- MFMAs of the current step ----
#pragma unroll
                for (int cb = 0; cb < NCBW; cb++) {
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[cb].x, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[cb].y, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[cb].z, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[cb].w, acc[cb], 0, 0, 0);
                }
                if (!more) break;
                a = an;
#pragma unroll
                for (int cb = 0; cb < NCBW; cb++) b[cb] = bn[cb];
                c = nc;
            }
        }
        // C/D layout: col = lane&15, row = (lane>>4)*4 + j
#pragma unroll
        for (int cb = 0; cb < NCBW; cb++) {
            const int col = (cb0 + cb) * 16 + r;
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

struct ConvArgs {
    const float* in;
    const float4* Wp;
    const int32_t* nbr;
    const uint32_t* gmask;
    int K, M_out, ld, Cin, Cout, NCH, NCB, nsplit;
    const float *sc, *sh, *res;
    float* out;
};

template <int NCBW>
static void launch_conv(bool vec, dim3 grid, hipStream_t st, const ConvArgs& a) {
    if (vec)
        hipLaunchKernelGGL((k_conv_os<NCBW, true>), grid, dim3(256), 0, st, a.in, a.Wp, a.nbr, a.gmask, a.K, a.M_out,
                           a.ld, a.Cin, a.Cout, a.NCH, a.NCB, a.nsplit, a.sc, a.sh, a.res, a.out);
    else
        hipLaunchKernelGGL((k_conv_os<NCBW, false>), grid, dim3(256), 0, st, a.in, a.Wp, a.nbr, a.gmask, a.K, a.M_out,
                           a.ld, a.Cin, a.Cout, a.NCH, a.NCB, a.nsplit, a.sc, a.sh, a.res, a.out);
}

extern "C" int gf_conv_fwd(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask, int K,
                           int M_out, int ld, int Cin, int Cout, const float* in_scale, const float* in_shift,
                           const float* residual, float* out, void* stream) {
    GF_CHECK_ARG(K >= 1 && K <= 32, "gf_conv_fwd: K=%d out of range [1,32]", K);
    GF_CHECK_ARG(Cin >= 1 && Cout >= 1 && Cout <= 128, "gf_conv_fwd: Cin=%d Cout=%d unsupported (Cout<=128)", Cin,
                 Cout);
    GF_CHECK_ARG(nbr != nullptr || K == 1, "gf_conv_fwd: nbr==NULL requires K==1");
    GF_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "gf_conv_fwd: in_scale/in_shift must come together");
    if (M_out <= 0) return GF_OK;
    const int ngroups = (M_out + 15) / 16;
    const int ncb = (Cout + 15) / 16, nch = (Cin + 15) / 16;
    const bool vec = (Cin % 16) == 0 && (((uintptr_t)in) % 16) == 0 &&
                     (in_scale == nullptr || ((((uintptr_t)in_scale) | ((uintptr_t)in_shift)) % 16) == 0);
    // column blocks per wave: as many as possible while the launch still has >= 2048 waves
    int ncbw = 1;
    for (int cand = ncb; cand >= 1; cand--) {
        const int split = (ncb + cand - 1) / cand;
        if ((long long)ngroups * split >= 2048) {
            ncbw = cand;
            break;
        }
    }
    const int nsplit = (ncb + ncbw - 1) / ncbw;
    long long nitems = (long long)ngroups * nsplit;
    int blocks = (int)((nitems + 3) / 4);  // 4 waves per 256-thread block, one item per wave
    if (blocks > 256 * 64) blocks = 256 * 64;
    ConvArgs a{in, reinterpret_cast<const float4*>(Wp), nbr, gmask, K, M_out, ld, Cin, Cout, nch, ncb, nsplit,
               in_scale, in_shift, residual, out};
    dim3 grid(blocks);
    hipStream_t st = (hipStream_t)stream;
    switch (ncbw) {
        case 1: launch_conv<1>(vec, grid, st, a); break;
        case 2: launch_conv<2>(vec, grid, st, a); break;
        case 3: launch_conv<3>(vec, grid, st, a); break;
        case 4: launch_conv<4>(vec, grid, st, a); break;
        case 5: launch_conv<5>(vec, grid, st, a); break;
        case 6: launch_conv<6>(vec, grid, st, a); break;
        case 7: launch_conv<7>(vec, grid, st, a); break;
        default: launch_conv<8>(vec, grid, st, a); break;
    }
    GF_CHECK_LAUNCH("gf_conv_fwd");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// weight gradient: dW[k] = sum_o in[nbr[k][o],:]^T dOut[o,:]
// One wave per (offset k, 16-channel input block, 16-channel output block, slice of rows).
// MFMA orientation: M = input channel, N = output channel, K = rows (4 per instruction), so
// both operands are 64-byte row segments; partial tiles are combined with fp32 atomics
// (K*Cin*Cout words only -- far below the atomic-rate ceiling).
// ------------------------------------------------------------------------------------
#define WG_ROWS 2048
__global__ __launch_bounds__(256) void k_conv_wgrad(const float* __restrict__ in, const float* __restrict__ dout,
                                                    const int32_t* __restrict__ nbr, int K, int M_out, int ld,
                                                    int Cin, int Cout, int NCI, int NCO, int nslices,
                                                    float* __restrict__ dW) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    long long item = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nitems = (long long)K * NCI * NCO * nslices;
    if (item >= nitems) return;
    const int sl = (int)(item % nslices);
    item /= nslices;
    const int cob = (int)(item % NCO);
    item /= NCO;
    const int cib = (int)(item % NCI);
    const int k = (int)(item / NCI);
    const int ci = cib * 16 + r, co = cob * 16 + r;
    const int row0 = sl * WG_ROWS, row1 = min(M_out, row0 + WG_ROWS);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int base = row0; base < row1; base += 4) {
        const int row = base + q;
        float a = 0.f, b = 0.f;
        if (row < row1) {
            const int idx = nbr ? nbr[(size_t)k * ld + row] : row;
            if (idx >= 0) {
                if (ci < Cin) a = in[(size_t)idx * Cin + ci];
                if (co < Cout) b = dout[(size_t)row * Cout + co];
            }
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    // D layout: col (output channel) = lane&15, row (input channel) = 4*(lane>>4) + j
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int cii = cib * 16 + q * 4 + j;
        if (cii < Cin && co < Cout && acc[j] != 0.f) atomicAdd(&dW[((size_t)k * Cin + cii) * Cout + co], acc[j]);
    }
}

extern "C" int gf_conv_wgrad(const float* in, const float* dout, const int32_t* nbr, int K, int M_out, int ld, int Cin,
                             int Cout, float* dW, void* stream) {
    GF_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1, "gf_conv_wgrad: bad sizes");
    GF_CHECK_ARG(nbr != nullptr || K == 1, "gf_conv_wgrad: nbr==NULL requires K==1");
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(dW, 0, (size_t)K * Cin * Cout * sizeof(float), st);
    if (M_out <= 0) return GF_OK;
    const int nci = (Cin + 15) / 16, nco = (Cout + 15) / 16;
    const int nslices = (M_out + WG_ROWS - 1) / WG_ROWS;
    const long long nitems = (long long)K * nci * nco * nslices;
    hipLaunchKernelGGL(k_conv_wgrad, dim3(gf_div_up(nitems, 4)), dim3(256), 0, st, in, dout, nbr, K, M_out, ld, Cin,
                       Cout, nci, nco, nslices, dW);
    GF_CHECK_LAUNCH("gf_conv_wgrad");
    return GF_OK;
}
