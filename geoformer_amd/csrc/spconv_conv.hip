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
#include <stdlib.h>

#include "common.h"
#include <hip/hip_ext.h>
#include "geoformer_hip_dev.h"
#include "conv_pack.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CONV_MAX_CIN 512

extern "C" size_t gf_conv_packed_floats(int K, int Cin, int Cout) {
    const size_t nch = (Cin + 15) / 16, ncb = (Cout + 15) / 16;
    return (size_t)K * nch * ncb * 64 * 4;
}

// Wp[(((k*NCH + ch)*NCB + cb)*64 + lane)*4 + kk] = W[k][ch*16 + 4*(lane>>4) + kk][cb*16 + (lane&15)]
// (element functions in conv_pack.h: the training executor packs all of a step's weights in one launch)
__global__ void k_pack_weights(const float* __restrict__ W, int K, int Cin, int Cout, int NCH, int NCB,
                               float* __restrict__ Wp) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)K * NCH * NCB * 64;
    if (t >= total) return;
    reinterpret_cast<float4*>(Wp)[t] = gf_pack_weights_elem(W, Cin, Cout, NCH, NCB, t);
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

// The input gradient's weights in one launch: Wp = pack(W'), W'[k] = W[flip ? K-1-k : k]^T  ([K,Cin,Cout] -> the packed
// stream of a [K,Cout,Cin] operand).  (A flip copy + a transpose copy + the pack: three launches per convolution and
// backward, 128 launches per training step.)
__global__ void k_pack_weights_t(const float* __restrict__ W, int K, int Cin, int Cout, int NCH, int NCB, int flip,
                                 float* __restrict__ Wp) {
    // the packed operand has Cout rows (input channels of the gradient convolution) and Cin columns
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)K * NCH * NCB * 64;
    if (t >= total) return;
    reinterpret_cast<float4*>(Wp)[t] = gf_pack_weights_t_elem(W, K, Cin, Cout, NCH, NCB, flip, t);
}

extern "C" int gf_conv_pack_weights_t(const float* W, int K, int Cin, int Cout, int flip, float* Wp, void* stream) {
    GF_CHECK_ARG(W && Wp && K >= 1 && Cin >= 1 && Cout >= 1, "gf_conv_pack_weights_t: bad arguments");
    const int nch = (Cout + 15) / 16, ncb = (Cin + 15) / 16;
    size_t total = (size_t)K * nch * ncb * 64;
    hipLaunchKernelGGL(k_pack_weights_t, dim3(gf_div_up((long long)total, 256)), dim3(256), 0, (hipStream_t)stream, W, K,
                       Cin, Cout, nch, ncb, flip, Wp);
    GF_CHECK_LAUNCH("gf_conv_pack_weights_t");
    return GF_OK;
}

// Raw gather of this lane's 4 channels of row idx (zeros for a missing neighbour).  The fused
// BatchNorm+ReLU is applied later (activate_a), next to the MFMAs, so that the gathers of a batch
// are issued back to back instead of each waiting for its own data.
template <bool VEC>
__device__ __forceinline__ float4 load_a(const float* __restrict__ in, int idx, int Cin, int ch) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx >= 0) {
        const float* src = in + (size_t)idx * Cin + ch;
        if (VEC) {
            a = *reinterpret_cast<const float4*>(src);
        } else {
            float* pa = reinterpret_cast<float*>(&a);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ch + j < Cin) pa[j] = src[j];
        }
    }
    return a;
}

// act(x) = max(x*scale + shift, 0) on a present row; sc/sh live in LDS, padded with zeros past Cin
__device__ __forceinline__ float4 activate_a(float4 a, bool present, const float* sc, const float* sh, int ch) {
    if (present) {
        const float4 s = *reinterpret_cast<const float4*>(sc + ch);
        const float4 t = *reinterpret_cast<const float4*>(sh + ch);
        a.x = fmaxf(fmaf(a.x, s.x, t.x), 0.f);
        a.y = fmaxf(fmaf(a.y, s.y, t.y), 0.f);
        a.z = fmaxf(fmaf(a.z, s.z, t.z), 0.f);
        a.w = fmaxf(fmaf(a.w, s.w, t.w), 0.f);
    }
    return a;
}

// Per-wave schedule.  A "step" is one (kernel offset k, 16-channel chunk c) pair of the group; the
// steps of a group are the present offsets (gmask) x NCH chunks.  The wave first stages the group's
// neighbour indices in LDS (only the present offsets), then walks its steps in batches of PF: all
// gathers and weight loads of a batch are issued back to back (PF x 1 KiB gathered rows in flight per
// wave), then the batch's MFMAs run.  With SPLIT the four waves of the workgroup share one item and
// take the steps round-robin (small levels: 4x more waves, 4x shorter dependent chains), and the
// partial accumulators are summed through LDS.
// Steps in flight per wave.  The kernel is bound by dependent latency, and every resident wave helps to hide
// the parts of a group's chain that prefetching cannot (mask -> indices -> first gathers, the epilogue):
// PF 8/6/4 cost 135-150 VGPRs = 3 waves per SIMD; PF 4/2/3 fit 96-112 = 5/5/4 waves and measured 18-23 % faster
// on every level (rocprofv3 kernel trace, S150k: level 1 31.9 -> 26.3 us, level 2 39.9 -> 30.8, level 3 30.3 -> 23.7).
#ifndef CONV_PF1
#define CONV_PF1 4
#endif
#ifndef CONV_PF2
#define CONV_PF2 2
#endif
#ifndef CONV_PF3
#define CONV_PF3 3
#endif
#ifndef CONV_WPE
#define CONV_WPE_ATTR
#else
#define CONV_WPE_ATTR __attribute__((amdgpu_waves_per_eu(CONV_WPE, CONV_WPE)))
#endif
template <int NCBW>
struct ConvPF {
    static constexpr int value = NCBW == 1 ? CONV_PF1 : (NCBW == 2 ? CONV_PF2 : CONV_PF3);
};

struct StepIter {
    uint32_t m;  // offsets not yet visited
    int k, c, nch;
    __device__ __forceinline__ void init(uint32_t mask, int nch_) {
        nch = nch_;
        c = 0;
        if (mask) {
            k = __builtin_ctz(mask);
            m = mask & (mask - 1);
        } else {
            k = -1;
            m = 0;
        }
    }
    __device__ __forceinline__ void next() {
        if (k < 0) return;
        if (++c == nch) {
            c = 0;
            if (m) {
                k = __builtin_ctz(m);
                m &= m - 1;
            } else {
                k = -1;
            }
        }
    }
};

template <int NCBW, int SW, bool VEC>
__global__ CONV_WPE_ATTR __launch_bounds__(SW > 8 ? 1024 : 256, (SW <= 8 && NCBW <= 2) ? 5 : 1) void k_conv_os(const float* __restrict__ in, const float4* __restrict__ Wp,
                                                 const int32_t* __restrict__ nbr, const uint32_t* __restrict__ gmask,
                                                 int K, int M_out, int ld, int Cin, int Cout, int NCH, int NCB,
                                                 int nsplit, unsigned in_bytes, const float* __restrict__ in_scale,
                                                 const float* __restrict__ in_shift,
                                                 const float* __restrict__ residual, const float* __restrict__ out_scale,
                                                 const float* __restrict__ out_shift, float* __restrict__ out) {
    constexpr int PF = ConvPF<NCBW>::value;
    constexpr bool SPLIT = SW > 0;  // SW waves of the workgroup share one item
    constexpr int NSW = SW > 0 ? SW : 1;
    __shared__ int s_idx[SPLIT ? NSW : 8][32 * 16];
    __shared__ int s_kl[SPLIT ? NSW : 8][32];  // present offsets of the wave's item, ascending
    __shared__ float4 s_red[SPLIT ? NSW * NCBW * 64 : 1];
    __shared__ __attribute__((aligned(16))) float s_aff[2][CONV_MAX_CIN];  // fused BN scale / shift
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int ngroups = (M_out + 15) >> 4;
    const int nitems = ngroups * nsplit;
    const int first = SPLIT ? blockIdx.x : ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int stride = SPLIT ? gridDim.x : ((gridDim.x * blockDim.x) >> 6);
    int* idx_l = s_idx[w];
    // VEC path: gathers and weight loads go through buffer descriptors -- a 32-bit per-lane offset instead of
    // 64-bit pointer arithmetic, a scalar offset for the (uniform) weight block, and the hardware range
    // check turns "missing neighbour" (offset 0xffffffff) into zeros without a branch
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, K * NCH * NCB * 1024, 0x00020000);
    const unsigned rowbytes = (unsigned)Cin * 4u;
    const float* sc_l = nullptr;
    const float* sh_l = nullptr;
    // SPLIT (all waves of the workgroup walk the same items): the prologue vectors are only REQUESTED here and land in
    // LDS behind the first item's mask / index requests -- one memory round trip for the three instead of two
    float av_s[2] = {0.f, 0.f}, av_t[2] = {0.f, 0.f};
    bool aff_pending = false;
    if (in_scale) {  // workgroup-uniform
        if (SPLIT && NCH * 16 <= 2 * (int)blockDim.x) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int c = threadIdx.x + e * blockDim.x;
                if (c < Cin) {
                    av_s[e] = in_scale[c];
                    av_t[e] = in_shift[c];
                }
            }
            aff_pending = true;
        } else {
            for (int c = threadIdx.x; c < NCH * 16; c += blockDim.x) {
                s_aff[0][c] = c < Cin ? in_scale[c] : 0.f;
                s_aff[1][c] = c < Cin ? in_shift[c] : 0.f;
            }
            __syncthreads();
        }
        sc_l = s_aff[0];
        sh_l = s_aff[1];
    }

    for (int item0 = first; item0 < nitems; item0 += stride) {
        const int item = item0;
        const int g = item / nsplit;
        const int cb0 = (item - g * nsplit) * NCBW;
        const int o = g * 16 + r;
        const bool row_ok = o < M_out;
        f32x4 acc[NCBW];
#pragma unroll
        for (int cb = 0; cb < NCBW; cb++) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // residual rows of this lane's accumulator elements, requested up front (epilogue add)
        float res[NCBW <= 2 ? NCBW : 1][4];
        if (NCBW <= 2 && residual) {
#pragma unroll
            for (int cb = 0; cb < (NCBW <= 2 ? NCBW : 1); cb++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int row = g * 16 + q * 4 + j, col = (cb0 + cb) * 16 + r;
                    res[cb][j] = (row < M_out && col < Cout && (!SPLIT || (cb % NSW) == w))
                                     ? residual[(size_t)row * Cout + col] : 0.f;
                }
        }
        uint32_t mask = nbr ? (gmask ? gmask[g] : ((K >= 32) ? 0xffffffffu : ((1u << K) - 1u))) : 1u;
        // epilogue vectors of this lane's columns, requested with everything else
        float osc_r[NCBW <= 2 ? NCBW : 1], osh_r[NCBW <= 2 ? NCBW : 1];
        if (NCBW <= 2) {
#pragma unroll
            for (int cb = 0; cb < (NCBW <= 2 ? NCBW : 1); cb++) {
                const int col = (cb0 + cb) * 16 + r;
                osc_r[cb] = (out_scale && col < Cout) ? out_scale[col] : 1.f;
                osh_r[cb] = (out_scale && col < Cout) ? out_shift[col] : 0.f;
            }
        }
        // stage the neighbour indices (lane (r,q) fetches offsets q, q+4, ...): all loads first -- one memory round trip
        // instead of up to eight dependent ones -- then the LDS writes.  The requests do not wait for the group mask (an
        // absent offset's entries are all -1 in the table anyway): mask and indices share a round trip
        {
            int iv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int k = q + 4 * i;
                const bool want = k < K && row_ok;
                iv[i] = want ? (nbr ? nbr[(size_t)k * ld + o] : o) : -1;
            }
            if (aff_pending) {  // workgroup-uniform, first item only
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int c = threadIdx.x + e * blockDim.x;
                    if (c < NCH * 16) {
                        s_aff[0][c] = av_s[e];
                        s_aff[1][c] = av_t[e];
                    }
                }
                __syncthreads();
                aff_pending = false;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int k = q + 4 * i;
                if (k < K) idx_l[k * 16 + r] = iv[i];
            }
        }
        // the steps of the item are (present offset i, chunk c), numbered s = i*NCH + c; a wave takes s = first, first +
        // stride, ...  (plain index arithmetic: the earlier bit-scanning iterator cost ~20 scalar branches per
        // step, 0.26 us per step and wave on the small levels -- more than the step's memory round trip)
        mask = __builtin_amdgcn_readfirstlane(mask);
        if (lane < 32 && ((mask >> lane) & 1u)) s_kl[w][__popc(mask & ((1u << lane) - 1u))] = lane;
        __builtin_amdgcn_wave_barrier();
        const int nsteps = __popc(mask) * NCH;
        const int sstride = SPLIT ? NSW : 1;
        const unsigned inv_nch = (65536u + (unsigned)NCH - 1u) / (unsigned)NCH;  // s / NCH for s < 4096, NCH <= 16
        for (int s0 = SPLIT ? w : 0; s0 < nsteps; s0 += PF * sstride) {
            float4 a[PF];
            float4 b[PF][NCBW];
            bool valid[PF];
            bool present[PF];
            int chv[PF];
            // offsets and neighbour rows of the whole batch first (independent LDS reads, one wait), then the loads
            int kq[PF], cq[PF], iq[PF];
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const int sj = min(s0 + j * sstride, nsteps - 1);
                const int ki = (int)(((unsigned)sj * inv_nch) >> 16);
                kq[j] = s_kl[w][ki];
                cq[j] = sj - ki * NCH;
            }
#pragma unroll
            for (int j = 0; j < PF; j++) {
                kq[j] = __builtin_amdgcn_readfirstlane(kq[j]);
                iq[j] = idx_l[kq[j] * 16 + r];
            }
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const int sj = s0 + j * sstride;
                valid[j] = sj < nsteps;
                present[j] = false;
                chv[j] = 0;
                if (valid[j]) {
                    struct { int k, c; } it;
                    it.k = kq[j];
                    it.c = cq[j];
                    const int idx = iq[j];
                    present[j] = idx >= 0;
                    chv[j] = it.c * 16 + 4 * q;
                    if (VEC) {
                        const unsigned voff = present[j] ? (unsigned)idx * rowbytes + (unsigned)chv[j] * 4u : 0xffffffffu;
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0);
                        a[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                                           __uint_as_float(v[3]));
                    } else {
                        a[j] = load_a<VEC>(in, idx, Cin, chv[j]);
                    }
                    if (VEC) {
                        const unsigned wblk = (unsigned)((it.k * NCH + it.c) * NCB + cb0);
#pragma unroll
                        for (int cb = 0; cb < NCBW; cb++) {
                            // blocks past NCB fall outside the descriptor range and read as zeros
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                rs_w, (unsigned)lane * 16u, (cb0 + cb < NCB) ? (wblk + cb) * 1024u : 0xfffffff0u, 0);
                            b[j][cb] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                                                   __uint_as_float(v[3]));
                        }
                    } else {
#pragma unroll
                        for (int cb = 0; cb < NCBW; cb++)
                            b[j][cb] = (cb0 + cb < NCB)
                                           ? Wp[(((size_t)it.k * NCH + it.c) * NCB + cb0 + cb) * 64 + lane]
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < PF; j++) {
                if (valid[j]) {
                    if (sc_l) a[j] = activate_a(a[j], present[j], sc_l, sh_l, chv[j]);
#pragma unroll
                    for (int cb = 0; cb < NCBW; cb++) {
                        const float4 bb = b[j][cb];
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, bb.x, acc[cb], 0, 0, 0);
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, bb.y, acc[cb], 0, 0, 0);
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, bb.z, acc[cb], 0, 0, 0);
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, bb.w, acc[cb], 0, 0, 0);
                    }
                }
            }
        }
        if (SPLIT) {
            __syncthreads();  // previous item's reduction reads are done
#pragma unroll
            for (int cb = 0; cb < NCBW; cb++)
                s_red[(w * NCBW + cb) * 64 + lane] = make_float4(acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]);
            __syncthreads();
        }
        // C/D layout: col = lane&15, row = (lane>>4)*4 + j.  SPLIT: wave w sums column blocks w, w+SW, ...
#pragma unroll
        for (int cb = 0; cb < NCBW; cb++) {
            float v[4] = {acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]};
            if (SPLIT) {
                if ((cb % NSW) != w) continue;
                float4 t[NSW / 4 > 0 ? NSW / 4 : 1];
#pragma unroll
                for (int e = 0; e < NSW / 4; e++) {
                    const float4 p0 = s_red[((4 * e + 0) * NCBW + cb) * 64 + lane];
                    const float4 p1 = s_red[((4 * e + 1) * NCBW + cb) * 64 + lane];
                    const float4 p2 = s_red[((4 * e + 2) * NCBW + cb) * 64 + lane];
                    const float4 p3 = s_red[((4 * e + 3) * NCBW + cb) * 64 + lane];
                    t[e] = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                                       (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
                }
                float4 sum = t[0];
#pragma unroll
                for (int e = 1; e < NSW / 4; e++)
                    sum = make_float4(sum.x + t[e].x, sum.y + t[e].y, sum.z + t[e].z, sum.w + t[e].w);
                v[0] = sum.x;
                v[1] = sum.y;
                v[2] = sum.z;
                v[3] = sum.w;
            }
            const int col = (cb0 + cb) * 16 + r;
            // epilogue activation (the consumer's eval-mode BatchNorm + ReLU applied once per output element)
            const float osc = NCBW <= 2 ? osc_r[NCBW <= 2 ? cb : 0] : ((out_scale && col < Cout) ? out_scale[col] : 1.f);
            const float osh = NCBW <= 2 ? osh_r[NCBW <= 2 ? cb : 0] : ((out_scale && col < Cout) ? out_shift[col] : 0.f);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = g * 16 + q * 4 + j;
                if (row < M_out && col < Cout) {
                    float x = v[j];
                    if (residual) x += (NCBW <= 2) ? res[NCBW <= 2 ? cb : 0][j] : residual[(size_t)row * Cout + col];
                    if (out_scale) x = fmaxf(fmaf(x, osc, osh), 0.f);
                    out[(size_t)row * Cout + col] = x;
                }
            }
        }
    }
}

// Flat-chain form for the deep levels (a few hundred items at most: S150k levels 5-7 and every strided / inverse /
// 1x1x1 launch between them).  Such a launch is ONE dependent chain per workgroup and nothing else -- the chip is
// nearly empty -- so what it costs is the number of memory round trips in that chain, ~1 us each after the kernel
// boundary's cache invalidation.  k_conv_os<1,16> has eight of them in a row: prologue vectors -> (barrier) -> group
// mask -> neighbour indices -> (LDS) -> first batch of gathers and weights -> second batch -> (barrier) -> epilogue
// vectors -> store.  Here a workgroup's sixteen waves take the item's steps (offset k, 16-channel chunk c) = s,
// s + 16, ... over ALL K offsets -- no group mask, an absent offset is a step of zeros: 10-25 % more steps on these
// levels, each ~0.1 us of MFMA -- which makes every address of the chain known at launch: prologue / epilogue vectors,
// residual rows, ALL neighbour indices of the wave's steps and the weights of its first two batches are requested
// together, the gathers follow as one dependent round trip (two batches of FLAT_PF steps in flight, the later ones
// behind the MFMAs), then the LDS reduction over the waves and the store: three round trips.
#define FLAT_PF 4
#define FLAT_MAXB 6  // most batches per wave: K * NCH <= 16 * FLAT_PF * FLAT_MAXB = 384 steps per item (27 x 14: the 224 -> 112 layer)
#define FLAT_MAXB_SMALL 4  // the instance for items of at most 256 steps (every layer but the 2C -> C ones of the two deepest
                           // levels with a tail): the per-step index / address setup is unrolled MAXB * PF times, and six
                           // batches of it cost every flat launch ~1 us (8.6 -> 9.6 us at level 5, PMC of round 4)
// One convolution of the flat form.  `in2` (optional): the input rows are the CONCATENATION (in[:, :Cin1], in2[:, :Cin - Cin1])
// -- the skip concatenation of UBlock.forward (geoformer_modules.py:116) read in place, Cin1 a multiple of 16.
struct FlatOp {
    const float* in;
    const float* in2;
    const float4* Wp;
    const int32_t* nbr;
    const float *in_scale, *in_shift, *residual, *out_scale, *out_shift;
    float* out;
    int K, M_out, ld, Cin, Cin1, Cout, NCH, NCB;
    unsigned in_bytes, in2_bytes;
    int items, pad_;
};
template <int MAXB_>
__device__ __forceinline__ void conv_flat_item(const FlatOp& A, int item, float4* s_red, float (*s_aff)[CONV_MAX_CIN]) {
    constexpr int PF = FLAT_PF, MAXB = MAXB_;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const float* __restrict__ in = A.in;
    const int K = A.K, M_out = A.M_out, ld = A.ld, Cin = A.Cin, Cout = A.Cout, NCH = A.NCH, NCB = A.NCB;
    const int32_t* __restrict__ nbr = A.nbr;
    const float* __restrict__ in_scale = A.in_scale;
    const float* __restrict__ in_shift = A.in_shift;
    const float* __restrict__ residual = A.residual;
    const float* __restrict__ out_scale = A.out_scale;
    const float* __restrict__ out_shift = A.out_shift;
    float* __restrict__ out = A.out;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int g = item / NCB, cb = item - g * NCB;
    const int o = g * 16 + r;
    const bool row_ok = o < M_out;
    const int nsteps = K * NCH;
    const bool two = A.in2 != nullptr;
    const int nch1 = two ? A.Cin1 / 16 : NCH;  // chunks that come from `in`
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)A.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_in2 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(two ? A.in2 : in), 0, (int)(two ? A.in2_bytes : A.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)A.Wp, 0, K * NCH * NCB * 1024, 0x00020000);
    const unsigned rowbytes = (unsigned)(two ? A.Cin1 : Cin) * 4u, rowbytes2 = (unsigned)(Cin - (two ? A.Cin1 : 0)) * 4u;
    const unsigned inv_nch = (65536u + (unsigned)NCH - 1u) / (unsigned)NCH;  // s / NCH for s < 4096, NCH <= 16

    // ---- everything whose address is known now ----
    float av_s = 0.f, av_t = 0.f;
    const int tch = threadIdx.x;
    if (in_scale && tch < NCH * 16 && tch < Cin) {
        av_s = in_scale[tch];
        av_t = in_shift[tch];
    }
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    float osc = 1.f, osh = 0.f;
    const int col = cb * 16 + r;
    if (w == 0 && col < Cout) {
        if (residual) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = g * 16 + q * 4 + j;
                if (row < M_out) {
                    res[j] = residual[(size_t)row * Cout + col];
                }
            }
        }
        if (out_scale) {
            osc = out_scale[col];
            osh = out_shift[col];
        }
    }
    int idx[MAXB * PF];
    int kk[MAXB * PF], cc[MAXB * PF];  // uniform (scalar registers)
#pragma unroll
    for (int j = 0; j < MAXB * PF; j++) {
        const int s = w + 16 * j;
        const int sj = min(s, nsteps - 1);
        kk[j] = (int)(((unsigned)sj * inv_nch) >> 16);
        cc[j] = sj - kk[j] * NCH;
        idx[j] = -1;
        if (s < nsteps && row_ok) idx[j] = nbr ? nbr[(size_t)kk[j] * ld + o] : o;
    }
    float4 a[2][PF], b[2][PF];
    auto issue_w = [&](int bt) {  // weights of batch bt (uniform block, lane's 16 bytes)
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const int e = bt * PF + j;
            const unsigned wblk = (unsigned)((kk[e] * NCH + cc[e]) * NCB + cb);
            // (a step past the wave's last one restates that one's block: loaded, never multiplied)
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)lane * 16u, wblk * 1024u, 0);
            b[bt & 1][j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto issue_a = [&](int bt) {  // gathers of batch bt (a missing neighbour is out of the descriptor's range: zeros)
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const int e = bt * PF + j;
            u32x4 v;
            if (cc[e] < nch1) {  // (uniform)
                const unsigned voff = idx[e] >= 0 ? (unsigned)idx[e] * rowbytes + (unsigned)(cc[e] * 64 + q * 16) : 0xffffffffu;
                v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0);
            } else {
                const unsigned voff = idx[e] >= 0 ? (unsigned)idx[e] * rowbytes2 + (unsigned)((cc[e] - nch1) * 64 + q * 16) : 0xffffffffu;
                v = __builtin_amdgcn_raw_buffer_load_b128(rs_in2, voff, 0, 0);
            }
            a[bt & 1][j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    const int nb = w < nsteps ? ((nsteps - w + 15) / 16 + PF - 1) / PF : 0;  // batches of this wave (uniform)
    if (nb > 0) issue_w(0);
    if (nb > 1) issue_w(1);
    if (nb > 0) issue_a(0);
    if (nb > 1) issue_a(1);
    const float* sc_l = nullptr;
    const float* sh_l = nullptr;
    if (in_scale) {  // workgroup-uniform
        if (tch < NCH * 16) {
            s_aff[0][tch] = av_s;
            s_aff[1][tch] = av_t;
        }
        __syncthreads();
        sc_l = s_aff[0];
        sh_l = s_aff[1];
    }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int bt = 0; bt < MAXB; bt++) {
        if (bt < nb) {
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const int e = bt * PF + j;
                if (w + 16 * e < nsteps) {
                    float4 av = a[bt & 1][j];
                    if (sc_l) av = activate_a(av, idx[e] >= 0, sc_l, sh_l, cc[e] * 16 + 4 * q);
                    const float4 bb = b[bt & 1][j];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bb.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bb.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bb.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bb.w, acc, 0, 0, 0);
                }
            }
            if (bt + 2 < MAXB && bt + 2 < nb) {
                issue_w(bt + 2);
                issue_a(bt + 2);
            }
        }
    }
    s_red[w * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (w != 0) return;
    // fixed-order sum over the sixteen waves; C/D layout: col = lane & 15, row = (lane >> 4) * 4 + j
    float4 t[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float4 p0 = s_red[(4 * e + 0) * 64 + lane];
        const float4 p1 = s_red[(4 * e + 1) * 64 + lane];
        const float4 p2 = s_red[(4 * e + 2) * 64 + lane];
        const float4 p3 = s_red[(4 * e + 3) * 64 + lane];
        t[e] = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                           (p0.w + p1.w) + (p2.w + p3.w));
    }
    const float v[4] = {(t[0].x + t[1].x) + (t[2].x + t[3].x), (t[0].y + t[1].y) + (t[2].y + t[3].y),
                        (t[0].z + t[1].z) + (t[2].z + t[3].z), (t[0].w + t[1].w) + (t[2].w + t[3].w)};
    if (col < Cout) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = g * 16 + q * 4 + j;
            if (row < M_out) {
                float x = v[j];
                if (residual) x += res[j];
                if (out_scale) x = fmaxf(fmaf(x, osc, osh), 0.f);
                out[(size_t)row * Cout + col] = x;
            }
        }
    }
}

template <int MAXB_>
__global__ __launch_bounds__(1024, 1) void k_conv_flat(const float* __restrict__ in, const float4* __restrict__ Wp,
                                                       const int32_t* __restrict__ nbr, int K, int M_out, int ld, int Cin,
                                                       int Cout, int NCH, int NCB, unsigned in_bytes,
                                                       const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                       const float* __restrict__ residual, const float* __restrict__ out_scale,
                                                       const float* __restrict__ out_shift, float* __restrict__ out) {
    __shared__ float4 s_red[16 * 64];
    __shared__ __attribute__((aligned(16))) float s_aff[2][CONV_MAX_CIN];
    const FlatOp A{in, nullptr, Wp, nbr, in_scale, in_shift, residual, out_scale, out_shift, out, K, M_out, ld, Cin, Cin, Cout,
                   NCH, NCB, in_bytes, 0u, 0, 0};
    conv_flat_item<MAXB_>(A, (int)blockIdx.x, s_red, s_aff);
}

// Two 16-row groups per wave for the big 16-output-channel levels (level 1 of the U-Net: ~9000 groups).
// Measured on the S150k level-1 launch (variants with parts compiled out, rocprofv3 kernel trace): index staging +
// output 4.3 us, gathers 3.3 us, weight fetches 0.3 us, and 13.8 us for the loop WITHOUT any memory access, i.e. the
// kernel is instruction-issue bound: ~6 us of MFMA (32 cycles each per SIMD) plus the per-step address / control
// VALU work.  Hence: two groups share every step's control flow and weight fetch, the neighbour table is staged in
// LDS as ready-made byte offsets (missing = out of the descriptor's range, which reads as zeros), and the presence
// flags are only materialised for the fused BatchNorm prologue.  The steps walk the UNION of the two groups' offsets.
#ifndef CONV_PAIR_PF
#define CONV_PAIR_PF 2
#endif
#define CONV_PAIR_MISSING 0xfffff000u
template <bool AFF>
__global__ __launch_bounds__(256, 5) void k_conv_pair(const float* __restrict__ in, const float4* __restrict__ Wp,
                                                      const int32_t* __restrict__ nbr,
                                                      const uint32_t* __restrict__ gmask, int K, int M_out, int ld,
                                                      int Cin, int Cout, int NCH, unsigned in_bytes,
                                                      const float* __restrict__ in_scale,
                                                      const float* __restrict__ in_shift,
                                                      const float* __restrict__ residual,
                                                      const float* __restrict__ out_scale,
                                                      const float* __restrict__ out_shift, float* __restrict__ out) {
    constexpr int PF = CONV_PAIR_PF;
    __shared__ unsigned s_off[4][32 * 32];
    __shared__ __attribute__((aligned(16))) float s_aff[2][CONV_MAX_CIN];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int ngroups = (M_out + 15) >> 4;
    const int npairs = (ngroups + 1) >> 1;
    const int first = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int stride = (gridDim.x * blockDim.x) >> 6;
    unsigned* off_l = s_off[w];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, K * NCH * 1024, 0x00020000);
    const unsigned rowbytes = (unsigned)Cin * 4u;
    const unsigned lane_ch = 16u * (unsigned)q;  // byte offset of this lane's 4 channels inside a 16-channel chunk
    if (AFF) {
        for (int c = threadIdx.x; c < NCH * 16; c += blockDim.x) {
            s_aff[0][c] = c < Cin ? in_scale[c] : 0.f;
            s_aff[1][c] = c < Cin ? in_shift[c] : 0.f;
        }
        __syncthreads();
    }
    for (int pr = first; pr < npairs; pr += stride) {
        const int g0 = 2 * pr, g1 = g0 + 1;
        const int o0 = g0 * 16 + r, o1 = g1 * 16 + r;
        const bool ok0 = o0 < M_out, ok1 = o1 < M_out;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // residual rows of this lane's accumulator elements, requested up front (epilogue add)
        float res0[4] = {0.f, 0.f, 0.f, 0.f}, res1[4] = {0.f, 0.f, 0.f, 0.f};
        if (residual && r < Cout) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row0 = g0 * 16 + q * 4 + j, row1 = g1 * 16 + q * 4 + j;
                if (row0 < M_out) res0[j] = residual[(size_t)row0 * Cout + r];
                if (row1 < M_out) res1[j] = residual[(size_t)row1 * Cout + r];
            }
        }
        const uint32_t full = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
        const uint32_t m0 = gmask ? gmask[g0] : full, m1 = g1 < ngroups ? (gmask ? gmask[g1] : full) : 0u;
        const uint32_t mask = __builtin_amdgcn_readfirstlane(m0 | m1);
        // byte offsets of the neighbour rows of the union offsets: lane (r,q) fetches offsets q, q+4, ... of both groups
        {
            // all loads first (one round trip), then the LDS writes
            int i0[8], i1[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int k = q + 4 * i;
                const bool want = k < K && ((mask >> k) & 1u);
                i0[i] = (want && ok0) ? nbr[(size_t)k * ld + o0] : -1;
                i1[i] = (want && ok1) ? nbr[(size_t)k * ld + o1] : -1;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int k = q + 4 * i;
                if (k < K) {
                    off_l[k * 32 + r] = i0[i] >= 0 ? (unsigned)i0[i] * rowbytes : CONV_PAIR_MISSING;
                    off_l[k * 32 + 16 + r] = i1[i] >= 0 ? (unsigned)i1[i] * rowbytes : CONV_PAIR_MISSING;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        StepIter it;
        it.init(mask, NCH);
        while (it.k >= 0) {
            u32x4 x0[PF], x1[PF], wv[PF];
            unsigned v0[PF], v1[PF];
            int chn[PF];
            int nv = 0;
#pragma unroll
            for (int j = 0; j < PF; j++) {
                if (it.k >= 0) {
                    nv = j + 1;
                    const unsigned cbyte = (unsigned)it.c * 64u + lane_ch;
                    chn[j] = (int)(cbyte >> 2);
                    v0[j] = off_l[it.k * 32 + r] + cbyte;
                    v1[j] = off_l[it.k * 32 + 16 + r] + cbyte;
                    x0[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, v0[j], 0, 0);
                    x1[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, v1[j], 0, 0);
                    wv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)lane * 16u,
                                                                  (unsigned)(it.k * NCH + it.c) * 1024u, 0);
                }
                it.next();
            }
#pragma unroll
            for (int j = 0; j < PF; j++) {
                if (j < nv) {
                    float4 a0 = make_float4(__uint_as_float(x0[j][0]), __uint_as_float(x0[j][1]),
                                            __uint_as_float(x0[j][2]), __uint_as_float(x0[j][3]));
                    float4 a1 = make_float4(__uint_as_float(x1[j][0]), __uint_as_float(x1[j][1]),
                                            __uint_as_float(x1[j][2]), __uint_as_float(x1[j][3]));
                    if (AFF) {
                        a0 = activate_a(a0, v0[j] < CONV_PAIR_MISSING, s_aff[0], s_aff[1], chn[j]);
                        a1 = activate_a(a1, v1[j] < CONV_PAIR_MISSING, s_aff[0], s_aff[1], chn[j]);
                    }
                    const float b0 = __uint_as_float(wv[j][0]), b1 = __uint_as_float(wv[j][1]),
                                b2 = __uint_as_float(wv[j][2]), b3 = __uint_as_float(wv[j][3]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b0, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b1, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b2, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b2, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b3, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b3, acc1, 0, 0, 0);
                }
            }
        }
        // C/D layout: col = lane&15, row = (lane>>4)*4 + j
        if (r < Cout) {
            const float osc = out_scale ? out_scale[r] : 1.f, osh = out_scale ? out_shift[r] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row0 = g0 * 16 + q * 4 + j, row1 = g1 * 16 + q * 4 + j;
                float y0 = acc0[j] + res0[j], y1 = acc1[j] + res1[j];
                if (out_scale) {
                    y0 = fmaxf(fmaf(y0, osc, osh), 0.f);
                    y1 = fmaxf(fmaf(y1, osc, osh), 0.f);
                }
                if (row0 < M_out) out[(size_t)row0 * Cout + r] = y0;
                if (row1 < M_out) out[(size_t)row1 * Cout + r] = y1;
            }
        }
        __builtin_amdgcn_wave_barrier();  // off_l is rewritten by the next pair
    }
}

// ------------------------------------------------------------------------------------
// Counted-loop kernel for the 16-output-channel levels (level 1 of the U-Net: ~9000 groups, 9 launches per scene).
//
// What bound k_conv_pair (profiles/r1_e_pmc_conv_l1.md): ~700 scalar/vector instructions per wave around ~100 MFMAs
// (bit-scanning step iterator, index staging through LDS, four dword stores per row block) and a chain of dependent
// memory round trips per wave (mask -> indices -> gathers -> store).  Here
//   * the rulebook hands over a STEP TABLE (spconv_rules.hip k_subm3): per 16-row group only the present offsets, four
//     steps per 16-byte entry, so lane (row, *) holds the indices of up to 12 steps after three loads that do not
//     depend on the group's mask -- no LDS staging, no iterator; missing neighbours are index -1, which the
//     shift-and-add address turns into an out-of-range buffer offset (reads as zeros, no memory traffic);
//   * all gathers of the first GF_STEP_PHA blocks are issued back to back (one round trip), then the MFMAs run in a
//     fully unrolled, per-step guarded sequence;
//   * the product is computed TRANSPOSED (A = packed weights, B = gathered rows: out^T = W^T in^T), so a lane ends
//     up with four consecutive output channels of ONE row: one 16-byte store (and one 16-byte residual load) per
//     lane instead of four strided dwords;
//   * LDSW: the packed weights (27 KiB at 16->16) are staged once per workgroup and read with conflict-free
//     ds_read_b128; a workgroup then walks `gpw` groups per wave so the staging is amortised.
// Same arithmetic as k_conv_os / k_conv_pair: fp32 MFMA, per-offset ascending order, channel order inside an offset
// given by the packed layout.
// ------------------------------------------------------------------------------------
#ifndef CONV_G16_PHA2
#define CONV_G16_PHA2 2  // blocks of four steps gathered up front when Cin = 32 (two 16-channel chunks per step)
#endif
template <int NCH, bool AFF, bool RES, bool LDSW>
__global__ __launch_bounds__(256) void k_conv_g16(const float* __restrict__ in, const float4* __restrict__ Wp,
                                                  const int4* __restrict__ steps, const uint32_t* __restrict__ gmask,
                                                  int K, int M_out, unsigned in_bytes, int gpw,
                                                  const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                  const float* __restrict__ residual, const float* __restrict__ out_scale,
                                                  const float* __restrict__ out_shift, float* __restrict__ out) {
    constexpr int PHA = NCH == 1 ? GF_STEP_PHA : CONV_G16_PHA2;
    constexpr unsigned ROWB = NCH * 64u;  // bytes per input row
    constexpr int SHIFT = NCH == 1 ? 6 : 7;
    extern __shared__ __attribute__((aligned(16))) float4 s_w[];
    __shared__ __attribute__((aligned(16))) float s_aff[2][NCH * 16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int ngroups = (M_out + 15) >> 4;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, K * NCH * 1024, 0x00020000);
    if (LDSW) {
        const int total = K * NCH * 64;
        for (int t = threadIdx.x; t < total; t += 256) s_w[t] = Wp[t];
    }
    if (AFF) {
        if (threadIdx.x < NCH * 16) {
            s_aff[0][threadIdx.x] = in_scale[threadIdx.x];
            s_aff[1][threadIdx.x] = in_shift[threadIdx.x];
        }
    }
    if (LDSW || AFF) __syncthreads();
    const unsigned lane_ch = 16u * (unsigned)q;  // byte offset of this lane's 4 channels inside a 16-channel chunk
    float4 sc[NCH], sh[NCH];
    if (AFF) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            sc[c] = *reinterpret_cast<const float4*>(&s_aff[0][c * 16 + 4 * q]);
            sh[c] = *reinterpret_cast<const float4*>(&s_aff[1][c * 16 + 4 * q]);
        }
    }

#ifdef CONV_G16_STRIP
#if CONV_G16_STRIP == 0
    return;
#endif
#endif
    for (int i = 0; i < gpw; i++) {
        const int g = (blockIdx.x * gpw + i) * 4 + w;
        if (g >= ngroups) break;
        const int row = g * 16 + r;
#ifdef CONV_G16_STRIP
#if CONV_G16_STRIP == 1  // stores only
        if (row < M_out) *reinterpret_cast<float4*>(out + (size_t)row * 16 + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
#endif
#endif
        const int4* rec = steps + (size_t)g * (GF_STEP_BLKS * 16) + r;
        int4 ib[PHA];
#pragma unroll
        for (int b = 0; b < PHA; b++) ib[b] = rec[b * 16];
        uint32_t m = __builtin_amdgcn_readfirstlane(gmask[g]);
        const int n = __popc(m);
        float4 resv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RES && row < M_out) resv = *reinterpret_cast<const float4*>(residual + (size_t)row * 16 + 4 * q);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#ifdef CONV_G16_STRIP
#if CONV_G16_STRIP == 2  // index loads + mask + store
        if (row < M_out) {
            float4 v = make_float4((float)(ib[0].x + ib[1].y + ib[PHA - 1].z), (float)n, resv.x, 0.f);
            *reinterpret_cast<float4*>(out + (size_t)row * 16 + 4 * q) = v;
        }
        continue;
#endif
#endif

        // ---- phase A: the first PHA blocks, every gather in flight at once ----
        u32x4 a[PHA * 4][NCH];
#pragma unroll
        for (int s = 0; s < PHA * 4; s++) {
            const int idx = reinterpret_cast<const int*>(&ib[s >> 2])[s & 3];
            const unsigned voff = ((unsigned)idx << SHIFT) + lane_ch;  // idx == -1: beyond the descriptor, reads zeros
#if defined(CONV_G16_STRIP) && CONV_G16_STRIP == 4  // no gathers: MFMA + weights on index-derived values
#pragma unroll
            for (int c = 0; c < NCH; c++) a[s][c] = (u32x4){voff, voff + 1u, voff + 2u, voff + 3u};
#else
#pragma unroll
            for (int c = 0; c < NCH; c++) a[s][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, c * 64, 0);
#endif
        }
#ifdef CONV_G16_STRIP
#if CONV_G16_STRIP == 3  // + gathers, summed on the VALU, no MFMA / weights
        {
            float4 v = resv;
#pragma unroll
            for (int s = 0; s < PHA * 4; s++)
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    v.x += __uint_as_float(a[s][c][0]); v.y += __uint_as_float(a[s][c][1]);
                    v.z += __uint_as_float(a[s][c][2]); v.w += __uint_as_float(a[s][c][3]);
                }
            if (row < M_out) *reinterpret_cast<float4*>(out + (size_t)row * 16 + 4 * q) = v;
            continue;
        }
#endif
#endif
#pragma unroll
        for (int s = 0; s < PHA * 4; s++) {
            if (s < n) {
                const int k = __builtin_ctz(m);
                m &= m - 1;
                const bool present = reinterpret_cast<const int*>(&ib[s >> 2])[s & 3] >= 0;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    float4 wv;
                    if (LDSW) {
                        wv = s_w[(k * NCH + c) * 64 + lane];
                    } else {
                        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)lane * 16u,
                                                                              (unsigned)(k * NCH + c) * 1024u, 0);
                        wv = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]),
                                         __uint_as_float(t[3]));
                    }
                    float4 x = make_float4(__uint_as_float(a[s][c][0]), __uint_as_float(a[s][c][1]),
                                           __uint_as_float(a[s][c][2]), __uint_as_float(a[s][c][3]));
                    if (AFF) {
                        x.x = present ? fmaxf(fmaf(x.x, sc[c].x, sh[c].x), 0.f) : 0.f;
                        x.y = present ? fmaxf(fmaf(x.y, sc[c].y, sh[c].y), 0.f) : 0.f;
                        x.z = present ? fmaxf(fmaf(x.z, sc[c].z, sh[c].z), 0.f) : 0.f;
                        x.w = present ? fmaxf(fmaf(x.w, sc[c].w, sh[c].w), 0.f) : 0.f;
                    }
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, x.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, x.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, x.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, x.w, acc1, 0, 0, 0);
                }
            }
        }
        // ---- phase B: groups with more than 4*PHA present offsets, one block at a time ----
        for (int b = PHA; b * 4 < n; b++) {
            const int4 jb = rec[b * 16];
            u32x4 e[4][NCH];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned voff = ((unsigned)reinterpret_cast<const int*>(&jb)[j] << SHIFT) + lane_ch;
#pragma unroll
                for (int c = 0; c < NCH; c++) e[j][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, c * 64, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (b * 4 + j < n) {
                    const int k = __builtin_ctz(m);
                    m &= m - 1;
                    const bool present = reinterpret_cast<const int*>(&jb)[j] >= 0;
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        float4 wv;
                        if (LDSW) {
                            wv = s_w[(k * NCH + c) * 64 + lane];
                        } else {
                            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)lane * 16u,
                                                                                  (unsigned)(k * NCH + c) * 1024u, 0);
                            wv = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]),
                                             __uint_as_float(t[3]));
                        }
                        float4 x = make_float4(__uint_as_float(e[j][c][0]), __uint_as_float(e[j][c][1]),
                                               __uint_as_float(e[j][c][2]), __uint_as_float(e[j][c][3]));
                        if (AFF) {
                            x.x = present ? fmaxf(fmaf(x.x, sc[c].x, sh[c].x), 0.f) : 0.f;
                            x.y = present ? fmaxf(fmaf(x.y, sc[c].y, sh[c].y), 0.f) : 0.f;
                            x.z = present ? fmaxf(fmaf(x.z, sc[c].z, sh[c].z), 0.f) : 0.f;
                            x.w = present ? fmaxf(fmaf(x.w, sc[c].w, sh[c].w), 0.f) : 0.f;
                        }
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, x.x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, x.y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, x.z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, x.w, acc1, 0, 0, 0);
                    }
                }
            }
        }
        // transposed C/D layout: lane (r, q) holds channels 4q..4q+3 of row g*16 + r
        if (row < M_out) {
            float4 v = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
            if (RES) {
                v.x += resv.x; v.y += resv.y; v.z += resv.z; v.w += resv.w;
            }
            if (out_scale) {  // epilogue activation: the consumer's BatchNorm + ReLU, once per output element
                const float4 os = *reinterpret_cast<const float4*>(out_scale + 4 * q);
                const float4 ot = *reinterpret_cast<const float4*>(out_shift + 4 * q);
                v.x = fmaxf(fmaf(v.x, os.x, ot.x), 0.f); v.y = fmaxf(fmaf(v.y, os.y, ot.y), 0.f);
                v.z = fmaxf(fmaf(v.z, os.z, ot.z), 0.f); v.w = fmaxf(fmaf(v.w, os.w, ot.w), 0.f);
            }
            *reinterpret_cast<float4*>(out + (size_t)row * 16 + 4 * q) = v;
        }
    }
}

// ------------------------------------------------------------------------------------
// Pipelined form of the kernel above (what bench.py's roofline line is quoted on).
//
// Measured on the S150k level-1 launch with parts of k_conv_g16 compiled out (rocprofv3 kernel trace, 2221
// workgroups): empty kernel 2.5 us, + the output stores 4.4, + index loads 5.8, + gathers 10.0, and the MFMA section
// alone adds 8.4 us (365 MFMAs per SIMD at the ~1.9 GHz the chip holds under load = 6.1 us at a fully paced matrix
// pipe) -- and the full kernel takes the SUM, 20-21 us: every resident wave of a SIMD starts together, so they all
// wait for memory together and then queue for the matrix pipe together; nothing overlaps.
// Here a wave owns a CHUNK of consecutive groups of equal total cost (k_group_chunks: 3072 chunks = 12 waves per
// compute unit, all resident at once, no tail) and software-pipelines over them: while the MFMAs of group i run,
// the gathers of group i+1 are in flight and the indices of group i+2 are being fetched.  Every load of the
// pipeline is issued unconditionally (a missing group or neighbour is index -1 = out of the descriptor's range, no
// memory traffic), so the waits are exact counted vmcnt's instead of vmcnt(0).
// ------------------------------------------------------------------------------------
#ifdef CONV_TRACE
// dev build: per-wave cycle stamps (s_memtime) of the pipelined kernel, written to a caller-set buffer
__device__ unsigned long long* g_conv_trace = nullptr;
extern "C" int gf_dev_conv_trace(void* p) {
    GF_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_trace), &p, sizeof(p)));
    return GF_OK;
}
#define TR_NOW() __builtin_amdgcn_s_memtime()
#endif
#ifndef CONV_G16P_WAVES
#define CONV_G16P_WAVES 2  // resident waves per SIMD the register budget is set for
#endif
template <int NCH, bool AFF, bool RES, bool LDSW, int WPB>
__global__ __launch_bounds__(64 * WPB, CONV_G16P_WAVES) void k_conv_g16p(const float* __restrict__ in, const float4* __restrict__ Wp,
                                                      const int32_t* __restrict__ steps_raw,
                                                      const uint32_t* __restrict__ gmask,
                                                      const int32_t* __restrict__ chunks, int K, int M_out,
                                                      unsigned in_bytes, unsigned steps_bytes,
                                                      const float* __restrict__ in_scale,
                                                      const float* __restrict__ in_shift,
                                                      const float* __restrict__ residual,
                                                      const float* __restrict__ out_scale,
                                                      const float* __restrict__ out_shift, float* __restrict__ out,
                                                      float* __restrict__ out2) {
    constexpr int PHA = NCH == 1 ? GF_STEP_PHA : CONV_G16_PHA2;
    constexpr int NS = PHA * 4;  // steps of the pipelined part
    constexpr int SHIFT = NCH == 1 ? 6 : 7;
    extern __shared__ __attribute__((aligned(16))) float4 s_w[];
    __shared__ __attribute__((aligned(16))) float s_aff[2][NCH * 16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, K * NCH * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_st =
        __builtin_amdgcn_make_buffer_rsrc((void*)steps_raw, 0, (int)steps_bytes, 0x00020000);
    const int cw = blockIdx.x * WPB + w;
#ifdef CONV_TRACE
    unsigned long long tr[8] = {TR_NOW(), 0, 0, 0, 0, 0, 0, 0};  // start, after barrier, first data, [wait a0, mfma, tail] sums, end, groups
#endif
    // the table's count and this wave's two boundaries as three INDEPENDENT loads (count -> boundaries was one more
    // dependent round trip in front of the first index load; the table's tail is sized for GF_CONV_CHUNKS_MAX entries)
    const int cwc = min(cw, GF_CONV_CHUNKS_MAX - 1);
    const int nchunks = chunks[-1];
    const int c0 = chunks[cwc], c1 = chunks[cwc + 1];
    if (blockIdx.x * WPB >= nchunks) return;  // whole workgroup past the table (uniform: no barrier is skipped by part of it)
    const bool live_chunk = cw < nchunks;     // (a wave past the count inside the last workgroup: empty range)
    int g = __builtin_amdgcn_readfirstlane(live_chunk ? c0 : 0);
    const int gend = __builtin_amdgcn_readfirstlane(live_chunk ? c1 : 0);
    // the offset masks of the chunk's groups, one per lane (a chunk holds ~3 groups): one load in the prologue instead
    // of a dependent load in front of every group's first weight fetch
    const int g_first = g;
    uint32_t mask_v = 0;
    if (g_first + lane < gend) mask_v = gmask[g_first + lane];
    const unsigned lane_ch = 16u * (unsigned)q;
    const unsigned rec_lane = (unsigned)r * 16u;  // this lane's 16-byte entry inside a 256-byte step block
    constexpr unsigned GROUP_BYTES = GF_STEP_BLKS * 256u;

    // index blocks of one group: a group past the chunk's end reads beyond the descriptor (zeros) and is turned into
    // "all missing" (-1) afterwards, so that every stage issues the same number of loads
    auto load_idx = [&](u32x4 (&ib)[PHA], int gg) {
        const unsigned base = gg < gend ? (unsigned)gg * GROUP_BYTES : 0xfffff000u;
#pragma unroll
        for (int b = 0; b < PHA; b++) ib[b] = __builtin_amdgcn_raw_buffer_load_b128(rs_st, rec_lane, base + b * 256u, 0);
    };
    auto gather = [&](u32x4 (&a)[NS][NCH], const u32x4 (&ib)[PHA], bool live, unsigned& pres) {
        pres = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int idx = live ? (int)ib[s >> 2][s & 3] : -1;
            if (AFF) pres |= (idx >= 0 ? 1u : 0u) << s;
            const unsigned voff = ((unsigned)idx << SHIFT) + lane_ch;
#pragma unroll
            for (int c = 0; c < NCH; c++) a[s][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, c * 64, 0);
        }
    };

    u32x4 ibX[PHA], ibY[PHA];
    u32x4 aX[NS][NCH], aY[NS][NCH];
    unsigned presX = 0, presY = 0;
    // prologue part 1 (before the weights are staged: these loads overlap the staging)
    // prologue: index blocks of the first two groups, the weights (all staging loads in flight together: a
    // load / wait / write loop costs one memory round trip per 4 KiB, 2.6 us measured), then the first group's
    // gathers -- all of it before the barrier the LDS weights need
    load_idx(ibX, g);
    load_idx(ibY, g + 1);
    if (LDSW) {
        const int total = K * NCH * 64;
        constexpr int T = 64 * WPB;
        constexpr int WST = WPB >= 8 ? 4 : 8;  // 27 x 64 x NCH float4 over T threads: 7 (NCH 1) / 14 (NCH 2) per thread at T = 256
        for (int base = 0; base < total; base += T * WST) {
            u32x4 tmp[WST];
#pragma unroll
            for (int j = 0; j < WST; j++)
                tmp[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(base + j * T + (int)threadIdx.x) * 16u, 0, 0);
#pragma unroll
            for (int j = 0; j < WST; j++) {
                const int t = base + j * T + (int)threadIdx.x;
                if (t < total)
                    s_w[t] = make_float4(__uint_as_float(tmp[j][0]), __uint_as_float(tmp[j][1]), __uint_as_float(tmp[j][2]),
                                         __uint_as_float(tmp[j][3]));
            }
        }
    }
    float4 sc[NCH], sh[NCH];
    if (AFF) {  // every lane reads its own channels' scale / shift directly (L2-resident, 16 bytes)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            sc[c] = *reinterpret_cast<const float4*>(in_scale + c * 16 + 4 * q);
            sh[c] = *reinterpret_cast<const float4*>(in_shift + c * 16 + 4 * q);
        }
    }
    // the epilogue activation's scale / shift of this lane's four channels, once per wave (not once per group)
    float4 os = make_float4(1.f, 1.f, 1.f, 1.f), ot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (out_scale) {
        os = *reinterpret_cast<const float4*>(out_scale + 4 * q);
        ot = *reinterpret_cast<const float4*>(out_shift + 4 * q);
    }
    if (g < gend) gather(aX, ibX, true, presX);
    if (LDSW) __syncthreads();
#ifdef CONV_TRACE
    tr[1] = TR_NOW();
#endif
    if (g >= gend) return;

    // one group's MFMAs + store; `a` holds its gathered rows (in flight), `pres` its presence bits
    auto compute = [&](u32x4 (&a)[NS][NCH], unsigned pres, int gg) {
        uint32_t m = (gg - g_first) < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)mask_v, gg - g_first)
                                         : __builtin_amdgcn_readfirstlane(gmask[gg]);
        const int n = __popc(m);
        const int row = gg * 16 + r;
        float4 resv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RES && row < M_out) resv = *reinterpret_cast<const float4*>(residual + (size_t)row * 16 + 4 * q);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // offsets of the pipelined steps up front (scalar), weights of step s+1 requested before the MFMAs of step s;
        // steps past the group's count multiply all-missing rows (zeros) by the weights of offset 0: skipped from
        // step 8 on (a group of a scanned room has at least 9 present offsets in 99.9 % of the cases)
        int ks[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            ks[s] = m ? __builtin_ctz(m) : 0;
            m &= m - 1;
        }
        float4 wq[NCH], wn[NCH];
        auto fetch_w = [&](float4 (&dst)[NCH], int k) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (LDSW) {
                    dst[c] = s_w[(k * NCH + c) * 64 + lane];
                } else {
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)lane * 16u,
                                                                          (unsigned)(k * NCH + c) * 1024u, 0);
                    dst[c] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]),
                                         __uint_as_float(t[3]));
                }
            }
        };
        auto mfma_step = [&](const u32x4 (&av)[NCH], const float4 (&wv)[NCH], bool present) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                float4 x = make_float4(__uint_as_float(av[c][0]), __uint_as_float(av[c][1]), __uint_as_float(av[c][2]),
                                       __uint_as_float(av[c][3]));
                if (AFF) {
                    x.x = present ? fmaxf(fmaf(x.x, sc[c].x, sh[c].x), 0.f) : 0.f;
                    x.y = present ? fmaxf(fmaf(x.y, sc[c].y, sh[c].y), 0.f) : 0.f;
                    x.z = present ? fmaxf(fmaf(x.z, sc[c].z, sh[c].z), 0.f) : 0.f;
                    x.w = present ? fmaxf(fmaf(x.w, sc[c].w, sh[c].w), 0.f) : 0.f;
                }
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c].x, x.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c].y, x.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c].z, x.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c].w, x.w, acc1, 0, 0, 0);
            }
        };
        fetch_w(wq, ks[0]);
#ifdef CONV_TRACE
        const unsigned long long tc0 = TR_NOW();
        {
            // wait for the first gathered row of this group (and only for it)
            float probe = __uint_as_float(a[0][0][0]);
            asm volatile("v_mov_b32 %0, %0" : "+v"(probe));
            a[0][0][0] = __float_as_uint(probe);
        }
        const unsigned long long tc1 = TR_NOW();
        tr[3] += tc1 - tc0;
        if (tr[2] == 0) tr[2] = tc1;
#endif
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (s + 1 < NS) fetch_w(wn, ks[s + 1]);
            if (s < 8 || s < n) mfma_step(a[s], wq, AFF ? ((pres >> s) & 1u) != 0 : true);
#pragma unroll
            for (int c = 0; c < NCH; c++) wq[c] = wn[c];
        }
        // groups with more than NS present offsets (16 % of the level-1 groups of a scanned room): the remaining index
        // blocks in one round trip, their gathers in another (not pipelined, but two exposed latencies per group
        // instead of two per block)
        if (n > NS) {
            constexpr int RB = GF_STEP_BLKS - PHA;  // remaining blocks
            u32x4 jb[RB];
#pragma unroll
            for (int b = 0; b < RB; b++)
                jb[b] = __builtin_amdgcn_raw_buffer_load_b128(rs_st, rec_lane, (unsigned)gg * GROUP_BYTES + (PHA + b) * 256u, 0);
            for (int h = 0; h < RB; h += 2) {  // two blocks (8 steps) of gathers in flight
                if ((PHA + h) * 4 >= n) break;
                u32x4 e[8][NCH];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int st = (PHA + h) * 4 + j;
                    const int bi = h + (j >> 2);
                    // blocks past the group's count were never written: treat as missing
                    const int idx = (bi < RB && st < n) ? (int)jb[bi < RB ? bi : 0][j & 3] : -1;
                    const unsigned voff = ((unsigned)idx << SHIFT) + lane_ch;
#pragma unroll
                    for (int c = 0; c < NCH; c++) e[j][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, c * 64, 0);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int st = (PHA + h) * 4 + j;
                    const int bi = h + (j >> 2);
                    if (st < n) {
                        const int k = __builtin_ctz(m);
                        m &= m - 1;
                        float4 wv[NCH];
                        fetch_w(wv, k);
                        mfma_step(e[j], wv, bi < RB ? (int)jb[bi < RB ? bi : 0][j & 3] >= 0 : false);
                    }
                }
            }
        }
#ifdef CONV_TRACE
        const unsigned long long tc2 = TR_NOW();
        tr[4] += tc2 - tc1;
        tr[7] += 1;
#endif
        if (row < M_out) {
            float4 v = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
            if (RES) {
                v.x += resv.x; v.y += resv.y; v.z += resv.z; v.w += resv.w;
            }
            if (out_scale) {  // epilogue activation: the consumer's BatchNorm + ReLU, once per output element
                if (out2) {
                    // both forms leave the kernel: the raw sum (residual operand of the next block) and the
                    // activated copy its first convolution gathers from -- one BatchNorm + ReLU per ELEMENT here
                    // instead of one per GATHERED element (6-10 x more) in that convolution's prologue
                    *reinterpret_cast<float4*>(out + (size_t)row * 16 + 4 * q) = v;
                }
                v.x = fmaxf(fmaf(v.x, os.x, ot.x), 0.f); v.y = fmaxf(fmaf(v.y, os.y, ot.y), 0.f);
                v.z = fmaxf(fmaf(v.z, os.z, ot.z), 0.f); v.w = fmaxf(fmaf(v.w, os.w, ot.w), 0.f);
            }
            *reinterpret_cast<float4*>((out2 ? out2 : out) + (size_t)row * 16 + 4 * q) = v;
        }
#ifdef CONV_TRACE
        tr[5] += TR_NOW() - tc2;
#endif
    };

    // steady state, unrolled by two so that the buffers keep static names:
    //   X.a = rows of group g (in flight), Y.ib = indices of group g+1 (in flight)
    while (true) {
        gather(aY, ibY, g + 1 < gend, presY);  // group g+1 (all -1 past the end)
        load_idx(ibX, g + 2);
        compute(aX, presX, g);
        if (++g >= gend) break;
        gather(aX, ibX, g + 1 < gend, presX);
        load_idx(ibY, g + 2);
        compute(aY, presY, g);
        if (++g >= gend) break;
    }
#ifdef CONV_TRACE
    tr[6] = TR_NOW();
    if (g_conv_trace && lane == 0)
        for (int i = 0; i < 8; i++) g_conv_trace[(size_t)cw * 8 + i] = tr[i];
#endif
}

template <int NCH, bool LDSW>
static void launch_g16(dim3 grid, size_t lds, hipStream_t st, const float* in, const float4* Wp, const int4* steps,
                       const uint32_t* gmask, int K, int M_out, unsigned in_bytes, int gpw, const float* sc,
                       const float* sh, const float* res, const float* osc, const float* osh, float* out) {
    if (LDSW) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)k_conv_g16<NCH, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_conv_g16<NCH, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_conv_g16<NCH, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_conv_g16<NCH, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            attr = true;
        }
    }
#define G16_LAUNCH(AFF_, RES_)                                                                                        \
    hipLaunchKernelGGL((k_conv_g16<NCH, AFF_, RES_, LDSW>), grid, dim3(256), lds, st, in, Wp, steps, gmask, K, M_out, \
                       in_bytes, gpw, sc, sh, res, osc, osh, out)
    if (sc && res) G16_LAUNCH(true, true);
    else if (sc) G16_LAUNCH(true, false);
    else if (res) G16_LAUNCH(false, true);
    else G16_LAUNCH(false, false);
#undef G16_LAUNCH
}

// dev hook (bench.py's roofline probe): two events that the NEXT launch of the pipelined kernel on this host thread
// binds to the kernel itself (hipExtLaunchKernelGGL: start and stop are the dispatch's own begin / end timestamps, what
// a profiler's kernel trace reports) instead of events recorded before and after it on the stream, which add the
// command processor's handling of two event packets (~1.3 us on a 20 us kernel)
thread_local hipEvent_t t_kev_start = nullptr, t_kev_stop = nullptr;
thread_local bool t_kev_taken = false;
extern "C" int gf_dev_conv_kernel_events(void* start, void* stop) {
    t_kev_start = (hipEvent_t)start;
    t_kev_stop = (hipEvent_t)stop;
    t_kev_taken = false;
    return GF_OK;
}
extern "C" int gf_dev_conv_kernel_events_taken(void) { return t_kev_taken ? 1 : 0; }

// waves per workgroup of the pipelined kernel.  12 = ONE workgroup per compute unit with the default 3072 chunks (256
// workgroups, the kernel's 133-148 VGPRs allow three waves per SIMD): the 27 KiB of packed weights are staged once per
// compute unit instead of three times.  Measured on the S150k level-1 launch (tools/conv_wpb_exp.py, back-to-back
// launches): residual epilogue 18.95 -> 17.32 us, activation epilogue 18.62 -> 18.23; 8 or 16 waves leave compute units
// with a second round (23-25 us).  0 = 12 when the table's chunk count is a multiple of 12, else 4.
static int g_g16p_wpb = 0;
extern "C" int gf_dev_conv_g16p_wpb(int wpb) {
    GF_CHECK_ARG(wpb == 0 || wpb == 4 || wpb == 8 || wpb == 12 || wpb == 16, "gf_dev_conv_g16p_wpb: %d (4, 8, 12 or 16)", wpb);
    g_g16p_wpb = wpb;
    return GF_OK;
}

template <int NCH, bool LDSW, int WPB>
static void launch_g16p_w(size_t lds, hipStream_t st, const float* in, const float4* Wp, const int32_t* steps,
                          const uint32_t* gmask, int K, int M_out, int ld, unsigned in_bytes, const float* sc,
                          const float* sh, const float* res, const float* osc, const float* osh, float* out,
                          float* out2) {
    if (LDSW) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)k_conv_g16p<NCH, false, false, true, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_conv_g16p<NCH, false, true, true, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_conv_g16p<NCH, true, false, true, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_conv_g16p<NCH, true, true, true, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            attr = true;
        }
    }
    const size_t step_words = (size_t)(ld / 16) * GF_STEP_BLKS * 64;
    const int32_t* chunks = steps + step_words + 1;  // [-1] = number of chunks the table was built with
    const unsigned steps_bytes = (unsigned)(step_words * 4);
    dim3 grid((GF_CONV_CHUNKS_MAX + WPB - 1) / WPB);  // waves of chunks past the table's count leave at once
#define G16P_LAUNCH(AFF_, RES_)                                                                                       \
    do {                                                                                                              \
        if (t_kev_start) {                                                                                            \
            hipExtLaunchKernelGGL((k_conv_g16p<NCH, AFF_, RES_, LDSW, WPB>), grid, dim3(64 * WPB), (std::uint32_t)lds, st, \
                                  t_kev_start, t_kev_stop, 0u, in, Wp, steps, gmask, chunks, K, M_out, in_bytes,      \
                                  steps_bytes, sc, sh, res, osc, osh, out, out2);                                     \
            t_kev_start = t_kev_stop = nullptr;                                                                       \
            t_kev_taken = true;                                                                                       \
        } else {                                                                                                      \
            hipLaunchKernelGGL((k_conv_g16p<NCH, AFF_, RES_, LDSW, WPB>), grid, dim3(64 * WPB), lds, st, in, Wp, steps, gmask,  \
                               chunks, K, M_out, in_bytes, steps_bytes, sc, sh, res, osc, osh, out, out2);            \
        }                                                                                                             \
    } while (0)
    if (sc && res) G16P_LAUNCH(true, true);
    else if (sc) G16P_LAUNCH(true, false);
    else if (res) G16P_LAUNCH(false, true);
    else G16P_LAUNCH(false, false);
#undef G16P_LAUNCH
}

template <int NCH, bool LDSW>
static void launch_g16p(size_t lds, hipStream_t st, const float* in, const float4* Wp, const int32_t* steps,
                        const uint32_t* gmask, int K, int M_out, int ld, unsigned in_bytes, const float* sc,
                        const float* sh, const float* res, const float* osc, const float* osh, float* out,
                        float* out2) {
#define G16P_W(W_) launch_g16p_w<NCH, LDSW, W_>(lds, st, in, Wp, steps, gmask, K, M_out, ld, in_bytes, sc, sh, res, osc, osh, out, out2)
    // (S150k level 1, back-to-back launches: residual epilogue 18.4 -> 17.0 us, activation epilogue 17.9 -> 17.4 us)
    // (the 32 -> 16 launch, NCH = 2, keeps four waves: 55.7 us against 59.9 us with twelve)
    const int wpb = g_g16p_wpb ? g_g16p_wpb : (NCH == 1 && gf_conv_chunks() % 12 == 0 ? 12 : 4);
    switch (wpb) {
        case 8: G16P_W(8); break;
        case 12: G16P_W(12); break;
        case 16: G16P_W(16); break;
        default: G16P_W(4); break;
    }
#undef G16P_W
}

struct ConvArgs {
    const float* in;
    const float4* Wp;
    const int32_t* nbr;
    const uint32_t* gmask;
    int K, M_out, ld, Cin, Cout, NCH, NCB, nsplit;
    unsigned in_bytes;
    const float *sc, *sh, *res, *osc, *osh;
    float* out;
};

static int g_conv_block = 256;
static int g_conv_chunks = GF_CONV_CHUNKS;
int gf_conv_chunks() { return g_conv_chunks; }
extern "C" int gf_dev_conv_chunks(int n) {
    if (n <= 0) n = GF_CONV_CHUNKS;
    GF_CHECK_ARG(n % 4 == 0 && n <= GF_CONV_CHUNKS_MAX, "gf_dev_conv_chunks: %d (multiple of 4, at most %d)", n, GF_CONV_CHUNKS_MAX);
    g_conv_chunks = n;
    return GF_OK;
}
template <int NCBW, int SW>
static void launch_conv(bool vec, dim3 grid, hipStream_t st, const ConvArgs& a) {
    const int bs = SW ? SW * 64 : g_conv_block;
    if (vec)
        hipLaunchKernelGGL((k_conv_os<NCBW, SW, true>), grid, dim3(bs), 0, st, a.in, a.Wp, a.nbr, a.gmask, a.K,
                           a.M_out, a.ld, a.Cin, a.Cout, a.NCH, a.NCB, a.nsplit, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh, a.out);
    else
        hipLaunchKernelGGL((k_conv_os<NCBW, SW, false>), grid, dim3(bs), 0, st, a.in, a.Wp, a.nbr, a.gmask, a.K,
                           a.M_out, a.ld, a.Cin, a.Cout, a.NCH, a.NCB, a.nsplit, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh, a.out);
}

template <int SW>
static void dispatch_conv(int ncbw, bool vec, dim3 grid, hipStream_t st, const ConvArgs& a) {
    switch (ncbw) {
        case 1: launch_conv<1, SW>(vec, grid, st, a); break;
        case 2: launch_conv<2, SW>(vec, grid, st, a); break;
        case 3: launch_conv<3, SW>(vec, grid, st, a); break;
        case 4: launch_conv<4, SW>(vec, grid, st, a); break;
        case 5: launch_conv<5, SW>(vec, grid, st, a); break;
        case 6: launch_conv<6, SW>(vec, grid, st, a); break;
        case 7: launch_conv<7, SW>(vec, grid, st, a); break;
        default: launch_conv<8, SW>(vec, grid, st, a); break;
    }
}

// Dev knobs (include/geoformer_hip_dev.h: gf_dev_conv_knobs*): force a launch shape regardless of the level's size.
struct ConvKnobs {
    int split = -1, wide = -1, block = 0, pair = -1;  // -1 / 0: not set
    int g16 = -1, g16_ldsw = -1, g16_gpw = 0;         // counted-loop kernel: use / weights in LDS / groups per wave
    int g16_pipe = -1;                                // its pipelined form (one chunk of groups per wave)
    int flat = -1, flat_items = 0;                    // flat-chain kernel: use / item bound of the size-based choice
};
static ConvKnobs& conv_knobs_mut() {
    static ConvKnobs k;
    return k;
}
static const ConvKnobs& conv_knobs() { return conv_knobs_mut(); }

// Dev hook (include/geoformer_hip_dev.h): force a launch shape regardless of the level's size; -1 = size-based.
extern "C" int gf_dev_conv_knobs_g16(int use, int ldsw, int gpw, int pipe) {
    ConvKnobs& k = conv_knobs_mut();
    k.g16_pipe = pipe < 0 ? -1 : (pipe != 0);
    k.g16 = use < 0 ? -1 : (use != 0);
    k.g16_ldsw = ldsw < 0 ? -1 : (ldsw != 0);
    k.g16_gpw = gpw > 0 ? gpw : 0;
    return GF_OK;
}
extern "C" int gf_dev_conv_knob_flat(int use, int max_items) {
    ConvKnobs& k = conv_knobs_mut();
    k.flat = use < 0 ? -1 : (use != 0);
    k.flat_items = max_items > 0 ? max_items : 0;
    return GF_OK;
}
extern "C" int gf_dev_conv_knobs(int split, int wide, int pair, int ldsw, int block) {
    ConvKnobs& k = conv_knobs_mut();
    k.split = split < 0 ? -1 : (split != 0);
    k.wide = wide < 0 ? -1 : (wide != 0);
    k.pair = pair < 0 ? -1 : (pair != 0);
    (void)ldsw;  // (the LDS-weight form of k_conv_os is gone: round 6)
    k.block = block <= 0 ? 0 : (block > 256 ? 256 : block);
    g_conv_block = k.block > 0 ? k.block : 256;
    return GF_OK;
}

static int conv_fwd_impl(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask,
                         const int32_t* steps, const int32_t* fsteps, int K, int M_in, int M_out, int ld, int Cin, int Cout,
                         const float* in_scale, const float* in_shift, const float* residual,
                         const float* out_scale, const float* out_shift, float* out, float* out2, void* stream) {
    GF_CHECK_ARG(K >= 1 && K <= 32, "gf_conv_fwd: K=%d out of range [1,32]", K);
    GF_CHECK_ARG(Cin >= 1 && Cout >= 1, "gf_conv_fwd: Cin=%d Cout=%d", Cin, Cout);
    GF_CHECK_ARG(in_scale == nullptr || Cin <= CONV_MAX_CIN - 16, "gf_conv_fwd: fused prologue supports Cin <= %d",
                 CONV_MAX_CIN - 16);
    GF_CHECK_ARG(nbr != nullptr || steps != nullptr || K == 1, "gf_conv_fwd: no table requires K==1");
    GF_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "gf_conv_fwd: in_scale/in_shift must come together");
    GF_CHECK_ARG((out_scale == nullptr) == (out_shift == nullptr), "gf_conv_fwd: out_scale/out_shift must come together");
    GF_CHECK_ARG(out_scale == nullptr || ((((uintptr_t)out_scale) | ((uintptr_t)out_shift)) % 16) == 0,
                 "gf_conv_fwd: out_scale/out_shift must be 16-byte aligned");
    if (M_out <= 0) return GF_OK;
    const int ngroups = (M_out + 15) / 16;
    const int ncb = (Cout + 15) / 16, nch = (Cin + 15) / 16;
    const unsigned long long in_bytes64 = (unsigned long long)M_in * Cin * 4ull;
    const bool vec = (Cin % 16) == 0 && (((uintptr_t)in) % 16) == 0 && in_bytes64 < 0xfffffff0ull &&
                     (in_scale == nullptr || ((((uintptr_t)in_scale) | ((uintptr_t)in_shift)) % 16) == 0);
    // Big levels: one wave per 16-row group owning every column block.  Small levels (not enough groups
    // to fill 1024 SIMDs with several waves each): a workgroup per (group, <=2 column blocks), steps split
    // over its four waves.
    bool split = ngroups < 6000;
    const ConvKnobs& knobs = conv_knobs();
    if (knobs.split >= 0) split = knobs.split != 0;
    // tiny levels (all items resident at once with room to spare): 16 waves per item, 4x shorter chains again
    // tiny levels (<= 256 workgroups of 16 waves, all resident at once): 16 waves share an item, each wave's chain
    // of gather batches is 4x shorter again (S150k levels 5-7: 13.5/15.8/16.0 -> 10.6/11.4/11.5 us)
    bool wide = split && (long long)ngroups * ncb <= 256;
    if (knobs.wide >= 0) wide = split && knobs.wide != 0;
    // the flat-chain kernel wherever the wide shape would go (every address known at launch: k_conv_flat)
    const bool flat_ok = vec && nch <= 16 && K * nch <= 16 * FLAT_PF * FLAT_MAXB && in_bytes64 < 0xfffffff0ull;
    bool flat = flat_ok && (knobs.flat < 0 ? (split && (long long)ngroups * ncb <= (knobs.flat_items > 0 ? knobs.flat_items : 256))
                                           : knobs.flat != 0);
    const int ncbw = split ? (!wide && ncb >= 2 && ngroups >= 2048 ? 2 : 1) : (ncb > 8 ? 8 : ncb);
    const int nsplit = (ncb + ncbw - 1) / ncbw;
    const long long nitems = (long long)ngroups * nsplit;
    if (knobs.block > 0) g_conv_block = knobs.block;
    const int wpb = g_conv_block / 64;
    long long blocks = split ? nitems : (nitems + wpb - 1) / wpb;
    if (blocks > 256 * 64) blocks = 256 * 64;
    ConvArgs a{in, reinterpret_cast<const float4*>(Wp), nbr, gmask, K, M_out, ld, Cin, Cout, nch, ncb, nsplit,
               (unsigned)(in_bytes64 < 0xfffffff0ull ? in_bytes64 : 0), in_scale, in_shift, residual, out_scale, out_shift, out};
    dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    const size_t wbytes = (size_t)K * nch * ncb * 1024;
    // counted-loop kernel over the step table: 16 output channels, one or two input chunks
    bool g16 = steps != nullptr && gmask != nullptr && vec && Cout == 16 && nch <= 2 && (((uintptr_t)out) % 16) == 0 &&
               (residual == nullptr || (((uintptr_t)residual) % 16) == 0) && in_bytes64 <= 0xffffff00ull;
    if (knobs.g16 >= 0) g16 = g16 && knobs.g16 != 0;
    else g16 = g16 && !split;
    // LDS-weight kernel over the flat step table (spconv_lw.hip), where the caller built one
    if (fsteps != nullptr && gmask != nullptr) {
        int forced = 0;
        const bool aligned = vec && (((uintptr_t)out) % 16) == 0 && (residual == nullptr || (((uintptr_t)residual) % 16) == 0) &&
                             (out2 == nullptr || (((uintptr_t)out2) % 16) == 0) &&
                             (out_scale == nullptr || ((((uintptr_t)out_scale) | ((uintptr_t)out_shift)) % 16) == 0);
        if (gf_conv_lw_supported(K, M_in, M_out, Cin, Cout, aligned, &forced) && (forced || !(g16 && nch == 1)))
            return gf_conv_lw(in, Wp, gmask, fsteps, K, M_in, M_out, Cin, Cout, in_scale, in_shift, residual, out_scale, out_shift,
                              out, out2, st);
    }
    if (flat && (nbr != nullptr || K == 1) && out2 == nullptr) {
        if (K * nch <= 16 * FLAT_PF * FLAT_MAXB_SMALL)
            hipLaunchKernelGGL(k_conv_flat<FLAT_MAXB_SMALL>, dim3((unsigned)((long long)ngroups * ncb)), dim3(1024), 0, st, a.in,
                               a.Wp, a.nbr, a.K, a.M_out, a.ld, a.Cin, a.Cout, a.NCH, a.NCB, a.in_bytes, a.sc, a.sh, a.res, a.osc,
                               a.osh, a.out);
        else
            hipLaunchKernelGGL(k_conv_flat<FLAT_MAXB>, dim3((unsigned)((long long)ngroups * ncb)), dim3(1024), 0, st, a.in, a.Wp,
                               a.nbr, a.K, a.M_out, a.ld, a.Cin, a.Cout, a.NCH, a.NCB, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh,
                               a.out);
        GF_CHECK_LAUNCH("gf_conv_fwd");
        return GF_OK;
    }
    if (g16) {
        bool gl = wbytes <= 64 * 1024;
        if (knobs.g16_ldsw >= 0) gl = gl && knobs.g16_ldsw != 0;
        const size_t step_bytes64 = (size_t)(ld / 16) * GF_STEP_BLKS * 256;
        bool pipe = step_bytes64 < 0xfffff000ull && M_out > 0 && ld >= M_out;
        if (knobs.g16_pipe >= 0) pipe = pipe && knobs.g16_pipe != 0;
        if (pipe) {
            if (nch == 1) {
                if (gl) launch_g16p<1, true>(wbytes, st, a.in, a.Wp, steps, gmask, K, M_out, ld, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh, out, out2);
                else launch_g16p<1, false>(0, st, a.in, a.Wp, steps, gmask, K, M_out, ld, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh, out, out2);
            } else {
                if (gl) launch_g16p<2, true>(wbytes, st, a.in, a.Wp, steps, gmask, K, M_out, ld, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh, out, out2);
                else launch_g16p<2, false>(0, st, a.in, a.Wp, steps, gmask, K, M_out, ld, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh, out, out2);
            }
            GF_CHECK_LAUNCH("gf_conv_fwd");
            return GF_OK;
        }
        GF_CHECK_ARG(out2 == nullptr, "gf_conv_fwd_dual: this launch shape has no second output");
        int gpw = knobs.g16_gpw > 0 ? knobs.g16_gpw : (gl ? 2 : 1);
        const long long wgs = ((long long)ngroups + 4 * gpw - 1) / (4 * gpw);
        dim3 gg((unsigned)wgs);
        const int4* st4 = reinterpret_cast<const int4*>(steps);
        if (nch == 1) {
            if (gl) launch_g16<1, true>(gg, wbytes, st, a.in, a.Wp, st4, gmask, K, M_out, a.in_bytes, gpw, a.sc, a.sh, a.res, a.osc, a.osh, out);
            else launch_g16<1, false>(gg, 0, st, a.in, a.Wp, st4, gmask, K, M_out, a.in_bytes, gpw, a.sc, a.sh, a.res, a.osc, a.osh, out);
        } else {
            if (gl) launch_g16<2, true>(gg, wbytes, st, a.in, a.Wp, st4, gmask, K, M_out, a.in_bytes, gpw, a.sc, a.sh, a.res, a.osc, a.osh, out);
            else launch_g16<2, false>(gg, 0, st, a.in, a.Wp, st4, gmask, K, M_out, a.in_bytes, gpw, a.sc, a.sh, a.res, a.osc, a.osh, out);
        }
        GF_CHECK_LAUNCH("gf_conv_fwd");
        return GF_OK;
    }
    GF_CHECK_ARG(out2 == nullptr, "gf_conv_fwd_dual: this launch shape has no second output");
    GF_CHECK_ARG(nbr != nullptr || K == 1, "gf_conv_fwd: this launch shape needs the [K,ld] neighbour table");
    bool pair = !split && vec && ncb == 1 && nbr != nullptr && K <= 32 && nch <= 8 && in_bytes64 <= 0xfffff000ull - 4096ull;
    if (knobs.pair >= 0) pair = pair && knobs.pair != 0;
    if (pair) {
        const long long npairs = (ngroups + 1) / 2;
        long long pb = (npairs + 3) / 4;
        if (pb > 256 * 64) pb = 256 * 64;
        if (a.sc)
            hipLaunchKernelGGL(k_conv_pair<true>, dim3((unsigned)pb), dim3(256), 0, st, a.in, a.Wp, a.nbr, a.gmask, a.K,
                               a.M_out, a.ld, a.Cin, a.Cout, a.NCH, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh, a.out);
        else
            hipLaunchKernelGGL(k_conv_pair<false>, dim3((unsigned)pb), dim3(256), 0, st, a.in, a.Wp, a.nbr, a.gmask, a.K,
                               a.M_out, a.ld, a.Cin, a.Cout, a.NCH, a.in_bytes, a.sc, a.sh, a.res, a.osc, a.osh, a.out);
    } else if (split && wide)
        launch_conv<1, 16>(vec, grid, st, a);
    else if (split)
        dispatch_conv<4>(ncbw, vec, grid, st, a);
    else
        dispatch_conv<0>(ncbw, vec, grid, st, a);
    GF_CHECK_LAUNCH("gf_conv_fwd");
    return GF_OK;
}

extern "C" int gf_conv_fwd(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask,
                           const int32_t* steps, int K, int M_in, int M_out, int ld, int Cin, int Cout,
                           const float* in_scale, const float* in_shift, const float* residual,
                           const float* out_scale, const float* out_shift, float* out, void* stream) {
    return conv_fwd_impl(in, Wp, nbr, gmask, steps, nullptr, K, M_in, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out_scale,
                         out_shift, out, nullptr, stream);
}

// gf_conv_fwd with the relation's flat step table (gf_rules_flat_steps; may be NULL) and an optional second output
// (out_act, as gf_conv_fwd_dual; NULL = one output).  With a flat table the shapes whose packed weights fit the LDS take
// the LDS-weight kernel (spconv_lw.hip); everything else is gf_conv_fwd / gf_conv_fwd_dual.
extern "C" int gf_conv_fwd_flat(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask,
                                const int32_t* steps, const int32_t* flat, int K, int M_in, int M_out, int ld, int Cin, int Cout,
                                const float* in_scale, const float* in_shift, const float* residual,
                                const float* out_scale, const float* out_shift, float* out, float* out_act, void* stream) {
    GF_CHECK_ARG(out_act == nullptr || (out_scale != nullptr && out_shift != nullptr && ((uintptr_t)out_act % 16) == 0),
                 "gf_conv_fwd_flat: the second output needs its scale / shift and 16-byte alignment");
    return conv_fwd_impl(in, Wp, nbr, gmask, steps, flat, K, M_in, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out_scale,
                         out_shift, out, out_act, stream);
}

// gf_conv_fwd with TWO outputs: out = the raw sums (+ residual), out_act = max(out*out_scale + out_shift, 0).  The next
// residual block reads `out` as its residual operand and gathers from `out_act`, which takes the BatchNorm + ReLU out
// of its first convolution's prologue.  Level-1 shape only (the pipelined counted-loop kernel); gf_conv_dual_supported
// tells beforehand.
extern "C" int gf_conv_dual_supported(int M_out, int ld, int Cin, int Cout, int has_steps) {
    const ConvKnobs& knobs = conv_knobs();
    const int ngroups = (M_out + 15) / 16;
    bool split = ngroups < 6000;
    if (knobs.split >= 0) split = knobs.split != 0;
    bool g16 = has_steps && Cout == 16 && (Cin == 16 || Cin == 32) && (knobs.g16 >= 0 ? knobs.g16 != 0 : !split);
    bool pipe = M_out > 0 && ld >= M_out && (knobs.g16_pipe < 0 || knobs.g16_pipe != 0);
    return g16 && pipe ? 1 : 0;
}
extern "C" int gf_conv_fwd_dual(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask,
                                const int32_t* steps, int K, int M_in, int M_out, int ld, int Cin, int Cout,
                                const float* in_scale, const float* in_shift, const float* residual,
                                const float* out_scale, const float* out_shift, float* out, float* out_act, void* stream) {
    GF_CHECK_ARG(out_act != nullptr && out_scale != nullptr && out_shift != nullptr,
                 "gf_conv_fwd_dual: the second output needs its scale / shift");
    GF_CHECK_ARG(((uintptr_t)out_act % 16) == 0, "gf_conv_fwd_dual: out_act must be 16-byte aligned");
    return conv_fwd_impl(in, Wp, nbr, gmask, steps, nullptr, K, M_in, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out_scale,
                         out_shift, out, out_act, stream);
}

// Pre-activation residual block of the U-Net in one call (ResidualBlock, geoformer_modules.py:10-35, eval):
//   out = conv1(relu(bn1(conv0(relu(bn0(x)))))) + (Wi ? x . Wi : x)
// Three (two) launches of the kernels above; exists because the host side of a 17-26 us launch matters: one
// boundary crossing per block instead of three.
extern "C" int gf_resblock_fwd(const float* x, const float* Wp0, const float* Wp1, const float* Wpi,
                               const int32_t* nbr, const uint32_t* gmask, const int32_t* steps, int K, int M, int ld,
                               int Cin, int Cout,
                               const float* s0, const float* t0, const float* s1, const float* t1, float* tmp,
                               float* idn, float* out, void* stream) {
    GF_CHECK_ARG(x && Wp0 && Wp1 && tmp && out, "gf_resblock_fwd: null argument");
    GF_CHECK_ARG(Wpi != nullptr || Cin == Cout, "gf_resblock_fwd: identity branch needs Cin == Cout");
    GF_CHECK_ARG((Wpi == nullptr) == (idn == nullptr), "gf_resblock_fwd: Wpi and idn come together");
    int rc;
    if (Wpi) {
        rc = gf_conv_fwd(x, Wpi, nullptr, nullptr, nullptr, 1, M, M, 0, Cin, Cout, nullptr, nullptr, nullptr, nullptr,
                         nullptr, idn, stream);
        if (rc != GF_OK) return rc;
    }
    // bn1 + ReLU ride in the first conv's EPILOGUE (once per element of tmp) instead of the second conv's prologue
    // (once per gathered element: 6-10 x more often); tmp has no other reader
    rc = gf_conv_fwd(x, Wp0, nbr, gmask, steps, K, M, M, ld, Cin, Cout, s0, t0, nullptr, s1, t1, tmp, stream);
    if (rc != GF_OK) return rc;
    return gf_conv_fwd(tmp, Wp1, nbr, gmask, steps, K, M, M, ld, Cout, Cout, nullptr, nullptr, Wpi ? idn : x, nullptr,
                       nullptr, out, stream);
}

// Same launch bracketed by two caller-owned hipEvent_t recorded back to back with the kernel on the
// same stream (bench.py's roofline probe: the kernel's own duration, not the host's launch gaps).
extern "C" int gf_dev_conv_fwd_timed(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask,
                                 const int32_t* steps, int K, int M_in, int M_out, int ld, int Cin, int Cout, const float* in_scale,
                                 const float* in_shift, const float* residual, const float* out_scale,
                                 const float* out_shift, float* out, void* ev_start, void* ev_stop, void* stream) {
    GF_TRY(hipEventRecord((hipEvent_t)ev_start, (hipStream_t)stream));
    const int rc =
        gf_conv_fwd(in, Wp, nbr, gmask, steps, K, M_in, M_out, ld, Cin, Cout, in_scale, in_shift, residual, out_scale,
                    out_shift, out, stream);
    GF_TRY(hipEventRecord((hipEvent_t)ev_stop, (hipStream_t)stream));
    return rc;
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
    // 8 MFMAs (32 rows) per trip: the neighbour rows, then both operand streams, go out back to back -- one
    // dependent round trip per 32 rows instead of per 4 (the loop is pure memory latency otherwise)
    constexpr int U = 8;
    for (int base = row0; base < row1; base += 4 * U) {
        int idx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int row = base + 4 * u + q;
            idx[u] = row < row1 ? (nbr ? nbr[(size_t)k * ld + row] : row) : -1;
        }
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int row = base + 4 * u + q;
            a[u] = (idx[u] >= 0 && ci < Cin) ? in[(size_t)idx[u] * Cin + ci] : 0.f;
            b[u] = (idx[u] >= 0 && co < Cout) ? dout[(size_t)row * Cout + co] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    // D layout: col (output channel) = lane&15, row (input channel) = 4*(lane>>4) + j
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int cii = cib * 16 + q * 4 + j;
        if (cii < Cin && co < Cout && acc[j] != 0.f) atomicAdd(&dW[((size_t)k * Cin + cii) * Cout + co], acc[j]);
    }
}

// The same with the table's group masks (gmask[g] bit k: some row of the 16-row group g has a neighbour at offset k)
// and 16-byte loads.  Two things bound the kernel above (523k voxels, 16 -> 16: 226 us = 2.6x the forward):
//   * 61 % of the (group, offset) pairs of a scanned room's level 1 are empty (6.3 of 27 offsets per voxel, 10.5 per
//     group), and it walks them all: here a wave first lists the groups of its slice that have offset k at all;
//   * the MFMA wants lane (r, q) to supply channel r of row 4u + q, so every operand was a 4-byte load per lane --
//     12 vector-memory instructions of 64 addresses each per (group, offset), and the compute unit's ONE address
//     pipeline is the limit (time scaled with the instruction count, not with rows in flight or atomics).  Here lane
//     (row = lane >> 2, quad = lane & 3) loads 16 bytes of its row -- one instruction per operand tile -- and the tile
//     changes orientation through a per-wave LDS buffer (two 16-byte writes, eight 4-byte reads per pair).
// Channel counts are multiples of 16 on this route (the 6-channel input convolution takes the kernel above).
// 523k voxels: 16 -> 16 91 us (was 226), 32 -> 16 140 (417), 32 -> 32 239 (700), 64 -> 32 465 (1380).  What bounds it
// now is L2 bandwidth: a wave per (offset, 16 x 16 tile pair, slice) reads 2 KB per (group, offset) pair -- 700 MB per
// tile pair, 7.7 TB/s at 91 us -- because the gradient rows are read again for every offset and both operands again
// for every tile pair; one wave per slice for all tile pairs (and several offsets, one accumulator each) would read
// 1.8x (16 -> 16) to 3.6x (32 -> 32) less.  Built in round 4 (a wave per (slice, block of 9 offsets, 16 input channels)
// with 9 x NCO accumulators: the gradient rows loaded once per group visit, per present offset only the input tile
// gathered; loads for all nine offsets issued unconditionally so that the waits stay counted ones) and 2x SLOWER at
// every shape (193 / 287 / 555 / 997 us): with the closing atomics compiled out this kernel loses 1-3 %, and with 1.6-2.6x
// less traffic the other one loses a factor of two -- it is the number of vector-memory INSTRUCTIONS per useful (group,
// offset) pair that binds (3 here; 4.6-5.6 there because the absent offsets' index and gather instructions are still
// issued), the limit the first paragraph above names, not L2 bandwidth.  A form that wins would have to issue loads for
// present offsets only without giving up the counted waits of the software pipeline.  Not in the tree.
#ifndef WGT_ROWS
#define WGT_ROWS 1024  // rows per slice: more, shorter waves beat fewer, longer ones (2048: +80 %, 4096: +180 %)
#endif
#ifndef WGT_NB
#define WGT_NB 2  // (group, offset) pairs per trip (three trips in flight per wave)
#endif
__global__ __launch_bounds__(256) void k_conv_wgrad_t(const float* __restrict__ in, const float* __restrict__ dout,
                                                      const int32_t* __restrict__ nbr,
                                                      const uint32_t* __restrict__ gmask, int K, int M_out, int ld,
                                                      int Cin, int Cout, int NCI, int NCO, int nslices, int wps,
                                                      float* __restrict__ dW) {
    __shared__ __attribute__((aligned(16))) float sA[4][WGT_NB][256], sB[4][WGT_NB][256];
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
    const int rho = lane >> 2, gam = lane & 3;  // loading role: row of the group, channel quad
    // neighbouring waves = the tile pairs of one (offset, slice): they read the same rows at the same time, and the
    // compute unit's L1 serves the repeats (slices innermost instead: 32 -> 32 490 us against 239, 64 -> 32 957 / 465)
    int cob, cib, sl, k;
    if (wps > 0) {
        // XCD-aware order: workgroup w runs on XCD w % 8 (round-robin dispatch); all K x NCI x NCO items of a slice go to
        // workgroups w, w + 8, w + 16, ... of ONE XCD, back to back, so the slice's gradient rows and the input rows
        // around it are fetched into that XCD's L2 once instead of once per offset from every XCD (523k rows: 16 -> 16
        // 93 -> 85 us, 32 -> 32 243 -> 227, 64 -> 32 478 -> 446; taken from 128 slices up -- with few slices the order
        // would leave XCDs without work, and the batch-4 step as a whole got 1.5 ms slower with it on every level)
        const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
        sl = (t / wps) * 8 + xcd;
        int it = (t % wps) * 4 + wv;
        if (sl >= nslices || it >= K * NCI * NCO) return;
        cob = it % NCO;
        it /= NCO;
        cib = it % NCI;
        k = it / NCI;
    } else {
        long long item = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const long long nitems = (long long)K * NCI * NCO * nslices;
        if (item >= nitems) return;
        cob = (int)(item % NCO);
        item /= NCO;
        cib = (int)(item % NCI);
        item /= NCI;
        sl = (int)(item % nslices);
        k = (int)(item / nslices);
    }
    constexpr int GPS = WGT_ROWS / 16;  // groups per slice
    constexpr int NPW = (GPS + 63) / 64;
    const int g0 = sl * GPS, ngroups = (M_out + 15) >> 4;
    unsigned long long present[NPW];
#pragma unroll
    for (int w = 0; w < NPW; w++) {
        const int g = g0 + w * 64 + lane;
        present[w] = __ballot(w * 64 + lane < GPS && g < ngroups && (gmask ? ((gmask[g] >> k) & 1u) : 1u));
    }
    auto next_group = [&]() -> int {  // wave-uniform: the next listed group of the slice, -1 when the lists are empty
#pragma unroll
        for (int w = 0; w < NPW; w++) {
            if (present[w]) {
                const int b = __builtin_ctzll(present[w]);
                present[w] &= present[w] - 1;
                return g0 + w * 64 + b;
            }
        }
        return -1;
    };
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* inc = in + cib * 16 + 4 * gam;
    const float* doc = dout + cob * 16 + 4 * gam;
    const int32_t* nk = nbr ? nbr + (size_t)k * ld : nullptr;  // no table: the rows themselves (a 1x1x1 convolution)
    float* la = &sA[wv][0][0];
    float* lb = &sB[wv][0][0];
    // Three trips in flight per wave (the loop is a chain of two dependent memory round trips per trip otherwise):
    // while trip i goes through LDS and the MFMAs, the gathers of trip i+1 and the neighbour indices + gradient rows
    // of trip i+2 are on their way.  Every load is issued unconditionally (clamped address, result masked), so the
    // waits are exact counted ones.
    struct Trip {
        int idx[WGT_NB];   // neighbour row (or -1) of this lane's row in each of the trip's groups
        float4 bv[WGT_NB];
    };
    auto request = [&](Trip& tr) -> bool {  // lists the trip's groups, requests indices and gradient rows
        bool any = false;
#pragma unroll
        for (int t = 0; t < WGT_NB; t++) {
            const int g = next_group();
            any = any || g >= 0;
            const int row = g * 16 + rho;
            const bool ok = g >= 0 && row < M_out;
            const int rc = ok ? row : 0;
            const int i = nk ? nk[rc] : rc;
            const float4 v = *reinterpret_cast<const float4*>(doc + (size_t)rc * Cout);
            tr.idx[t] = ok ? i : -1;
            tr.bv[t] = v;
        }
        return any;
    };
    auto gather = [&](float4 (&av)[WGT_NB], const Trip& tr) {
#pragma unroll
        for (int t = 0; t < WGT_NB; t++) {
            const int i = tr.idx[t];
            const float4 v = *reinterpret_cast<const float4*>(inc + (size_t)(i >= 0 ? i : 0) * Cin);
            av[t] = i >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    Trip t0, t1, t2;
    float4 a0[WGT_NB], a1[WGT_NB];
    bool live0 = request(t0);
    bool live1 = request(t1);
    gather(a0, t0);
    while (live0) {
        const bool live2 = request(t2);
        gather(a1, t1);
#pragma unroll
        for (int t = 0; t < WGT_NB; t++) {
            *reinterpret_cast<float4*>(la + t * 256 + rho * 16 + 4 * gam) = a0[t];
            *reinterpret_cast<float4*>(lb + t * 256 + rho * 16 + 4 * gam) = t0.bv[t];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < WGT_NB; t++)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float a = la[t * 256 + (4 * u + q) * 16 + r];
                const float b = lb[t * 256 + (4 * u + q) * 16 + r];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        t0 = t1;
        t1 = t2;
#pragma unroll
        for (int t = 0; t < WGT_NB; t++) a0[t] = a1[t];
        live0 = live1;
        live1 = live2;
    }
    // D layout: col (output channel) = lane&15, row (input channel) = 4*(lane>>4) + j
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int cii = cib * 16 + q * 4 + j, co = cob * 16 + r;
        if (acc[j] != 0.f) atomicAdd(&dW[((size_t)k * Cin + cii) * Cout + co], acc[j]);
    }
}

static int conv_wgrad_impl(const float* in, const float* dout, const int32_t* nbr, int K, int M_out, int ld, int Cin,
                           int Cout, float* dW, bool zero, void* stream) {
    GF_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1, "gf_conv_wgrad: bad sizes");
    GF_CHECK_ARG(nbr != nullptr || K == 1, "gf_conv_wgrad: nbr==NULL requires K==1");
    hipStream_t st = (hipStream_t)stream;
    if (zero) GF_TRY(hipMemsetAsync(dW, 0, (size_t)K * Cin * Cout * sizeof(float), st));
    if (M_out <= 0) return GF_OK;
    const int nci = (Cin + 15) / 16, nco = (Cout + 15) / 16;
    const int nslices = (M_out + WG_ROWS - 1) / WG_ROWS;
    const long long nitems = (long long)K * nci * nco * nslices;
    hipLaunchKernelGGL(k_conv_wgrad, dim3(gf_div_up(nitems, 4)), dim3(256), 0, st, in, dout, nbr, K, M_out, ld, Cin,
                       Cout, nci, nco, nslices, dW);
    GF_CHECK_LAUNCH("gf_conv_wgrad");
    return GF_OK;
}

extern "C" int gf_conv_wgrad(const float* in, const float* dout, const int32_t* nbr, int K, int M_out, int ld, int Cin,
                             int Cout, float* dW, void* stream) {
    return conv_wgrad_impl(in, dout, nbr, K, M_out, ld, Cin, Cout, dW, true, stream);
}

static int conv_wgrad_masked_impl(const float* in, const float* dout, const int32_t* nbr, const uint32_t* gmask, int K,
                                  int M_out, int ld, int Cin, int Cout, float* dW, bool zero, void* stream) {
    // (K == 1 without a table -- a 1x1x1 convolution -- takes the tiled kernel as well: every group present, rows as
    // they lie; 165 -> ~30 us for the [560k, 32] x [560k, 16] product of a training batch's first level)
    if (((gmask == nullptr || nbr == nullptr) && !(K == 1 && nbr == nullptr)) || (Cin & 15) || (Cout & 15))
        return conv_wgrad_impl(in, dout, nbr, K, M_out, ld, Cin, Cout, dW, zero, stream);
    if (nbr == nullptr) gmask = nullptr;
    GF_CHECK_ARG(K >= 1 && K <= 32, "gf_conv_wgrad_masked: K=%d (at most 32 offsets)", K);
    hipStream_t st = (hipStream_t)stream;
    if (zero) GF_TRY(hipMemsetAsync(dW, 0, (size_t)K * Cin * Cout * sizeof(float), st));
    if (M_out <= 0) return GF_OK;
    const int nci = Cin / 16, nco = Cout / 16;
    const int nslices = (M_out + WGT_ROWS - 1) / WGT_ROWS;
    const long long nitems = (long long)K * nci * nco * nslices;
    if (nslices >= 128) {  // (few slices: the order would leave XCDs without work)
        const int wps = gf_div_up((long long)K * nci * nco, 4);  // workgroups per slice
        const long long wgs = (long long)gf_div_up(nslices, 8) * 8 * wps;
        GF_LAUNCH_OP(GF_OP_WGRAD, k_conv_wgrad_t, dim3((unsigned)wgs), dim3(256), 0, st, in, dout, nbr, gmask, K, M_out, ld, Cin,
                     Cout, nci, nco, nslices, wps, dW);
        GF_CHECK_LAUNCH("gf_conv_wgrad_masked");
        return GF_OK;
    }
    GF_LAUNCH_OP(GF_OP_WGRAD, k_conv_wgrad_t, dim3(gf_div_up(nitems, 4)), dim3(256), 0, st, in, dout, nbr, gmask, K, M_out, ld,
                 Cin, Cout, nci, nco, nslices, 0, dW);
    GF_CHECK_LAUNCH("gf_conv_wgrad_masked");
    return GF_OK;
}

extern "C" int gf_conv_wgrad_masked(const float* in, const float* dout, const int32_t* nbr, const uint32_t* gmask, int K,
                                    int M_out, int ld, int Cin, int Cout, float* dW, void* stream) {
    return conv_wgrad_masked_impl(in, dout, nbr, gmask, K, M_out, ld, Cin, Cout, dW, true, stream);
}

extern "C" int gf_conv_wgrad_masked_acc(const float* in, const float* dout, const int32_t* nbr, const uint32_t* gmask,
                                        int K, int M_out, int ld, int Cin, int Cout, float* dW, void* stream) {
    return conv_wgrad_masked_impl(in, dout, nbr, gmask, K, M_out, ld, Cin, Cout, dW, false, stream);
}


