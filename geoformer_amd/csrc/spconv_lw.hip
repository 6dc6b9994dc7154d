// LDS-weight sparse convolution over a flat step table (round 6): the middle U-Net level whose packed weights fit the
// compute unit's LDS (32 -> 32 at 27 offsets: 108 KiB), and the 32 -> 16 / 16 -> 16 shapes of the first level.
//
//   out[o,:] = sum_k act(in[nbr[k][o], window]) @ W[k][window, :]  (+ residual)
//
// What bounds the other forward kernels (profiles/r5_pmc_forward.md, profiles/r6_conv_rw_experiment.txt): a f32 MFMA
// 16x16x4 is 2048 flop in 32 cycles -- only twice a VALU fma -- so ~6 non-MFMA instructions per MFMA and SIMD is the
// whole budget.  k_conv_os spends 14 (bit-scanning, LDS index staging, per-trip address arithmetic, 1 KiB of weights
// per 4 MFMAs through the L1) and issues nothing of trip i + 1 under the MFMAs of trip i.  Here
//   * the rulebook hands over a FLAT STEP TABLE (spconv_rules.hip gf_rules_flat_steps): one 64-byte record of input rows
//     per (16-row group, present offset) in the order the MFMAs consume them, so gathers and index loads are linear in the
//     step number: a ring of D steps of gathers and D more of indices is in flight ACROSS group boundaries, every load
//     issued unconditionally (absent neighbour / step past the chunk = out-of-range buffer offset: zeros, no traffic),
//     all waits exact counted vmcnt's;
//   * a wave owns a chunk of consecutive groups of equal cost and ALL column blocks: one gather feeds 4 * NCB MFMAs;
//   * the packed weights of all offsets live in LDS (staged once per workgroup = compute unit), B operands are
//     conflict-free ds_read_b128 at (offset * NCH * NCB + block) KiB;
//   * the offset of a step comes from the group's mask (s_ff1 + s_and): ~20 non-MFMA instructions per 16 MFMAs at
//     32 -> 32, three waves per SIMD to fill the gaps.
// Transposed product (A = weights, B = gathered rows) as k_conv_g16p: a lane ends with four consecutive output channels
// of one row, one 16-byte store per column block.  fp32 MFMA (k-ordered fmaf chain), offsets ascending, two interleaved
// accumulators per column block.  More input chunks than fit are passes over channel windows (residual = the output).
#include "common.h"
#include "geoformer_hip_dev.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct LwArgs {
    const float* in;        // first channel of the window
    const float* Wp;        // packed weights of the whole convolution (gf_conv_pack_weights)
    const uint32_t* gmask;  // offset mask per 16-row group
    const int32_t* flat;    // flat step table (common.h GF_FLAT_*)
    const float *in_scale, *in_shift;  // of the window's channels, or null
    const float* residual;
    const float *out_scale, *out_shift;
    float* out;
    float* out2;
    unsigned in_bytes, row_bytes, steps_bytes;
    int nch_total, ch0, ncb_total, M_out, Cout;
    int nbins, ngroups, rounds;  // of the flat table (host-known: gf_rules_flat_steps' default bins)
};

#define LW_FENCE() asm volatile("" ::: "memory")

#ifdef LW_TRACE
// dev build: per-wave cycle stamps (s_memtime): start, after the prologue, end; steps, groups
__device__ unsigned long long* g_lw_trace = nullptr;
extern "C" int gf_dev_lw_trace(void* p) {
    GF_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lw_trace), &p, sizeof(p)));
    return GF_OK;
}
#endif

template <int K, int NCH, int NCB, int D, bool AFF, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_conv_lw(const LwArgs A) {
    constexpr int NBLK = NCH * NCB;  // weight blocks (1 KiB each) per offset
    constexpr int WPS = WPB / 4;     // waves per SIMD: they share one bin of the table, round j to wave j % WPS
    static_assert(NCB <= 2, "residual requests are written for one or two column blocks");
    static_assert(WPB % 4 == 0, "a workgroup is whole SIMD rounds");
    extern __shared__ __attribute__((aligned(16))) float4 s_w[];
#ifdef LW_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int32_t* __restrict__ flat = A.flat;
    const int nbins = A.nbins, ngroups = A.ngroups, rounds = A.rounds;
    // this wave's groups: rounds t, t + WPS, ... of bin (4 * workgroup + SIMD slot), one descriptor per lane
    const int bin = (int)blockIdx.x * 4 + (w & 3), t = w >> 2;
    int4 dsc = make_int4(-1, 0, 0, 0);
    {
        const int j = t + WPS * lane;
        if (j < rounds && bin < nbins)
            dsc = reinterpret_cast<const int4*>(flat + gf_flat_desc_at(ngroups))[(size_t)j * nbins + bin];
    }
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)A.in, 0, (int)A.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_st =
        __builtin_amdgcn_make_buffer_rsrc((void*)(flat + gf_flat_steps_at(ngroups, nbins)), 0, (int)A.steps_bytes, 0x00020000);
    const unsigned out_bytes = (unsigned)A.M_out * (unsigned)A.Cout * 4u;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)A.out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out2 = __builtin_amdgcn_make_buffer_rsrc((void*)(A.out2 ? A.out2 : A.out), 0, (int)out_bytes, 0x00020000);
    const unsigned row_bytes = A.row_bytes;
    const unsigned lane_c = 16u * (unsigned)q;
    const unsigned lane_r = 4u * (unsigned)r;

    // ---- the weights: every staging load in flight together, written to LDS behind the first index records ----
    constexpr int T = 64 * WPB;
    constexpr int total = K * NBLK * 64;
    constexpr int PER = (total + T - 1) / T;
    float4 tmp[PER];
    {
        const float4* __restrict__ Wp4 = reinterpret_cast<const float4*>(A.Wp);
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int e = j * T + (int)threadIdx.x;  // (k, c, cb, lane) of the window's image
            const int el = e & 63, blk = e >> 6;
            const int k = blk / NBLK, rem = blk - k * NBLK, c = rem / NCB, cb = rem - c * NCB;
            const int src = ((k * A.nch_total + A.ch0 + c) * A.ncb_total + cb) * 64 + el;
            tmp[j] = e < total ? Wp4[src] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 sc[AFF ? NCH : 1], sh[AFF ? NCH : 1];
    if (AFF) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            sc[c] = *reinterpret_cast<const float4*>(A.in_scale + c * 16 + 4 * q);
            sh[c] = *reinterpret_cast<const float4*>(A.in_shift + c * 16 + 4 * q);
        }
    }
    float4 os[NCB], ot[NCB];
    const bool has_res = A.residual != nullptr, has_act = A.out_scale != nullptr, has_out2 = A.out2 != nullptr;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        os[cb] = make_float4(1.f, 1.f, 1.f, 1.f);
        ot[cb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_act) {
            os[cb] = *reinterpret_cast<const float4*>(A.out_scale + cb * 16 + 4 * q);
            ot[cb] = *reinterpret_cast<const float4*>(A.out_shift + cb * 16 + 4 * q);
        }
    }

    // ---- the wave's step stream: its groups' records one after the other ----
    // number of groups (descriptors with a group), total steps
    const int ng = __builtin_amdgcn_readfirstlane(__popcll(__ballot(dsc.x >= 0)));
    int tot = dsc.x >= 0 ? dsc.z : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
    const int total_steps = __builtin_amdgcn_readfirstlane(tot);
    // index-load cursor: record ls of range li (ends at le); past the last range it stays on a valid record
    int li = 0;
    int ls = __builtin_amdgcn_readlane(dsc.y, 0);
    int le = ls + __builtin_amdgcn_readlane(dsc.z, 0);
    auto next_record = [&]() -> unsigned {  // byte offset of the stream's next record
        while (ls >= le && li + 1 < ng) {
            li++;
            ls = __builtin_amdgcn_readlane(dsc.y, li);
            le = ls + __builtin_amdgcn_readlane(dsc.z, li);
        }
        const unsigned off = (unsigned)ls * 64u;
        if (ls < le) ls++;
        return off;
    };
    int idx0[D], idxr[D];
#pragma unroll
    for (int d = 0; d < D; d++) idx0[d] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_st, lane_r, next_record(), 0);
#pragma unroll
    for (int d = 0; d < D; d++) idxr[d] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_st, lane_r, next_record(), 0);
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int e = j * T + (int)threadIdx.x;
        if (e < total) s_w[e] = tmp[j];
    }
    // gathers of the first D steps
    u32x4 ring[D][NCH];
    unsigned absent = 0;  // bit d: the row in ring slot d was absent (AFF)
#pragma unroll
    for (int d = 0; d < D; d++) {
        const unsigned voff = __umul24((unsigned)idx0[d], row_bytes) + lane_c;
        if (AFF) absent |= ((unsigned)idx0[d] >> 31) << d;
        const unsigned sb = d < total_steps ? 0u : 0x80000000u;
#pragma unroll
        for (int c = 0; c < NCH; c++) ring[d][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, sb + c * 64, 0);
    }
#ifdef LW_TRACE
    const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();  // the weights are in LDS
#ifdef LW_TRACE
    const unsigned long long tr2 = __builtin_amdgcn_s_memtime();
#endif
    if (ng == 0) return;

    const unsigned out_rowb = (unsigned)A.Cout * 4u;
    const unsigned lane_w = (unsigned)lane * 16u;
    f32x4 acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) acc[cb][0] = acc[cb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Residual rows are requested ONE GROUP AHEAD and waited for by hand.  As an intrinsic load next to its use the compiler
    // puts a vmcnt(0) in front of the add (the path from a request to its use may hold no other load: an empty group), which
    // drains the whole gather ring at every group boundary.  As inline asm the compiler does not see the load (its own
    // counted waits only get stricter by the hidden loads: still correct), and the wave waits itself: a request that is at
    // least D consumed steps old has landed -- the counted wait of the D-th step covers a gather issued AFTER it -- and only
    // a shorter group pays a vmcnt(0).
    u32x4 resv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) resv[cb] = (u32x4){0u, 0u, 0u, 0u};
    int since = D;
    // (the residual's buffer descriptor as four plain words for the asm operand: base, base high 16 bits, bytes, format)
    const unsigned long long res_base = (unsigned long long)(A.residual ? A.residual : A.out);
    const u32x4 rs_res_v = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)res_base),
                            (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(res_base >> 32) & 0xffffu)),
                            (unsigned)__builtin_amdgcn_readfirstlane((int)out_bytes), 0x00020000u};
    auto request_res = [&](int ci_) {  // for the wave's ci_-th group
        if (has_res && ci_ < ng) {
            const int gg = __builtin_amdgcn_readlane(dsc.x, ci_);
            const unsigned ooff = (unsigned)(16 * gg + r) * out_rowb + 16u * (unsigned)q;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(resv[0]) : "v"(ooff), "s"(rs_res_v) : "memory");
            if (NCB > 1)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:64" : "=v"(resv[NCB > 1 ? 1 : 0]) : "v"(ooff), "s"(rs_res_v) : "memory");
            since = 0;
        }
    };
    // the group's sums leave the wave: residual, epilogue activation, one 16-byte store per column block
    auto finish_group = [&](int gg, int ci_) {
        const unsigned ooff = (unsigned)(16 * gg + r) * out_rowb + 16u * (unsigned)q;  // (a row past M_out: beyond the descriptors)
        if (has_res) {
            if (since < D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) asm volatile("" : "+v"(resv[cb]));
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            float4 v = make_float4(acc[cb][0][0] + acc[cb][1][0], acc[cb][0][1] + acc[cb][1][1], acc[cb][0][2] + acc[cb][1][2],
                                   acc[cb][0][3] + acc[cb][1][3]);
            acc[cb][0] = acc[cb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (has_res) {
                v.x += __uint_as_float(resv[cb][0]); v.y += __uint_as_float(resv[cb][1]);
                v.z += __uint_as_float(resv[cb][2]); v.w += __uint_as_float(resv[cb][3]);
            }
            if (has_act) {
                if (has_out2) {
                    const u32x4 raw = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_out, ooff, cb * 64, 0);
                }
                v.x = fmaxf(fmaf(v.x, os[cb].x, ot[cb].x), 0.f); v.y = fmaxf(fmaf(v.y, os[cb].y, ot[cb].y), 0.f);
                v.z = fmaxf(fmaf(v.z, os[cb].z, ot[cb].z), 0.f); v.w = fmaxf(fmaf(v.w, os[cb].w, ot[cb].w), 0.f);
            }
            const u32x4 o4 = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            if (has_act && has_out2) __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out2, ooff, cb * 64, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out, ooff, cb * 64, 0);
        }
        request_res(ci_ + 1);
    };

    // consumer state: the wave's ci-th group g, the offsets of it that are still to come (m, the current step's included),
    // the current step's offset kc with its weights already in wv[parity]
    int ci = 0;
    int g = __builtin_amdgcn_readlane(dsc.x, 0);
    uint32_t m = (uint32_t)__builtin_amdgcn_readlane(dsc.w, 0);
    request_res(0);
    since = -1;  // (the prologue's D gathers went out BEFORE this request: the first gather behind it belongs to step D,
                 //  whose counted wait runs when `since` would read D + 1)
    auto next_group = [&]() {
        ci++;
        g = __builtin_amdgcn_readlane(dsc.x, min(ci, 63));
        m = ci < ng ? (uint32_t)__builtin_amdgcn_readlane(dsc.w, min(ci, 63)) : 0u;
    };
    // groups without any offset (an inverse convolution's rows outside the coarse grid): zeros (+ residual)
    while (m == 0u && ci < ng) {
        finish_group(g, ci);
        next_group();
    }
    float4 wv[2][NCH][NCB];
    auto fetch_w = [&](float4 (&dst)[NCH][NCB], int k) {
        const unsigned wb = (unsigned)k * (NBLK * 1024u) + lane_w;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#if defined(LW_STRIP) && LW_STRIP == 3  // no LDS weight reads
                dst[c][cb] = make_float4(__uint_as_float(wb), 1.f, 2.f, (float)cb);
#else
                dst[c][cb] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_w) + wb + (c * NCB + cb) * 1024);
#endif
    };
    int kc = m ? __builtin_ctz(m) : 0;
    fetch_w(wv[0], kc);

    for (int s = 0; s < total_steps; s += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (s + d < total_steps) {
                // ---- consume stream step s + d: offset kc of group g; the next step's weights are requested first ----
                m &= m - 1;
                since++;
                uint32_t mm = m;  // the mask the next step's offset comes from: this group's rest, or the next group's
                if (mm == 0u && ci + 1 < ng) mm = (uint32_t)__builtin_amdgcn_readlane(dsc.w, min(ci + 1, 63));
                kc = mm ? __builtin_ctz(mm) : 0;
                fetch_w(wv[(d + 1) & 1], kc);
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    float4 x = make_float4(__uint_as_float(ring[d][c][0]), __uint_as_float(ring[d][c][1]),
                                           __uint_as_float(ring[d][c][2]), __uint_as_float(ring[d][c][3]));
                    if (AFF) {
                        const bool present = ((absent >> d) & 1u) == 0u;
                        x.x = fmaxf(fmaf(x.x, sc[c].x, present ? sh[c].x : 0.f), 0.f);
                        x.y = fmaxf(fmaf(x.y, sc[c].y, present ? sh[c].y : 0.f), 0.f);
                        x.z = fmaxf(fmaf(x.z, sc[c].z, present ? sh[c].z : 0.f), 0.f);
                        x.w = fmaxf(fmaf(x.w, sc[c].w, present ? sh[c].w : 0.f), 0.f);
                    }
                    const float4 (&wc)[NCB] = wv[d & 1][c];
#if defined(LW_STRIP) && LW_STRIP == 1  // no MFMA
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) { acc[cb][0][0] += x.x * wc[cb].x; acc[cb][0][1] += x.y * wc[cb].y; }
                    continue;
#endif
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) acc[cb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[cb].x, x.x, acc[cb][0], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) acc[cb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[cb].y, x.y, acc[cb][1], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) acc[cb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[cb].z, x.z, acc[cb][0], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) acc[cb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[cb].w, x.w, acc[cb][1], 0, 0, 0);
                }
                if (m == 0u) {
                    finish_group(g, ci);
                    next_group();
                    if (m == 0u && ci < ng) {  // (empty groups in between: the weights requested above belong to none of them)
                        do {
                            finish_group(g, ci);
                            next_group();
                        } while (m == 0u && ci < ng);
                        kc = m ? __builtin_ctz(m) : 0;
                        fetch_w(wv[(d + 1) & 1], kc);
                    }
                }
            }
            // ---- produce: the gathers of stream step s + d + D, the index record of step s + d + 2 D ----
            {
                const unsigned voff = __umul24((unsigned)idxr[d], row_bytes) + lane_c;
                if (AFF) absent = (absent & ~(1u << d)) | (((unsigned)idxr[d] >> 31) << d);
                const unsigned sb = s + d + D < total_steps ? 0u : 0x80000000u;
#if defined(LW_STRIP) && LW_STRIP == 2  // no gathers
#pragma unroll
                for (int c = 0; c < NCH; c++) ring[d][c] = (u32x4){voff, sb, voff + 1u, (unsigned)c};
#else
#pragma unroll
                for (int c = 0; c < NCH; c++) ring[d][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, sb + c * 64, 0);
#endif
                idxr[d] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_st, lane_r, next_record(), 0);
                LW_FENCE();
            }
        }
    }
#ifdef LW_TRACE
    if (g_lw_trace && lane == 0) {
        unsigned long long* tt = g_lw_trace + (size_t)((int)blockIdx.x * WPB + w) * 8;
        tt[0] = tr0; tt[1] = tr1; tt[2] = tr2; tt[3] = __builtin_amdgcn_s_memtime(); tt[4] = (unsigned long long)total_steps;
        tt[5] = (unsigned long long)ng;
    }
#endif
}

namespace {

int g_lw_use = -1;         // -1 size-based, 0 never, 1 whenever the shape allows
int g_lw_min_groups = 0;   // size-based choice: at least this many 16-row groups (0 = default)

template <int K, int NCH, int NCB, int D>
int lw_launch(bool aff, hipStream_t st, const LwArgs& a) {
    constexpr int WPB = 12;
    constexpr size_t lds = (size_t)K * NCH * NCB * 1024;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_conv_lw<K, NCH, NCB, D, true, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_conv_lw<K, NCH, NCB, D, false, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const dim3 grid((unsigned)(a.nbins / 4));
    if (aff) hipLaunchKernelGGL((k_conv_lw<K, NCH, NCB, D, true, WPB>), grid, dim3(64 * WPB), lds, st, a);
    else hipLaunchKernelGGL((k_conv_lw<K, NCH, NCB, D, false, WPB>), grid, dim3(64 * WPB), lds, st, a);
    return 0;
}

}  // namespace

extern "C" int gf_dev_conv_knob_lw(int use, int min_groups) {
    g_lw_use = use < 0 ? -1 : (use != 0);
    g_lw_min_groups = min_groups > 0 ? min_groups : 0;
    return GF_OK;
}

// chunks per input pass and column blocks the kernel is instantiated for: (window chunks, column blocks)
static bool lw_shape(int nchw, int ncb) { return (nchw == 1 || nchw == 2) && (ncb == 1 || ncb == 2); }

// 1 if the LDS-weight kernel takes this convolution (gf_conv_fwd's dispatch, given a flat table); `forced`: the dev knob said so
int gf_conv_lw_supported(int K, int M_in, int M_out, int Cin, int Cout, bool aligned, int* forced) {
    if (forced) *forced = g_lw_use == 1;
    if (g_lw_use == 0) return 0;
    if (!aligned || K != 27) return 0;
    if ((Cin & 15) || (Cout & 15)) return 0;
    const int nch = Cin / 16, ncb = Cout / 16;
    const int npass = (nch + 1) / 2;
    if (!lw_shape((nch + npass - 1) / npass, ncb) || !lw_shape(nch / npass, ncb)) return 0;
    // measured on S150k (tools/conv_lw_exp.py, profiles/r6_conv_lw_notes.md): 32 -> 32 21.8 against 27.1 us, 32 -> 16 30.2
    // against 46.6; 16 -> 16 equal to k_conv_g16p (17.7 / 17.9), two passes (64 -> 32) slower than k_conv_os (49 / 44)
    const bool pays = npass == 1 && nch == 2;
    // 24-bit row multiply; an absent row's offset 0xFFFFFF * row_bytes (mod 2^32) >= 2^30 - row_bytes must lie beyond the buffer
    if (M_in >= (1 << 24) || (unsigned long long)M_in * Cin * 4ull > (1ull << 30) - 4096ull) return 0;
    if ((M_out + 15) / 16 > 64 * 3 * GF_FLAT_BINS) return 0;  // (a wave keeps its groups' descriptors one per lane)
    if (g_lw_use == 1) return 1;
    return pays && (M_out + 15) / 16 >= (g_lw_min_groups > 0 ? g_lw_min_groups : 1500);
}

int gf_conv_lw(const float* in, const float* Wp, const uint32_t* gmask, const int32_t* flat, int K, int M_in, int M_out, int Cin,
               int Cout, const float* in_scale, const float* in_shift, const float* residual, const float* out_scale,
               const float* out_shift, float* out, float* out2, hipStream_t st) {
    const int nch = Cin / 16, ncb = Cout / 16;
    const int ngroups = (M_out + 15) / 16;
    const int npass = (nch + 1) / 2;
    int ch0 = 0;
    for (int p = 0; p < npass; p++) {
        const int n = (nch - ch0 + (npass - p) - 1) / (npass - p);
        const bool first = p == 0, last = p == npass - 1;
        LwArgs a;
        a.in = in + ch0 * 16;
        a.Wp = Wp;
        a.gmask = gmask;
        a.flat = flat;
        a.in_scale = in_scale ? in_scale + ch0 * 16 : nullptr;
        a.in_shift = in_shift ? in_shift + ch0 * 16 : nullptr;
        a.residual = first ? residual : out;
        a.out_scale = last ? out_scale : nullptr;
        a.out_shift = last ? out_shift : nullptr;
        a.out = out;
        a.out2 = last ? out2 : nullptr;
        a.in_bytes = (unsigned)((unsigned long long)M_in * Cin * 4ull - (unsigned long long)ch0 * 64ull);
        a.row_bytes = (unsigned)Cin * 4u;
        a.steps_bytes = (unsigned)(((size_t)K * ngroups + GF_FLAT_PAD) * 64);
        a.nch_total = nch;
        a.ch0 = ch0;
        a.ncb_total = ncb;
        a.M_out = M_out;
        a.Cout = Cout;
        a.nbins = GF_FLAT_BINS;  // (tables are built with the default bins: gf_rules_flat_steps(..., 0, ...))
        a.ngroups = ngroups;
        a.rounds = (ngroups + a.nbins - 1) / a.nbins;
        const bool aff = in_scale != nullptr;
        if (n == 1 && ncb == 1) lw_launch<27, 1, 1, 8>(aff, st, a);
        else if (n == 2 && ncb == 1) lw_launch<27, 2, 1, 8>(aff, st, a);
        else if (n == 1 && ncb == 2) lw_launch<27, 1, 2, 8>(aff, st, a);
        else lw_launch<27, 2, 2, 6>(aff, st, a);
        ch0 += n;
    }
    GF_CHECK_LAUNCH("gf_conv_fwd (LDS-weight kernel)");
    return GF_OK;
}
