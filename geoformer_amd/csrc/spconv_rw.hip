// Register-weight sparse convolution for the middle U-Net levels (round 6).
//
//   out[o, cb*16 .. cb*16+15] = sum_k act(in[nbr[k][o], window]) @ W[k][window, cb*16 .. cb*16+15]  (+ residual)
//
// What bound k_conv_os on levels 2-4 (spconv_conv.hip; PMC of round 5: 358 scalar + 349 vector instructions around 49 MFMAs
// per wave, 38 % of the padded tiles' matrix time at every problem size): a trip of its loop is LDS lookup -> readfirstlane
// -> LDS lookup -> address arithmetic -> gather + two 1-KiB weight loads -> s_waitcnt -> 16 MFMAs, nothing of trip i + 1 in
// flight under the MFMAs of trip i, and two thirds of what it pulls through the compute unit's L1 are weights.
//
// Here ONE wave owns (a contiguous range of 16-row groups) x (one 16-column block of the output) and keeps that column
// block's weights of ALL K offsets and its NCH <= 3 input chunks in REGISTERS (K * NCH * 4 VGPRs: 216 at 32 channels, 324 at
// 48; one wave per SIMD, the register file is the weight store), so the loop over a group's offsets is fully unrolled
// with static weight operands:
//   * no weight traffic at all after the prologue, no LDS, no barrier, no cross-wave reduction;
//   * per offset: one address multiply, NCH 16-byte gathers (lane (r, q) = row r, channels 4q..4q+3 of a chunk: the B
//     operand of the transposed product out^T = W^T in^T, as k_conv_g16p), one 4-byte index load, 4 * NCH MFMAs behind a
//     scalar branch on the group's offset mask;
//   * a ring of D offsets of gathers in flight that runs across group boundaries; every load is issued unconditionally
//     (an absent neighbour or a group past the range's end is an out-of-range buffer offset: zeros, no traffic), so all
//     waits are exact counted vmcnt's placed by the compiler;
//   * the neighbour index of (group g + 1, offset j) is re-loaded into the register that held (g, j) right after its
//     last use, a whole group ahead of its next one.
// The other column blocks of the same rows are other waves (neighbouring waves of a workgroup: their gathers hit L1).
// Convolutions with more than three input chunks run as passes over channel windows of the input (second pass: residual =
// the first pass's output, same element by the same lane).
//
// Arithmetic: fp32 MFMA 16x16x4 (k-ordered fmaf chain), offsets ascending, two interleaved accumulators (even / odd
// channel pairs) summed at the end -- not the summation order of k_conv_os, the same 1e-4 bound against the oracle.
#include "common.h"
#include "geoformer_hip_dev.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Compiler fences.  Left alone, hipcc (ROCm 7.2) (1) RE-MATERIALISES the register-resident weights -- re-issues their loads
// inside the loop behind a vmcnt(0) -- and (2) sinks the gathers of several ring slots into one cluster, which shortens
// their lead.  RW_PIN makes a loaded value opaque (defined by an asm: nothing to re-materialise; it also waits for the
// load), RW_FENCE is a memory clobber no load moves across.  (The intrinsics' "volatile" aux bit is no alternative: on
// gfx950 it is lowered to sc0 sc1, i.e. every gather would bypass the caches.)
#define RW_PIN4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#define RW_FENCE() asm volatile("" ::: "memory")

struct RwArgs {
    const float* in;        // first channel of the window
    const float* Wp;        // packed weights of the whole convolution (gf_conv_pack_weights)
    const int32_t* nbr;     // [K, ld]
    const uint32_t* gmask;  // per 16-row group, or null (every offset present)
    const float *in_scale, *in_shift;  // of the window's channels, or null
    const float* residual;
    const float *out_scale, *out_shift;
    float* out;
    float* out2;
    const int32_t* bounds;  // optional [nranges + 1]: first group of every range (equal-cost ranges); null: equal counts
    unsigned in_bytes, row_bytes, w_bytes, nbr_bytes;
    int nch_total, ch0, NCB, ld, M_out, nranges, Cout;
};

// act(x) = max(x * s + t, 0) on a present row, 0 on an absent one
__device__ __forceinline__ float4 rw_act(float4 x, bool present, float4 s, float4 t) {
    float4 y;
    y.x = fmaxf(fmaf(x.x, s.x, present ? t.x : 0.f), 0.f);
    y.y = fmaxf(fmaf(x.y, s.y, present ? t.y : 0.f), 0.f);
    y.z = fmaxf(fmaf(x.z, s.z, present ? t.z : 0.f), 0.f);
    y.w = fmaxf(fmaf(x.w, s.w, present ? t.w : 0.f), 0.f);
    return y;  // (an absent row was gathered as zeros: fma(0, s, 0) = 0)
}

#ifdef RW_TRACE
// dev build: per-wave cycle stamps (s_memtime): start, after the prologue, end; groups, present offsets
__device__ unsigned long long* g_rw_trace = nullptr;
extern "C" int gf_dev_rw_trace(void* p) {
    GF_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_rw_trace), &p, sizeof(p)));
    return GF_OK;
}
#endif

template <int K, int NCH, int D, bool AFF>
__global__ __launch_bounds__(256, NCH == 1 ? 2 : 1) void k_conv_rw(const RwArgs A) {
    static_assert(K % D == 0, "the ring must divide the offsets");
    __shared__ __attribute__((aligned(16))) float s_aff[2][NCH * 16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    if (AFF) {
        if (threadIdx.x < NCH * 16) {
            s_aff[0][threadIdx.x] = A.in_scale[threadIdx.x];
            s_aff[1][threadIdx.x] = A.in_shift[threadIdx.x];
        }
        __syncthreads();
    }
#ifdef RW_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
    unsigned long long tr_n = 0;
#endif
    const int NCB = A.NCB;
    const int item = blockIdx.x * 4 + w;
    if (item >= A.nranges * NCB) return;
    const int range = item / NCB, cb = item - range * NCB;
    const int ngroups = (A.M_out + 15) >> 4;
    int g0 = (int)((long long)range * ngroups / A.nranges);
    int g1 = (int)((long long)(range + 1) * ngroups / A.nranges);
    if (A.bounds) {
        g0 = A.bounds[range];
        g1 = A.bounds[range + 1];
    }
    g0 = __builtin_amdgcn_readfirstlane(g0);
    g1 = __builtin_amdgcn_readfirstlane(g1);
    if (g0 >= g1) return;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)A.in, 0, (int)A.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)A.Wp, 0, (int)A.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nbr = __builtin_amdgcn_make_buffer_rsrc((void*)A.nbr, 0, (int)A.nbr_bytes, 0x00020000);
    const unsigned row_bytes = A.row_bytes;
    const unsigned lane_c = 16u * (unsigned)q;
    const unsigned ldb = (unsigned)A.ld * 4u;
    const int M_out = A.M_out;
    // A row past M_out in the last group needs no care: whatever its table entries hold, its gathers are range-checked
    // buffer loads and its column of the transposed product is never stored.
    const __amdgpu_buffer_rsrc_t rs_gm = __builtin_amdgcn_make_buffer_rsrc((void*)A.gmask, 0, ((M_out + 15) >> 4) * 4, 0x00020000);
    const unsigned out_bytes = (unsigned)M_out * (unsigned)A.Cout * 4u;
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(A.residual ? A.residual : A.out), 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)A.out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out2 = __builtin_amdgcn_make_buffer_rsrc((void*)(A.out2 ? A.out2 : A.out), 0, (int)out_bytes, 0x00020000);

    // neighbour indices of the first group
    int idx[K];
    {
        const unsigned vrow = (unsigned)(16 * g0 + r) * 4u;
#pragma unroll
        for (int j = 0; j < K; j++) idx[j] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_nbr, vrow, (unsigned)j * ldb, 0);
    }
    // the offset masks of the range's groups, one per lane (a range holds at most 64 groups: gf_conv_rw)
    const int mask_v = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_gm, (unsigned)(g0 + lane) * 4u, 0, 0);
    // the first D offsets of the first group, then their registers take the second group's indices
    u32x4 ring[D][NCH];
    unsigned absent = 0;  // bit s: the row gathered into ring slot s was absent (AFF)
    {
        const unsigned vrow = (unsigned)(16 * min(g0 + 1, g1 - 1) + r) * 4u;
#pragma unroll
        for (int s = 0; s < D; s++) {
            const unsigned voff = __umul24((unsigned)idx[s], row_bytes) + lane_c;
            if (AFF) absent |= ((unsigned)idx[s] >> 31) << s;
#pragma unroll
            for (int c = 0; c < NCH; c++) ring[s][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, c * 64, 0);
            idx[s] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_nbr, vrow, (unsigned)s * ldb, 0);
        }
    }
    // this column block's weights: wr[k][c] = lane's 16 bytes of block (k, ch0 + c, cb)
    float4 wr[K][NCH];
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(
                rs_w, (unsigned)lane * 16u, (unsigned)(((k * A.nch_total + A.ch0 + c) * NCB + cb)) * 1024u, 0);
            wr[k][c] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
        }
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int c = 0; c < NCH; c++) RW_PIN4(wr[k][c]);
    float4 os = make_float4(1.f, 1.f, 1.f, 1.f), ot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A.out_scale) {
        os = *reinterpret_cast<const float4*>(A.out_scale + cb * 16 + 4 * q);
        ot = *reinterpret_cast<const float4*>(A.out_shift + cb * 16 + 4 * q);
    }
    const unsigned out_colb = ((unsigned)cb * 16u + 4u * (unsigned)q) * 4u;
    const unsigned out_rowb = (unsigned)A.Cout * 4u;
    const bool has_res = A.residual != nullptr, has_act = A.out_scale != nullptr, has_out2 = A.out2 != nullptr;

#ifdef RW_TRACE
    const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
#endif
    for (int g = g0; g < g1; g++) {
        const unsigned sb_next = g + 1 < g1 ? 0u : 0x80000000u;  // gathers of a group past the range: out of range
        const int gq1 = min(g + 1, g1 - 1), gq2 = min(g + 2, g1 - 1);
        const unsigned vrow1 = (unsigned)(16 * gq1 + r) * 4u, vrow2 = (unsigned)(16 * gq2 + r) * 4u;
        const uint32_t m = (uint32_t)__builtin_amdgcn_readlane(mask_v, g - g0);
#if defined(RW_STRIP) && RW_STRIP == 4
        const uint32_t m1 = (uint32_t)__builtin_amdgcn_readlane(mask_v, min(g + 1, g1 - 1) - g0);
#endif
        const unsigned ooff = (unsigned)(16 * g + r) * out_rowb + out_colb;  // (a row past M_out: beyond the descriptors)
        u32x4 resv = {0u, 0u, 0u, 0u};
        if (has_res) resv = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ooff, 0, 0);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int slot = k % D;
            if ((m >> k) & 1u) {
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    float4 x = make_float4(__uint_as_float(ring[slot][c][0]), __uint_as_float(ring[slot][c][1]),
                                           __uint_as_float(ring[slot][c][2]), __uint_as_float(ring[slot][c][3]));
#if defined(RW_STRIP) && RW_STRIP == 1  // no MFMA: the memory pipeline alone
                    acc0[0] += x.x; acc0[1] += x.y; acc0[2] += x.z; acc0[3] += x.w;
                    continue;
#endif
                    if (AFF) {
                        const float4 s4 = *reinterpret_cast<const float4*>(&s_aff[0][c * 16 + 4 * q]);
                        const float4 t4 = *reinterpret_cast<const float4*>(&s_aff[1][c * 16 + 4 * q]);
                        x = rw_act(x, ((absent >> slot) & 1u) == 0u, s4, t4);
                    }
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[k][c].x, x.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[k][c].y, x.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[k][c].z, x.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[k][c].w, x.w, acc1, 0, 0, 0);
                }
            }
            // the slot's next tenant: offset j of this group or of the next one
            const int j = (k + D) % K;
            const bool nxt = k + D >= K;
            const unsigned voff = __umul24((unsigned)idx[j], row_bytes) + lane_c;
            if (AFF) absent = (absent & ~(1u << slot)) | (((unsigned)idx[j] >> 31) << slot);
#if defined(RW_STRIP) && (RW_STRIP == 2 || RW_STRIP == 3)  // no gathers
#pragma unroll
            for (int c = 0; c < NCH; c++) ring[slot][c] = (u32x4){voff, voff + 1u, voff + 2u, voff + (unsigned)c};
#elif defined(RW_STRIP) && RW_STRIP == 4  // gathers of absent offsets skipped (the waits are the compiler's)
            if ((((nxt ? m1 : m) >> j) & 1u) != 0u) {
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    ring[slot][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, (nxt ? sb_next : 0u) + c * 64, 0);
            }
#else
#pragma unroll
            for (int c = 0; c < NCH; c++)
                ring[slot][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, (nxt ? sb_next : 0u) + c * 64, 0);
#endif
            // ... and the register that held its index takes the same offset of the group after
#if defined(RW_STRIP) && (RW_STRIP == 3 || RW_STRIP == 5)  // no index loads
            idx[j] = idx[j] + 16;
#else
            idx[j] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_nbr, nxt ? vrow2 : vrow1, (unsigned)j * ldb, 0);
#endif
            RW_FENCE();
        }
        // transposed C/D layout: lane (r, q) holds channels 4q..4q+3 of row 16g + r
        float4 v = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
        if (has_res) {
            v.x += __uint_as_float(resv[0]); v.y += __uint_as_float(resv[1]);
            v.z += __uint_as_float(resv[2]); v.w += __uint_as_float(resv[3]);
        }
        if (has_act) {  // epilogue activation: the consumer's BatchNorm + ReLU, once per output element
            if (has_out2) {
                const u32x4 raw = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                __builtin_amdgcn_raw_buffer_store_b128(raw, rs_out, ooff, 0, 0);
            }
            v.x = fmaxf(fmaf(v.x, os.x, ot.x), 0.f); v.y = fmaxf(fmaf(v.y, os.y, ot.y), 0.f);
            v.z = fmaxf(fmaf(v.z, os.z, ot.z), 0.f); v.w = fmaxf(fmaf(v.w, os.w, ot.w), 0.f);
        }
        const u32x4 o4 = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        if (has_act && has_out2) __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out2, ooff, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out, ooff, 0, 0);
#ifdef RW_TRACE
        tr_n += (unsigned long long)__popc(m);
#endif
    }
#ifdef RW_TRACE
    if (g_rw_trace && lane == 0) {
        unsigned long long* t = g_rw_trace + (size_t)item * 8;
        t[0] = tr0; t[1] = tr1; t[2] = __builtin_amdgcn_s_memtime(); t[3] = (unsigned long long)(g1 - g0); t[4] = tr_n;
        t[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

namespace {

int g_rw_use = -1;       // -1 size-based, 0 never, 1 whenever the shape allows
int g_rw_min_items = 0;  // size-based choice: at least this many (group, column block) items (0 = default)
int g_rw_maxch = 3;      // input chunks per pass (GF_CONV_RW_MAXCH: 2 = a 48-channel input as 2 + 1)

template <int K, int NCH, int D>
void rw_launch(bool aff, dim3 grid, hipStream_t st, const RwArgs& a) {
    if (aff) hipLaunchKernelGGL((k_conv_rw<K, NCH, D, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_rw<K, NCH, D, false>), grid, dim3(256), 0, st, a);
}

}  // namespace

// dev hook (tools/conv_rw_exp.py): range boundaries computed by the caller, [n + 1] device ints; null = off
static const int32_t* g_rw_bounds = nullptr;
static int g_rw_nbounds = 0;
extern "C" int gf_dev_conv_rw_bounds(const int32_t* bounds, int n) {
    g_rw_bounds = bounds;
    g_rw_nbounds = bounds ? n : 0;
    return GF_OK;
}

extern "C" int gf_dev_conv_knob_rw(int use, int min_items) {
    g_rw_use = use < 0 ? -1 : (use != 0);
    g_rw_min_items = min_items > 0 ? min_items : 0;
    return GF_OK;
}

// 1 if the register-weight kernel takes this convolution (gf_conv_fwd's dispatch); `forced`: the dev knob said so
int gf_conv_rw_supported(int K, int M_in, int M_out, int Cin, int Cout, bool has_nbr, bool aligned, int* forced) {
    static bool env_read = false;
    if (!env_read) {
        env_read = true;
        if (const char* e = getenv("GF_CONV_RW")) g_rw_use = atoi(e) != 0;
        if (const char* e = getenv("GF_CONV_RW_MAXCH")) g_rw_maxch = atoi(e) == 2 ? 2 : 3;
    }
    if (forced) *forced = g_rw_use == 1;
    if (g_rw_use == 0) return 0;
    if (!has_nbr || !aligned || (K != 27 && K != 8)) return 0;  // (has_nbr: table AND group masks)
    if ((Cin & 15) || (Cout & 15) || Cin < 16 || Cout < 16) return 0;
    if (M_in >= (1 << 24) || (unsigned long long)M_in * Cin * 4ull > (1ull << 30) - 4096ull) return 0;  // 24-bit multiply; an absent row's
    // offset 0xFFFFFF * row_bytes (mod 2^32) >= 2^30 - row_bytes must lie beyond the buffer
    if (g_rw_use == 1) return 1;
    // size-based: enough (range, column block) waves for one per SIMD, and a level where k_conv_os / the flat form lose
    const long long items = (long long)((M_out + 15) / 16) * (Cout / 16);
    const long long need = g_rw_min_items > 0 ? g_rw_min_items : 1024;
    return items >= need && Cin >= 32;
}

// the launch(es); arguments as conv_fwd_impl (spconv_conv.hip) has checked them
int gf_conv_rw(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask, int K, int M_in, int M_out, int ld,
               int Cin, int Cout, const float* in_scale, const float* in_shift, const float* residual, const float* out_scale,
               const float* out_shift, float* out, float* out2, hipStream_t st) {
    const int nch = Cin / 16, ncb = Cout / 16;
    const int ngroups = (M_out + 15) / 16;
    // passes over channel windows of at most three chunks, as even as possible (4 -> 2 + 2, 6 -> 3 + 3, 7 -> 3 + 2 + 2)
    const int npass = (nch + g_rw_maxch - 1) / g_rw_maxch;
    int ch0 = 0;
    for (int p = 0; p < npass; p++) {
        const int n = (nch - ch0 + (npass - p) - 1) / (npass - p);
        const bool first = p == 0, last = p == npass - 1;
        // waves: one per SIMD (two at one chunk); ranges = waves / column blocks, at most one per group
        const int waves = (n == 1 ? 2048 : 1024);
        int nranges = waves / ncb;
        if (nranges < 1) nranges = 1;
        if (nranges > ngroups) nranges = ngroups;
        if (nranges < (ngroups + 63) / 64) nranges = (ngroups + 63) / 64;  // (the kernel keeps a range's group masks one per lane)
        RwArgs a;
        a.bounds = nullptr;
        if (g_rw_bounds) {
            nranges = g_rw_nbounds;
            a.bounds = g_rw_bounds;
        }
        a.in = in + ch0 * 16;
        a.Wp = Wp;
        a.nbr = nbr;
        a.gmask = gmask;
        a.in_scale = in_scale ? in_scale + ch0 * 16 : nullptr;
        a.in_shift = in_shift ? in_shift + ch0 * 16 : nullptr;
        a.residual = first ? residual : out;
        a.out_scale = last ? out_scale : nullptr;
        a.out_shift = last ? out_shift : nullptr;
        a.out = out;
        a.out2 = last ? out2 : nullptr;
        a.in_bytes = (unsigned)((unsigned long long)M_in * Cin * 4ull - (unsigned long long)ch0 * 64ull);
        a.row_bytes = (unsigned)Cin * 4u;
        a.w_bytes = (unsigned)((size_t)K * nch * ncb * 1024);
        a.nbr_bytes = (unsigned)((size_t)K * ld * 4);
        a.nch_total = nch;
        a.ch0 = ch0;
        a.NCB = ncb;
        a.ld = ld;
        a.M_out = M_out;
        a.nranges = nranges;
        a.Cout = Cout;
        const dim3 grid((unsigned)(((long long)nranges * ncb + 3) / 4));
        const bool aff = in_scale != nullptr;
        if (K == 27) {
            if (n == 1) rw_launch<27, 1, 9>(aff, grid, st, a);
            else if (n == 2) rw_launch<27, 2, 9>(aff, grid, st, a);
            else rw_launch<27, 3, 3>(aff, grid, st, a);
        } else {
            if (n == 1) rw_launch<8, 1, 8>(aff, grid, st, a);
            else if (n == 2) rw_launch<8, 2, 8>(aff, grid, st, a);
            else rw_launch<8, 3, 8>(aff, grid, st, a);
        }
        ch0 += n;
    }
    GF_CHECK_LAUNCH("gf_conv_fwd (register-weight kernel)");
    return GF_OK;
}
