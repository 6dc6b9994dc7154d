// Shared device/host helpers for the gfx950 kernels of libgeoformer_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "geoformer_hip.h"

#define GF_WAVE 64

// step table of the counted-loop conv kernel (spconv_rules.hip k_subm3 writes it, spconv_conv.hip reads it):
// blocks of four steps per 16-row group; GF_STEP_BLKS * 4 >= 27 offsets; the first GF_STEP_PHA blocks always exist
#define GF_STEP_BLKS 7
#define GF_STEP_PHA 3
// equal-cost chunks of consecutive groups (one per wave of the pipelined level-1 conv kernel: 12 waves on each of
// the 256 compute units = 3 per SIMD, what its 131-158 registers allow; 2048 measured 3 % slower per launch in the
// forward, 3584 15 % slower); k_group_chunks writes GF_CONV_CHUNKS + 1 boundaries behind the step table
#define GF_CONV_CHUNKS 3072      // default number of chunks
#define GF_CONV_CHUNKS_MAX 4096  // capacity of the table tail: [count, boundary 0 .. boundary count]
int gf_conv_chunks();            // current setting (dev knob gf_dev_conv_chunks, spconv_conv.hip)

// ---- error plumbing: every entry point returns 0 or a negative gf_status ----
void gf_set_error(const char* fmt, ...);

#define GF_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            gf_set_error(__VA_ARGS__);   \
            return GF_ERR_INVALID_ARG;   \
        }                                \
    } while (0)

// a HIP runtime call whose failure must not pass silently (memsets / copies queued next to the launches)
#define GF_TRY(call)                                                             \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) {                                                 \
            gf_set_error("%s failed: %s", #call, hipGetErrorString(e__));        \
            return GF_ERR_LAUNCH;                                                \
        }                                                                        \
    } while (0)

#define GF_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            gf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return GF_ERR_LAUNCH;                                                    \
        }                                                                            \
    } while (0)

// levels [l_begin, l_end) of gf_rules_down2_chain (spconv_rules.hip; used by unet_exec.hip)
int gf_rules_down2_chain_range(const int32_t* coords, int M0, int B, int X, int Y, int Z, int nlevels, int l_begin,
                               int l_end, int32_t* ws, int32_t* counts, hipStream_t st);

// exclusive int32 scan (geodesic.hip): start[0..n] (start[n] = total), cursor[0..n) = start; block_sums and block_off
// hold gf_iscan_blocks(n) words each
int gf_iscan_blocks(int n);
void gf_iscan(const int32_t* v, int n, int32_t* start, int32_t* cursor, int32_t* block_sums, int32_t* block_off,
              hipStream_t st);

// the whole chain with every stage as one launch over all levels (spconv_rules.hip); gf_rules_level_parallel(): the
// default, unless GF_RULES_SERIAL=1
int gf_rules_down2_chain_all(const int32_t* coords, int M0, int B, int X, int Y, int Z, int nlevels, int32_t* ws,
                             int32_t* counts, hipStream_t st);
bool gf_rules_level_parallel();

static inline int gf_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// lane i <- lane i ^ D of a 64-lane wave WITHOUT the LDS crossbar (__shfl_xor is ds_bpermute_b32: ~120 cycles of latency per
// step, and a wave reduction is six dependent steps): v_permlane32_swap / v_permlane16_swap (gfx950) for the two steps across
// 16-lane rows, DPP row operations inside a row.  Same values as __shfl_xor(v, D, 64) -- a reduction keeps its order.
template <int D>
__device__ __forceinline__ float gf_shfl_xor(float v) {
    static_assert(D == 32 || D == 16 || D == 8 || D == 4 || D == 2 || D == 1, "gf_shfl_xor: D = 1, 2, 4, 8, 16, 32");
    const unsigned x = __float_as_uint(v);
    if constexpr (D == 32) {
        // swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second; with both = x the pair holds x[i ^ 32]
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return __uint_as_float((__lane_id() & 32) ? r[0] : r[1]);
    } else if constexpr (D == 16) {
        // swap: odd 16-lane rows of the first operand <-> even rows of the second
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return __uint_as_float((__lane_id() & 16) ? r[0] : r[1]);
    } else if constexpr (D == 8) {
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xF, 0xF, false));  // row_ror:8
    } else if constexpr (D == 4) {
        // banks 0, 2 (lanes 0-3, 8-11 of a row) read lane + 4 (row_shl:4), banks 1, 3 read lane - 4 (row_shr:4)
        const int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false));
    } else if constexpr (D == 2) {
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    } else {
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    }
}
template <int D>
__device__ __forceinline__ int gf_shfl_xor_i(int v) {  // (pure lane moves: the bit pattern travels unchanged)
    return __float_as_int(gf_shfl_xor<D>(__int_as_float(v)));
}
__device__ __forceinline__ int gf_wave_sum_i(int s) {
    s += gf_shfl_xor_i<32>(s);
    s += gf_shfl_xor_i<16>(s);
    s += gf_shfl_xor_i<8>(s);
    s += gf_shfl_xor_i<4>(s);
    s += gf_shfl_xor_i<2>(s);
    s += gf_shfl_xor_i<1>(s);
    return s;
}
__device__ __forceinline__ float gf_wave_max(float m) {
    m = fmaxf(m, gf_shfl_xor<32>(m));
    m = fmaxf(m, gf_shfl_xor<16>(m));
    m = fmaxf(m, gf_shfl_xor<8>(m));
    m = fmaxf(m, gf_shfl_xor<4>(m));
    m = fmaxf(m, gf_shfl_xor<2>(m));
    m = fmaxf(m, gf_shfl_xor<1>(m));
    return m;
}
// sum over the wave with the order of `for (d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d)`
__device__ __forceinline__ float gf_wave_sum(float s) {
    s += gf_shfl_xor<32>(s);
    s += gf_shfl_xor<16>(s);
    s += gf_shfl_xor<8>(s);
    s += gf_shfl_xor<4>(s);
    s += gf_shfl_xor<2>(s);
    s += gf_shfl_xor<1>(s);
    return s;
}

// ---- LDS-weight convolution over a flat step table (spconv_lw.hip; dispatched by gf_conv_fwd when a flat table is given) ----
// flat step table (spconv_rules.hip gf_rules_flat_steps writes it):
//   header   [0] steps S, [1] bins NB, [2] groups, [3] K, [4] rounds J = ceil(groups / NB)
//   sizes    [GF_FLAT_HIST + n] groups with n present offsets; [GF_FLAT_BSTART + n] first sorted position of that size
//            (sizes descending); [GF_FLAT_BCUR + n] the fill kernel's cursor
//   goff     [GF_FLAT_GOFF + g] first step of group g (g <= groups); behind it ppos[g]: the group's sorted position
//   desc     int4 per (round j, bin b) at gf_flat_desc_at(): {group or -1, first step, steps, offset mask}: the groups
//            sorted by size (descending) and dealt to the NB bins in snake order, so every bin (= one SIMD's waves of the
//            conv kernel) carries the same number of steps to within a few
//   steps    one 64-byte record of 16 input rows per (16-row group, present offset) in group-major, offset-ascending order,
//            GF_FLAT_PAD records of -1 behind the last
#define GF_FLAT_HIST 8
#define GF_FLAT_BSTART 40
#define GF_FLAT_BCUR 72
#define GF_FLAT_GOFF 128
#define GF_FLAT_BINS 1024  // default number of bins: four SIMDs of 256 compute units
#define GF_FLAT_PAD 32     // records of -1 behind the last step (the conv kernel's loads run ahead)
__host__ __device__ static inline size_t gf_flat_ppos_at(int ngroups) { return GF_FLAT_GOFF + (size_t)ngroups + 1; }
__host__ __device__ static inline size_t gf_flat_desc_at(int ngroups) { return (GF_FLAT_GOFF + 2 * (size_t)ngroups + 1 + 63) / 64 * 64; }
__host__ __device__ static inline size_t gf_flat_steps_at(int ngroups, int nbins) {
    return gf_flat_desc_at(ngroups) + (size_t)((ngroups + nbins - 1) / nbins + 1) * nbins * 4;
}
int gf_conv_lw_supported(int K, int M_in, int M_out, int Cin, int Cout, bool aligned, int* forced);
int gf_conv_lw(const float* in, const float* Wp, const uint32_t* gmask, const int32_t* flat, int K, int M_in, int M_out, int Cin,
               int Cout, const float* in_scale, const float* in_shift, const float* residual, const float* out_scale,
               const float* out_shift, float* out, float* out2, hipStream_t st);

// ---- dev hook: events BOUND to the next launch of an operator's main kernel (include/geoformer_hip_dev.h:
// gf_dev_op_kernel_events).  hipExtLaunchKernelGGL's start / stop events are the dispatch's own begin / end timestamps,
// i.e. what a profiler's kernel trace reports for that kernel -- no host time, no neighbouring launches inside.
// Process-wide (backward kernels are launched from the autograd thread); defined in spconv_rules.hip. ----
enum { GF_OP_BFS = 0, GF_OP_CROSS_ATTN = 1, GF_OP_MASK_HEAD = 2, GF_OP_FPS = 3, GF_OP_CROSS_ATTN_BWD = 4,
       GF_OP_MASK_HEAD_BWD_FEAT = 5, GF_OP_MASK_HEAD_BWD_PARAM = 6, GF_OP_WGRAD = 7, GF_OP_COUNT = 8 };
struct GfOpEvents {
    hipEvent_t start, stop;
    int taken;
};
GfOpEvents* gf_op_events(int op);
#define GF_LAUNCH_OP(op, kernel, grid, block, lds, st, ...)                                              \
    do {                                                                                                 \
        GfOpEvents* oe__ = gf_op_events(op);                                                             \
        if (oe__->start) {                                                                               \
            hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds), st, oe__->start, oe__->stop, 0u, __VA_ARGS__); \
            oe__->start = oe__->stop = nullptr;                                                          \
            oe__->taken = 1;                                                                             \
        } else {                                                                                         \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                               \
        }                                                                                                \
    } while (0)

// ---- occupancy-bitmap rank index (see spconv_rules.hip) ----
struct GfIndex {
    const uint32_t* bitmap;  // one bit per cell of the [B,X,Y,Z] grid
    const int32_t* prefix;   // exclusive popcount prefix per 32-bit word
    const int32_t* perm;     // rank -> row, or nullptr when rows are already in rank order
    int X, Y, Z;
};

__device__ __forceinline__ int gf_index_lookup(const GfIndex& ix, int b, int x, int y, int z) {
    if ((unsigned)x >= (unsigned)ix.X || (unsigned)y >= (unsigned)ix.Y || (unsigned)z >= (unsigned)ix.Z) return -1;
    unsigned long long lin = (((unsigned long long)b * ix.X + x) * ix.Y + y) * ix.Z + z;
    unsigned long long w = lin >> 5;
    unsigned bit = (unsigned)(lin & 31);
    uint32_t word = ix.bitmap[w];
    if (!((word >> bit) & 1u)) return -1;
    int rank = ix.prefix[w] + __popc(word & ((1u << bit) - 1u));
    return ix.perm ? ix.perm[rank] : rank;
}

// ---- block-wide exclusive scan (256 threads), shared by the rulebook and kNN-grid builders ----
#define SCAN_THREADS 256
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan over a 256-thread block; returns the exclusive prefix, *total gets the block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
    __shared__ int wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = wave_incl_scan(v);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        int s = wsum[w];
        if (w < wid) off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

// backbone_attn.hip, for unet_exec.hip: the transformer's tile tables built ahead (side stream), then the transformer without them
int gf_backbone_transformer_tables(const int* scene_offsets, int n_scenes, int M, void* scratch, void* stream);
int gf_backbone_transformer_prepared(const float* feats, const int* coords, const int* scene_offsets, int n_scenes, int M, int c,
                                     int n_layers, const float* const* params, void* scratch, float* out, void* stream);
