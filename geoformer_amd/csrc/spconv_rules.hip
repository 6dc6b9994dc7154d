// Rulebook construction for the sparse-voxel U-Net (gfx950).
//
// MI355X-first design: instead of a probing hash table (what spconv 1.0's GPU path and
// the north-star wording suggest) the voxel set of every level is indexed by an
// OCCUPANCY BITMAP + POPCOUNT RANK: one bit per grid cell, an exclusive popcount prefix per
// 32-bit word.  A lookup is one bitmap word + one prefix word (both L2-resident: 0.86 MB +
// 0.86 MB for a 299x179x128 ScanNet scene) and is collision-free, and the rank of a set bit
// IS the row index in ascending linearised (b,x,y,z) order -- which is exactly the canonical
// output order of a strided convolution, so the down-sampled
// voxel sets need no sort and no hash at all.  Level 1 (rows in voxelize first-occurrence
// order) adds a rank->row permutation.  Everything here is integer work and bit-exact
// against oracle/gf_oracle.c (orc_rules_subm3 / orc_rules_down2).
#include <stdarg.h>

#include <cstdlib>
#include "common.h"

// ------------------------------------------------------------------------------------
// error plumbing (shared by all translation units)
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void gf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gf_last_error(void) { return g_err; }

// dev hook (common.h: GF_LAUNCH_OP): events bound to the next launch of an operator's main kernel.  Process-wide, not
// per host thread: a backward kernel is launched from the framework's autograd thread, not from the thread that armed
// the hook (measurement runs are single-stepped, nothing else launches these operators meanwhile).
static GfOpEvents t_op_events[GF_OP_COUNT] = {};
GfOpEvents* gf_op_events(int op) { return &t_op_events[op]; }
extern "C" int gf_dev_op_kernel_events(int op, void* start, void* stop) {
    GF_CHECK_ARG(op >= 0 && op < GF_OP_COUNT, "gf_dev_op_kernel_events: operator %d (0..%d)", op, GF_OP_COUNT - 1);
    GF_CHECK_ARG((start == nullptr) == (stop == nullptr), "gf_dev_op_kernel_events: start and stop come together");
    t_op_events[op].start = (hipEvent_t)start;
    t_op_events[op].stop = (hipEvent_t)stop;
    t_op_events[op].taken = 0;
    return GF_OK;
}
extern "C" int gf_dev_op_kernel_events_taken(int op) {
    return op >= 0 && op < GF_OP_COUNT ? t_op_events[op].taken : 0;
}
// timing events for the hook above, owned by the caller (bench.py has no other way to hold a raw hipEvent_t)
extern "C" void* gf_dev_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        gf_set_error("gf_dev_event_create: hipEventCreate failed");
        return nullptr;
    }
    return e;
}
extern "C" int gf_dev_event_destroy(void* e) {
    if (e) GF_TRY(hipEventDestroy((hipEvent_t)e));
    return GF_OK;
}
// waits for `stop`; microseconds between the two events in *us
extern "C" int gf_dev_event_elapsed_us(void* start, void* stop, float* us) {
    GF_CHECK_ARG(start && stop && us, "gf_dev_event_elapsed_us: null argument");
    GF_TRY(hipEventSynchronize((hipEvent_t)stop));
    float ms = 0.f;
    GF_TRY(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
    *us = ms * 1e3f;
    return GF_OK;
}
extern "C" int gf_abi_version(void) { return GF_ABI_VERSION; }

// ------------------------------------------------------------------------------------
// bitmap + scan
// ------------------------------------------------------------------------------------
#define SCAN_WPT 4  // words per thread
#define SCAN_WPB (SCAN_THREADS * SCAN_WPT)

extern "C" size_t gf_index_words(int B, int X, int Y, int Z) {
    unsigned long long cells = (unsigned long long)B * X * Y * Z;
    return (size_t)((cells + 31) / 32);
}
static size_t scan_blocks(size_t words) { return (words + SCAN_WPB - 1) / SCAN_WPB; }
extern "C" size_t gf_index_scratch_bytes(size_t words) { return (2 * scan_blocks(words) + 64) * sizeof(int32_t); }

// Set bit `lin` of a bitmap from every active lane of a wave.  Rows arrive in a spatially coherent order, so
// neighbouring lanes mostly hit the same 32-bit word (a z-run of cells, or the eight children of one coarse voxel):
// the bits of each run of equal word indices are OR-ed together across the lanes and only the last lane of a run
// issues the atomic -- a wave-wide atomic whose lanes share an address is serialised lane by lane otherwise
// (k_down_bits on 142k voxels: 32 us before).  Lanes without work pass valid = false.
__device__ __forceinline__ void bitmap_set_coalesced(uint32_t* __restrict__ bitmap, unsigned long long lin, bool valid) {
    const int lane = threadIdx.x & 63;
    const unsigned long long word = valid ? (lin >> 5) : ~0ull;
    uint32_t bits = valid ? (1u << (lin & 31)) : 0u;
    const unsigned lo = (unsigned)word, hi = (unsigned)(word >> 32);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned plo = __shfl_up(lo, d, 64), phi = __shfl_up(hi, d, 64);
        const uint32_t pb = __shfl_up(bits, d, 64);
        // inside a run of equal words the partial ORs double their reach each step (segmented scan)
        if (lane >= d && plo == lo && phi == hi) bits |= pb;
    }
    const unsigned nlo = __shfl_down(lo, 1, 64), nhi = __shfl_down(hi, 1, 64);
    const bool last = lane == 63 || nlo != lo || nhi != hi;
    if (valid && last) atomicOr(&bitmap[word], bits);
}

__global__ void k_set_bits(const int32_t* __restrict__ coords, int Mcap, const int32_t* __restrict__ d_M, int X,
                           int Y, int Z, uint32_t* __restrict__ bitmap) {
    const int M = d_M ? *d_M : Mcap;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < M;
    unsigned long long lin = 0;
    if (valid) {
        int4 c = reinterpret_cast<const int4*>(coords)[i];
        lin = (((unsigned long long)c.x * X + c.y) * Y + c.z) * Z + c.w;
    }
    bitmap_set_coalesced(bitmap, lin, valid);
}

__global__ void k_block_popc(const uint32_t* __restrict__ bitmap, size_t words, int32_t* __restrict__ block_sums) {
    size_t base = (size_t)blockIdx.x * SCAN_WPB + (size_t)threadIdx.x * SCAN_WPT;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_WPT; j++)
        if (base + j < words) s += __popc(bitmap[base + j]);
    int tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums -> block_off, grand total -> *total (and *total2 if given)
__global__ void k_scan_block_sums(const int32_t* __restrict__ block_sums, int nblocks, int32_t* __restrict__ block_off,
                                  int32_t* __restrict__ total, int32_t* __restrict__ total2) {
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += SCAN_THREADS) {
        int i = base + threadIdx.x;
        int v = i < nblocks ? block_sums[i] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot);
        int carry = carry_s;
        if (i < nblocks) block_off[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (total) *total = carry_s;
        if (total2) *total2 = carry_s;
    }
}

__global__ void k_word_prefix(const uint32_t* __restrict__ bitmap, size_t words, const int32_t* __restrict__ block_off,
                              int32_t* __restrict__ prefix) {
    size_t base = (size_t)blockIdx.x * SCAN_WPB + (size_t)threadIdx.x * SCAN_WPT;
    int c[SCAN_WPT];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_WPT; j++) {
        c[j] = (base + j < words) ? __popc(bitmap[base + j]) : 0;
        s += c[j];
    }
    int tot;
    int ex = block_excl_scan(s, &tot) + block_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_WPT; j++) {
        if (base + j < words) prefix[base + j] = ex;
        ex += c[j];
    }
}

__global__ void k_perm(const int32_t* __restrict__ coords, int Mcap, const int32_t* __restrict__ d_M, GfIndex ix,
                       int32_t* __restrict__ perm) {
    const int M = d_M ? *d_M : Mcap;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    int4 c = reinterpret_cast<const int4*>(coords)[i];
    unsigned long long lin = (((unsigned long long)c.x * ix.X + c.y) * ix.Y + c.z) * ix.Z + c.w;
    uint32_t word = ix.bitmap[lin >> 5];
    int rank = ix.prefix[lin >> 5] + __popc(word & ((1u << (lin & 31)) - 1u));
    perm[rank] = i;
}

static int run_scan(uint32_t* bitmap, size_t words, int32_t* prefix, void* scratch, int32_t* d_total,
                    hipStream_t st) {
    int nb = (int)scan_blocks(words);
    int32_t* block_sums = (int32_t*)scratch;
    int32_t* block_off = block_sums + nb;
    int32_t* total = block_off + nb;
    hipLaunchKernelGGL(k_block_popc, dim3(nb), dim3(SCAN_THREADS), 0, st, bitmap, words, block_sums);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(SCAN_THREADS), 0, st, block_sums, nb, block_off, total,
                       d_total);
    hipLaunchKernelGGL(k_word_prefix, dim3(nb), dim3(SCAN_THREADS), 0, st, bitmap, words, block_off, prefix);
    return 0;
}

extern "C" int gf_index_build(const int32_t* coords, int M, const int32_t* d_M, int B, int X, int Y, int Z,
                              uint32_t* bitmap, int32_t* prefix, int32_t* perm, void* scratch, void* stream) {
    GF_CHECK_ARG(M >= 0 && B > 0 && X > 0 && Y > 0 && Z > 0, "gf_index_build: bad sizes M=%d B=%d shape=%dx%dx%d", M,
                 B, X, Y, Z);
    size_t words = gf_index_words(B, X, Y, Z);
    GF_CHECK_ARG(words < (1ull << 31), "gf_index_build: grid of %zu words is too large for the bitmap index", words);
    hipStream_t st = (hipStream_t)stream;
    GF_TRY(hipMemsetAsync(bitmap, 0, words * sizeof(uint32_t), st));
    if (M > 0)
        hipLaunchKernelGGL(k_set_bits, dim3(gf_div_up(M, 256)), dim3(256), 0, st, coords, M, d_M, X, Y, Z, bitmap);
    run_scan(bitmap, words, prefix, scratch, nullptr, st);
    if (perm && M > 0) {
        GfIndex ix{bitmap, prefix, nullptr, X, Y, Z};
        hipLaunchKernelGGL(k_perm, dim3(gf_div_up(M, 256)), dim3(256), 0, st, coords, M, d_M, ix, perm);
    }
    GF_CHECK_LAUNCH("gf_index_build");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// submanifold 3x3x3: one thread per output row, 27 index lookups, coalesced column writes
// ------------------------------------------------------------------------------------
// `steps` (optional) is the same relation laid out for the counted-loop conv kernel (spconv_conv.hip, k_conv_g16):
// per 16-row group the PRESENT offsets only, in ascending offset order, as blocks of four steps
//     steps[((g * GF_STEP_BLKS + s / 4) * 16 + row) * 4 + s % 4] = input row of (row, s-th present offset) or -1
// so that lane (row, *) fetches four steps with one 16-byte load and the 16 rows of a block are one contiguous
// 256-byte segment.  The first GF_STEP_PHA blocks of a group are always written (padded with -1): the kernel loads
// them before it knows the group's mask.
__global__ void k_subm3(const int32_t* __restrict__ coords, int Mcap, const int32_t* __restrict__ d_M, GfIndex ix,
                        int32_t* __restrict__ nbr, int ld, uint32_t* __restrict__ gmask, int32_t* __restrict__ steps) {
    const int M = d_M ? *d_M : Mcap;
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mask = 0;
    int rk[27];
#pragma unroll
    for (int k = 0; k < 27; k++) rk[k] = -1;
    if (o < M) {
        int4 c = reinterpret_cast<const int4*>(coords)[o];
#pragma unroll
        for (int k = 0; k < 27; k++) {
            const int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;
            int r = (k == 13) ? o : gf_index_lookup(ix, c.x, c.y + dx, c.z + dy, c.w + dz);
            rk[k] = r;
            if (nbr) nbr[(size_t)k * ld + o] = r;
            if (r >= 0) mask |= 1u << k;
        }
    } else if (o < ld && nbr) {
#pragma unroll
        for (int k = 0; k < 27; k++) nbr[(size_t)k * ld + o] = -1;
    }
    // OR over each 16-row group (16 consecutive lanes)
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) mask |= __shfl_xor(mask, d, 64);
    if ((threadIdx.x & 15) == 0 && o < ld) gmask[o >> 4] = mask;
    if (steps && o < ld) {
        int32_t* rec = steps + ((size_t)(o >> 4) * GF_STEP_BLKS * 16 + (o & 15)) * 4;
        int s = 0;
#pragma unroll
        for (int k = 0; k < 27; k++) {
            if ((mask >> k) & 1u) {
                rec[(s >> 2) * 64 + (s & 3)] = rk[k];
                s++;
            }
        }
        const int n = __popc(mask);
        int end = (n + 3) & ~3;
        if (end < 4 * GF_STEP_PHA) end = 4 * GF_STEP_PHA;
        for (s = n; s < end; s++) rec[(s >> 2) * 64 + (s & 3)] = -1;
    }
}

// Equal-cost chunks of consecutive 16-row groups for the pipelined conv kernel (k_conv_g16p): chunk c covers groups
// [chunks[c], chunks[c + 1]) and every chunk carries about the same number of steps (+ a fixed cost per group), so
// one wave per chunk keeps all SIMDs busy for the same time without any dynamic scheduling; consecutive groups
// stay together (they share neighbour rows in L1/L2).  One workgroup of 1024 threads: block scan of the costs,
// then every group writes the chunk boundaries that fall into its cost interval.
#define CHUNK_GROUP_COST 3
// cost of a group in steps: its present offsets + a fixed part + the two exposed round trips of a group whose offsets
// do not fit the pipelined part of the kernel
__device__ __forceinline__ int chunk_cost(uint32_t mask) {
    const int n = __popc(mask);
    return n + CHUNK_GROUP_COST + (n > 4 * GF_STEP_PHA ? 8 : 0);
}
__global__ __launch_bounds__(1024) void k_group_chunks(const uint32_t* __restrict__ gmask, int ngroups, int nchunks,
                                                       int32_t* __restrict__ chunks) {
    __shared__ long long s_wave[16];
    __shared__ long long s_total;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (ngroups + 1023) / 1024;
    const int lo = min(ngroups, t * per), hi = min(ngroups, lo + per);
    long long sum = 0;
    for (int g = lo; g < hi; g++) sum += chunk_cost(gmask[g]);
    long long inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (t == 0) {
        long long acc = 0;
        for (int i = 0; i < 16; i++) {
            const long long v = s_wave[i];
            s_wave[i] = acc;
            acc += v;
        }
        s_total = acc;
    }
    __syncthreads();
    const long long C = s_total > 0 ? s_total : 1;
    long long before = s_wave[wv] + inc - sum;  // cost of all groups in front of this thread's first one
    for (int g = lo; g < hi; g++) {
        // chunk c starts at the first group whose preceding cost reaches c * C / nchunks
        const long long c_hi = before * nchunks / C;
        long long c_lo = 0;
        if (g > 0) {
            const long long prev = before - chunk_cost(gmask[g - 1]);
            c_lo = prev * nchunks / C + 1;
        }
        for (long long c = c_lo; c <= c_hi && c < nchunks; c++) chunks[c] = g;
        before += chunk_cost(gmask[g]);
    }
    if (t == 0) {
        // boundaries behind the last group's start (and everything when there are no groups) end the table
        const long long last = ngroups > 0 ? (C - chunk_cost(gmask[ngroups - 1])) * nchunks / C + 1 : 0;
        for (long long c = last; c <= nchunks; c++) chunks[c] = ngroups;
        chunks[-1] = nchunks;  // the count sits in front of the boundaries
    }
}

extern "C" size_t gf_rules_steps_words(int ld) { return (size_t)(ld / 16) * GF_STEP_BLKS * 64 + GF_CONV_CHUNKS_MAX + 16; }

extern "C" int gf_rules_subm3(const int32_t* coords, int M, const int32_t* d_M, int X, int Y, int Z,
                              const uint32_t* bitmap, const int32_t* prefix, const int32_t* perm, int32_t* nbr, int ld,
                              uint32_t* gmask, int32_t* steps, void* stream) {
    GF_CHECK_ARG(ld >= M && (ld % 16) == 0, "gf_rules_subm3: ld=%d must be a multiple of 16 and >= M=%d", ld, M);
    GF_CHECK_ARG(nbr != nullptr || steps != nullptr, "gf_rules_subm3: neither table requested");
    if (ld == 0) return GF_OK;
    GfIndex ix{bitmap, prefix, perm, X, Y, Z};
    hipLaunchKernelGGL(k_subm3, dim3(gf_div_up(ld, 256)), dim3(256), 0, (hipStream_t)stream, coords, M, d_M, ix, nbr,
                       ld, gmask, steps);
    if (steps)  // chunk boundaries live behind the step blocks (M known on the host here; groups = ceil(M / 16))
        hipLaunchKernelGGL(k_group_chunks, dim3(1), dim3(1024), 0, (hipStream_t)stream, gmask, (M + 15) / 16,
                           gf_conv_chunks(), steps + (size_t)(ld / 16) * GF_STEP_BLKS * 64 + 1);
    GF_CHECK_LAUNCH("gf_rules_subm3");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// Flat step table of a [K, ld] relation for the LDS-weight conv kernel (spconv_lw.hip, k_conv_lw): one 64-byte record
// of 16 input rows per (16-row group, PRESENT offset), group-major and offset-ascending -- the order the kernel's
// consumer walks with its mask iterator, so its gathers and index loads are linear in the step number and can run any
// distance ahead of the MFMAs -- plus the groups SORTED BY SIZE and dealt to NB bins in snake order: a level of a scanned
// room has ~3 groups of 6-27 steps per SIMD, and contiguous equal-cost chunks of whole groups left the slowest SIMD with
// 1.7x the mean (profiles/r6_conv_lw_notes.md).  Layout: common.h (GF_FLAT_*).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_flat_scan(const uint32_t* __restrict__ gmask, int ngroups, int nbins, int K,
                                                    int32_t* __restrict__ flat) {
    __shared__ int s_wave[16];
    __shared__ int s_cnt[16][32];   // groups of size n seen by wave w (its lanes' ranks inside the size come from here)
    __shared__ int s_base[16][32];  // sorted position of wave w's first group of size n
    int32_t* goff = flat + GF_FLAT_GOFF;
    int32_t* ppos = flat + gf_flat_ppos_at(ngroups);
    const uint32_t kmask = K >= 32 ? 0xffffffffu : ((1u << K) - 1u);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t < 512) (&s_cnt[0][0])[t] = 0;
    __syncthreads();
    const int per = (ngroups + 1023) / 1024;
    const int lo = min(ngroups, t * per), hi = min(ngroups, lo + per);
    int sum = 0;
    for (int g = lo; g < hi; g++) {
        const int n = __popc(gmask[g] & kmask);
        sum += n;
        // rank among the wave's groups of this size (LDS atomic on a wave-private counter: no cross-wave contention; the
        // same ranking with one global atomic per group took 35-70 us on 9000 groups -- ~20 addresses, serialised at L2)
        ppos[g] = atomicAdd(&s_cnt[wv][n & 31], 1);
    }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (t < 32) {  // thread n: the sizes' totals over the waves
        int c = 0;
        for (int w = 0; w < 16; w++) {
            s_base[w][t] = c;
            c += s_cnt[w][t];
        }
        flat[GF_FLAT_HIST + t] = c;
        s_cnt[0][t] = c;  // (the counts are no longer needed: row 0 carries the totals to thread 0)
    }
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int i = 0; i < 16; i++) {
            const int v = s_wave[i];
            s_wave[i] = acc;
            acc += v;
        }
        goff[ngroups] = acc;
        flat[0] = acc;
        flat[1] = nbins;
        flat[2] = ngroups;
        flat[3] = K;
        flat[4] = (ngroups + nbins - 1) / nbins;
        int pos = 0;
        for (int n = 31; n >= 0; n--) {  // sorted positions: sizes descending
            flat[GF_FLAT_BSTART + n] = pos;
            flat[GF_FLAT_BCUR + n] = 0;
            const int c = s_cnt[0][n];
            s_cnt[0][n] = pos;
            pos += c;
        }
    }
    __syncthreads();
    int before = s_wave[wv] + inc - sum;  // steps of all groups in front of this thread's first one
    for (int g = lo; g < hi; g++) {
        goff[g] = before;
        const int n = __popc(gmask[g] & kmask);
        before += n;
        ppos[g] += s_cnt[0][n & 31] + s_base[wv][n & 31];
    }
    // slots of the last round that no group takes
    int32_t* desc = flat + gf_flat_desc_at(ngroups);
    const int rounds = (ngroups + nbins - 1) / nbins;
    for (int i = t; i < nbins; i += 1024) {
        if (rounds > 0) reinterpret_cast<int4*>(desc)[(size_t)(rounds - 1) * nbins + i] = make_int4(-1, 0, 0, 0);
        reinterpret_cast<int4*>(desc)[(size_t)rounds * nbins + i] = make_int4(-1, 0, 0, 0);
    }
}

// one thread per (group, row): the group's present offsets in ascending order; row 0 also files the group's descriptor
// under its sorted position
__global__ void k_flat_fill(const int32_t* __restrict__ nbr, const uint32_t* __restrict__ gmask, int ngroups, int nbins, int M,
                            int ld, int K, int32_t* __restrict__ flat) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t >> 4, r = t & 15;
    if (g >= ngroups) return;
    const int32_t* goff = flat + GF_FLAT_GOFF;
    int32_t* steps = flat + gf_flat_steps_at(ngroups, nbins);
    const uint32_t kmask = K >= 32 ? 0xffffffffu : ((1u << K) - 1u);
    const uint32_t mask = gmask[g] & kmask;
    uint32_t m = mask;
    const int s0 = goff[g];
    int s = s0;
    const int o = g * 16 + r;
    while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1;
        steps[(size_t)s * 16 + r] = o < M ? nbr[(size_t)k * ld + o] : -1;
        s++;
    }
    if (g == ngroups - 1)
        for (int p = 0; p < GF_FLAT_PAD; p++) steps[(size_t)(s + p) * 16 + r] = -1;
    if (r == 0) {
        const int p = flat[gf_flat_ppos_at(ngroups) + g];
        const int j = p / nbins, x = p - j * nbins;
        const int b = (j & 1) ? nbins - 1 - x : x;
        int32_t* desc = flat + gf_flat_desc_at(ngroups);
        reinterpret_cast<int4*>(desc)[(size_t)j * nbins + b] = make_int4(g, s0, s - s0, (int)mask);
    }
}

extern "C" size_t gf_rules_flat_words(int K, int ld) {
    const int ngroups = ld / 16;
    return gf_flat_steps_at(ngroups, GF_FLAT_BINS) + ((size_t)K * ngroups + GF_FLAT_PAD) * 16;
}

extern "C" int gf_rules_flat_steps(const int32_t* nbr, const uint32_t* gmask, int K, int M, int ld, int nbins,
                                   int32_t* flat, void* stream) {
    GF_CHECK_ARG(nbr && gmask && flat, "gf_rules_flat_steps: null argument");
    GF_CHECK_ARG(K >= 1 && K <= 31 && M >= 0 && ld >= M && (ld % 16) == 0, "gf_rules_flat_steps: K=%d M=%d ld=%d", K, M, ld);
    if (nbins <= 0) nbins = GF_FLAT_BINS;
    GF_CHECK_ARG(nbins % 4 == 0 && nbins <= GF_FLAT_BINS, "gf_rules_flat_steps: %d bins (a multiple of 4, at most %d)", nbins,
                 GF_FLAT_BINS);
    const int ngroups = (M + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_flat_scan, dim3(1), dim3(1024), 0, st, gmask, ngroups, nbins, K, flat);
    if (ngroups > 0)
        hipLaunchKernelGGL(k_flat_fill, dim3(gf_div_up((long long)ngroups * 16, 256)), dim3(256), 0, st, nbr, gmask, ngroups, nbins,
                           M, ld, K, flat);
    GF_CHECK_LAUNCH("gf_rules_flat_steps");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// strided 2x2x2 / stride 2
// ------------------------------------------------------------------------------------
__global__ void k_down_bits(const int32_t* __restrict__ coords, int Mcap, const int32_t* __restrict__ d_M, int OX,
                            int OY, int OZ, uint32_t* __restrict__ bitmap) {
    const int M = d_M ? *d_M : Mcap;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = i < M;
    unsigned long long lin = 0;
    if (valid) {
        int4 c = reinterpret_cast<const int4*>(coords)[i];
        int ox = c.y >> 1, oy = c.z >> 1, oz = c.w >> 1;
        valid = ox < OX && oy < OY && oz < OZ;  // dropped otherwise: candidate output outside out_shape
        lin = (((unsigned long long)c.x * OX + ox) * OY + oy) * OZ + oz;
    }
    bitmap_set_coalesced(bitmap, lin, valid);
}

__global__ void k_down_fill(const int32_t* __restrict__ coords, int Mcap, const int32_t* __restrict__ d_M, GfIndex ox_,
                            int32_t* __restrict__ out_coords, int32_t* __restrict__ child, int ld,
                            int32_t* __restrict__ parent, int32_t* __restrict__ koff, int32_t* __restrict__ up,
                            int ld_up, uint32_t* __restrict__ gmask_up) {
    const int M = d_M ? *d_M : Mcap;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mask = 0;
    if (i < M) {
        int4 c = reinterpret_cast<const int4*>(coords)[i];
        int ox = c.y >> 1, oy = c.z >> 1, oz = c.w >> 1;
        int k = ((c.y & 1) * 2 + (c.z & 1)) * 2 + (c.w & 1);
        int r = gf_index_lookup(ox_, c.x, ox, oy, oz);
        parent[i] = r;
        koff[i] = k;
        if (r >= 0) {
            child[(size_t)k * ld + r] = i;
            reinterpret_cast<int4*>(out_coords)[r] = make_int4(c.x, ox, oy, oz);  // same value from every child
            mask = 1u << k;
        }
#pragma unroll
        for (int kk = 0; kk < 8; kk++) up[(size_t)kk * ld_up + i] = (kk == k) ? r : -1;
    } else if (i < ld_up) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) up[(size_t)kk * ld_up + i] = -1;
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) mask |= __shfl_xor(mask, d, 64);
    if ((threadIdx.x & 15) == 0 && i < ld_up) gmask_up[i >> 4] = mask;
}

// group masks of a [K,ld] table (one thread per output row)
__global__ void k_table_gmask(const int32_t* __restrict__ tbl, int K, int ld, uint32_t* __restrict__ gmask) {
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mask = 0;
    if (o < ld)
        for (int k = 0; k < K; k++)
            if (tbl[(size_t)k * ld + o] >= 0) mask |= 1u << k;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) mask |= __shfl_xor(mask, d, 64);
    if ((threadIdx.x & 15) == 0 && o < ld) gmask[o >> 4] = mask;
}

extern "C" int gf_rules_down2(const int32_t* coords, int M, const int32_t* d_M, int B, int X, int Y, int Z,
                              uint32_t* bitmap_out, int32_t* prefix_out, void* scratch, int32_t* out_coords,
                              int32_t* d_M_out, int32_t* child, int ld, int32_t* parent, int32_t* koff, int32_t* up,
                              int ld_up, uint32_t* gmask_down, uint32_t* gmask_up, void* stream) {
    GF_CHECK_ARG(X >= 2 && Y >= 2 && Z >= 2, "gf_rules_down2: spatial shape %dx%dx%d too small for k=2,s=2", X, Y, Z);
    GF_CHECK_ARG((ld % 16) == 0 && (ld_up % 16) == 0 && ld_up >= M, "gf_rules_down2: bad leading dims ld=%d ld_up=%d",
                 ld, ld_up);
    const int OX = (X - 2) / 2 + 1, OY = (Y - 2) / 2 + 1, OZ = (Z - 2) / 2 + 1;
    size_t words = gf_index_words(B, OX, OY, OZ);
    GF_CHECK_ARG(words < (1ull << 31), "gf_rules_down2: output grid too large");
    hipStream_t st = (hipStream_t)stream;
    GF_TRY(hipMemsetAsync(bitmap_out, 0, words * sizeof(uint32_t), st));
    GF_TRY(hipMemsetAsync(child, 0xff, (size_t)8 * ld * sizeof(int32_t), st));
    if (M > 0)
        hipLaunchKernelGGL(k_down_bits, dim3(gf_div_up(M, 256)), dim3(256), 0, st, coords, M, d_M, OX, OY, OZ,
                           bitmap_out);
    run_scan(bitmap_out, words, prefix_out, scratch, d_M_out, st);
    if (ld_up > 0) {
        GfIndex oix{bitmap_out, prefix_out, nullptr, OX, OY, OZ};
        hipLaunchKernelGGL(k_down_fill, dim3(gf_div_up(ld_up, 256)), dim3(256), 0, st, coords, M, d_M, oix, out_coords,
                           child, ld, parent, koff, up, ld_up, gmask_up);
    }
    if (ld > 0)
        hipLaunchKernelGGL(k_table_gmask, dim3(gf_div_up(ld, 256)), dim3(256), 0, st, child, 8, ld, gmask_down);
    GF_CHECK_LAUNCH("gf_rules_down2");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// The whole chain of k=2/s=2 down-sampling rulebooks of the U-Net in one call.
//
// All tables live in one caller-provided workspace (bitmaps first, child tables next, so two memsets clear them for
// every level at once instead of two per level), and a level's tables are sized by a host-known bound of its voxel
// count: min(bound of the level above, grid cells).  (Walking the tail of small levels in ONE single-workgroup
// kernel, with workgroup barriers in place of kernel boundaries, was tried: 165 us against ~50 us for the two dozen
// 1-3 us launches it replaces -- a lone workgroup pays a memory round trip per loop iteration.)
// ------------------------------------------------------------------------------------
#define DOWN_MAX_LEVELS 8
#define DOWN_FIELDS 10  // bitmap prefix scratch out_coords child parent koff up gmask_down gmask_up

struct DownPlan {
    int nl;
    int shape[DOWN_MAX_LEVELS + 1][3];
    int cap[DOWN_MAX_LEVELS + 1];
    size_t words[DOWN_MAX_LEVELS];
    long long off[DOWN_MAX_LEVELS][DOWN_FIELDS];  // int32-element offsets into the workspace
    long long bitmaps_begin, bitmaps_end, child_begin, child_end, total;
    // (for the level-parallel build: the one-hot `up` tables of the levels below the first start as -1 everywhere and
    //  their group masks as 0, so both sit next to the blocks the two memsets clear anyway)
    long long zero_end, ff_end;
};

static inline long long pad64(long long n) { return (n + 63) / 64 * 64; }

static int plan_down_chain(int M0, int B, int X, int Y, int Z, int nlevels, DownPlan& P) {
    P.nl = 0;
    P.shape[0][0] = X; P.shape[0][1] = Y; P.shape[0][2] = Z;
    P.cap[0] = M0 < 16 ? 16 : (M0 + 15) / 16 * 16;
    for (int l = 0; l < nlevels && l < DOWN_MAX_LEVELS; l++) {
        const int x = P.shape[l][0], y = P.shape[l][1], z = P.shape[l][2];
        if (x < 2 || y < 2 || z < 2) break;
        const int ox = (x - 2) / 2 + 1, oy = (y - 2) / 2 + 1, oz = (z - 2) / 2 + 1;
        P.shape[l + 1][0] = ox; P.shape[l + 1][1] = oy; P.shape[l + 1][2] = oz;
        const unsigned long long cells = (unsigned long long)B * ox * oy * oz;
        unsigned long long c = (cells + 15) / 16 * 16;
        if (c > (unsigned long long)P.cap[l]) c = P.cap[l];
        if (c < 16) c = 16;
        P.cap[l + 1] = (int)c;
        P.words[l] = gf_index_words(B, ox, oy, oz);
        if (P.words[l] >= (1ull << 31)) return -1;
        P.nl = l + 1;
    }
    long long cur = 0;
    P.bitmaps_begin = cur;
    for (int l = 0; l < P.nl; l++) { P.off[l][0] = cur; cur += pad64((long long)P.words[l]); }
    P.bitmaps_end = cur;
    for (int l = 1; l < P.nl; l++) { P.off[l][9] = cur; cur += pad64(P.cap[l] / 16); }
    P.zero_end = cur;
    P.child_begin = cur;
    for (int l = 0; l < P.nl; l++) { P.off[l][4] = cur; cur += pad64(8ll * P.cap[l + 1]); }
    P.child_end = cur;
    for (int l = 1; l < P.nl; l++) { P.off[l][7] = cur; cur += pad64(8ll * P.cap[l]); }
    P.ff_end = cur;
    for (int l = 0; l < P.nl; l++) {
        const long long ci = P.cap[l], co = P.cap[l + 1];
        const long long sz[DOWN_FIELDS] = {0, (long long)P.words[l], (long long)(gf_index_scratch_bytes(P.words[l]) + 3) / 4,
                                           4 * co, 0, ci, ci, 8 * ci, co / 16, ci / 16};
        for (int f = 0; f < DOWN_FIELDS; f++) {
            if (f == 0 || f == 4 || (l >= 1 && (f == 7 || f == 9))) continue;
            P.off[l][f] = cur;
            cur += pad64(sz[f]);
        }
    }
    P.total = cur;
    return 0;
}

extern "C" int gf_rules_down2_chain_plan(int M0, int B, int X, int Y, int Z, int nlevels, long long* offsets,
                                         int* caps, int* shapes, long long* ws_elems, int* nlevels_out) {
    GF_CHECK_ARG(offsets && caps && shapes && ws_elems && nlevels_out, "gf_rules_down2_chain_plan: null argument");
    GF_CHECK_ARG(M0 >= 0 && B > 0 && nlevels >= 0 && nlevels <= DOWN_MAX_LEVELS,
                 "gf_rules_down2_chain_plan: M0=%d B=%d nlevels=%d (at most %d levels)", M0, B, nlevels, DOWN_MAX_LEVELS);
    DownPlan P;
    GF_CHECK_ARG(plan_down_chain(M0, B, X, Y, Z, nlevels, P) == 0, "gf_rules_down2_chain_plan: grid too large");
    for (int l = 0; l <= P.nl; l++) {
        caps[l] = P.cap[l];
        for (int a = 0; a < 3; a++) shapes[3 * l + a] = P.shape[l][a];
    }
    for (int l = 0; l < P.nl; l++)
        for (int f = 0; f < DOWN_FIELDS; f++) offsets[l * DOWN_FIELDS + f] = P.off[l][f];
    *ws_elems = P.total;
    *nlevels_out = P.nl;
    return GF_OK;
}

// levels [l_begin, l_end) of the chain (internal: gf_unet_fwd issues the first level, whose tables and count the
// convolutions need first, ahead of the rest); the memsets that clear every level's bitmaps and child tables go
// with level 0
int gf_rules_down2_chain_range(const int32_t* coords, int M0, int B, int X, int Y, int Z, int nlevels, int l_begin,
                               int l_end, int32_t* ws, int32_t* counts, hipStream_t st) {
    GF_CHECK_ARG(coords && ws && counts, "gf_rules_down2_chain: null argument");
    GF_CHECK_ARG(M0 >= 0 && B > 0 && nlevels >= 0 && nlevels <= DOWN_MAX_LEVELS, "gf_rules_down2_chain: bad sizes");
    DownPlan P;
    GF_CHECK_ARG(plan_down_chain(M0, B, X, Y, Z, nlevels, P) == 0, "gf_rules_down2_chain: grid too large");
    if (P.nl == 0 || M0 == 0) return GF_OK;
    if (l_end > P.nl) l_end = P.nl;
    if (l_begin == 0) {
        GF_TRY(hipMemsetAsync(ws + P.bitmaps_begin, 0, (size_t)(P.bitmaps_end - P.bitmaps_begin) * 4, st));
        GF_TRY(hipMemsetAsync(ws + P.child_begin, 0xff, (size_t)(P.child_end - P.child_begin) * 4, st));
    }
    for (int l = l_begin; l < l_end; l++) {
        const int32_t* cur = l ? ws + P.off[l - 1][3] : coords;
        int32_t* F[DOWN_FIELDS];
        for (int f = 0; f < DOWN_FIELDS; f++) F[f] = ws + P.off[l][f];
        const int Mcap = l ? P.cap[l] : M0;
        const int32_t* d_M = l ? counts + l : nullptr;
        const int OX = P.shape[l + 1][0], OY = P.shape[l + 1][1], OZ = P.shape[l + 1][2];
        const int ld = P.cap[l + 1], ld_up = P.cap[l];
        if (Mcap > 0)
            hipLaunchKernelGGL(k_down_bits, dim3(gf_div_up(Mcap, 256)), dim3(256), 0, st, cur, Mcap, d_M, OX, OY, OZ,
                               (uint32_t*)F[0]);
        run_scan((uint32_t*)F[0], P.words[l], F[1], F[2], counts + l + 1, st);
        GfIndex oix{(uint32_t*)F[0], F[1], nullptr, OX, OY, OZ};
        hipLaunchKernelGGL(k_down_fill, dim3(gf_div_up(ld_up, 256)), dim3(256), 0, st, cur, Mcap, d_M, oix, F[3], F[4],
                           ld, F[5], F[6], F[7], ld_up, (uint32_t*)F[9]);
        hipLaunchKernelGGL(k_table_gmask, dim3(gf_div_up(ld, 256)), dim3(256), 0, st, F[4], 8, ld, (uint32_t*)F[8]);
    }
    GF_CHECK_LAUNCH("gf_rules_down2_chain");
    return GF_OK;
}

// ------------------------------------------------------------------------------------
// The same chain with every STAGE as one launch over all levels.  A coarse level's voxel set is a function of the first
// level's coordinates alone (halve l + 1 times, dropping a voxel at the first level whose cropped grid it leaves), and a
// level's rows are the set bits of the level above's bitmap in rank order -- so the bitmaps of all levels are set from
// the first level's rows, scanned side by side, and every level's tables are filled by enumerating the bits of its
// input bitmap: 6 launches instead of 6 per level (36 dependent launches of 2-9 us each held the second level's
// convolutions back by ~260 us per forward: profiles/r4_b_forward_full_timeline.txt).  Same tables bit for bit
// (tests/test_gpu_spconv.py::test_down_rules_chain_bit_exact).
// ------------------------------------------------------------------------------------
struct DownAll {
    int nl, B, M0;
    int shape[DOWN_MAX_LEVELS + 1][3];
    int cap[DOWN_MAX_LEVELS + 1];
    unsigned words[DOWN_MAX_LEVELS];
    int32_t* f[DOWN_MAX_LEVELS][DOWN_FIELDS];
    int scan_begin[DOWN_MAX_LEVELS + 1];  // blocks of the popcount / prefix launches, per level
    int fill_begin[DOWN_MAX_LEVELS + 1];  // blocks of the fill launch
    int gm_begin[DOWN_MAX_LEVELS + 1];    // blocks of the group-mask launch
    const int32_t* coords0;
    int32_t* counts;
};

__global__ void k_down_bits_all(DownAll A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = i < A.M0;
    int b = 0, x = 0, y = 0, z = 0;
    if (valid) {
        const int4 c = reinterpret_cast<const int4*>(A.coords0)[i];
        b = c.x; x = c.y; y = c.z; z = c.w;
    }
    for (int l = 0; l < A.nl; l++) {
        x >>= 1; y >>= 1; z >>= 1;
        const int OX = A.shape[l + 1][0], OY = A.shape[l + 1][1], OZ = A.shape[l + 1][2];
        valid = valid && x < OX && y < OY && z < OZ;  // dropped for good: candidate output outside out_shape
        const unsigned long long lin = valid ? (((unsigned long long)b * OX + x) * OY + y) * OZ + z : 0ull;
        bitmap_set_coalesced((uint32_t*)A.f[l][0], lin, valid);
    }
}

__device__ __forceinline__ int down_all_level(const int* begin, int nl, int blk) {
    int l = 0;
    while (l + 1 < nl && blk >= begin[l + 1]) l++;
    return l;
}

__global__ void k_block_popc_all(DownAll A) {
    const int l = down_all_level(A.scan_begin, A.nl, blockIdx.x);
    const uint32_t* bitmap = (const uint32_t*)A.f[l][0];
    const size_t words = A.words[l];
    int32_t* block_sums = A.f[l][2];
    const int blk = blockIdx.x - A.scan_begin[l];
    size_t base = (size_t)blk * SCAN_WPB + (size_t)threadIdx.x * SCAN_WPT;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_WPT; j++)
        if (base + j < words) s += __popc(bitmap[base + j]);
    int tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blk] = tot;
}

// one block per level: exclusive scan of that level's block sums; the total is the next level's voxel count
__global__ void k_scan_block_sums_all(DownAll A) {
    __shared__ int carry_s;
    const int l = blockIdx.x;
    const int nblocks = A.scan_begin[l + 1] - A.scan_begin[l];
    const int32_t* block_sums = A.f[l][2];
    int32_t* block_off = A.f[l][2] + nblocks;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += SCAN_THREADS) {
        int i = base + threadIdx.x;
        int v = i < nblocks ? block_sums[i] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot);
        int carry = carry_s;
        if (i < nblocks) block_off[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        block_off[nblocks] = carry_s;  // (the slot run_scan keeps the total in)
        A.counts[l + 1] = carry_s;
    }
}

__global__ void k_word_prefix_all(DownAll A) {
    const int l = down_all_level(A.scan_begin, A.nl, blockIdx.x);
    const uint32_t* bitmap = (const uint32_t*)A.f[l][0];
    const size_t words = A.words[l];
    const int nblocks = A.scan_begin[l + 1] - A.scan_begin[l];
    const int32_t* block_off = A.f[l][2] + nblocks;
    int32_t* prefix = A.f[l][1];
    const int blk = blockIdx.x - A.scan_begin[l];
    size_t base = (size_t)blk * SCAN_WPB + (size_t)threadIdx.x * SCAN_WPT;
    int c[SCAN_WPT];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_WPT; j++) {
        c[j] = (base + j < words) ? __popc(bitmap[base + j]) : 0;
        s += c[j];
    }
    int tot;
    int ex = block_excl_scan(s, &tot) + block_off[blk];
#pragma unroll
    for (int j = 0; j < SCAN_WPT; j++) {
        if (base + j < words) prefix[base + j] = ex;
        ex += c[j];
    }
}

// what k_down_fill writes for input row i of level l with coordinates (b, x, y, z); `up` rows start as -1 and
// gmask_up as 0 for the levels below the first (memsets), so only the one live entry / bit is written there
__device__ __forceinline__ void down_fill_row(const DownAll& A, int l, int i, int b, int x, int y, int z) {
    const int ox = x >> 1, oy = y >> 1, oz = z >> 1;
    const int k = ((x & 1) * 2 + (y & 1)) * 2 + (z & 1);
    const GfIndex oix{(const uint32_t*)A.f[l][0], A.f[l][1], nullptr, A.shape[l + 1][0], A.shape[l + 1][1], A.shape[l + 1][2]};
    const int r = gf_index_lookup(oix, b, ox, oy, oz);
    const int ld = A.cap[l + 1], ld_up = A.cap[l];
    A.f[l][5][i] = r;
    A.f[l][6][i] = k;
    if (r >= 0) {
        A.f[l][4][(size_t)k * ld + r] = i;
        reinterpret_cast<int4*>(A.f[l][3])[r] = make_int4(b, ox, oy, oz);  // same value from every child
        A.f[l][7][(size_t)k * ld_up + i] = r;
        atomicOr((uint32_t*)A.f[l][9] + (i >> 4), 1u << k);
    }
}

__global__ void k_down_fill_all(DownAll A) {
    const int l = down_all_level(A.fill_begin, A.nl, blockIdx.x);
    const int blk = blockIdx.x - A.fill_begin[l];
    if (l == 0) {
        // the first level's rows come in the caller's order: one thread per row, as k_down_fill
        const int i = blk * blockDim.x + threadIdx.x;
        const int ld = A.cap[1], ld_up = A.cap[0];
        uint32_t mask = 0;
        if (i < A.M0) {
            const int4 c = reinterpret_cast<const int4*>(A.coords0)[i];
            const int ox = c.y >> 1, oy = c.z >> 1, oz = c.w >> 1;
            const int k = ((c.y & 1) * 2 + (c.z & 1)) * 2 + (c.w & 1);
            const GfIndex oix{(const uint32_t*)A.f[0][0], A.f[0][1], nullptr, A.shape[1][0], A.shape[1][1], A.shape[1][2]};
            const int r = gf_index_lookup(oix, c.x, ox, oy, oz);
            A.f[0][5][i] = r;
            A.f[0][6][i] = k;
            if (r >= 0) {
                A.f[0][4][(size_t)k * ld + r] = i;
                reinterpret_cast<int4*>(A.f[0][3])[r] = make_int4(c.x, ox, oy, oz);
                mask = 1u << k;
            }
#pragma unroll
            for (int kk = 0; kk < 8; kk++) A.f[0][7][(size_t)kk * ld_up + i] = (kk == k) ? r : -1;
        } else if (i < ld_up) {
#pragma unroll
            for (int kk = 0; kk < 8; kk++) A.f[0][7][(size_t)kk * ld_up + i] = -1;
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) mask |= __shfl_xor(mask, d, 64);
        if ((threadIdx.x & 15) == 0 && i < ld_up) ((uint32_t*)A.f[0][9])[i >> 4] = mask;
        return;
    }
    // a level below: its rows are the set bits of the level above's output bitmap, in rank order; one thread per word
    const unsigned w = (unsigned)blk * blockDim.x + threadIdx.x;
    if (w >= A.words[l - 1]) return;
    uint32_t word = ((const uint32_t*)A.f[l - 1][0])[w];
    if (!word) return;
    int i = A.f[l - 1][1][w];
    const int SX = A.shape[l][0], SY = A.shape[l][1], SZ = A.shape[l][2];
    (void)SX;
    while (word) {
        const int bit = __builtin_ctz(word);
        word &= word - 1;
        unsigned long long lin = (unsigned long long)w * 32 + bit;
        const int z = (int)(lin % SZ); lin /= SZ;
        const int y = (int)(lin % SY); lin /= SY;
        const int x = (int)(lin % SX);
        const int b = (int)(lin / SX);
        down_fill_row(A, l, i, b, x, y, z);
        i++;
    }
}

__global__ void k_table_gmask_all(DownAll A) {
    const int l = down_all_level(A.gm_begin, A.nl, blockIdx.x);
    const int o = (blockIdx.x - A.gm_begin[l]) * blockDim.x + threadIdx.x;
    const int ld = A.cap[l + 1];
    const int32_t* tbl = A.f[l][4];
    uint32_t mask = 0;
    if (o < ld)
        for (int k = 0; k < 8; k++)
            if (tbl[(size_t)k * ld + o] >= 0) mask |= 1u << k;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) mask |= __shfl_xor(mask, d, 64);
    if ((threadIdx.x & 15) == 0 && o < ld) ((uint32_t*)A.f[l][8])[o >> 4] = mask;
}

// the chain is built with every stage as one launch over all levels; gf_rules_down2_chain_range (level by level, the
// round-4 form) remains for callers that want a sub-range
bool gf_rules_level_parallel() { return true; }

int gf_rules_down2_chain_all(const int32_t* coords, int M0, int B, int X, int Y, int Z, int nlevels, int32_t* ws,
                             int32_t* counts, hipStream_t st) {
    GF_CHECK_ARG(coords && ws && counts, "gf_rules_down2_chain: null argument");
    GF_CHECK_ARG(M0 >= 0 && B > 0 && nlevels >= 0 && nlevels <= DOWN_MAX_LEVELS, "gf_rules_down2_chain: bad sizes");
    DownPlan P;
    GF_CHECK_ARG(plan_down_chain(M0, B, X, Y, Z, nlevels, P) == 0, "gf_rules_down2_chain: grid too large");
    if (P.nl == 0 || M0 == 0) return GF_OK;
    GF_TRY(hipMemsetAsync(ws + P.bitmaps_begin, 0, (size_t)(P.zero_end - P.bitmaps_begin) * 4, st));
    GF_TRY(hipMemsetAsync(ws + P.child_begin, 0xff, (size_t)(P.ff_end - P.child_begin) * 4, st));
    DownAll A;
    A.nl = P.nl; A.B = B; A.M0 = M0; A.coords0 = coords; A.counts = counts;
    for (int l = 0; l <= P.nl; l++) {
        A.cap[l] = P.cap[l];
        for (int a = 0; a < 3; a++) A.shape[l][a] = P.shape[l][a];
    }
    A.scan_begin[0] = A.fill_begin[0] = A.gm_begin[0] = 0;
    for (int l = 0; l < P.nl; l++) {
        A.words[l] = (unsigned)P.words[l];
        for (int f = 0; f < DOWN_FIELDS; f++) A.f[l][f] = ws + P.off[l][f];
        A.scan_begin[l + 1] = A.scan_begin[l] + (int)scan_blocks(P.words[l]);
        const long long items = l == 0 ? (long long)P.cap[0] : (long long)P.words[l - 1];
        A.fill_begin[l + 1] = A.fill_begin[l] + gf_div_up(items, 256);
        A.gm_begin[l + 1] = A.gm_begin[l] + gf_div_up(P.cap[l + 1], 256);
    }
    hipLaunchKernelGGL(k_down_bits_all, dim3(gf_div_up(M0, 256)), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_block_popc_all, dim3(A.scan_begin[P.nl]), dim3(SCAN_THREADS), 0, st, A);
    hipLaunchKernelGGL(k_scan_block_sums_all, dim3(P.nl), dim3(SCAN_THREADS), 0, st, A);
    hipLaunchKernelGGL(k_word_prefix_all, dim3(A.scan_begin[P.nl]), dim3(SCAN_THREADS), 0, st, A);
    hipLaunchKernelGGL(k_down_fill_all, dim3(A.fill_begin[P.nl]), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_table_gmask_all, dim3(A.gm_begin[P.nl]), dim3(256), 0, st, A);
    GF_CHECK_LAUNCH("gf_rules_down2_chain");
    return GF_OK;
}

extern "C" int gf_rules_down2_chain(const int32_t* coords, int M0, int B, int X, int Y, int Z, int nlevels,
                                    int32_t* ws, int32_t* counts, void* stream) {
    if (gf_rules_level_parallel())
        return gf_rules_down2_chain_all(coords, M0, B, X, Y, Z, nlevels, ws, counts, (hipStream_t)stream);
    return gf_rules_down2_chain_range(coords, M0, B, X, Y, Z, nlevels, 0, nlevels, ws, counts, (hipStream_t)stream);
}
