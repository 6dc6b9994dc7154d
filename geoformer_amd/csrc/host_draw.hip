// Host-side helper (no device code): numpy's legacy sampling draw, bit for bit, at half the host time.
//
//   reference: geoformer.py:575-577 draws `np.random.choice(n_b, npoint, replace=False)` once per scene.  With the
//   legacy global generator that is `permutation(n_b)[:npoint]`: arange(n) shuffled by Fisher-Yates from the top,
//   `j = random_interval(i)` = 32-bit MT19937 outputs masked to the bit length of i and rejected while > i
//   (numpy/random/mtrand.pyx: RandomState.choice / permutation / _shuffle_raw; _legacy/legacy-distributions +
//   src/mt19937/mt19937.c for the generator).  The forward cannot start sampling before this draw and the device
//   idles meanwhile (0.43 ms for 60k points), so it is worth restating natively:
//     * the generator advances a whole 624-word block at a time and tempers it in one vectorisable pass;
//     * rejection is branch-free (the candidate is stored every time, the index only moves on acceptance) --
//       numpy's loop mispredicts on every fourth draw;
//     * the swap indices are known before the swaps start, so the swap pass prefetches its targets.
//   The caller moves the generator state in and out (np.random.get_state / set_state), so the stream of random
//   numbers stays the one the reference consumes.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define GF_DRAW_AVX2 1
#endif

namespace {
constexpr int MT_N = 624, MT_M = 397;

__attribute__((always_inline)) inline void mt_advance_body(uint32_t* key) {
    constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    int i = 0;
    for (; i < MT_N - MT_M; i++) {
        const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
        key[i] = key[i + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    for (; i < MT_N - 1; i++) {
        const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
        key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    const uint32_t y = (key[MT_N - 1] & UP) | (key[0] & LO);
    key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
}

__attribute__((always_inline)) inline void mt_temper_body(const uint32_t* __restrict__ key, uint32_t* __restrict__ out) {
    for (int i = 0; i < MT_N; i++) {
        uint32_t y = key[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[i] = y;
    }
}

// the block update and the tempering are element-wise over 624 words: compiled once more for AVX2 (8 words per
// instruction) and picked at run time
__attribute__((target("avx2"))) void mt_next_block_avx2(uint32_t* key, uint32_t* out) {
    mt_advance_body(key);
    mt_temper_body(key, out);
}
void mt_next_block_base(uint32_t* key, uint32_t* out) {
    mt_advance_body(key);
    mt_temper_body(key, out);
}
inline void mt_next_block(uint32_t* key, uint32_t* out) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) mt_next_block_avx2(key, out);
    else mt_next_block_base(key, out);
}
inline void mt_temper_block(const uint32_t* __restrict__ key, uint32_t* __restrict__ out) { mt_temper_body(key, out); }
}  // namespace

// ---- the generator's output stream drawn AHEAD of the draw (round 6) ----
// The raw 32-bit words do not depend on n -- only the masks, the rejections and the swaps do -- and the forward's host has
// nothing to do while the backbone runs: gf_host_legacy_prefetch tempers the next `nwords` outputs of the given state into
// a per-thread buffer (with the key after every block), and a gf_host_legacy_choice that starts from exactly that state
// reads its words from there.  A draw that needs more than was drawn ahead starts over on the generator itself.
namespace {
struct Prefetch {
    uint32_t key0[MT_N];
    int pos0 = -1;
    uint32_t* words = nullptr;  // outputs following (key0, pos0)
    uint32_t* keys = nullptr;   // key after the j-th block advance, j = 1 .. nblocks
    size_t nwords = 0, cap_words = 0, cap_blocks = 0;
    int nblocks = 0;
};
thread_local Prefetch t_pf;

// the draw's two work arrays (swap partners, the permutation as 32-bit words), kept per thread and grown on demand: two
// fresh allocations of 240 KB each are mapped pages the kernel has to fault in one by one -- at every draw
struct DrawScratch {
    uint32_t* J = nullptr;
    uint32_t* x = nullptr;
    size_t cap = 0;
    ~DrawScratch() {
        free(J);
        free(x);
    }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        free(J);
        free(x);
        const size_t c = n + n / 4 + 4096;
        J = (uint32_t*)malloc(c * sizeof(uint32_t));
        x = (uint32_t*)malloc(c * sizeof(uint32_t));
        cap = (J && x) ? c : 0;
        return cap != 0;
    }
};
thread_local DrawScratch t_ds;

#ifdef GF_DRAW_AVX2
// Pass 1, 32 words per trip.  Word t of a trip is accepted iff (w & mask) <= i_t, and i_t lies in [i - 31, i] (it drops by
// one per acceptance): a value <= i - 32 is accepted and a value > i is rejected whatever came before it, so unless one of
// the 32 falls into the band in between (32 * 32 / 2^bits of the trips) the trip is a plain stream compaction -- four
// independent 8-lane compares against the SAME threshold (the loop-carried chain broadcast -> compare -> movemask ->
// popcount -> subtract, ~12 cycles, is paid once per 32 words instead of once per 8), accepted lanes permuted to the front,
// one unaligned store each.  A trip with a value in the band takes the scalar steps.  The caller's loop guarantees
// i - lo >= 31 on entry of every trip (all acceptances stay under this mask).  Returns the words consumed (a multiple of 32).
struct CompactLut {
    alignas(32) uint32_t idx[256][8];
    CompactLut() {
        for (int m = 0; m < 256; m++) {
            int c = 0;
            for (int b = 0; b < 8; b++)
                if (m >> b & 1) idx[m][c++] = (uint32_t)b;
            for (; c < 8; c++) idx[m][c] = 0;
        }
    }
};
__attribute__((target("avx2,popcnt"))) int pass1_blocks_avx2(const uint32_t* __restrict__ b, int avail, uint32_t mask,
                                                               uint32_t lo, uint32_t* i_io, uint32_t* __restrict__ Jq,
                                                               size_t* q_io) {
    static const CompactLut lut;
    uint32_t i = *i_io;
    size_t q = *q_io;
    const __m256i vmask = _mm256_set1_epi32((int)mask);
    int t = 0;
    while (avail - t >= 32 && i >= lo + 31) {
        // (values and thresholds are below 2^31: signed compares)
        const __m256i hi = _mm256_set1_epi32((int)i), band_lo = _mm256_set1_epi32((int)i - 32);
        __m256i v[4];
        int rej[4], amb = 0;
        for (int u = 0; u < 4; u++) {
            v[u] = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(b + t + 8 * u)), vmask);
            rej[u] = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(v[u], hi)));
            amb |= rej[u] ^ _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(v[u], band_lo)));
        }
        if (amb) {  // a value inside (i - 32, i]: its fate depends on the words before it
            for (int u = 0; u < 32; u++) {
                const uint32_t x = b[t + u] & mask;
                Jq[q] = x;
                const uint32_t acc = x <= i;
                q += acc;
                i -= acc;
            }
        } else {
            for (int u = 0; u < 4; u++) {
                const int acc = ~rej[u] & 0xff;
                const __m256i packed = _mm256_permutevar8x32_epi32(v[u], _mm256_load_si256((const __m256i*)lut.idx[acc]));
                _mm256_storeu_si256((__m256i*)(Jq + q), packed);  // (up to 7 words past the accepted ones: scratch has the room)
                const int c = __builtin_popcount((unsigned)acc);
                q += (size_t)c;
                i -= (uint32_t)c;
            }
        }
        t += 32;
    }
    *i_io = i;
    *q_io = q;
    return t;
}
#endif
}  // namespace

extern "C" int gf_host_legacy_prefetch(const uint32_t* key, int pos, long long nwords) {
    GF_CHECK_ARG(key && pos >= 0 && pos <= MT_N && nwords >= 0, "gf_host_legacy_prefetch: bad arguments");
    Prefetch& P = t_pf;
    const size_t first = (size_t)(MT_N - pos);
    const size_t more = (size_t)nwords > first ? (size_t)nwords - first : 0;
    const int nb = (int)((more + MT_N - 1) / MT_N);
    const size_t total = first + (size_t)nb * MT_N;
    if (total > P.cap_words) {
        free(P.words);
        P.words = (uint32_t*)malloc(total * sizeof(uint32_t));
        P.cap_words = P.words ? total : 0;
    }
    if ((size_t)nb > P.cap_blocks) {
        free(P.keys);
        P.keys = (uint32_t*)malloc((size_t)(nb > 0 ? nb : 1) * MT_N * sizeof(uint32_t));
        P.cap_blocks = P.keys ? (size_t)nb : 0;
    }
    if (!P.words || (nb > 0 && !P.keys)) {
        P.pos0 = -1;
        gf_set_error("gf_host_legacy_prefetch: out of memory");
        return GF_ERR_LAUNCH;
    }
    memcpy(P.key0, key, sizeof(P.key0));
    P.pos0 = pos;
    uint32_t cur[MT_N], blk[MT_N];
    memcpy(cur, key, sizeof(cur));
    if (pos < MT_N) {
        mt_temper_block(cur, blk);
        memcpy(P.words, blk + pos, first * sizeof(uint32_t));
    }
    for (int j = 0; j < nb; j++) {
        mt_next_block(cur, P.words + first + (size_t)j * MT_N);
        memcpy(P.keys + (size_t)j * MT_N, cur, sizeof(cur));
    }
    P.nwords = total;
    P.nblocks = nb;
    return GF_OK;
}

// key[624], *pos: numpy's MT19937 state (pos == 624: the block is used up).  out[k] = permutation(n)[:k].
static int legacy_choice_impl(uint32_t* key, int32_t* pos_io, long long n, long long k, long long* out, int32_t* out32,
                              long long cap32, bool use_prefetch);
extern "C" int gf_host_legacy_choice(uint32_t* key, int32_t* pos_io, long long n, long long k, long long* out) {
    GF_CHECK_ARG(key && pos_io && out, "gf_host_legacy_choice: null argument");
    GF_CHECK_ARG(n >= 1 && n <= 0x7fffffffLL && k >= 0 && k <= n, "gf_host_legacy_choice: n=%lld k=%lld", n, k);
    GF_CHECK_ARG(*pos_io >= 0 && *pos_io <= MT_N, "gf_host_legacy_choice: generator position %d", (int)*pos_io);
    const Prefetch& P = t_pf;
    if (P.pos0 == *pos_io && P.words && memcmp(P.key0, key, sizeof(P.key0)) == 0) {
        const int rc = legacy_choice_impl(key, pos_io, n, k, out, nullptr, 0, true);
        t_pf.pos0 = -1;  // (consumed)
        if (rc != 1) return rc;  // 1: the words drawn ahead did not suffice -- nothing was changed: the plain draw
    }
    return legacy_choice_impl(key, pos_io, n, k, out, nullptr, 0, false);
}

// the same draw as 32-bit indices: out32[0..k); with room for n entries the permutation is shuffled in place there
static int legacy_choice32(uint32_t* key, int32_t* pos_io, long long n, long long k, int32_t* out32, long long cap32) {
    const Prefetch& P = t_pf;
    if (P.pos0 == *pos_io && P.words && memcmp(P.key0, key, sizeof(P.key0)) == 0) {
        const int rc = legacy_choice_impl(key, pos_io, n, k, nullptr, out32, cap32, true);
        t_pf.pos0 = -1;
        if (rc != 1) return rc;
    }
    return legacy_choice_impl(key, pos_io, n, k, nullptr, out32, cap32, false);
}

static int legacy_choice_impl(uint32_t* key, int32_t* pos_io, long long n, long long k, long long* out, int32_t* out32,
                              long long cap32, bool use_prefetch) {
    int pos = *pos_io;
    uint32_t block[MT_N];
    const Prefetch& P = t_pf;
    size_t used = 0;  // words of the stream drawn ahead that pass 1 has consumed
    if (!use_prefetch && pos < MT_N) mt_temper_block(key, block);
    if (!t_ds.reserve((size_t)n + 16)) {  // (+ the overhang of pass 1's 8-word stores)
        gf_set_error("gf_host_legacy_choice: out of memory");
        return GF_ERR_LAUNCH;
    }
    uint32_t* J = t_ds.J;
    // 32-bit working set: half the cache footprint (the caller's own buffer when it takes 32-bit indices and has the room)
    uint32_t* x = (out32 && cap32 >= n) ? (uint32_t*)out32 : t_ds.x;
    // pass 1: the swap partner of every position, top down: Jq[q] belongs to position n - 1 - q
#ifdef GF_DRAW_AVX2
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
#endif
    uint32_t i = (uint32_t)(n - 1);
    size_t q = 0;
    while (i >= 1) {
        uint32_t mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const uint32_t lo = (mask >> 1) + 1;  // the mask serves positions lo..mask
        while (i >= lo) {
            int avail;
            const uint32_t* b;
            if (use_prefetch) {
                if (used >= P.nwords) return 1;  // (drawn ahead too little: the caller starts over on the generator)
                const size_t left = P.nwords - used;
                avail = left > 65536 ? 65536 : (int)left;
                b = P.words + used;
            } else {
                if (pos == MT_N) {
                    mt_next_block(key, block);
                    pos = 0;
                }
                avail = MT_N - pos;
                b = block + pos;
            }
            int t0 = 0;
#ifdef GF_DRAW_AVX2
            if (avx2) t0 = pass1_blocks_avx2(b, avail, mask, lo, &i, J, &q);
#endif
            // i drops by at most one per draw, so the next min(words left, i - lo + 1) draws cannot leave the mask's
            // range: a check-free inner loop (the only loop-carried chain is compare -> subtract)
            const uint32_t room = i - lo + 1;
            const int left_w = avail - t0;
            const int m = room < (uint32_t)left_w ? (int)room : left_w;
            for (int t = t0; t < t0 + m; t++) {
                const uint32_t v = b[t] & mask;
                J[q] = v;
                const uint32_t acc = v <= i;
                q += acc;
                i -= acc;
            }
            if (use_prefetch) used += (size_t)(t0 + m);
            else pos += t0 + m;
        }
    }
    if (use_prefetch) {
        // the generator's state behind `used` outputs: position inside the block they end in, that block's key
        const size_t first = (size_t)(MT_N - P.pos0);
        if (used <= first) {
            pos = P.pos0 + (int)used;
        } else {
            const size_t more = used - first;
            const size_t adv = (more + MT_N - 1) / MT_N;  // block advances (>= 1)
            memcpy(key, P.keys + (adv - 1) * MT_N, MT_N * sizeof(uint32_t));
            pos = (int)(more - (adv - 1) * MT_N);
        }
    }
    // pass 2: the swaps
    for (long long t = 0; t < n; t++) x[t] = (uint32_t)t;
    for (long long p = n - 1, s = 0; p >= 1; p--, s++) {
        if (p >= 24) __builtin_prefetch(&x[J[s + 24]], 1, 1);
        const uint32_t j = J[s];
        const uint32_t tmp = x[j];
        x[j] = x[p];
        x[p] = tmp;
    }
    if (out)
        for (long long t = 0; t < k; t++) out[t] = (long long)x[t];
    if (out32 && (uint32_t*)out32 != x) memcpy(out32, x, (size_t)k * sizeof(int32_t));
    *pos_io = pos;
    return GF_OK;
}

// ---- draw + upload + gather in one call (round 6) ----
// The eval forward's device idles from the moment the foreground count reaches the host until the first sampling launch:
// the draw itself, then -- through the framework -- the upload of the indices, the gather of the drawn points and ~80 us
// of interpreter time around them.  Here the host goes from the count to the queued gather in one native call: the draw
// as 32-bit indices straight into the caller's pinned buffer, one asynchronous copy, one launch that writes the drawn
// points' coordinates and the 64-bit copy of the indices the model keeps (geoformer.py:575-579: sampling_indices, xyz).
__global__ void k_take_drawn(const int32_t* __restrict__ idx, int k, int n, const float* __restrict__ xyz_src,
                             long long* __restrict__ idx64, float* __restrict__ xyz_dst, int32_t* __restrict__ idx32) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k) return;
    const int j = idx[t];
    idx64[t] = j;
    idx32[t] = j;
    if ((unsigned)j >= (unsigned)n) return;  // (cannot happen: a permutation of 0..n-1)
    xyz_dst[3 * t + 0] = xyz_src[3 * (size_t)j + 0];
    xyz_dst[3 * t + 1] = xyz_src[3 * (size_t)j + 1];
    xyz_dst[3 * t + 2] = xyz_src[3 * (size_t)j + 2];
}

extern "C" int gf_host_draw_sample(uint32_t* key, int32_t* pos_io, long long n, long long k, int32_t* pinned,
                                   long long pinned_cap, int32_t* d_idx32, long long* d_idx64, const float* xyz_src,
                                   float* xyz_dst, int fps_m, int32_t* fps_idx, void* fps_scratch, void* stream) {
    GF_CHECK_ARG(key && pos_io && pinned && d_idx32 && d_idx64 && xyz_src && xyz_dst, "gf_host_draw_sample: null argument");
    GF_CHECK_ARG(n >= 1 && n <= 0x7fffffffLL && k >= 1 && k <= n && pinned_cap >= k, "gf_host_draw_sample: n=%lld k=%lld cap=%lld",
                 n, k, pinned_cap);
    GF_CHECK_ARG(*pos_io >= 0 && *pos_io <= MT_N, "gf_host_draw_sample: generator position %d", (int)*pos_io);
    const int rc = legacy_choice32(key, pos_io, n, k, pinned, pinned_cap);
    if (rc != GF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // the gather reads the indices straight out of the pinned buffer (device-visible host memory: 200 KB over the bus inside
    // the kernel) instead of behind a copy command of its own -- one queue entry and its latency less on the path to the
    // first sampling launch; d_idx32 receives the device copy from the same kernel
    hipLaunchKernelGGL(k_take_drawn, dim3((unsigned)gf_div_up(k, 256)), dim3(256), 0, st, pinned, (int)k, (int)n, xyz_src, d_idx64,
                       xyz_dst, d_idx32);
    GF_CHECK_LAUNCH("gf_host_draw_sample");
    // (... and the first sampling launch over the drawn points, when the caller wants it queued right behind them)
    if (fps_m > 0) {
        GF_CHECK_ARG(fps_idx && fps_scratch, "gf_host_draw_sample: sampling outputs missing");
        return gf_furthest_point_sampling(xyz_dst, 1, (int)k, fps_m, fps_idx, fps_scratch, stream);
    }
    return GF_OK;
}
