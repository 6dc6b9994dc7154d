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
static int legacy_choice_impl(uint32_t* key, int32_t* pos_io, long long n, long long k, long long* out, bool use_prefetch);
extern "C" int gf_host_legacy_choice(uint32_t* key, int32_t* pos_io, long long n, long long k, long long* out) {
    GF_CHECK_ARG(key && pos_io && out, "gf_host_legacy_choice: null argument");
    GF_CHECK_ARG(n >= 1 && n <= 0x7fffffffLL && k >= 0 && k <= n, "gf_host_legacy_choice: n=%lld k=%lld", n, k);
    GF_CHECK_ARG(*pos_io >= 0 && *pos_io <= MT_N, "gf_host_legacy_choice: generator position %d", (int)*pos_io);
    const Prefetch& P = t_pf;
    if (P.pos0 == *pos_io && P.words && memcmp(P.key0, key, sizeof(P.key0)) == 0) {
        const int rc = legacy_choice_impl(key, pos_io, n, k, out, true);
        t_pf.pos0 = -1;  // (consumed)
        if (rc != 1) return rc;  // 1: the words drawn ahead did not suffice -- nothing was changed: the plain draw
    }
    return legacy_choice_impl(key, pos_io, n, k, out, false);
}

static int legacy_choice_impl(uint32_t* key, int32_t* pos_io, long long n, long long k, long long* out, bool use_prefetch) {
    int pos = *pos_io;
    uint32_t block[MT_N];
    const Prefetch& P = t_pf;
    size_t used = 0;  // words of the stream drawn ahead that pass 1 has consumed
    if (!use_prefetch && pos < MT_N) mt_temper_block(key, block);
    uint32_t* J = (uint32_t*)malloc((size_t)(n + 1) * sizeof(uint32_t));
    uint32_t* x = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));  // 32-bit working set: half the cache footprint
    if (!J || !x) {
        free(J);
        free(x);
        gf_set_error("gf_host_legacy_choice: out of memory");
        return GF_ERR_LAUNCH;
    }
    // pass 1: the swap partner of every position, top down
    uint32_t i = (uint32_t)(n - 1);
    while (i >= 1) {
        uint32_t mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const uint32_t lo = (mask >> 1) + 1;  // the mask serves positions lo..mask
        while (i >= lo) {
            int avail;
            const uint32_t* b;
            if (use_prefetch) {
                if (used >= P.nwords) {  // (drawn ahead too little: the caller starts over on the generator)
                    free(J);
                    free(x);
                    return 1;
                }
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
            // i drops by at most one per draw, so the next min(words left, i - lo + 1) draws cannot leave the mask's
            // range: a check-free inner loop (the only loop-carried chain is compare -> subtract)
            const uint32_t room = i - lo + 1;
            const int m = room < (uint32_t)avail ? (int)room : avail;
            for (int t = 0; t < m; t++) {
                const uint32_t v = b[t] & mask;
                J[i] = v;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
                // i -= (v <= i) as compare + add-with-carry: two cycles of loop-carried latency instead of the four of
                // the compare / set / extend / subtract sequence the compiler emits (this loop is 80k iterations of
                // nothing else)
                asm("cmp %1, %0\n\tadc $-1, %0" : "+r"(i) : "r"(v) : "cc");
#else
                i -= (v <= i);
#endif
            }
            if (use_prefetch) used += (size_t)m;
            else pos += m;
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
    for (long long p = n - 1; p >= 1; p--) {
        if (p >= 24) __builtin_prefetch(&x[J[p - 24]], 1, 1);
        const uint32_t j = J[p];
        const uint32_t tmp = x[j];
        x[j] = x[p];
        x[p] = tmp;
    }
    for (long long t = 0; t < k; t++) out[t] = (long long)x[t];
    free(J);
    free(x);
    *pos_io = pos;
    return GF_OK;
}
