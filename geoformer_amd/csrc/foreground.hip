// Foreground selection of the forward, fused (GeoFormer.forward, model/geoformer/geoformer.py:423-439):
//     semantic_preds = semantic_scores.max(1)[1];  fg = preds >= 4 (or == 3 for the other fold);
//     fg_idxs = nonzero(fg);  batch_idxs_, locs_float_, output_feats_, semantic_scores_ = <...>[fg_idxs]
// The reference (and PyTorch) does this with an arg-max, a compare, nonzero() -- count, host read-back, partition --
// and four gathers issued after the read-back, i.e. in a stretch where the host is the bottleneck.  Here three small
// launches do all of it BEFORE the one read-back of the count: per-point class decision + per-block counts, a scan
// of the block counts, and an ordered compaction that writes the index list and the four gathered tensors in place
// (outputs are allocated at capacity N, the caller slices them to the count).
//   arg-max: first maximal class (strict '>' in ascending class order), as torch.max on a row without ties.
#include <time.h>

#include "common.h"

#define FG_THREADS 256
#define FG_PER 4  // points per thread: a workgroup owns FG_THREADS * FG_PER consecutive points

__device__ __forceinline__ bool fg_decide(const float* __restrict__ row, int C, int cls, int mode) {
    float mx = row[0];
    int arg = 0;
    for (int k = 1; k < C; k++) {
        const float v = row[k];
        if (v > mx) {
            mx = v;
            arg = k;
        }
    }
    return mode ? (arg == cls) : (arg >= cls);
}

__global__ __launch_bounds__(FG_THREADS) void k_fg_flags(const float* __restrict__ scores, int N, int C, int cls, int mode,
                                                        unsigned char* __restrict__ flags,
                                                        int32_t* __restrict__ block_counts) {
    __shared__ int s_cnt[FG_THREADS / 64];
    const int base = blockIdx.x * (FG_THREADS * FG_PER);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < FG_PER; i++) {
        const int p = base + i * FG_THREADS + threadIdx.x;
        bool f = false;
        if (p < N) {
            f = fg_decide(scores + (size_t)p * C, C, cls, mode);
            flags[p] = f ? 1 : 0;
        }
        cnt += __popcll(__ballot(f));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < FG_THREADS / 64; w++) t += s_cnt[w];
        block_counts[blockIdx.x] = t;
    }
}

// single workgroup: exclusive scan of the block counts, total -> *d_count
__global__ __launch_bounds__(SCAN_THREADS) void k_fg_scan(const int32_t* __restrict__ block_counts, int nb,
                                                         int32_t* __restrict__ block_offs, int32_t* __restrict__ d_count,
                                                         int32_t* __restrict__ h_count) {
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += SCAN_THREADS) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? block_counts[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, &tot);
        const int carry = carry_s;
        if (i < nb) block_offs[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *d_count = carry_s;
        if (h_count) {  // the host polls this word: out of the device now, not at the end of the launch
            __hip_atomic_store(h_count, carry_s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
        }
    }
}

__global__ __launch_bounds__(FG_THREADS) void k_fg_compact(const unsigned char* __restrict__ flags, int N,
                                                          const int32_t* __restrict__ block_offs,
                                                          const float* __restrict__ scores, int C,
                                                          const float* __restrict__ locs,
                                                          const int32_t* __restrict__ batch_idxs,
                                                          const float* __restrict__ feats,
                                                          const int32_t* __restrict__ feat_rows, int F,
                                                          long long* __restrict__ fg_idxs, float* __restrict__ locs_out,
                                                          int32_t* __restrict__ bidx_out, float* __restrict__ feats_out,
                                                          float* __restrict__ scores_out) {
    __shared__ int s_w[FG_THREADS / 64];
    const int base = blockIdx.x * (FG_THREADS * FG_PER);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int run = block_offs[blockIdx.x];
    for (int i = 0; i < FG_PER; i++) {
        const int p = base + i * FG_THREADS + threadIdx.x;
        const bool f = p < N && flags[p];
        const unsigned long long bal = __ballot(f);
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < FG_THREADS / 64; w++) {
            const int c = s_w[w];
            if (w < wave) before += c;
            total += c;
        }
        if (f) {
            const size_t pos = (size_t)(run + before + __popcll(bal & ((1ull << lane) - 1ull)));
            fg_idxs[pos] = p;
            if (locs_out) {
                locs_out[pos * 3 + 0] = locs[(size_t)p * 3 + 0];
                locs_out[pos * 3 + 1] = locs[(size_t)p * 3 + 1];
                locs_out[pos * 3 + 2] = locs[(size_t)p * 3 + 2];
            }
            if (bidx_out) bidx_out[pos] = batch_idxs[p];
            if (scores_out)
                for (int k = 0; k < C; k++) scores_out[pos * C + k] = scores[(size_t)p * C + k];
            if (feats_out) {
                const size_t fr = feat_rows ? (size_t)feat_rows[p] : (size_t)p;  // feats_[i] = feats[feat_rows[fg_idxs[i]]]
                if ((F & 3) == 0) {
                    const float4* src = reinterpret_cast<const float4*>(feats + fr * F);
                    float4* dst = reinterpret_cast<float4*>(feats_out + pos * F);
                    for (int k = 0; k < (F >> 2); k++) dst[k] = src[k];
                } else {
                    for (int k = 0; k < F; k++) feats_out[pos * F + k] = feats[fr * F + k];
                }
            }
        }
        run += total;
        __syncthreads();
    }
}

static int fg_blocks(int N) { return (N + FG_THREADS * FG_PER - 1) / (FG_THREADS * FG_PER); }

extern "C" size_t gf_fg_scratch_bytes(int N) {
    const size_t nb = (size_t)fg_blocks(N > 0 ? N : 0);
    return ((size_t)(N > 0 ? N : 0) + 63) / 64 * 64 + (2 * nb + 16) * sizeof(int32_t);
}

extern "C" int gf_fg_select(const float* scores, int N, int C, int cls, int mode, const float* locs,
                            const int32_t* batch_idxs, const float* feats, const int32_t* feat_rows, int F,
                            void* scratch, long long* fg_idxs,
                            float* locs_out, int32_t* bidx_out, float* feats_out, float* scores_out, int32_t* d_count,
                            int32_t* h_count, void* stream) {
    GF_CHECK_ARG(scores && scratch && fg_idxs && d_count, "gf_fg_select: null argument");
    GF_CHECK_ARG(N >= 0 && C >= 1 && (mode == 0 || mode == 1), "gf_fg_select: N=%d C=%d mode=%d", N, C, mode);
    GF_CHECK_ARG((locs_out == nullptr || locs) && (bidx_out == nullptr || batch_idxs) && (feats_out == nullptr || (feats && F >= 1)),
                 "gf_fg_select: an output was requested without its source");
    GF_CHECK_ARG(feats_out == nullptr || (F & 3) != 0 || ((((uintptr_t)feats) | ((uintptr_t)feats_out)) & 15) == 0,
                 "gf_fg_select: feature rows must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nb = fg_blocks(N);
    unsigned char* flags = (unsigned char*)scratch;
    int32_t* block_counts = (int32_t*)(flags + ((size_t)N + 63) / 64 * 64);
    int32_t* block_offs = block_counts + nb;
    if (N == 0) {
        GF_TRY(hipMemsetAsync(d_count, 0, sizeof(int32_t), st));
        if (h_count) *h_count = 0;
        return GF_OK;
    }
    hipLaunchKernelGGL(k_fg_flags, dim3(nb), dim3(FG_THREADS), 0, st, scores, N, C, cls, mode, flags, block_counts);
    hipLaunchKernelGGL(k_fg_scan, dim3(1), dim3(SCAN_THREADS), 0, st, block_counts, nb, block_offs, d_count, h_count);
    hipLaunchKernelGGL(k_fg_compact, dim3(nb), dim3(FG_THREADS), 0, st, flags, N, block_offs, scores, C, locs, batch_idxs,
                       feats, feat_rows, F, fg_idxs, locs_out, bidx_out, feats_out, scores_out);
    GF_CHECK_LAUNCH("gf_fg_select");
    return GF_OK;
}

// Host helper of the h_count protocol above: wait for a word a kernel stores to host memory, without a system call
extern "C" int gf_host_wait_word(const volatile int32_t* word, int pending, long long timeout_us) {
    if (!word) return pending;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; spins++) {
        const int v = *word;
        if (v != pending) return v;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        __builtin_ia32_pause();
#endif
        if ((spins & 1023u) == 1023u) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const long long us = (long long)(t1.tv_sec - t0.tv_sec) * 1000000LL + (t1.tv_nsec - t0.tv_nsec) / 1000;
            if (us >= timeout_us) return *word;
        }
    }
}
