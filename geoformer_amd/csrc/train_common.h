// Shared by the training kernels of the two transformer stacks (backbone_attn.hip, decoder_layer_train.hip): the hashed
// dropout masks and the one-launch weight / bias gradient kernel.
#pragma once
#include "common.h"

// ---- dropout: the keep decision of an element is a hash of (seed, site, row, column), so a backward recomputes its
// masks instead of storing them and a test can build the same masks on the host (include/geoformer_hip.h) ----
struct GfDrop {
    uint32_t seed, thresh;
    float inv;
};

static inline GfDrop gf_drop_make(float p, unsigned seed) {
    GfDrop d;
    d.seed = seed;
    d.thresh = p <= 0.f ? 0u : (uint32_t)((double)p * 16777216.0);
    d.inv = p <= 0.f ? 1.f : 1.f / (1.f - p);
    return d;
}

__device__ __forceinline__ uint32_t gf_fmix32(uint32_t x) {  // MurmurHash3's finaliser
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

// 1 / (1 - p) if the element is kept, else 0
__device__ __forceinline__ float gf_drop_keep(const GfDrop& d, uint32_t site, uint32_t row, uint32_t col) {
    uint32_t h = gf_fmix32(d.seed ^ (row * 64u + site));
    h = gf_fmix32(h + col * 0x9E3779B1u);
    return (h >> 8) >= d.thresh ? d.inv : 0.f;
}

// ---- every weight / bias / norm gradient of a call in one launch: dst[o][i] = sum_t A[t][o] B[t][i] over the M rows
// (a workgroup per 16 x 16 tile, its four waves take interleaved row groups, summed in a fixed order) and column sums ----
struct GfGemmJob {
    const float *A, *B;
    float* dst;
    int lda, ldb, O, I, ldd, ivalid;  // O % 16 == 0; B readable up to column 16 * ceil(I / 16); dst[o][i], i < ivalid
};
struct GfColJob {
    const float* src;
    float* dst;
    int ld, n, rows, pad;  // dst[c] = sum_{r < rows} src[r * ld + c], c < n
};
#define GF_MAX_GEMM 28
#define GF_MAX_COL 48
struct GfWJobs {
    int ng, nc;
    int gstart[GF_MAX_GEMM + 1], cstart[GF_MAX_COL + 1];
    GfGemmJob g[GF_MAX_GEMM];
    GfColJob c[GF_MAX_COL];
};
static_assert(sizeof(GfWJobs) <= 3800, "kernel argument block");

int gf_wgrad_launch(const GfWJobs& J, int M, hipStream_t st);  // backbone_attn.hip; gemm jobs sum over M rows

struct GfWJobBuilder {
    GfWJobs J;
    int gt = 0, ct = 0;
    GfWJobBuilder() { J.ng = J.nc = 0; }
    void gemm(const float* A, int lda, const float* B, int ldb, int O, int I, int ivalid, float* dst, int ldd) {
        GfGemmJob& g = J.g[J.ng];
        g.A = A; g.B = B; g.dst = dst; g.lda = lda; g.ldb = ldb; g.O = O; g.I = I; g.ldd = ldd; g.ivalid = ivalid;
        J.gstart[J.ng++] = gt;
        gt += (O / 16) * ((I + 15) / 16);
    }
    void cols(const float* src, int ld, int n, int rows, float* dst) {
        GfColJob& q = J.c[J.nc];
        q.src = src; q.dst = dst; q.ld = ld; q.n = n; q.rows = rows; q.pad = 0;
        J.cstart[J.nc++] = ct;
        ct += (n + 63) / 64;
    }
    int launch(int M, hipStream_t st) {
        J.gstart[J.ng] = gt;
        J.cstart[J.nc] = ct;
        return gf_wgrad_launch(J, M, st);
    }
};
