/*
 * gf_oracle.c -- CPU restatement of the GeoFormer hot-path native operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under geoformer_amd/ may import, link or call this
 * file; it is the checker used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Every function cites the reference lines it restates (paths are
 * relative to the reference checkout; the sources are NOT copied here, the algorithm is
 * written out again in plain scalar C).
 *
 * PARITY PINNING
 *   - pointnet2 / PG_OP operators: pinned by construction against the cited CUDA/C++
 *     sources (the reference holds no golden vectors for them; its only unit test is a
 *     CUDA gradcheck of three_interpolate, lib/pointnet2/pointnet2_test.py:15-27), and
 *     by the hand-derived known-answer tests in tests/test_oracle_kats.py.
 *   - geodesic BFS: pinned against the reference's own Python (cal_geodesic_vectorize,
 *     model/geoformer/geodesic_utils.py:91-164) imported in the build container; the
 *     generated vectors live in tests/golden/ (generator: tests/golden/make_golden.py).
 *   - sparse convolution (spconv 1.0, llijiang/spconv@740a5b7) and kNN (faiss-gpu,
 *     unpinned version) are third-party code ABSENT from the reference tree and the
 *     reference holds no test for either: PARITY UNPINNED for those two.  The oracle
 *     restates their published semantics and is cross-checked
 *     against torch.nn.functional.conv3d on densified scenes.
 *
 * Floating point: built with -ffp-contract=off; every fused multiply-add that decides
 * an integer output is an explicit fmaf() in the order documented at the call site
 * (the CUDA build of the reference contracts the same expressions).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* Host threads for the embarrassingly parallel outer loops (output rows of a convolution, kNN queries, BFS
 * sources, ball-query centres).  Every output element is still produced by ONE thread in the documented
 * order, so results do not depend on the thread count.  Used by bench.py's cpu_baseline ("cores"). */
ORC_API int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * small open-addressing hash (uint64 key -> int32 value), used where the reference uses
 * google::dense_hash_map (lib/pointgroup_ops/src/datatype/datatype.h:7,24)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint64_t *keys;
    int32_t *vals;
    uint64_t cap; /* power of two */
} orc_map;

static uint64_t orc_mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static int orc_map_init(orc_map *m, uint64_t n) {
    uint64_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    m->cap = cap;
    m->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    m->vals = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!m->keys || !m->vals) return -1;
    memset(m->keys, 0xff, cap * sizeof(uint64_t));
    return 0;
}
static void orc_map_free(orc_map *m) { free(m->keys); free(m->vals); }
/* returns slot; *found says whether key was present */
static uint64_t orc_map_slot(const orc_map *m, uint64_t key, int *found) {
    uint64_t s = orc_mix(key) & (m->cap - 1);
    while (m->keys[s] != UINT64_MAX && m->keys[s] != key) s = (s + 1) & (m->cap - 1);
    *found = (m->keys[s] == key);
    return s;
}
static int32_t orc_map_get(const orc_map *m, uint64_t key) {
    int f; uint64_t s = orc_map_slot(m, key, &f);
    return f ? m->vals[s] : -1;
}

ORC_API void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------
 * a1  voxelize_idx   (lib/pointgroup_ops/src/voxelize/voxelize.cpp:10-31, 34-49, 58-152)
 *   voxel ids are handed out by an insertion counter in point order (:96-104); the rule
 *   table row of voxel v is [count, point ids in point order..., 0 padding] (:143-149);
 *   output coords are those of rule[1], the first point of the voxel (:39-48).
 *   mode 1 keeps front(), mode 2 back() (:125-136), modes 3/4 keep every point.
 * ---------------------------------------------------------------------------------- */
static uint64_t orc_pack4(int64_t b, int64_t x, int64_t y, int64_t z) {
    return ((uint64_t)(b & 0xffff) << 48) | ((uint64_t)(x & 0xffff) << 32) | ((uint64_t)(y & 0xffff) << 16) |
           (uint64_t)(z & 0xffff);
}

ORC_API int orc_voxelize_idx(const int64_t *coords, int32_t N, int32_t ncol, int32_t mode, int32_t *input_map,
                             int32_t *M_out, int32_t *maxActive_out, int64_t **out_coords, int32_t **out_map) {
    orc_map mp;
    if (orc_map_init(&mp, (uint64_t)N)) return -1;
    int32_t nActive = 0;
    int32_t *cnt = (int32_t *)calloc((size_t)N + 1, sizeof(int32_t));
    for (int32_t i = 0; i < N; i++) {
        const int64_t *c = coords + (size_t)i * ncol;
        uint64_t key = ncol == 4 ? orc_pack4(c[0], c[1], c[2], c[3]) : orc_pack4(0, c[0], c[1], c[2]);
        int f; uint64_t s = orc_map_slot(&mp, key, &f);
        if (!f) { mp.keys[s] = key; mp.vals[s] = nActive++; }
        input_map[i] = mp.vals[s];
        cnt[mp.vals[s]]++;
    }
    int32_t maxActive = 1;
    if (mode == 3 || mode == 4)
        for (int32_t v = 0; v < nActive; v++) if (cnt[v] > maxActive) maxActive = cnt[v];
    int32_t ld = maxActive + 1;
    int32_t *om = (int32_t *)calloc((size_t)nActive * ld + 1, sizeof(int32_t));
    int64_t *oc = (int64_t *)calloc((size_t)nActive * ncol + 1, sizeof(int64_t));
    int32_t *fill = (int32_t *)calloc((size_t)nActive + 1, sizeof(int32_t));
    for (int32_t i = 0; i < N; i++) {
        int32_t v = input_map[i];
        if (mode == 3 || mode == 4) {
            om[(size_t)v * ld + 1 + fill[v]] = i;
            fill[v]++;
            om[(size_t)v * ld] = fill[v];
        } else if (mode == 2) { /* back() */
            om[(size_t)v * ld] = 1; om[(size_t)v * ld + 1] = i;
        } else { /* modes 0,1: front() */
            if (!fill[v]) { om[(size_t)v * ld] = 1; om[(size_t)v * ld + 1] = i; fill[v] = 1; }
        }
    }
    for (int32_t v = 0; v < nActive; v++) {
        int32_t first = om[(size_t)v * ld + 1];
        memcpy(oc + (size_t)v * ncol, coords + (size_t)first * ncol, sizeof(int64_t) * ncol);
    }
    free(cnt); free(fill); orc_map_free(&mp);
    *M_out = nActive; *maxActive_out = maxActive; *out_coords = oc; *out_map = om;
    return 0;
}

/* a2  voxelize_fp / voxelize_bp  (lib/pointgroup_ops/src/voxelize/voxelize.cu:9-22, 34-47)
 *   out[row] = sum_i (1/cnt) * feats[r[i]] accumulated in i order: each term is the
 *   rounded product multiplier*x added to the running sum (the CUDA code passes the
 *   product to atomicAdd, so product and add are never fused). */
ORC_API void orc_voxelize_fp(const float *feats, const int32_t *rules, int32_t M, int32_t maxActive, int32_t C,
                             int32_t average, float *out) {
    for (int32_t row = 0; row < M; row++) {
        const int32_t *r = rules + (size_t)row * (maxActive + 1);
        int32_t n = r[0];
        float mult = (average && n > 0) ? 1.0f / (float)n : 1.0f;
        float *o = out + (size_t)row * C;
        for (int32_t c = 0; c < C; c++) o[c] = 0.0f;
        for (int32_t i = 1; i <= n; i++) {
            const float *inp = feats + (size_t)r[i] * C;
            for (int32_t c = 0; c < C; c++) { volatile float p = mult * inp[c]; o[c] = o[c] + p; }
        }
    }
}
ORC_API void orc_voxelize_bp(const float *d_out, const int32_t *rules, int32_t M, int32_t maxActive, int32_t C,
                             int32_t average, float *d_feats /* [N,C], pre-zeroed */) {
    for (int32_t row = 0; row < M; row++) {
        const int32_t *r = rules + (size_t)row * (maxActive + 1);
        int32_t n = r[0];
        float mult = (average && n > 0) ? 1.0f / (float)n : 1.0f;
        const float *o = d_out + (size_t)row * C;
        for (int32_t i = 1; i <= n; i++) {
            float *inp = d_feats + (size_t)r[i] * C;
            for (int32_t c = 0; c < C; c++) { volatile float p = mult * o[c]; inp[c] = inp[c] + p; }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * a4  sparse-conv rulebooks (spconv 1.0 get_indice_pairs; third-party, see header).
 *   Canonical, output-stationary form used by the whole build:
 *     nbr[k*ld + o] = input row feeding output row o through kernel offset k, or -1.
 *   kernel offset index k = (kx*K + ky)*K + kz over the coordinate columns (x,y,z);
 *   pair (i,o) under k means coord_in[i] = coord_out[o]*stride - pad + k_vec.
 *   Call sites that define the geometry: geoformer.py:42-44 (subm k3 p1),
 *   geoformer_modules.py:77-84 (SparseConv3d k2 s2), :94-96 (SparseInverseConv3d).
 * ---------------------------------------------------------------------------------- */
static uint64_t orc_lin(int64_t b, int64_t x, int64_t y, int64_t z, int64_t X, int64_t Y, int64_t Z) {
    return (uint64_t)(((b * X + x) * Y + y) * Z + z);
}

/* submanifold 3x3x3, padding 1: outputs == inputs (same rows). */
ORC_API int orc_rules_subm3(const int32_t *coords, int32_t M, int32_t X, int32_t Y, int32_t Z, int32_t ld,
                            int32_t *nbr /* [27*ld], filled */) {
    orc_map mp;
    if (orc_map_init(&mp, (uint64_t)M)) return -1;
    for (int32_t i = 0; i < M; i++) {
        const int32_t *c = coords + (size_t)i * 4;
        int f; uint64_t s = orc_map_slot(&mp, orc_lin(c[0], c[1], c[2], c[3], X, Y, Z), &f);
        mp.keys[s] = orc_lin(c[0], c[1], c[2], c[3], X, Y, Z); mp.vals[s] = i;
    }
    for (size_t t = 0; t < (size_t)27 * ld; t++) nbr[t] = -1;
    for (int32_t o = 0; o < M; o++) {
        const int32_t *c = coords + (size_t)o * 4;
        for (int kx = 0; kx < 3; kx++) for (int ky = 0; ky < 3; ky++) for (int kz = 0; kz < 3; kz++) {
            int k = (kx * 3 + ky) * 3 + kz;
            int64_t x = c[1] - 1 + kx, y = c[2] - 1 + ky, z = c[3] - 1 + kz;
            if (x < 0 || y < 0 || z < 0 || x >= X || y >= Y || z >= Z) continue;
            nbr[(size_t)k * ld + o] = orc_map_get(&mp, orc_lin(c[0], x, y, z, X, Y, Z));
        }
    }
    orc_map_free(&mp);
    return 0;
}

static int orc_cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static int orc_cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* strided 2x2x2, stride 2, padding 0.  out = floor(in/2); k_vec = in - 2*out.
 * Output voxels in ascending linearised (b,x,y,z) order over the OUTPUT shape
 * (canonical order); out_shape = floor((S-2)/2)+1; inputs whose
 * only candidate output lies outside out_shape are dropped.
 * Returns M_out (<= M); out_coords [M,4] capacity, child [8*ld_out] with ld_out given,
 * parent[M] (output row or -1), koff[M]. */
ORC_API int32_t orc_rules_down2(const int32_t *coords, int32_t M, int32_t X, int32_t Y, int32_t Z,
                                int32_t *out_coords, int32_t ld_out, int32_t *child, int32_t *parent,
                                int32_t *koff) {
    int64_t OX = (X - 2) / 2 + 1, OY = (Y - 2) / 2 + 1, OZ = (Z - 2) / 2 + 1;
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)M + 1));
    int32_t n = 0;
    for (int32_t i = 0; i < M; i++) {
        const int32_t *c = coords + (size_t)i * 4;
        int64_t ox = c[1] >> 1, oy = c[2] >> 1, oz = c[3] >> 1;
        if (ox >= OX || oy >= OY || oz >= OZ) continue;
        keys[n++] = orc_lin(c[0], ox, oy, oz, OX, OY, OZ);
    }
    qsort(keys, (size_t)n, sizeof(uint64_t), orc_cmp_u64);
    int32_t mo = 0;
    for (int32_t i = 0; i < n; i++) if (i == 0 || keys[i] != keys[i - 1]) keys[mo++] = keys[i];
    for (size_t t = 0; t < (size_t)8 * ld_out; t++) child[t] = -1;
    for (int32_t i = 0; i < M; i++) {
        const int32_t *c = coords + (size_t)i * 4;
        int64_t ox = c[1] >> 1, oy = c[2] >> 1, oz = c[3] >> 1;
        koff[i] = ((c[1] & 1) * 2 + (c[2] & 1)) * 2 + (c[3] & 1);
        if (ox >= OX || oy >= OY || oz >= OZ) { parent[i] = -1; continue; }
        uint64_t key = orc_lin(c[0], ox, oy, oz, OX, OY, OZ);
        int32_t lo = 0, hi = mo - 1, r = -1;
        while (lo <= hi) { int32_t mid = (lo + hi) / 2; if (keys[mid] == key) { r = mid; break; } if (keys[mid] < key) lo = mid + 1; else hi = mid - 1; }
        parent[i] = r;
        child[(size_t)koff[i] * ld_out + r] = i;
        out_coords[(size_t)r * 4 + 0] = c[0]; out_coords[(size_t)r * 4 + 1] = (int32_t)ox;
        out_coords[(size_t)r * 4 + 2] = (int32_t)oy; out_coords[(size_t)r * 4 + 3] = (int32_t)oz;
    }
    free(keys);
    return mo;
}

/* a5  gather-GEMM-scatter, output-stationary statement (spconv 1.0 indice_conv family;
 *   SURVEY.md App. A #5): out[o,:] = sum_k in[nbr[k][o],:] @ W[k]  (W is [K,Cin,Cout]),
 *   k ascending, fp32 accumulate.  With nbr = subm table this is indice_subm_conv, with
 *   the child table indice_conv (k2 s2), and with the one-hot parent table
 *   (nbr[k][i] = parent[i] iff k == koff[i]) indice_inverse_conv. */
ORC_API void orc_conv_fwd(const float *in, const float *W, const int32_t *nbr, int32_t K, int32_t M_out, int32_t ld,
                          int32_t Cin, int32_t Cout, float *out) {
#pragma omp parallel for schedule(static, 256)
    for (int32_t o = 0; o < M_out; o++) {
        float *dst = out + (size_t)o * Cout;
        for (int32_t c = 0; c < Cout; c++) dst[c] = 0.0f;
        for (int32_t k = 0; k < K; k++) {
            int32_t i = nbr[(size_t)k * ld + o];
            if (i < 0) continue;
            const float *src = in + (size_t)i * Cin;
            const float *w = W + (size_t)k * Cin * Cout;
            for (int32_t ci = 0; ci < Cin; ci++) {
                float a = src[ci];
                const float *wr = w + (size_t)ci * Cout;
                for (int32_t c = 0; c < Cout; c++) dst[c] = fmaf(a, wr[c], dst[c]);
            }
        }
    }
}
/* The same sum in double precision: the ARBITER of the full-size float comparisons (tests/test_gpu_fullsize.py).  Two
 * fp32 evaluations with different summation orders (this file's sequential fmaf chain, the HIP kernel's MFMA k-blocks)
 * are each compared with this one instead of with each other. */
ORC_API void orc_conv_fwd_f64(const double *in, const double *W, const int32_t *nbr, int32_t K, int32_t M_out, int32_t ld,
                              int32_t Cin, int32_t Cout, double *out) {
#pragma omp parallel for schedule(static, 256)
    for (int32_t o = 0; o < M_out; o++) {
        double *dst = out + (size_t)o * Cout;
        for (int32_t c = 0; c < Cout; c++) dst[c] = 0.0;
        for (int32_t k = 0; k < K; k++) {
            int32_t i = nbr[(size_t)k * ld + o];
            if (i < 0) continue;
            const double *src = in + (size_t)i * Cin;
            const double *w = W + (size_t)k * Cin * Cout;
            for (int32_t ci = 0; ci < Cin; ci++) {
                double a = src[ci];
                const double *wr = w + (size_t)ci * Cout;
                for (int32_t c = 0; c < Cout; c++) dst[c] += a * wr[c];
            }
        }
    }
}
/* backward wrt input: dIn[i,:] += dOut[o,:] @ W[k]^T over all pairs (i = nbr[k][o]). dIn pre-zeroed. */
ORC_API void orc_conv_dgrad(const float *dout, const float *W, const int32_t *nbr, int32_t K, int32_t M_out,
                            int32_t ld, int32_t Cin, int32_t Cout, float *din) {
    for (int32_t k = 0; k < K; k++) {
        const float *w = W + (size_t)k * Cin * Cout;
        for (int32_t o = 0; o < M_out; o++) {
            int32_t i = nbr[(size_t)k * ld + o];
            if (i < 0) continue;
            const float *g = dout + (size_t)o * Cout;
            float *dst = din + (size_t)i * Cin;
            for (int32_t ci = 0; ci < Cin; ci++) {
                float acc = 0.0f;
                for (int32_t c = 0; c < Cout; c++) acc = fmaf(g[c], w[(size_t)ci * Cout + c], acc);
                dst[ci] += acc;
            }
        }
    }
}
/* backward wrt weights: dW[k] = sum_o in[nbr[k][o],:]^T dOut[o,:]  (double accumulate). */
ORC_API void orc_conv_wgrad(const float *in, const float *dout, const int32_t *nbr, int32_t K, int32_t M_out,
                            int32_t ld, int32_t Cin, int32_t Cout, float *dW) {
    double *acc = (double *)malloc(sizeof(double) * (size_t)Cin * Cout);
    for (int32_t k = 0; k < K; k++) {
        memset(acc, 0, sizeof(double) * (size_t)Cin * Cout);
        for (int32_t o = 0; o < M_out; o++) {
            int32_t i = nbr[(size_t)k * ld + o];
            if (i < 0) continue;
            const float *src = in + (size_t)i * Cin;
            const float *g = dout + (size_t)o * Cout;
            for (int32_t ci = 0; ci < Cin; ci++)
                for (int32_t c = 0; c < Cout; c++) acc[(size_t)ci * Cout + c] += (double)src[ci] * (double)g[c];
        }
        for (size_t t = 0; t < (size_t)Cin * Cout; t++) dW[(size_t)k * Cin * Cout + t] = (float)acc[t];
    }
    free(acc);
}

/* ------------------------------------------------------------------------------------
 * a10  furthest point sampling   (lib/pointnet2/_ext_src/src/sampling_gpu.cu:72-176,
 *      launch geometry include/cuda_utils.h:17-21, temp pre-fill src/sampling.cpp:75-77)
 *   The CUDA block is emulated thread by thread: thread tid scans k = tid, tid+bs, ...
 *   with strict '>' (:111-112), threads without an eligible point contribute
 *   (best=-1, besti=0) (:93-94), points with |p|^2 <= 1e-3 are skipped (:104), and the
 *   shared-memory tree (:62-68, :118-170) keeps slot idx1 on ties.
 *   fp32 expression order: the CUDA build contracts a*a+b*b+c*c; fixed here (and in the
 *   HIP kernel) as fmaf(c,c, fmaf(b,b, a*a)).
 * ---------------------------------------------------------------------------------- */
static int orc_fps_block(int n) {
    int p = 1;
    while (p * 2 <= n && p < 512) p *= 2;
    return p < 1 ? 1 : p;
}
ORC_API void orc_fps(const float *xyz /*[b,n,3]*/, int32_t b, int32_t n, int32_t m, int32_t *idxs /*[b,m]*/) {
    if (m <= 0) return;
    int bs = orc_fps_block(n);
    float *temp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float dists[512]; int32_t dists_i[512];
    for (int32_t bi = 0; bi < b; bi++) {
        const float *ds = xyz + (size_t)bi * n * 3;
        int32_t *out = idxs + (size_t)bi * m;
        for (int32_t k = 0; k < n; k++) temp[k] = 1e10f;
        int32_t old = 0;
        out[0] = 0;
        for (int32_t j = 1; j < m; j++) {
            float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
            for (int tid = 0; tid < bs; tid++) {
                int32_t besti = 0; float best = -1.0f;
                for (int32_t k = tid; k < n; k += bs) {
                    float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
                    float mag = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
                    if ((double)mag <= 1e-3) continue;
                    float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
                    float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    float d2 = fminf(d, temp[k]);
                    temp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best; dists_i[tid] = besti;
            }
            for (int half = bs / 2; half >= 1; half /= 2)
                for (int tid = 0; tid < half; tid++) {
                    float v1 = dists[tid], v2 = dists[tid + half];
                    int32_t i1 = dists_i[tid], i2 = dists_i[tid + half];
                    dists[tid] = fmaxf(v1, v2);
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            old = dists_i[0];
            out[j] = old;
        }
    }
    free(temp);
}

/* a11  gather_points / grad   (sampling_gpu.cu:11-23, 37-50) */
ORC_API void orc_gather_points(const float *points /*[b,c,n]*/, const int32_t *idx /*[b,m]*/, int32_t b, int32_t c,
                               int32_t n, int32_t m, float *out /*[b,c,m]*/) {
    for (int32_t i = 0; i < b; i++) for (int32_t l = 0; l < c; l++) for (int32_t j = 0; j < m; j++)
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]];
}
ORC_API void orc_gather_points_grad(const float *grad_out /*[b,c,m]*/, const int32_t *idx, int32_t b, int32_t c,
                                    int32_t n, int32_t m, float *grad_points /*[b,c,n] zeroed*/) {
    for (int32_t i = 0; i < b; i++) for (int32_t l = 0; l < c; l++) for (int32_t j = 0; j < m; j++)
        grad_points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]] += grad_out[((size_t)i * c + l) * m + j];
}

/* a12  ball query   (ball_query_gpu.cu:12-47; idx pre-zeroed ball_query.cpp:22-24)
 *   first nsample indices in ascending k with d2 < radius^2 (strict, radius2 in fp32);
 *   on the first hit every slot is filled with that index; rows without a hit stay 0.
 *   d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx)), dx = new_x - x (contracted form of :34-35). */
ORC_API void orc_ball_query(const float *new_xyz /*[b,m,3]*/, const float *xyz /*[b,n,3]*/, int32_t b, int32_t n,
                            int32_t m, float radius, int32_t nsample, int32_t *idx /*[b,m,nsample]*/) {
    float radius2 = radius * radius;
    memset(idx, 0, sizeof(int32_t) * (size_t)b * m * nsample);
    for (int32_t bi = 0; bi < b; bi++) {
        const float *P = xyz + (size_t)bi * n * 3;
        const float *Q = new_xyz + (size_t)bi * m * 3;
        int32_t *I = idx + (size_t)bi * m * nsample;
#pragma omp parallel for schedule(dynamic, 16)
        for (int32_t j = 0; j < m; j++) {
            float nx = Q[j * 3 + 0], ny = Q[j * 3 + 1], nz = Q[j * 3 + 2];
            int32_t cnt = 0;
            for (int32_t k = 0; k < n && cnt < nsample; k++) {
                float dx = nx - P[k * 3 + 0], dy = ny - P[k * 3 + 1], dz = nz - P[k * 3 + 2];
                float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (d2 < radius2) {
                    if (cnt == 0) for (int32_t l = 0; l < nsample; l++) I[(size_t)j * nsample + l] = k;
                    I[(size_t)j * nsample + cnt] = k;
                    cnt++;
                }
            }
        }
    }
}

/* a13  group_points / grad   (group_points_gpu.cu:11-31, 46-67) */
ORC_API void orc_group_points(const float *points /*[b,c,n]*/, const int32_t *idx /*[b,np,ns]*/, int32_t b, int32_t c,
                              int32_t n, int32_t np, int32_t ns, float *out /*[b,c,np,ns]*/) {
    for (int32_t i = 0; i < b; i++) for (int32_t l = 0; l < c; l++) for (int32_t j = 0; j < np; j++)
        for (int32_t k = 0; k < ns; k++)
            out[(((size_t)i * c + l) * np + j) * ns + k] =
                points[((size_t)i * c + l) * n + idx[((size_t)i * np + j) * ns + k]];
}
ORC_API void orc_group_points_grad(const float *grad_out, const int32_t *idx, int32_t b, int32_t c, int32_t n,
                                   int32_t np, int32_t ns, float *grad_points /*[b,c,n] zeroed*/) {
    for (int32_t i = 0; i < b; i++) for (int32_t l = 0; l < c; l++) for (int32_t j = 0; j < np; j++)
        for (int32_t k = 0; k < ns; k++)
            grad_points[((size_t)i * c + l) * n + idx[((size_t)i * np + j) * ns + k]] +=
                grad_out[(((size_t)i * c + l) * np + j) * ns + k];
}

/* a25  three_nn / three_interpolate (+grad)   (interpolate_gpu.cu:12-62, 75-104, 119-146) */
ORC_API void orc_three_nn(const float *unknown /*[b,n,3]*/, const float *known /*[b,m,3]*/, int32_t b, int32_t n,
                          int32_t m, float *dist2 /*[b,n,3]*/, int32_t *idx /*[b,n,3]*/) {
    for (int32_t bi = 0; bi < b; bi++) for (int32_t j = 0; j < n; j++) {
        const float *u = unknown + ((size_t)bi * n + j) * 3;
        double best1 = 1e40, best2 = 1e40, best3 = 1e40;
        int32_t b1 = 0, b2 = 0, b3 = 0;
        for (int32_t k = 0; k < m; k++) {
            const float *p = known + ((size_t)bi * m + k) * 3;
            float dx = u[0] - p[0], dy = u[1] - p[1], dz = u[2] - p[2];
            float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (d < best1) { best3 = best2; b3 = b2; best2 = best1; b2 = b1; best1 = d; b1 = k; }
            else if (d < best2) { best3 = best2; b3 = b2; best2 = d; b2 = k; }
            else if (d < best3) { best3 = d; b3 = k; }
        }
        float *d2 = dist2 + ((size_t)bi * n + j) * 3; int32_t *ix = idx + ((size_t)bi * n + j) * 3;
        d2[0] = (float)best1; d2[1] = (float)best2; d2[2] = (float)best3; ix[0] = b1; ix[1] = b2; ix[2] = b3;
    }
}
ORC_API void orc_three_interpolate(const float *points /*[b,c,m]*/, const int32_t *idx /*[b,n,3]*/,
                                   const float *weight /*[b,n,3]*/, int32_t b, int32_t c, int32_t m, int32_t n,
                                   float *out /*[b,c,n]*/) {
    for (int32_t bi = 0; bi < b; bi++) for (int32_t l = 0; l < c; l++) for (int32_t j = 0; j < n; j++) {
        const float *w = weight + ((size_t)bi * n + j) * 3; const int32_t *ix = idx + ((size_t)bi * n + j) * 3;
        const float *p = points + ((size_t)bi * c + l) * m;
        /* points[i1]*w1 + points[i2]*w2 + points[i3]*w3, contracted left to right */
        out[((size_t)bi * c + l) * n + j] = fmaf(p[ix[2]], w[2], fmaf(p[ix[1]], w[1], p[ix[0]] * w[0]));
    }
}
ORC_API void orc_three_interpolate_grad(const float *grad_out /*[b,c,n]*/, const int32_t *idx, const float *weight,
                                        int32_t b, int32_t c, int32_t n, int32_t m, float *grad_points /*[b,c,m] zeroed*/) {
    for (int32_t bi = 0; bi < b; bi++) for (int32_t l = 0; l < c; l++) for (int32_t j = 0; j < n; j++) {
        const float *w = weight + ((size_t)bi * n + j) * 3; const int32_t *ix = idx + ((size_t)bi * n + j) * 3;
        float g = grad_out[((size_t)bi * c + l) * n + j];
        float *gp = grad_points + ((size_t)bi * c + l) * m;
        gp[ix[0]] += g * w[0]; gp[ix[1]] += g * w[1]; gp[ix[2]] += g * w[2];
    }
}

/* ------------------------------------------------------------------------------------
 * a15  kNN   (call sites model/geoformer/geodesic_utils.py:11-24, geoformer.py:172-177;
 *   faiss-gpu GpuIndexFlatL2 is third-party and absent -> parity unpinned, see header)
 *   exact fp32 squared L2, d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx)); the k smallest by
 *   (d2, index) ascending; D receives SQUARED distances like faiss; rows with fewer than
 *   k candidates are padded with (inf, -1).
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_knn(const float *base /*[n,3]*/, int32_t n, const float *query /*[nq,3]*/, int32_t nq, int32_t k,
                     float *D /*[nq,k]*/, int64_t *I /*[nq,k]*/) {
#pragma omp parallel
    {
        float *bd = (float *)malloc(sizeof(float) * (size_t)k);
        int64_t *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
#pragma omp for schedule(dynamic, 64)
        for (int32_t q = 0; q < nq; q++) {
            int32_t cnt = 0;
            float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
            for (int32_t p = 0; p < n; p++) {
                float dx = qx - base[p * 3 + 0], dy = qy - base[p * 3 + 1], dz = qz - base[p * 3 + 2];
                float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (cnt == k && !(d < bd[k - 1])) continue; /* ascending p: ties keep the earlier index */
                int32_t pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && d < bd[pos - 1]) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
                bd[pos] = d; bi[pos] = p;
                if (cnt < k) cnt++;
            }
            for (int32_t j = 0; j < k; j++) {
                D[(size_t)q * k + j] = j < cnt ? bd[j] : INFINITY;
                I[(size_t)q * k + j] = j < cnt ? bi[j] : -1;
            }
        }
        free(bd); free(bi);
    }
}

/* ------------------------------------------------------------------------------------
 * a16  geodesic distance, hop-synchronous BFS
 *      (model/geoformer/geodesic_utils.py:91-164, unique_with_inds :4-8)
 *   dist_arr/idx_arr are the kNN lists with the self column already dropped (:110-111)
 *   and distances already square-rooted (:22).  Per query q (queries never interact:
 *   the reference de-duplicates (point,query) columns, :131-136):
 *     geo[q][src] = 0, visited[src] = 1 (:118-119)
 *     seed list = in-radius neighbours of src in rank order, NOT visited-filtered (:121-127)
 *     each hop: keep the FIRST list entry of every point (torch.unique sorts the columns
 *     by (point,query) and unique_with_inds returns the first original position), walk
 *     the survivors in ascending point order, assign geo/visited (:139-140), and append
 *     every in-radius, unvisited neighbour with distance D[p][r] + geo (:143-161).
 *   Hence a point's value is  geo[parent] + D[parent][rank]  for the lowest-index parent
 *   of the previous hop, lowest rank.  At most max_step hops are assigned.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_geodesic(const float *dist_arr /*[n,kk]*/, const int64_t *idx_arr /*[n,kk]*/, int32_t n, int32_t kk,
                          const int64_t *query_inds /*[nq]*/, int32_t nq, float radius, int32_t max_step,
                          float *geo /*[nq,n]*/) {
#pragma omp parallel
    {
    uint8_t *visited = (uint8_t *)malloc((size_t)n);
    int32_t *first = (int32_t *)malloc(sizeof(int32_t) * (size_t)n); /* stamp of the hop that listed the point */
    float *cand = (float *)malloc(sizeof(float) * (size_t)n);
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *nxt = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
#pragma omp for schedule(dynamic, 1)
    for (int32_t q = 0; q < nq; q++) {
        float *g = geo + (size_t)q * n;
        for (int32_t p = 0; p < n; p++) { g[p] = -1.0f; first[p] = -1; }
        memset(visited, 0, (size_t)n);
        int64_t src = query_inds[q];
        g[src] = 0.0f; visited[src] = 1;
        int32_t ncur = 0;
        for (int32_t r = 0; r < kk; r++) {
            float d = dist_arr[(size_t)src * kk + r]; int64_t v = idx_arr[(size_t)src * kk + r];
            if (d <= radius && v >= 0 && first[v] != 0) { first[v] = 0; cand[v] = d; cur[ncur++] = (int32_t)v; }
        }
        for (int32_t step = 0; step < max_step && ncur > 0; step++) {
            /* ascending point order == the order torch.unique leaves the frontier in */
            qsort(cur, (size_t)ncur, sizeof(int32_t), orc_cmp_i32);
            for (int32_t t = 0; t < ncur; t++) { int32_t p = cur[t]; g[p] = cand[p]; visited[p] = 1; }
            int32_t nn = 0;
            for (int32_t t = 0; t < ncur; t++) {
                int32_t p = cur[t];
                float base = g[p];
                for (int32_t r = 0; r < kk; r++) {
                    float d = dist_arr[(size_t)p * kk + r]; int64_t v = idx_arr[(size_t)p * kk + r];
                    if (!(d <= radius) || v < 0 || visited[v]) continue;
                    if (first[v] == step + 1) continue; /* an earlier (lower parent, lower rank) entry wins */
                    first[v] = step + 1; cand[v] = d + base; nxt[nn++] = (int32_t)v;
                }
            }
            int32_t *tmp = cur; cur = nxt; nxt = tmp; ncur = nn;
        }
    }
    free(visited); free(first); free(cand); free(cur); free(nxt);
    }
}

/* ------------------------------------------------------------------------------------
 * a25  dormant PG_OP natives
 * ---------------------------------------------------------------------------------- */
/* sec_mean/min/max  (lib/pointgroup_ops/src/sec_mean/sec_mean.cu:12-27, 38-53, 64-79) */
ORC_API void orc_sec_mean(const float *inp /*[N,C]*/, const int32_t *offsets /*[nP+1]*/, int32_t nP, int32_t C,
                          float *out /*[nP,C]*/) {
    for (int32_t p = 0; p < nP; p++) {
        int32_t s = offsets[p], e = offsets[p + 1];
        float cnt = (float)(e - s);
        for (int32_t c = 0; c < C; c++) {
            float mean = 0.0f;
            for (int32_t i = s; i < e; i++) { volatile float t = inp[(size_t)i * C + c] / cnt; mean = mean + t; }
            out[(size_t)p * C + c] = mean;
        }
    }
}
ORC_API void orc_sec_min(const float *inp, const int32_t *offsets, int32_t nP, int32_t C, float *out) {
    for (int32_t p = 0; p < nP; p++) for (int32_t c = 0; c < C; c++) {
        float v = INFINITY; /* the CUDA initialiser 1e50 converts to +inf in fp32 (:45) */
        for (int32_t i = offsets[p]; i < offsets[p + 1]; i++) if (inp[(size_t)i * C + c] < v) v = inp[(size_t)i * C + c];
        out[(size_t)p * C + c] = v;
    }
}
ORC_API void orc_sec_max(const float *inp, const int32_t *offsets, int32_t nP, int32_t C, float *out) {
    for (int32_t p = 0; p < nP; p++) for (int32_t c = 0; c < C; c++) {
        float v = -INFINITY;
        for (int32_t i = offsets[p]; i < offsets[p + 1]; i++) if (inp[(size_t)i * C + c] > v) v = inp[(size_t)i * C + c];
        out[(size_t)p * C + c] = v;
    }
}

/* roipool_fp  (lib/pointgroup_ops/src/roipool/roipool.cu:12-31): segment max + arg-max (first max wins) */
ORC_API void orc_roipool_fp(const float *feats, const int32_t *offsets, int32_t nP, int32_t C, float *out,
                            int32_t *maxidx) {
    for (int32_t p = 0; p < nP; p++) for (int32_t c = 0; c < C; c++) {
        int32_t arg = -1; float v = -INFINITY;
        for (int32_t i = offsets[p]; i < offsets[p + 1]; i++)
            if (feats[(size_t)i * C + c] > v) { v = feats[(size_t)i * C + c]; arg = i; }
        out[(size_t)p * C + c] = v; maxidx[(size_t)p * C + c] = arg;
    }
}
/* get_iou  (lib/pointgroup_ops/src/get_iou/get_iou.cu:12-29) */
ORC_API void orc_get_iou(const int32_t *pidx, const int32_t *poff, const int64_t *inst_labels,
                         const int32_t *inst_pointnum, int32_t nInst, int32_t nP, float *iou) {
    for (int32_t p = 0; p < nP; p++) for (int32_t k = 0; k < nInst; k++) {
        int32_t inter = 0;
        for (int32_t i = poff[p]; i < poff[p + 1]; i++) if ((int32_t)inst_labels[pidx[i]] == k) inter++;
        float denom = (float)((poff[p + 1] - poff[p]) + inst_pointnum[k] - inter) + 1e-5f;
        iou[(size_t)p * nInst + k] = (float)inter / denom;
    }
}
/* ballquery_batch_p  (bfs_cluster.cu:15-60) in canonical order: starts assigned in point order (the CUDA code
 * hands them out with an atomic cursor).  Returns the total pair count; idx holds at most n*meanActive. */
ORC_API int32_t orc_ballquery_batch_p(const float *xyz, const int32_t *batch_idxs, const int32_t *batch_offsets,
                                      int32_t n, int32_t meanActive, float radius, int32_t *idx, int32_t *start_len) {
    float r2 = radius * radius;
    int64_t thre = (int64_t)n * meanActive;
    int32_t cum = 0;
    for (int32_t i = 0; i < n; i++) {
        int32_t b = batch_idxs[i], cnt = 0;
        start_len[i * 2] = cum;
        for (int32_t k = batch_offsets[b]; k < batch_offsets[b + 1]; k++) {
            float dx = xyz[i * 3] - xyz[k * 3], dy = xyz[i * 3 + 1] - xyz[k * 3 + 1], dz = xyz[i * 3 + 2] - xyz[k * 3 + 2];
            if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < r2) {
                if (cnt >= 1000) break;
                if ((int64_t)cum + cnt < thre) idx[cum + cnt] = k;
                cnt++;
            }
        }
        start_len[i * 2 + 1] = cnt;
        cum += cnt;
    }
    return cum;
}
/* bfs_cluster  (bfs_cluster.cpp:28-111): scan order + FIFO BFS over same-label ball-query edges */
ORC_API void orc_bfs_cluster(const int32_t *sem, const int32_t *bq_idx, const int32_t *start_len, int32_t N,
                             int32_t threshold, int32_t *cluster_idxs /*[N,2]*/, int32_t *cluster_offsets /*[N+1]*/,
                             int32_t *nCluster, int32_t *sumNPoint) {
    uint8_t *vis = (uint8_t *)calloc((size_t)N + 1, 1);
    int32_t *cc = (int32_t *)malloc(sizeof(int32_t) * ((size_t)N + 1));
    int32_t nc = 0, sum = 0;
    cluster_offsets[0] = 0;
    for (int32_t i = 0; i < N; i++) {
        if (vis[i]) continue;
        int32_t head = 0, tail = 0;
        cc[tail++] = i; vis[i] = 1;
        while (head < tail) {
            int32_t cur = cc[head++];
            for (int32_t j = start_len[cur * 2]; j < start_len[cur * 2] + start_len[cur * 2 + 1]; j++) {
                int32_t v = bq_idx[j];
                if (sem[v] != sem[cur] || vis[v]) continue;
                cc[tail++] = v; vis[v] = 1;
            }
        }
        if (tail >= threshold) {
            for (int32_t j = 0; j < tail; j++) { cluster_idxs[(size_t)(sum + j) * 2] = nc; cluster_idxs[(size_t)(sum + j) * 2 + 1] = cc[j]; }
            sum += tail; nc++; cluster_offsets[nc] = sum;
        }
    }
    free(vis); free(cc);
    *nCluster = nc; *sumNPoint = sum;
}

/* generate_proposal statistics + membership rows  (model/geoformer/geoformer.py:206-262), batch 1.
 * sem_prob = softmax of the semantic scores of the foreground points (computed by the caller like the
 * reference's first line).  Sequential fp32 sums. */
ORC_API void orc_proposal_stats(const float *mask_logits, const float *cls_logits, const float *sem_prob, int32_t nq,
                                int32_t N, int32_t ncls, float logit_thresh, float score_thresh,
                                int32_t npoint_thresh, int32_t min_class, int32_t *cls_pred, int32_t *npoints,
                                float *scores, int32_t *final_mask) {
    for (int32_t q = 0; q < nq; q++) {
        const float *c = cls_logits + (size_t)q * ncls;
        float mx = c[0]; int32_t arg = 0;
        for (int32_t k = 1; k < ncls; k++) if (c[k] > mx) { mx = c[k]; arg = k; }
        float den = 0.f;
        for (int32_t k = 0; k < ncls; k++) den += expf(c[k] - mx);
        float cls_score = 1.0f / den;
        int32_t n = 0; float sp = 0.f, ss = 0.f;
        for (int32_t p = 0; p < N; p++) {
            float pr = 1.0f / (1.0f + expf(-mask_logits[(size_t)q * N + p]));
            if (pr >= logit_thresh) { n++; sp += pr; ss += sem_prob[(size_t)p * ncls + arg]; }
        }
        float d = (float)n + 1e-6f, mask_score = sp / d, sem_score = ss / d;
        cls_pred[q] = arg; npoints[q] = n;
        scores[q] = mask_score * sqrtf(cls_score) * sem_score;
        final_mask[q] = (arg >= min_class) && (n >= npoint_thresh) && (mask_score >= score_thresh);
    }
}
/* few-shot form: GeoFormerFS.generate_proposal (model/geoformer/geoformer_fs.py:205-222), the query's similarity to
 * the support prototype in the class score's place */
ORC_API void orc_proposal_stats_fs(const float *mask_logits, const float *sim, int32_t nq, int32_t N, float logit_thresh,
                                   float score_thresh, int32_t npoint_thresh, float sim_thresh, int32_t *npoints,
                                   float *scores, int32_t *final_mask) {
    for (int32_t q = 0; q < nq; q++) {
        int32_t n = 0; float sp = 0.f;
        for (int32_t p = 0; p < N; p++) {
            float pr = 1.0f / (1.0f + expf(-mask_logits[(size_t)q * N + p]));
            if (pr >= logit_thresh) { n++; sp += pr; }
        }
        float mask_score = sp / ((float)n + 1e-6f);
        npoints[q] = n;
        scores[q] = mask_score * sqrtf(sim[q]);
        final_mask[q] = (sim[q] >= sim_thresh) && (n >= npoint_thresh) && (mask_score >= score_thresh);
    }
}
ORC_API void orc_proposal_scatter(const float *mask_logits, const int32_t *sel, int32_t n_sel, int32_t N,
                                  const int64_t *fg_idxs, float logit_thresh, int32_t num_points, int32_t *proposals) {
    for (int32_t i = 0; i < n_sel; i++) for (int32_t p = 0; p < N; p++)
        if (1.0f / (1.0f + expf(-mask_logits[(size_t)sel[i] * N + p])) >= logit_thresh)
            proposals[(size_t)i * num_points + fg_idxs[p]] = 1;
}

/* pairwise intersections of 0/1 proposal masks = the einsum("nc,mc->nm") of matrix_non_max_suppression
 * (util/utils_3d.py:104) */
ORC_API void orc_mask_intersections(const int32_t *masks, int32_t n, int32_t N, int32_t *inter) {
    for (int32_t i = 0; i < n; i++) for (int32_t j = 0; j < n; j++) {
        int32_t c = 0;
        for (int32_t p = 0; p < N; p++) c += (masks[(size_t)i * N + p] != 0) && (masks[(size_t)j * N + p] != 0);
        inter[(size_t)i * n + j] = c;
    }
}
