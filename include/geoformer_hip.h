/*
 * geoformer_hip.h -- C ABI of libgeoformer_hip.so, the MI355X (gfx950) native layer under
 * GeoFormer's per-scene forward/backward hot path.
 *
 * Boundary rules
 *   - plain pointers and sizes only (no torch types); every pointer is a DEVICE pointer
 *     unless the parameter name starts with h_;
 *   - every call is asynchronous on the hipStream_t passed as `stream` (void*), allocates
 *     nothing and keeps no global state: the caller owns outputs and scratch;
 *   - return value 0 = ok, negative = gf_status; gf_last_error() gives the message of the
 *     calling thread's last failure.
 *
 * Each entry point names the reference interface it stands in for.  Reference paths are
 * relative to the VinAIResearch/GeoFormer checkout; "spconv" is llijiang/spconv@740a5b7
 * (un-vendored dependency of the reference, docs/INSTALL.md:27-47) and "faiss" the
 * faiss-gpu package (docs/INSTALL.md:69-73).
 */
#ifndef GEOFORMER_HIP_H
#define GEOFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_ABI_VERSION 5

typedef enum {
    GF_OK = 0,
    GF_ERR_INVALID_ARG = -1,
    GF_ERR_LAUNCH = -2,
    GF_ERR_UNSUPPORTED = -3,
    GF_ERR_CALLBACK = -4, /* a caller-supplied callback reported a failure (gf_unet_fwd_phased) */
} gf_status;

int gf_abi_version(void);
const char* gf_last_error(void);

/* ===================================================================================
 * Foreground selection of the forward, fused (geoformer.py:423-439): arg-max over the class scores, the
 * foreground test (mode 0: class >= cls, mode 1: class == cls), the ascending index list of the foreground points
 * and the gathered rows of the per-point tensors, all before the one read-back of the count.
 *   scores fp32 [N,C]; locs fp32 [N,3]; batch_idxs int32 [N]; feats fp32 [N,F] or, with feat_rows int32 [N]
 *   (the p2v map), fp32 [M,F] voxel rows read as feats[feat_rows[p]]  (sources, any may be NULL together with its
 *   output);  fg_idxs int64 [N], locs_out [N,3], bidx_out [N], feats_out [N,F],
 *   scores_out [N,C]: capacity N, the first *d_count rows are valid;  scratch: gf_fg_scratch_bytes(N).
 *   h_count (may be NULL): a word of PINNED host memory the count is ALSO stored to, with system scope, by the scan that
 *   finds it -- i.e. before the gathers of the four outputs run, and without a copy command or an event in between: the
 *   caller sets it to a negative value before the call and polls it (gf_host_wait_word); everything it then queues on
 *   `stream` is behind the gathers by stream order.  N = 0 stores 0 to it from the host.
 * =================================================================================== */
size_t gf_fg_scratch_bytes(int N);
int gf_fg_select(const float* scores, int N, int C, int cls, int mode, const float* locs, const int32_t* batch_idxs,
                 const float* feats, const int32_t* feat_rows, int F, void* scratch, long long* fg_idxs, float* locs_out,
                 int32_t* bidx_out, float* feats_out, float* scores_out, int32_t* d_count, int32_t* h_count, void* stream);
/* Host helper: spin (pause instructions, no system call) until *word != pending or timeout_us have passed; returns the
 * word's value (== pending: timed out).  `word`: host memory a device kernel stores to with system scope (h_count above). */
int gf_host_wait_word(const volatile int32_t* word, int pending, long long timeout_us);

/* ===================================================================================
 * Host helper (CPU code, no launch): the reference's per-scene sampling draw
 * np.random.choice(n, k, replace=False) (geoformer.py:575-577) on numpy's legacy MT19937 state, bit for bit.
 *   key uint32[624], *pos: the state as np.random.get_state() returns it (updated in place, hand it back with
 *   set_state);  out int64[k] = permutation(n)[:k].
 * =================================================================================== */
int gf_host_legacy_choice(uint32_t* key, int32_t* pos, long long n, long long k, long long* out);
/* Draw the next `nwords` outputs of the state (key, pos) AHEAD into a per-thread buffer (the caller's state is not
 * touched): a gf_host_legacy_choice that starts from exactly this state reads its words from there -- the draw of the
 * forward's sampling indices (geoformer.py:575-577) then only pays the rejections and the swaps behind the count's
 * read-back.  Too few words drawn ahead: the draw runs on the generator itself.  Same values, same final state. */
int gf_host_legacy_prefetch(const uint32_t* key, int pos, long long nwords);
/* The draw and what geoformer.py:575-579 does with it, from the host's side in ONE call (the device idles between the
 * arrival of the foreground count and the first sampling launch): the same draw as 32-bit indices into the caller's PINNED
 * buffer `pinned` (pinned_cap entries, >= k; with >= n the shuffle runs in place there) and ONE launch that reads them
 * there (device-visible host memory) and writes d_idx32[k], d_idx64[k] (the model's `sampling_indices`) and
 * xyz_dst[k,3] = xyz_src[idx] (n rows).
 * key / pos: the generator's state, advanced in place.  The pinned buffer may be rewritten once the copy has left it
 * (stream order).  fps_m > 0: gf_furthest_point_sampling(xyz_dst, 1, k, fps_m, fps_idx, fps_scratch) is queued behind the
 * gather in the same call (geoformer.py:580-581 / pointnet2_utils.furthest_point_sample on the drawn points). */
int gf_host_draw_sample(uint32_t* key, int32_t* pos, long long n, long long k, int32_t* pinned, long long pinned_cap,
                        int32_t* d_idx32, long long* d_idx64, const float* xyz_src, float* xyz_dst, int fps_m,
                        int32_t* fps_idx, void* fps_scratch, void* stream);

/* ===================================================================================
 * Sparse convolution (stands in for spconv.ops.get_indice_pairs / indice_conv /
 * indice_subm_conv / indice_inverse_conv, reached from geoformer.py:42-53 and
 * geoformer_modules.py:15-35,63-105 through SubMConv3d / SparseConv3d /
 * SparseInverseConv3d).
 *
 * Rulebooks are OUTPUT-STATIONARY neighbour tables: nbr[k*ld + o] = input row feeding
 * output row o through kernel offset k (k = (kx*K+ky)*K+kz), or -1.  The canonical
 * (in,out) pair lists of spconv are the non-negative entries of row k in ascending o.
 * =================================================================================== */

/* Number of 32-bit words in the occupancy bitmap of a [B,X,Y,Z] grid. */
size_t gf_index_words(int B, int X, int Y, int Z);
/* Scratch bytes gf_index_build / gf_rules_down2 need for a bitmap of `words` words. */
size_t gf_index_scratch_bytes(size_t words);

/* Build the occupancy-bitmap rank index of a voxel set.
 *   coords  int32 [M,4] (batch,x,y,z), unique rows          (SparseConvTensor.indices)
 *   bitmap  uint32[words]  out      prefix int32[words] out
 *   perm    int32 [M] out: rank (ascending linear index) -> row
 *   d_M     optional device int32 overriding M (M is then a capacity) */
int gf_index_build(const int32_t* coords, int M, const int32_t* d_M, int B, int X, int Y, int Z, uint32_t* bitmap,
                   int32_t* prefix, int32_t* perm, void* scratch, void* stream);

/* Submanifold 3x3x3 (padding 1) neighbour table from an index.
 *   perm may be NULL when rows are already in ascending linear order (levels >= 2).
 *   nbr    int32 [27*ld] out, ld >= M rounded up to 16 (may be NULL when only `steps` is wanted)
 *   gmask  uint32[ceil(M/16)] out: OR of the offsets present in each 16-row group
 *   steps  optional int32 [gf_rules_steps_words(ld)] out: the same relation as a STEP TABLE -- per 16-row group g only
 *          the present offsets, ascending, four steps per 16-byte entry:
 *              steps[((g*7 + s/4)*16 + row)*4 + s%4] = input row of (row, s-th present offset of g) or -1;
 *          the first three blocks of every group are always written (-1 padded).  gf_conv_fwd's counted-loop
 *          kernel reads this instead of nbr. */
size_t gf_rules_steps_words(int ld);
int gf_rules_subm3(const int32_t* coords, int M, const int32_t* d_M, int X, int Y, int Z, const uint32_t* bitmap,
                   const int32_t* prefix, const int32_t* perm, int32_t* nbr, int ld, uint32_t* gmask, int32_t* steps,
                   void* stream);

/* Strided 2x2x2 / stride 2 rulebook: builds the OUTPUT level's index and tables.
 *   in shape (X,Y,Z) -> out shape (X/2,Y/2,Z/2) (floor; inputs mapping outside are dropped)
 *   bitmap_out/prefix_out: index of the output voxel set (rows in ascending linear order)
 *   out_coords int32 [cap,4] out      d_M_out device int32 out (number of output voxels)
 *   child  int32 [8*ld] out: child[k*ld + o] = input row under offset k, or -1
 *   parent int32 [M] out (output row or -1)      koff int32 [M] out (kernel offset 0..7)
 *   up     int32 [8*ld_up] out: one-hot table of the inverse conv (up[k*ld_up+i] = parent[i] iff k==koff[i])
 *   gmask_down uint32[ceil(cap/16)], gmask_up uint32[ceil(M/16)] out */
int gf_rules_down2(const int32_t* coords, int M, const int32_t* d_M, int B, int X, int Y, int Z,
                   uint32_t* bitmap_out, int32_t* prefix_out, void* scratch, int32_t* out_coords,
                   int32_t* d_M_out, int32_t* child, int ld, int32_t* parent, int32_t* koff, int32_t* up, int ld_up,
                   uint32_t* gmask_down, uint32_t* gmask_up, void* stream);

/* The chain of `nlevels` successive down-sampling rulebooks (the six SparseConv3d(k=2,s=2) of the U-Net,
 * geoformer_modules.py:83-96) in one call and one workspace; the tail of small levels runs in a single launch.
 *   plan (host only): offsets[l*10 + f] = int32-element offset into the workspace of field f of level l, f in
 *     (bitmap, prefix, scratch, out_coords[cap_{l+1},4], child[8,cap_{l+1}], parent[cap_l], koff[cap_l],
 *      up[8,cap_l], gmask_down[cap_{l+1}/16], gmask_up[cap_l/16]);  caps[l] = capacity (leading dimension) of
 *     level l (caps[0] = M0 rounded up to 16), shapes[3*l..] = grid of level l; *ws_elems = workspace size in
 *     int32 elements; *nlevels_out = levels actually built (stops when a grid side drops below 2).
 *   chain: counts[l+1] (device) = number of voxels of level l+1; tables as gf_rules_down2 writes them. */
int gf_rules_down2_chain_plan(int M0, int B, int X, int Y, int Z, int nlevels, long long* offsets, int* caps, int* shapes,
                              long long* ws_elems, int* nlevels_out);
int gf_rules_down2_chain(const int32_t* coords, int M0, int B, int X, int Y, int Z, int nlevels, int32_t* ws,
                         int32_t* counts, void* stream);

/* Weight pre-packing: spconv's parameter layout [k,k,k,Cin,Cout] (= [K,Cin,Cout], kept as the
 * state-dict layout, checkpoint.py:47-49) -> the per-lane MFMA B-operand order the conv kernel
 * streams with one 16-byte load per lane.  Wp holds gf_conv_packed_floats(K,Cin,Cout) floats
 * (channels zero-padded to multiples of 16). */
size_t gf_conv_packed_floats(int K, int Cin, int Cout);
int gf_conv_pack_weights(const float* W, int K, int Cin, int Cout, float* Wp, void* stream);
/* Packed weights of the INPUT GRADIENT's convolution in one launch: the pack of W'[k] = W[flip ? K-1-k : k]^T, a
 * [K,Cout,Cin] operand (gf_conv_packed_floats(K, Cout, Cin) floats); flip = 1 for submanifold tables (gf_conv_wgrad's
 * comment: weights W[K-1-k]^T over the same table), 0 for the child <-> up tables. */
int gf_conv_pack_weights_t(const float* W, int K, int Cin, int Cout, int flip, float* Wp, void* stream);

/* FLAT STEP TABLE of a [K,ld] relation (nbr + gmask as gf_rules_subm3 / gf_rules_down2 write them; the reference's
 * spconv keeps this as its indice pairs, spconv/ops.py get_indice_pairs via geoformer_modules.py:52-129): one 64-byte
 * record of 16 input rows per (16-row group, PRESENT offset), group-major and offset-ascending, behind a header, the
 * per-group step offsets and a bin table: the groups sorted by their number of steps (descending) and dealt to `nbins`
 * bins (0 = the default 1024 = one per SIMD; gf_conv_fwd_flat expects the default) in snake order, one
 * {group, first step, steps, offset mask} descriptor per (round, bin).  gf_conv_fwd_flat's LDS-weight kernel walks a
 * bin's records linearly.  K <= 31.  flat: int32 [gf_rules_flat_words(K, ld)] out. */
size_t gf_rules_flat_words(int K, int ld);
int gf_rules_flat_steps(const int32_t* nbr, const uint32_t* gmask, int K, int M, int ld, int nbins, int32_t* flat,
                        void* stream);

/* Forward gather-GEMM (output-stationary): out[o,:] = sum_k act(in[nbr[k][o],:]) @ W[k] (+ residual[o,:])
 *   in fp32 [M_in,Cin]   Wp = packed W (gf_conv_pack_weights)   out fp32 [M_out,Cout]
 *   (M_in bounds the buffer descriptor the gathers go through)
 *   nbr may be NULL with K == 1 (identity map: plain GEMM, the k=1 "i_branch" conv)
 *   in_scale/in_shift  optional fp32 [Cin]: act(x) = max(x*scale + shift, 0) fused on the
 *                      gathered rows (eval-mode BatchNorm1d + ReLU, geoformer_modules.py:19-26)
 *   residual           optional fp32 [M_out,Cout] added in the epilogue (geoformer_modules.py:33)
 *   out_scale/out_shift optional fp32 [Cout], 16-byte aligned: out = max(out*scale + shift, 0) in the epilogue (the
 *                      CONSUMER's BatchNorm + ReLU applied once per output element; gf_resblock_fwd uses it for bn1)
 *   steps              optional step table of the same relation (gf_rules_subm3); nbr may then be NULL for the
 *                      launch shapes that read it (16 output channels, Cin 16 or 32, level-1 sized inputs)
 */
int gf_conv_fwd(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask, const int32_t* steps, int K,
                int M_in, int M_out, int ld, int Cin, int Cout, const float* in_scale, const float* in_shift,
                const float* residual, const float* out_scale, const float* out_shift, float* out, void* stream);

/* gf_conv_fwd with two outputs: out = the raw sums (+ residual), out_act = max(out*out_scale + out_shift, 0).  A
 * pre-activation residual block (geoformer_modules.py:10-35) reads its input twice -- raw as the residual operand,
 * through BatchNorm + ReLU as the first convolution's input -- so the producer writes both and the consumer's
 * convolution gathers activated rows (one activation per element instead of one per gathered element).  Implemented for
 * the level-1 launch shape (16 output channels over a step table); gf_conv_dual_supported() tells beforehand. */
int gf_conv_dual_supported(int M_out, int ld, int Cin, int Cout, int has_steps);
int gf_conv_fwd_dual(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask, const int32_t* steps,
                     int K, int M_in, int M_out, int ld, int Cin, int Cout, const float* in_scale, const float* in_shift,
                     const float* residual, const float* out_scale, const float* out_shift, float* out, float* out_act,
                     void* stream);

/* gf_conv_fwd with the relation's flat step table (gf_rules_flat_steps; NULL = none) and an optional second output
 * (out_act as gf_conv_fwd_dual; NULL = one output).  With a table, the shapes whose packed weights fit the LDS
 * (K = 27, 16 / 32 output channels, 16-channel multiples in) run the LDS-weight kernel, which also has both outputs;
 * every other shape is gf_conv_fwd / gf_conv_fwd_dual (same results to fp32 summation order). */
int gf_conv_fwd_flat(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask, const int32_t* steps,
                     const int32_t* flat, int K, int M_in, int M_out, int ld, int Cin, int Cout, const float* in_scale,
                     const float* in_shift, const float* residual, const float* out_scale, const float* out_shift,
                     float* out, float* out_act, void* stream);

/* Pre-activation residual block (ResidualBlock, model/geoformer/geoformer_modules.py:10-35) in eval mode, one
 * call:  out = conv1(relu(bn1(conv0(relu(bn0(x)))))) + (Wpi ? x . Wi : x).
 *   x fp32 [M,Cin]; Wp0 (K,Cin,Cout), Wp1 (K,Cout,Cout), Wpi (1,Cin,Cout) or NULL: gf_conv_pack_weights output;
 *   nbr/gmask/K/ld: the level's submanifold table; s0,t0 [Cin], s1,t1 [Cout]: folded BatchNorm (scale, shift);
 *   tmp, idn (NULL iff Wpi NULL), out fp32 [M,Cout]. */
int gf_resblock_fwd(const float* x, const float* Wp0, const float* Wp1, const float* Wpi, const int32_t* nbr,
                    const uint32_t* gmask, const int32_t* steps, int K, int M, int ld, int Cin, int Cout, const float* s0, const float* t0,
                    const float* s1, const float* t1, float* tmp, float* idn, float* out, void* stream);

/* Weight gradient of the same operator: dW[k] = sum_o in[nbr[k][o],:]^T dOut[o,:]  (dW fp32
 * [K,Cin,Cout], zeroed by the call).  The input gradient needs no entry point of its own: it is
 * gf_conv_fwd over the transposed table with per-offset transposed weights
 * (submanifold: same table, weights W[K-1-k]^T; strided/inverse: child <-> up tables). */
int gf_conv_wgrad(const float* in, const float* dout, const int32_t* nbr, int K, int M_out, int ld, int Cin, int Cout,
                  float* dW, void* stream);
/* The same with the table's group masks (gmask as gf_conv_fwd takes it, K <= 32): (group, offset) pairs without a
 * single neighbour are skipped -- 61 % of them on a scanned room's level 1.  gmask or nbr NULL: gf_conv_wgrad. */
int gf_conv_wgrad_masked(const float* in, const float* dout, const int32_t* nbr, const uint32_t* gmask, int K, int M_out,
                         int ld, int Cin, int Cout, float* dW, void* stream);
/* dW += ... : the same without the zero fill (a caller that clears all of a step's weight gradients at once). */
int gf_conv_wgrad_masked_acc(const float* in, const float* dout, const int32_t* nbr, const uint32_t* gmask, int K,
                             int M_out, int ld, int Cin, int Cout, float* dW, void* stream);

/* The sparse U-Net in TRAINING mode (batch statistics, saved activations, backward) as a layer program run from native
 * code: GeoFormer.input_conv -> UBlock x7 -> output_layer with requires_grad (model/geoformer/geoformer.py:39-53,
 * 398-401; ResidualBlock / UBlock: model/geoformer/geoformer_modules.py:10-35,52-129; the training loop's backward:
 * train.py:63-75).  The host compiles the module tree ONCE into a list of ops over numbered feature buffers; a call
 * runs a contiguous range of ops (the two voxel transformers of the deepest levels stay framework modules, so a step
 * is three ranges) with exactly the launches the per-module route makes through gf_conv_fwd, gf_conv_pack_weights(_t),
 * gf_conv_wgrad_masked and gf_bn_relu_train_fwd / _bwd -- without ~1 500 framework calls and ~230 autograd nodes per
 * step.
 *   op kinds     0  dst = relu(batchnorm(src))      batch statistics; running statistics updated; mean / invstd saved
 *                1  dst = conv(src) (+ aux)         table 0: 1x1x1 (plain rows), 1: submanifold 3x3x3 of `level`,
 *                                                   2: 2x2x2 stride 2 from `level` to level+1, 3: its inverse
 *                2  dst = [src | aux]               the skip concatenation (UBlock.forward, geoformer_modules.py:116)
 *   buffers      act[i] / grad[i]: fp32 [rows(level of i), C of i], 16-byte aligned, caller-owned for the whole step
 *   backward     ops in reverse; an op ACCUMULATES into a source's gradient when ghas[src] != 0 and sets ghas[src];
 *                a residual operand's gradient is the output's gradient itself: grad[aux] is re-pointed at grad[dst]
 *                (copied when ghas[dst] == 2: a gradient tensor the caller does not own).  Parameter gradients land
 *                in pgrad at the op's pgrad_off: dW [K,Cin,Cout], or dgamma [C] followed by dbeta [C]. */
typedef struct GfTrainOp {
    int kind, level, table, src, dst, aux, Cin, Cout, no_dgrad, pad_;
    const float* w;                                 /* convolution weights [K,Cin,Cout] */
    const float *gamma, *beta;                      /* BatchNorm */
    float *running_mean, *running_var;
    float eps, momentum;
    long long wp_off;                               /* packed forward weights: floats into `wp` */
    long long pgrad_off;                            /* floats into `pgrad` */
    long long stats_off;                            /* save_mean [C] then save_invstd [C]: floats into `stats` */
} GfTrainOp;
typedef struct GfTrainLevel {
    int M, ld;                                      /* voxels of the level; leading dimension of its 27-offset table */
    const int32_t* nbr; const uint32_t* gmask; const int32_t* steps;
    int M_coarse, ld_down;                          /* the level below: rows, leading dimension of child */
    const int32_t* child; const uint32_t* gmask_down;
    int ld_up, pad_; const int32_t* up; const uint32_t* gmask_up;
    const int32_t* flat;                            /* flat step table of the 27-offset relation (gf_rules_flat_steps) or NULL */
} GfTrainLevel;
/* floats of `scratch` a range of ops needs (BatchNorm partials + the transposed weight pack of the widest op) */
size_t gf_unet_train_scratch_floats(const GfTrainOp* ops, int nops, const GfTrainLevel* levels);
int gf_unet_train_fwd(const GfTrainOp* ops, int op_begin, int op_end, const GfTrainLevel* levels, float* const* act,
                      float* wp, float* stats, float* scratch, void* stream);
int gf_unet_train_bwd(const GfTrainOp* ops, int op_begin, int op_end, const GfTrainLevel* levels, float* const* act,
                      float** grad, unsigned char* ghas, const float* stats, float* pgrad, float* scratch, void* stream);

/* The whole sparse U-Net of the eval forward in one call: input conv -> UBlock x nlevels -> output BatchNorm + ReLU
 * (GeoFormer.input_conv / unet / output_layer, model/geoformer/geoformer.py:42-53,398-401; UBlock and ResidualBlock,
 * model/geoformer/geoformer_modules.py:10-35,52-129; two blocks per level before and after the inner level, all
 * BatchNorm in eval mode).  Issues exactly the launches the per-module route makes through gf_index_build,
 * gf_rules_subm3, gf_rules_down2_chain, gf_conv_fwd, gf_resblock_fwd and gf_backbone_transformer, from native code
 * and out of one workspace; the rulebook chain runs on `side_stream` beside the level-1 convolutions and the call
 * waits once on the host for the chain's voxel counts.
 *   All pointers inside the parameter structs are DEVICE pointers except tr_params (host array of device pointers,
 *   the table gf_backbone_transformer takes); packed weights come from gf_conv_pack_weights; (s, t) pairs are
 *   eval-mode BatchNorm folded to y = max(x*s + t, 0) and must be 16-byte aligned. */
#define GF_UNET_MAX_LEVELS 8
typedef struct GfResBlockParams {
    const float *wp0, *wp1, *wpi;   /* conv_branch.2 / conv_branch.5 / i_branch.0 weights; wpi NULL iff Cin == Cout */
    const float *s0, *t0, *s1, *t1; /* conv_branch.0 [Cin] and conv_branch.3 [Cout] */
} GfResBlockParams;
typedef struct GfUnetLevelParams {
    int C;                          /* channel width of the level (multiple of 16) */
    int tr_layers;                  /* > 0: the level ends with the voxel transformer of that many layers */
    GfResBlockParams blocks[2];     /* UBlock.blocks: C -> C */
    GfResBlockParams tail[2];       /* UBlock.blocks_tail: 2C -> C, C -> C (unused on the deepest level) */
    const float *down_wp, *down_s, *down_t; /* UBlock.conv: BN(C) + ReLU + SparseConv3d(C, C_next, k=2, s=2) */
    const float *up_wp, *up_s, *up_t;       /* UBlock.deconv: BN(C_next) + ReLU + SparseInverseConv3d(C_next, C, k=2) */
    const float* const* tr_params;  /* see gf_backbone_transformer */
} GfUnetLevelParams;
typedef struct GfUnetParams {
    int nlevels;                    /* 7 in GeoFormer */
    int cin;                        /* channels of the voxel features handed in (<= 16) */
    const float* input_wp;          /* packed [27,16,16] input-conv weights, input channels zero-padded to 16 */
    const float *out_s, *out_t;     /* output_layer BatchNorm folded, [16] */
    GfUnetLevelParams level[GF_UNET_MAX_LEVELS];
} GfUnetParams;
/* bytes of workspace gf_unet_fwd needs for a scene of M0 voxels on a [B,X,Y,Z] grid (capacity bound, host only) */
size_t gf_unet_ws_bytes(const GfUnetParams* P, int M0, int B, int X, int Y, int Z);
/*   feats fp32 [M0,cin] voxel features, coords int32 [M0,4] (b,x,y,z) unique rows, ws 256-byte aligned device
 *   workspace, host_counts PINNED host int32 [nlevels]: receives the voxel count of every level,
 *   out fp32 [M0,16]; side_stream may be NULL (everything on `stream`).  Everything queued on side_stream is
 *   joined into `stream` before the call returns control of the tables to later launches. */
int gf_unet_fwd(const GfUnetParams* P, const float* feats, const int32_t* coords, int M0, int B, int X, int Y, int Z,
                void* ws, size_t ws_bytes, int32_t* host_counts, float* out, void* stream, void* side_stream);
/* The same in two phases, for the staggered serving loop (geoformer_amd/serving.py):
 *   phase A = level-1 index and table, the first level's convolutions (input conv + two residual blocks) and the
 *             down-sampling rulebook chain, all queued without a host wait; the CONVOLUTIONS wait for the n_gate recorded
 *             events (hipEvent_t handles), the integer work does not;
 *   `between(user, events_out, max_events)` is then called on the host (may be NULL): it may queue any other work on
 *             other streams and returns the number of recorded events it stored in events_out (<= max_events = 8), or a
 *             negative number to abort the call (GF_ERR_CALLBACK);
 *   phase B = everything below the first level and the whole up pass, behind those events.
 * The loop uses `between` to issue the PREVIOUS scene's sampling / BFS stretch (whose events phase B then waits for):
 * this scene's phase A runs under that scene's foreground read-back and first sampling picks, when the chip is idle. */
typedef int (*GfUnetBetween)(void* user, void** events_out, int max_events);
int gf_unet_fwd_phased(const GfUnetParams* P, const float* feats, const int32_t* coords, int M0, int B, int X, int Y, int Z,
                       void* ws, size_t ws_bytes, int32_t* host_counts, float* out, void* stream, void* side_stream,
                       void* const* gate_events, int n_gate, GfUnetBetween between, void* user);

/* The same with the rulebooks AHEAD of `stream`: index, tables and the down-sampling chain read the voxel coordinates and
 * nothing else.  input_events: the n_input (>= 0) recorded hipEvent_t the COORDINATES wait for (none: they have been
 * resident all along); every rulebook launch goes to side_stream behind those events and behind the end of this host
 * thread's previous gf_unet_fwd* call (which read the same workspace) -- not behind what `stream` still has queued.  In a
 * loop of forwards they then run under the previous scene's sampling / BFS stretch (test.py:52-96 calls the model scene
 * after scene).  `feats` is read on side_stream as well (the zero-padded input rows are made there): it waits for the same
 * events, or was produced on side_stream itself.  Same launches and results; side_stream NULL: gf_unet_fwd. */
int gf_unet_fwd_ahead(const GfUnetParams* P, const float* feats, const int32_t* coords, int M0, int B, int X, int Y, int Z,
                      void* ws, size_t ws_bytes, int32_t* host_counts, float* out, void* stream, void* side_stream,
                      void* const* input_events, int n_input);

/* ===================================================================================
 * Training criterion: Hungarian matching on the device (model/matcher.py:79-126 moves the cost matrix to the host
 * and calls scipy.optimize.linear_sum_assignment per scene)
 * =================================================================================== */

/* Linear sum assignment of queries to ground-truth instances, minimising the total cost.
 *   cost fp32 [nq, K] row-major (device); present int32 [K]: instances that take part (ascending order = the column
 *   order scipy sees); out: match_q int32 [K] = query assigned to instance k (-1: absent or left over),
 *   match_of_q int32 [nq] = instance of query q (-1: none), n_match int32 [1] = min(nq, number present),
 *   status int32 [1]: 0 ok, 1 problem larger than the kernel's tables (512 x 1024), 2 infeasible (non-finite costs).
 * Same solver as scipy (shortest augmenting paths in float64, same scan order and tie-breaking, transposed when there
 * are fewer instances than queries), so the assignment equals scipy's on the same fp32 matrix. */
int gf_lsap(const float* cost, int nq, int K, const int32_t* present, int32_t* match_q, int32_t* match_of_q,
            int32_t* n_match, int32_t* status, void* stream);

/* Dice and focal loss of one scene's matched (query, instance) pairs for one decoder layer
 * (compute_dice_loss / compute_sigmoid_focal_loss on the matched rows, /root/reference/criterion.py:26-58,137-190), and
 * their gradient.  mask_logits fp32 [nq,n]; inst_masks fp32 0/1 [K,n]; match_q [K] / match_of_q [nq] / n_match [1] as
 * gf_lsap wrote them.  fwd: sums fp32, gf_pair_losses_sums_floats(K) floats (first [K,4] row sums: p t, p, t, focal
 * term -- kept for the backward --, then the partial sums of the row segments),
 * out[0] = sum_k dice_k / (n_match + 1e-6), out[1] = sum_k mean_j focal_kj / (n_match + 1e-6).
 * bwd: grad_out fp32 [2] (d loss / d out), d_logits fp32 [nq,n] written in full (zero rows for unmatched queries). */
size_t gf_pair_losses_sums_floats(int K);
int gf_pair_losses_fwd(const float* mask_logits, const float* inst_masks, const int32_t* match_q, int nq, int K, int n,
                       const int32_t* n_match, float* sums, float* out, void* stream);
int gf_pair_losses_bwd(const float* mask_logits, const float* inst_masks, const int32_t* match_of_q, const float* sums,
                       int nq, int K, int n, const int32_t* n_match, const float* grad_out, float* d_logits, void* stream);

/* Training-mode BatchNorm1d (+ ReLU) over voxel rows x[M,C], forward and backward: the pre-activation pair in front of
 * every sparse convolution of the U-Net (geoformer_modules.py:10-35,52-129; BatchNorm1d(eps 1e-4, momentum 0.1),
 * geoformer.py:39), three launches per direction (csrc/bn_train.hip).  C a multiple of 4, M >= 2, pointers 16-byte aligned.
 *   fwd: y = relu((x - mean) * invstd * gamma + beta) with the batch's biased variance; running_mean / running_var
 *        (optional, both or none) updated in place with `momentum` (unbiased variance), save_mean / save_invstd [C]
 *        written for the backward.
 *   bwd: dx (optional) [M,C], dgamma / dbeta (optional) [C] from x, y (the forward's output: ReLU mask), dy.
 * scratch: gf_bn_train_scratch_floats(M, C) floats. */
size_t gf_bn_train_scratch_floats(int M, int C);
int gf_bn_relu_train_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                         int relu, float* running_mean, float* running_var, float* y, float* save_mean,
                         float* save_invstd, float* scratch, void* stream);
int gf_bn_relu_train_bwd(const float* x, const float* y, const float* dy, int M, int C, const float* gamma,
                         const float* save_mean, const float* save_invstd, int relu, float* dx, float* dgamma,
                         float* dbeta, float* scratch, void* stream);
/* gf_bn_relu_train_bwd with dx = (the layer's input gradient) + addend[M,C] (the gradient the input already received
 * through another consumer, e.g. a residual block's identity branch); addend may be NULL or dx itself. */
int gf_bn_relu_train_bwd_add(const float* x, const float* y, const float* dy, int M, int C, const float* gamma,
                             const float* save_mean, const float* save_invstd, int relu, const float* addend, float* dx,
                             float* dgamma, float* dbeta, float* scratch, void* stream);

/* The same pair for the channel-major layouts x[B,C,L]: nn.BatchNorm1d over [B,C,L] (semantic head, mask tower:
 * geoformer.py:55-70, B = 1, L = number of points) and nn.BatchNorm2d over [B,C,H,W] with L = H*W (the set-abstraction
 * MLP, lib/pointnet2/pytorch_utils.py:9-32).  No alignment requirement on L.  scratch: gf_bn_train_cl_scratch_floats. */
size_t gf_bn_train_cl_scratch_floats(int B, int C, long long L);
int gf_bn_relu_train_cl_fwd(const float* x, int B, int C, long long L, const float* gamma, const float* beta, float eps,
                            float momentum, int relu, float* running_mean, float* running_var, float* y,
                            float* save_mean, float* save_invstd, float* scratch, void* stream);
int gf_bn_relu_train_cl_bwd(const float* x, const float* y, const float* dy, int B, int C, long long L,
                            const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dx,
                            float* dgamma, float* dbeta, float* scratch, void* stream);

/* ===================================================================================
 * PG_OP (lib/pointgroup_ops/src/pointgroup_ops_api.cpp:6-23)
 * =================================================================================== */

/* voxelize_idx on the GPU (reference: PG_OP.voxelize_idx, lib/pointgroup_ops/src/voxelize/voxelize.cpp:10-152, CPU):
 * voxel ids in order of first occurrence, input_map[N] point -> voxel, rule rows [count, point ids ascending, 0 pad]
 * (modes 3/4) or [1, front()/back()] (modes 0,1 / 2), output coords = coords of rule[1].
 *   coords int64 [N,ncol] on the device (ncol 4 = (b,x,y,z) or 3), every field in [0, 65535];
 *   _count: fills input_map int32 [N] and d_M_maxActive int32[3] = {M, maxActive, error flag} (device);
 *   _fill (after the caller read M and maxActive): output_coords int64 [M,ncol], output_map int32 [M,1+maxActive];
 *   the same scratch (gf_voxelize_idx_scratch_bytes(N)) must be handed to both calls. */
size_t gf_voxelize_idx_scratch_bytes(int N);
int gf_voxelize_idx_count(const long long* coords, int N, int ncol, int mode, void* scratch, int32_t* input_map,
                          int32_t* d_M_maxActive, void* stream);
int gf_voxelize_idx_fill(const long long* coords, int N, int ncol, int mode, void* scratch, const int32_t* input_map,
                         int M, int maxActive, long long* out_coords, int32_t* out_map, void* stream);

/* The host side of feeding a scene (what the reference's drivers do with blocking `.cuda()` calls from their own thread,
 * test.py:56 / train.py:63-75, after a host voxelize_idx in the dataset's collate, datasets/scannetv2_inst.py:389-455) on
 * a NATIVE worker thread: per job, n_copies x (memcpy src -> pinned, hipMemcpyAsync pinned -> dev on `stream`), an event
 * ("copied": the pinned buffers may be rewritten), then -- coords_dev != NULL -- gf_voxelize_idx_count on the uploaded
 * coordinates, the read-back of its three words into head_host (pinned) and a second event ("head").  gf_feeder_submit
 * returns at once; the caller's thread never touches the bytes.  slot 0..3 names the job's state and events; a slot takes
 * a new job after gf_feeder_wait_issued(slot).  Waits release the interpreter lock when called through ctypes. */
#define GF_FEEDER_MAX_COPIES 16
typedef struct {
    int slot, n_copies;
    const void* src[GF_FEEDER_MAX_COPIES];
    void* pinned[GF_FEEDER_MAX_COPIES];
    void* dev[GF_FEEDER_MAX_COPIES];
    size_t bytes[GF_FEEDER_MAX_COPIES];
    const long long* coords_dev; /* device int64 [N,ncol]: one of the `dev` targets, or NULL (no voxelisation) */
    int N, ncol, mode, pad_;
    void* scratch;               /* gf_voxelize_idx_scratch_bytes(N) */
    int32_t* input_map;          /* device int32 [N] */
    int32_t* head_dev;           /* device int32 [3] */
    int32_t* head_host;          /* pinned int32 [3] */
    void* stream;
} GfFeederJob;
void* gf_feeder_create(int device);
int gf_feeder_submit(void* feeder, const GfFeederJob* job);
int gf_feeder_wait_issued(void* feeder, int slot); /* the job's calls are queued on its stream (or: its error status) */
int gf_feeder_wait_copied(void* feeder, int slot); /* after wait_issued: the uploads have left the pinned buffers */
int gf_feeder_wait_head(void* feeder, int slot);   /* after wait_issued: head_host holds {M, maxActive, error} */
int gf_feeder_destroy(void* feeder);

/* PG_OP.voxelize_fp (voxelize.cu:9-31): out[row,:] = sum_i mult * feats[rules[row,i],:], i in rule
 * order, mult = 1/count when average (mode 4).  rules int32 [M, 1+maxActive].  out fp32 [M,C]. */
int gf_voxelize_fp(const float* feats, const int32_t* rules, int M, int maxActive, int C, int average, float* out,
                   void* stream);
/* PG_OP.voxelize_bp / point_recover_fp (voxelize.cu:34-53): d_feats[rules[row,i],:] += mult*d_out[row,:];
 * d_feats must be zeroed by the caller (pointgroup_ops.py:69). */
int gf_voxelize_bp(const float* d_out, const int32_t* rules, int M, int maxActive, int C, int average, float* d_feats,
                   void* stream);

/* ===================================================================================
 * pointnet2._ext (lib/pointnet2/_ext_src/src/bindings.cpp:8-21)
 * =================================================================================== */

/* gather_points (sampling_gpu.cu:11-33): out[b,c,j] = points[b,c,idx[b,j]] */
int gf_gather_points(const float* points, const int32_t* idx, int b, int c, int n, int m, float* out, void* stream);
/* gather_points_grad (sampling_gpu.cu:37-60): grad_points[b,c,idx[b,j]] += grad_out[b,c,j] (zeroed by caller) */
int gf_gather_points_grad(const float* grad_out, const int32_t* idx, int b, int c, int n, int m, float* grad_points,
                          void* stream);
/* group_points (group_points_gpu.cu:11-42): out[b,c,j,s] = points[b,c,idx[b,j,s]] */
int gf_group_points(const float* points, const int32_t* idx, int b, int c, int n, int npoints, int nsample, float* out,
                    void* stream);
/* group_points_grad (group_points_gpu.cu:46-78) */
int gf_group_points_grad(const float* grad_out, const int32_t* idx, int b, int c, int n, int npoints, int nsample,
                         float* grad_points, void* stream);
/* ball_query (ball_query_gpu.cu:12-57): first nsample indices in ascending order with d2 < radius^2,
 * padded with the first hit; rows without a hit are all zero.  idx int32 [b,m,nsample] (fully written). */
int gf_ball_query(const float* new_xyz, const float* xyz, int b, int n, int m, float radius, int nsample, int32_t* idx,
                  void* stream);
/* furthest_point_sampling (sampling_gpu.cu:72-232) incl. the |p|^2 <= 1e-3 skip, the m > n padding
 * and the launch-geometry tie-break of the reference.  xyz fp32 [b,n,3] -> idxs int32 [b,m].
 * scratch: gf_fps_scratch_bytes(b) bytes (zeroed by the call). */
size_t gf_fps_scratch_bytes(int b);
int gf_furthest_point_sampling(const float* xyz, int b, int n, int m, int32_t* idxs, void* scratch, void* stream);
/* Same sequence, continued: idxs[b, 0..m_known) already hold its first m_known picks (an earlier call with a
 * smaller m on the same points); fills idxs[b, m_known..m).  Lets a consumer of the first picks (the geodesic BFS
 * needs 256 of 2048) start while the rest is still being drawn. */
int gf_furthest_point_sampling_resume(const float* xyz, int b, int n, int m, int m_known, int32_t* idxs,
                                      void* scratch, void* stream);

/* ===================================================================================
 * Geodesic stage (model/geoformer/geodesic_utils.py)
 * =================================================================================== */

/* Radius-limited kNN graph (stands in for find_knn, geodesic_utils.py:11-24, on the geodesic path):
 * for every point the k nearest points with sqrt(d2) <= radius, ordered by (d2, index), itself
 * included (normally column 0); missing entries are (inf, -1).
 *   D fp32 [n,k] (sqrt(d2) when sqrt_out, else squared like faiss)   I int32 [n,k]
 *   deg int32 [n] optional: number of valid entries after column 0
 *   scratch: gf_knn_scratch_bytes(n); gf_knn_error_flag() points at a device int set to 1 when some
 *   point had more in-radius neighbours than the kernel's list capacity (rows then truncated). */
size_t gf_knn_scratch_bytes(int n);
int gf_knn_radius(const float* xyz, int n, int k, float radius, int sqrt_out, float* D, int32_t* I, int32_t* deg,
                  void* scratch, void* stream);
const int32_t* gf_knn_error_flag(void* scratch, int n);

/* Hop-synchronous geodesic BFS (cal_geodesic_vectorize, geodesic_utils.py:91-164) for nq sources of
 * one scene.  D/I are kNN rows INCLUDING column 0 (which is skipped, :110-111), D already sqrt'ed.
 *   geo fp32 [nq,n] out (-1 = not reached within max_step hops)
 *   keys_ws: nq*n uint64, queue_ws: nq * gf_geodesic_bfs_queue_words(n) int32 (scratch).
 *   D/I rows must be sorted by distance with (inf,-1) padding (the order gf_knn_radius and faiss emit). */
size_t gf_geodesic_bfs_queue_words(int n);
int gf_geodesic_bfs(const float* D, const int32_t* I, const int32_t* deg, int n, int K, const int32_t* src, int nq,
                    float radius, int max_step, float* geo, void* keys_ws, void* queue_ws, void* stream);
/* Same, with the workgroup size per query chosen by the caller: wg_threads = 1024 (one query per compute unit,
 * fastest when the launch has the device to itself), 512 or 256 (several queries share a compute unit, so the
 * launch fits beside another resident kernel -- the host runs it next to furthest point sampling).
 * queue_words: int32 words of queue_ws per query as allocated (checked: at least gf_geodesic_bfs_queue_words(n) = 4 n).
 * (ABI 3.) */
int gf_geodesic_bfs_cfg(const float* D, const int32_t* I, const int32_t* deg, int n, int K, const int32_t* src, int nq,
                        float radius, int max_step, float* geo, void* keys_ws, void* queue_ws, size_t queue_words,
                        int wg_threads, void* stream);
/* ===================================================================================
 * Mask head (GeoFormer.mask_heads_forward, model/geoformer/geoformer.py:286-324), fused
 * =================================================================================== */

/* logits[q,p] = W2_q relu(W1_q [rel(q,p) ; feat_p] + b1_q) + b2_q,  rel = qxyz_q - coords_p and, where
 * geo[q,p] < 0, rel += sqrt_max_geo[q] * sign(rel).
 *   feat fp32 [N,C] (C = 16), coords fp32 [N,3], geo fp32 [nq,N] or NULL (then no fix-up),
 *   qxyz fp32 [nq,3], sqrt_max_geo fp32 [nq] (NULL iff geo NULL),
 *   w1 fp32 [nq,C,3+C] (row c = [3 coordinate taps, C feature taps], the layout
 *   parse_dynamic_params produces, geoformer.py:264-284), b1 [nq,C], w2 [nq,C], b2 [nq]
 *   out fp32 [nq,N]. */
int gf_mask_head(const float* feat, const float* coords, const float* geo, const float* qxyz,
                 const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2, const float* b2, int N,
                 int nq, int C, float* out, void* stream);
/* Same, with w1 / b1 / w2 / b2 pointing INTO one parameter matrix of row stride ldp floats (the controller's
 * [nq, 16*19+16+16+1] output, split as geoformer.py:264-284 does: w1 | w2 | b1 | b2), read in place; ldp = 0: dense. */
int gf_mask_head_packed(const float* feat, const float* coords, const float* geo, const float* qxyz,
                        const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2, const float* b2,
                        int ldp, int N, int nq, int C, float* out, void* stream);
/* E episodes over ONE scene in one launch: the few-shot test loop re-queries a cached scene once per (label, run)
 * (test_fs.py:157-174; GeoFormerFS.get_mask_prediction, model/geoformer/geoformer_fs.py:326-355 once per call).
 * Parameters [E*nq, ...] and logits out fp32 [E*nq, N] are episode-major (row e*nq + q); feat / coords are the
 * scene's, geo [nq,N], qxyz [nq,3] and sqrt_max_geo [nq] the scene's queries', shared by every episode.  Row e*nq + q
 * of `out` equals what gf_mask_head_packed writes to row q for episode e's parameters (split_ws = NULL).
 * split_ws: NULL, or gf_mask_head_split_bytes(N) bytes of scratch (8-byte aligned): the 16-channel feature product then
 * runs on the bf16 matrix pipe over the EXACT three-piece bf16 split of both fp32 operands (six of the nine piece products;
 * what is dropped is below 3 * 2^-24 of the result: one fp32 rounding), 1.3x faster; NULL keeps the fp32 MFMA. */
size_t gf_mask_head_split_bytes(int N);
int gf_mask_head_episodes(const float* feat, const float* coords, const float* geo, const float* qxyz,
                          const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2, const float* b2,
                          int ldp, int N, int nq, int E, int C, void* split_ws, float* out, void* stream);

/* Backward of the fused mask head (training): given gout = dL/dlogits fp32 [nq,N], writes the gradient of the packed
 * per-query parameters (w1 | w2 | b1 | b2 columns, row stride ldp >= 337; the w1/b1/w2 pointers point into the packed
 * parameter matrix like for gf_mask_head_packed) into dparams fp32 [nq,ldp] and of the mask features into dfeat fp32
 * [N,16] (zero on entry).  The hidden activations are recomputed, nothing of the forward is kept; no gradient flows
 * into coords / geo / qxyz (geoformer.py:296-310 treats them as data).  scratch: gf_mask_head_bwd_scratch_floats. */
size_t gf_mask_head_bwd_scratch_floats(int N, int nq);
int gf_mask_head_bwd(const float* feat, const float* coords, const float* geo, const float* qxyz,
                     const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2, const float* gout,
                     int ldp, int N, int nq, int C, float* dparams, float* dfeat, float* scratch, void* stream);
/* The same for E episodes over one scene (the decoder layers of a training step, geoformer.py:286-324 once per layer:
 * same features / coordinates / geodesic rows / query positions, E sets of generated parameters): parameters, gout and
 * dparams have E * nq rows, episode-major; dfeat is the sum over all of them; scratch:
 * gf_mask_head_bwd_scratch_floats(N, E * nq). */
int gf_mask_head_bwd_episodes(const float* feat, const float* coords, const float* geo, const float* qxyz,
                              const float* sqrt_max_geo, const float* w1, const float* b1, const float* w2,
                              const float* gout, int ldp, int N, int nq, int E, int C, float* dparams, float* dfeat,
                              float* scratch, void* stream);

/* ===================================================================================
 * Per-point MLP chains of the eval forward, fused: mask_tower (geoformer.py:64-71), semantic head
 * (geoformer.py:54-62): Conv1d(k=1)/Linear + eval BatchNorm1d + ReLU stacks over the rows of x
 * =================================================================================== */

/* out[p,:] = L_n(... L_1(x[p,:])) with L_l(h) = act_l((W_l h) * scale_l + shift_l), act = ReLU where relu[l] != 0.
 * The caller folds bias and eval-mode BatchNorm into (scale, shift).
 *   x fp32 [N, channels[0]], out fp32 [N, channels[n_layers]], n_layers <= 4,
 *   W / scale / shift: HOST arrays of n_layers DEVICE pointers (W_l row-major [channels[l+1], channels[l]]),
 *   channels: HOST int[n_layers+1] (inputs multiples of 16, all widths <= 64; the last output width is free), relu: HOST int[n_layers]. */
int gf_pointwise_mlp(const float* x, int N, int n_layers, const float* const* W, const float* const* scale,
                     const float* const* shift, const int* channels, const int* relu, float* out, void* stream);
/* Same with a row indirection of the input: out[p] = MLP(x[rows[p]]), rows int32 [N] (the semantic head reading the
 * voxel features through p2v_map, geoformer.py:541-547, without materialising the gathered tensor). */
int gf_pointwise_mlp_rows(const float* x, const int32_t* rows, int N, int n_layers, const float* const* W,
                          const float* const* scale, const float* const* shift, const int* channels, const int* relu,
                          float* out, void* stream);

/* Set-abstraction MLP + max-pool, fused (PointnetSAModuleVotes: SharedMLP + max_pool2d over the samples,
 * lib/pointnet2/pointnet2_modules.py:335-349):  out[b,:,i] = max_s L_n(...L_1(grouped[b,:,i,s])), layers as above
 * (Conv2d 1x1 + eval BatchNorm2d + ReLU folded into W / scale / shift).
 *   grouped fp32 [B, channels[0], npoint, nsample] (gf_group_points layout), out fp32 [B, channels[n], npoint];
 *   channels[0] arbitrary <= 64 (e.g. 3 + 16), hidden widths multiples of 16, all <= 64. */
int gf_group_mlp_max(const float* grouped, int B, int npoint, int nsample, int n_layers, const float* const* W,
                     const float* const* scale, const float* const* shift, const int* channels, const int* relu,
                     float* out, void* stream);

/* The whole set-abstraction stage for given sample indices in two launches (PointnetSAModuleVotes.forward with
 * inds, lib/pointnet2/pointnet2_modules.py:305-349; QueryAndGroup, pointnet2_utils.py:326-356):
 *   new_xyz[b,i] = xyz[b, inds[b,i]];  idx = ball_query(radius, nsample, xyz, new_xyz);
 *   input channels of sample s of centre i: (xyz[idx] - new_xyz[i]) (* 1/radius with normalize_xyz) if use_xyz,
 *   then feats[:, idx];  out[b,:,i] = max_s MLP(...)   -- without materialising the grouped tensor.
 *   xyz fp32 [B,n,3], feats fp32 [B,C,n], inds int32 [B,npoint]; outputs new_xyz fp32 [B,npoint,3],
 *   idx int32 [B,npoint,nsample] (the ball-query result), out fp32 [B,channels[n_layers],npoint];
 *   channels[0] must equal C + 3*use_xyz. */
int gf_ball_query_centres(const float* xyz, const int32_t* centre_idx, int b, int n, int m, float radius, int nsample,
                          float* new_xyz, int32_t* idx, void* stream);
/* The same query for ONE point set through a hash grid with cells of `radius` (27 buckets around each centre instead
 * of every point): identical rows, ~8x less work at 2048 centres x 50 000 points.  Centres as indices into xyz
 * (centre_idx, new_xyz then receives their coordinates) or as coordinates (centres fp32 [m,3], new_xyz may be NULL).
 *   scratch: gf_knn_scratch_bytes(n). */
int gf_point_grid_build(const float* xyz, int n, float radius, void* scratch, void* stream);
int gf_ball_query_grid(const float* xyz, int n, const int32_t* centre_idx, const float* centres, int m, float radius,
                       int nsample, void* scratch, int grid_ready, float* new_xyz, int32_t* idx, void* stream);
/* gf_point_grid_build: the grid alone (same points, radius and scratch), e.g. early on another stream; the query is
 * then called with grid_ready = 1 and launches nothing but itself. */
int gf_sa_group_mlp_max(const float* xyz, const float* feats, const int32_t* inds, int B, int n, int C, int npoint,
                        float radius, int nsample, int use_xyz, int normalize_xyz, int n_layers,
                        const float* const* W, const float* const* scale, const float* const* shift,
                        const int* channels, const int* relu, float* new_xyz, int32_t* idx, float* out, void* scratch,
                        int grid_ready, void* stream);
/* scratch: NULL, or gf_knn_scratch_bytes(n) bytes -> grid ball query (B = 1; grid_ready as above) */

/* Soft-max over the middle dimension of x[n0,n1,inner] (training path of the decoder's vector cross-attention,
 * model/transformer_detr.py:449: F.softmax(sim / sqrt(d), dim=1) on [nq,nc,B,d]):
 *   y = softmax_{n1}(scale * x);   gx = scale * y * (gy - sum_{n1} gy * y). */
int gf_softmax_dim1_fwd(const float* x, int n0, int n1, int inner, float scale, float* y, void* stream);
int gf_softmax_dim1_bwd(const float* y, const float* gy, int n0, int n1, int inner, float scale, float* gx,
                        void* stream);

/* ===================================================================================
 * Token-side stages of the decoder between two cross-attentions, fused (inference)
 * (TransformerDecoderLayer.forward_pre_rel, model/transformer_detr.py:425-463; TransformerDecoder.forward,
 *  model/transformer_detr.py:130-166)
 * =================================================================================== */

/* One launch = [post part of layer l] + [pre part of layer l+1]; either half may be absent (NULL table).
 *   post: tgt = relu(out_mlp(attn_out)) + tgt2; tgt += linear2(relu(linear1(norm3(tgt)))); inter_out = norm(tgt)
 *   pre : t2 = norm1(tgt); q = k = t2 + query_pos; tgt += self_attn(q, k, t2); tgt2 = norm2(tgt);
 *         q1_out = attn_mlp[0](tgt2)   (the query half of the next cross-attention's first linear)
 *   attn_out fp32 [B,nq,64] (gf_decoder_cross_attn output), tgt_in fp32 [nq,B,64] (first stage only),
 *   query_pos fp32 [nq,B,64], inter_out fp32 [nq,B,64], q1_out fp32 [B,nq,64],
 *   state: gf_decoder_token_state_bytes(nq,B) bytes, carried unchanged from one stage to the next,
 *   post_params (HOST array of 10 device pointers): out_mlp.W[64,64] .b | norm3.w .b | linear1.W[ff,64] .b |
 *                linear2.W[64,ff] .b | decoder.norm.w .b
 *   pre_params  (10): norm1.w .b | self_attn.in_proj_weight[192,64] in_proj_bias | out_proj.W .b | norm2.w .b |
 *                attn_mlp[0].W[64,64] .b.       d = 64, nhead = 4, ff % 16 == 0, ff <= 256. */
size_t gf_decoder_token_state_bytes(int nq, int B);
int gf_decoder_token_stage(const float* attn_out, const float* tgt_in, const float* query_pos, int nq, int B, int d,
                           int nhead, int ff, const float* const* post_params, const float* const* pre_params,
                           void* state, float* inter_out, float* q1_out, void* stream);

/* Training form of the decoder's token-side stages (what train.py:63-75 runs through TransformerDecoderLayer
 * .forward_pre_rel, model/transformer_detr.py:425-463, with its dropouts): forward with hashed dropout masks and a native
 * backward, 2 + 1 launches forward and 4 + 2 backward per layer instead of ~95 framework launches.  All tensors fp32
 * [B, T, 64] row-major (row = b * T + t); d_model 64, 4 heads.
 *   pre  (x, query_pos) -> (t2n, q1):  t2 = norm1(x); q = k = t2 + query_pos;
 *                                      x1 = x + drop(out_proj(MHA(q, k, t2))); t2n = norm2(x1); q1 = W1 t2n + b1
 *   post (ca, t2n) -> (x3, inter):     x2 = relu(out_mlp(ca)) + drop(t2n);
 *                                      x3 = x2 + drop(linear2(drop(relu(linear1(norm3(x2)))))); inter = norm(x3)
 *   params: HOST arrays of 10 DEVICE pointers in gf_decoder_token_stage's pre / post order.
 *   Dropout: element kept iff the hash of gf_backbone_transformer_train_fwd's comment says so, with
 *     site = 8 * layer + {0 attention weights, 1 attention branch, 2 the normed query added after the cross-attention,
 *     3 hidden layer, 4 feed-forward branch}, row = b * T + t, col = channel (attention weights: 4 * key + head).
 *   save / work: the *_save_bytes / *_work_bytes sizes; grads: *_grad_floats floats, the 10 parameters' gradients back
 *   to back in table order (every value written; post: the last two are this layer's share of decoder.norm's).
 *   Backward inputs that are NULL count as zero (at least one must be given).  Fixed summation orders. */
size_t gf_decoder_pre_train_save_bytes(int T, int B);
size_t gf_decoder_pre_train_work_bytes(int T, int B);
long long gf_decoder_pre_grad_floats(void);
int gf_decoder_pre_train_fwd(const float* x, const float* qpos, int T, int B, const float* const* params, float p,
                             unsigned seed, int layer, void* save, float* t2n, float* q1, void* stream);
int gf_decoder_pre_train_bwd(const float* x, const float* qpos, const float* t2n, const float* d_t2n, const float* d_q1,
                             int T, int B, const float* const* params, float p, unsigned seed, int layer, void* save,
                             void* work, float* dx, float* dqpos, float* grads, void* stream);
size_t gf_decoder_post_train_save_bytes(int T, int B, int ff);
size_t gf_decoder_post_train_work_bytes(int T, int B, int ff);
long long gf_decoder_post_grad_floats(int ff);
int gf_decoder_post_train_fwd(const float* ca, const float* t2n, int T, int B, int ff, const float* const* params,
                              float p, unsigned seed, int layer, void* save, float* x3, float* inter, void* stream);
int gf_decoder_post_train_bwd(const float* ca, const float* x3, const float* d_x3, const float* d_inter, int T, int B,
                              int ff, const float* const* params, float p, unsigned seed, int layer, void* save,
                              void* work, float* d_ca, float* d_t2n, float* grads, void* stream);

/* ===================================================================================
 * Proposal extraction of the eval forward (GeoFormer.generate_proposal,
 * model/geoformer/geoformer.py:193-262), fused
 * =================================================================================== */

/* Per query q over its mask-logit row: member(p) = 1/(1+exp(-logit)) >= logit_thresh,
 *   npoints[q] = #members, mask_score = sum prob / (npoints + 1e-6),
 *   cls_pred[q] = argmax cls_logits[q], cls_score = softmax(cls_logits[q])[cls_pred],
 *   sem_score = sum_{members} sem_prob[p, cls_pred] / (npoints + 1e-6),
 *   scores[q] = mask_score * sqrt(cls_score) * sem_score,
 *   final[q] = cls_pred >= min_class && npoints >= npoint_thresh && mask_score >= score_thresh.
 *   mask_logits fp32 [nq,N], cls_logits fp32 [nq,ncls], sem_prob fp32 [ncls,N] CLASS-MAJOR (soft-max of the
 *   semantic scores of the foreground points, transposed); outputs int32 [nq] / fp32 [nq]. */
int gf_proposal_stats(const float* mask_logits, const float* cls_logits, const float* sem_prob, int nq, int N,
                      int ncls, float logit_thresh, float score_thresh, int npoint_thresh, int min_class,
                      int* cls_pred, int* npoints, float* scores, int* final_mask, void* stream);

/* Few-shot form (GeoFormerFS.generate_proposal, model/geoformer/geoformer_fs.py:205-238): members as above,
 *   mask_score = sum prob / (npoints + 1e-6), scores[q] = mask_score * sqrt(sim[q]),
 *   final[q] = sim[q] >= sim_thresh && npoints >= npoint_thresh && mask_score >= score_thresh.
 *   mask_logits fp32 [nq,N], sim fp32 [nq] (cosine similarity of the query to the support prototype). */
int gf_proposal_stats_fs(const float* mask_logits, const float* sim, int nq, int N, float logit_thresh,
                         float score_thresh, int npoint_thresh, float sim_thresh, int* npoints, float* scores,
                         int* final_mask, void* stream);

/* proposals[i, fg_idxs[p]] = 1 for every member p of query sel[i]; proposals int32 [n_sel,num_points] must be
 * zero-filled by the caller, fg_idxs int64 [N] (row of each foreground point in the scene), sel int32 [n_sel]. */
int gf_proposal_scatter(const float* mask_logits, const int* sel, int n_sel, int N, const long long* fg_idxs,
                        float logit_thresh, int num_points, int* proposals, void* stream);

/* Ingredients of the decoder's relative position embedding for one scene (geoformer.py:619-651):
 *   geo_ctx[q,j] = geo[q, inds[j]]   (geo fp32 [nq,n], inds int32 [nc] = the context points' FPS indices),
 *   max_geo[q]   = max_j geo_ctx[q,j], or the largest such maximum where row q is entirely unreachable (< 0). */
int gf_relpos_prepare(const float* geo, const int32_t* inds, int nq, int n, int nc, float* geo_ctx, float* max_geo,
                      void* stream);

/* The accepted queries of gf_proposal_stats in ascending order with their classes / scores (geoformer.py:236-243),
 * compacted on the device: sel int32 [nq], cls_out int64 [nq], scores_out fp32 [nq] (capacity nq), *d_count = how many. */
int gf_proposal_select(const int32_t* final_, const int32_t* cls_pred, const float* scores, int nq, int32_t* sel,
                       long long* cls_out, float* scores_out, int32_t* d_count, void* stream);

/* inter[i,j] = number of points in both proposal i and proposal j (matrix NMS, util/utils_3d.py:95-141: the
 * einsum over the [n,N] float masks; exact because the masks are 0/1).  masks int32 [n,N] (gf_proposal_scatter
 * output), inter int32 [n,n], scratch: gf_mask_intersections_scratch_bytes(n, N). */
size_t gf_mask_intersections_scratch_bytes(int n, int N);
int gf_mask_intersections(const int32_t* masks, int n, int N, void* scratch, int32_t* inter, void* stream);

/* ===================================================================================
 * Backbone voxel transformer of the two deepest U-Net levels, fused (inference)
 * (UBlock: model/geoformer/geoformer_modules.py:64-68,120-127; TransformerEncoder(d_model=128, N,
 *  heads=4, d_ff=64): model/transformer.py:62-188)
 * =================================================================================== */

/* out = after(TransformerEncoder(xyz, before(feats))) per scene in n_layers + 2 launches (n_scenes <= 4096).
 *   feats fp32 [M,c] (c % 16 == 0), coords int32 [M,4] (b,x,y,z) with the rows of a scene contiguous,
 *   scene_offsets int32 [n_scenes+1] (device), out fp32 [M,c],
 *   scratch: gf_backbone_transformer_scratch_bytes(M) bytes,
 *   params: HOST array of gf_backbone_transformer_num_params(n_layers) DEVICE pointers, nn.Linear weights
 *   row-major [out,in], in this order:
 *     before.W[128,c] before.b | position.W[128,3] position.b |
 *     per layer: norm1.alpha norm1.bias  q.W q.b  k.W k.b  v.W v.b  out.W out.b  norm2.alpha norm2.bias
 *                ff1.W[64,128] ff1.b  ff2.W[128,64] ff2.b |
 *     norm.alpha norm.bias | after.W[c,128] after.b */
size_t gf_backbone_transformer_scratch_bytes(int M);
int gf_backbone_transformer_num_params(int n_layers);
int gf_backbone_transformer(const float* feats, const int* coords, const int* scene_offsets, int n_scenes, int M,
                            int c, int n_layers, const float* const* params, void* scratch, float* out,
                            void* stream);

/* Training form of the same stack (what train.py:63-75 runs through the framework modules of
 * geoformer_modules.py:64-68,120-127 and transformer.py:62-188 with their dropouts, p = transformer.py's 0.1): the
 * forward keeps what the backward needs in `save`, the backward returns the gradient of the input rows and of every
 * parameter -- n_layers + 2 and 2 n_layers + 2 launches instead of ~250 framework launches per level and step.
 *   coords int32 [M,4] with the scene id in column 0, ascending (scene offsets are found on the device);
 *   p: dropout probability of the four dropout sites of a layer (0: every element kept -- modules in eval mode);
 *   seed: the call's dropout seed (the same value must be passed to the backward): an element is kept iff
 *     (h >> 8) >= (uint32)(p * 2^24),  h = fmix32(fmix32(seed ^ (row * 64 + site)) + col * 0x9E3779B1),
 *     fmix32 = MurmurHash3's finaliser, site = 4 * layer + {0 attention weights, 1 attention branch, 2 hidden layer,
 *     3 feed-forward branch}, row = the token's row in the batch, col = the channel (attention weights: 4 * key's
 *     index in its scene + head); kept elements are scaled by 1 / (1 - p);
 *   save: gf_backbone_transformer_train_save_bytes(M, n_layers) bytes, written by the forward, read by the backward;
 *   work: gf_backbone_transformer_train_work_bytes(M, n_layers, n_scenes) bytes of scratch for the backward;
 *   dfeats fp32 [M,c]; grads: gf_backbone_transformer_grad_floats(c, n_layers) floats, the parameters' gradients
 *   back to back in the order of the parameter table, each in its parameter's layout (every value is written).
 *   c % 16 == 0, c <= 384.  Summation orders are fixed: the same inputs give the same bits. */
size_t gf_backbone_transformer_train_save_bytes(int M, int n_layers);
size_t gf_backbone_transformer_train_work_bytes(int M, int n_layers, int n_scenes);
long long gf_backbone_transformer_grad_floats(int c, int n_layers);
int gf_backbone_transformer_train_fwd(const float* feats, const int* coords, int n_scenes, int M, int c, int n_layers,
                                      const float* const* params, float p, unsigned seed, void* save, float* out,
                                      void* stream);
int gf_backbone_transformer_train_bwd(const float* feats, const float* dout, int n_scenes, int M, int c, int n_layers,
                                      const float* const* params, float p, unsigned seed, void* save, void* work,
                                      float* dfeats, float* grads, void* stream);

/* ===================================================================================
 * Decoder cross-attention (TransformerDecoderLayer.forward_pre_rel, model/transformer_detr.py:443-454
 * with the relative embedding of GeoFormer.forward_decoder, model/geoformer/geoformer.py:619-651), fused
 * =================================================================================== */

/* Packs attn_mlp.0.weight (W1), attn_mlp.2.weight (W2) and v_mlp.0.weight (Wv), each [64,64] (out,in),
 * into MFMA A-operand lane order; Wpack holds gf_decoder_wpack_floats() floats. */
size_t gf_decoder_wpack_floats(void);
int gf_decoder_pack_weights(const float* W1, const float* W2, const float* Wv, float* Wpack, void* stream);

/* out[b,i,:] = sum_j softmax_j(sim_ij / 8) * (Kv[b,j,:] + Wv r_ij),   sim_ij = W2 relu(Q1[b,i,:] - K1[b,j,:] + W1 r_ij) + b2,
 * r_ij = [sin(P), cos(P)], P = 2*pi*((g3 - lo)/(hi - lo)) . gaussB,  g3 = geo_ctx[b,i,j] on all three axes, or
 * max_geo[b,i] + |qloc[b,i] - cloc[b,j]| per axis where geo_ctx < 0.
 *   geo_ctx [B,nq,nc], max_geo [B,nq], qloc [B,nq,3], cloc [B,nc,3], lo/hi [B,3] (the pc_dims pair as the
 *   caller passes it -- GeoFormer passes [maxs, mins]), gaussB [3,32],
 *   Q1 = W1 q + b1 [B,nq,64], K1 = W1 k [B,nc,64], Kv = Wv k + bv [B,nc,64], b2 [64];  out [B,nq,64]. */
int gf_decoder_cross_attn(const float* geo_ctx, const float* max_geo, const float* qloc, const float* cloc,
                          const float* lo, const float* hi, const float* gaussB, const float* Q1, const float* K1,
                          const float* Kv, const float* Wpack, const float* b2, int B, int nq, int nc, int d, float* out,
                          float* stat_m, float* stat_l, void* stream);
/* The same with the workgroup shape chosen: wg_waves = 16 (gf_decoder_cross_attn) or 8 -- 512 threads at 120 registers
 * fit on a compute unit BESIDE a 512-thread workgroup of gf_geodesic_bfs_cfg, for a serving loop that runs one scene's
 * decoder under the next scene's sampling / BFS stretch (GeoFormer.forward_split).  Results agree to rounding (the
 * per-wave partial soft-max states are merged over 8 instead of 16 waves). */
int gf_decoder_cross_attn_cfg(const float* geo_ctx, const float* max_geo, const float* qloc, const float* cloc,
                              const float* lo, const float* hi, const float* gaussB, const float* Q1, const float* K1,
                              const float* Kv, const float* Wpack, const float* b2, int B, int nq, int nc, int d,
                              float* out, float* stat_m, float* stat_l, int wg_waves, void* stream);

/* Backward of gf_decoder_cross_attn (training).  stat_m / stat_l fp32 [B,nq,64]: per (query, channel) maximum and
 * sum of the scaled soft-max logits, written by the forward when the two pointers are given (NULL otherwise); out =
 * the forward's output, gout = dL/dout fp32 [B,nq,64]; W2 fp32 [64,64] the second pair-MLP weight (unpacked).
 * Outputs: dQ1 fp32 [B,nq,64] (overwritten), dK1 / dKv fp32 [B,nc,64] (zero on entry), dW fp32 [3,64,64] = the PAIR
 * parts of dW1, dW2, dWv (overwritten; the hoisted parts W1 q, W1 k, Wv k are the caller's GEMMs).  Everything of size
 * nq*nc*64 is recomputed.  scratch: gf_decoder_cross_attn_bwd_scratch_floats(B,nq,nc) floats. */
size_t gf_decoder_cross_attn_bwd_scratch_floats(int B, int nq, int nc);
int gf_decoder_cross_attn_bwd(const float* geo_ctx, const float* max_geo, const float* qloc, const float* cloc,
                              const float* lo, const float* hi, const float* gaussB, const float* Q1, const float* K1,
                              const float* Kv, const float* Wpack, const float* W2, const float* out, const float* stat_m,
                              const float* stat_l, const float* gout, int B, int nq, int nc, int d, float* dQ1, float* dK1,
                              float* dKv, float* dW, float* scratch, void* stream);

/* ===================================================================================
 * Natives GeoFormer inherits but never executes -- binding completeness
 * =================================================================================== */

/* PG_OP.sec_mean / sec_min / sec_max (sec_mean.cu:12-86): kind 0/1/2; offsets int32 [nProposal+1]; out [nProposal,C] */
int gf_sec_op(int kind, const float* inp, const int32_t* offsets, int nProposal, int C, float* out, void* stream);
/* PG_OP.roipool_fp/bp (roipool.cu:12-57): segment max + arg-max; bp adds d_out to d_feats[argmax] (caller zeroes) */
int gf_roipool_fp(const float* feats, const int32_t* offsets, int nProposal, int C, float* out, int32_t* maxidx,
                  void* stream);
int gf_roipool_bp(const float* d_out, const int32_t* maxidx, int nProposal, int C, float* d_feats, void* stream);
/* PG_OP.get_iou (get_iou.cu:12-38): iou [nProposal,nInstance]; instance_labels int64 [N] */
int gf_get_iou(const int32_t* proposals_idx, const int32_t* proposals_offset, const long long* instance_labels,
               const int32_t* instance_pointnum, int nInstance, int nProposal, float* iou, void* stream);
/* PG_OP.ballquery_batch_p (bfs_cluster.cu:15-89): idx int32 [n*meanActive], start_len int32 [n,2], starts in point
 * order; d_cumsum (device int32) = total pair count (the reference's return value). */
size_t gf_ballquery_batch_p_scratch_bytes(int n);
int gf_ballquery_batch_p(const float* xyz, const int32_t* batch_idxs, const int32_t* batch_offsets, int n, int meanActive,
                         float radius, int32_t* idx, int32_t* start_len, int32_t* d_cumsum, void* scratch, void* stream);
/* PG_OP.bfs_cluster (bfs_cluster.cpp:28-111): HOST pointers (the reference runs it on CPU tensors).
 * h_cluster_idxs capacity [N,2], h_cluster_offsets capacity [N+1]. */
int gf_bfs_cluster_host(const int32_t* h_semantic_label, const int32_t* h_ball_query_idxs, const int32_t* h_start_len,
                        int N, int threshold, int32_t* h_cluster_idxs, int32_t* h_cluster_offsets, int32_t* h_nCluster,
                        int32_t* h_sumNPoint);
/* pointnet2._ext.three_nn / three_interpolate / three_interpolate_grad (interpolate_gpu.cu:12-157) */
int gf_three_nn(const float* unknown, const float* known, int b, int n, int m, float* dist2, int32_t* idx, void* stream);
int gf_three_interpolate(const float* points, const int32_t* idx, const float* weight, int b, int c, int m, int n,
                         float* out, void* stream);
int gf_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight, int b, int c, int n, int m,
                              float* grad_points, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOFORMER_HIP_H */
