/*
 * geoformer_hip_dev.h -- development / measurement hooks of libgeoformer_hip.so.
 *
 * NOT part of the drop-in boundary (include/geoformer_hip.h): nothing on the product path calls these.
 * They exist for bench.py's roofline probe, for the parity tests that force every launch shape on a small
 * input, and for the dev tools under tools/.  Same conventions as the product header (device pointers,
 * hipStream_t as void*, 0 = ok).
 */
#ifndef GEOFORMER_HIP_DEV_H
#define GEOFORMER_HIP_DEV_H

#include "geoformer_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* gf_conv_fwd with two caller-owned hipEvent_t recorded immediately before/after the launch on `stream`
 * (the kernel's own duration on the stream it runs on; bench.py's roofline probe). */
int gf_dev_conv_fwd_timed(const float* in, const float* Wp, const int32_t* nbr, const uint32_t* gmask,
                          const int32_t* steps, int K, int M_in, int M_out, int ld, int Cin, int Cout, const float* in_scale, const float* in_shift,
                          const float* residual, const float* out_scale, const float* out_shift, float* out, void* ev_start,
                          void* ev_stop, void* stream);

/* Force gf_conv_fwd's launch shape (it normally follows the level's size): split / wide / pair: 0, 1 or -1 (size
 * based); ldsw: 1 = weights staged in LDS where supported; block: threads per workgroup of the one-wave-per-group
 * shape (0 = default).  Process-wide; tests reset with (-1,-1,-1,0,0).  The same knobs can be set once from the
 * environment (GF_CONV_SPLIT / _WIDE / _PAIR / _LDSW / _BLOCK). */
int gf_dev_conv_knobs(int split, int wide, int pair, int ldsw, int block);

/* The counted-loop kernel over the step table: use (0 / 1 / -1 = size based), weights staged in LDS (0 / 1 / -1 =
 * when they fit), groups walked per wave of the non-pipelined form (0 = default), pipelined form (0 / 1 / -1 =
 * default on).  Environment: GF_CONV_G16 / _G16_LDSW / _G16_GPW / _G16_PIPE. */
int gf_dev_conv_knobs_g16(int use, int ldsw, int gpw, int pipe);

/* The flat-chain kernel of the deep levels (k_conv_flat): use 0 / 1 (whenever the shape allows: 16-channel multiples,
 * K * Cin / 16 <= 256) / -1 = size based, i.e. launches of at most max_items (group, column block) items (0 = the
 * default bound of 256). */
int gf_dev_conv_knob_flat(int use, int max_items);

/* The LDS-weight kernel over a flat step table (k_conv_lw, spconv_lw.hip; only where gf_conv_fwd_flat is given such a
 * table): use 0 / 1 (whenever the shape allows) or -1 (size-based: at least `min_groups` 16-row groups, 0 = default). */
int gf_dev_conv_knob_lw(int use, int min_groups);

/* Number of equal-cost chunks (= waves of the pipelined kernel) the NEXT rulebooks are built with: a multiple of 4,
 * at most 4096; 0 = default (3072 = 12 waves per compute unit). */
int gf_dev_conv_chunks(int n);
/* waves per workgroup of the pipelined level-1 kernel (4, 8, 12, 16; 0 = default). */
int gf_dev_conv_g16p_wpb(int wpb);

/* Events around the convolution launches of gf_unet_fwd, recorded on the stream the kernels run on (bench.py's
 * roofline probes).  mode 0 = off, 1 = the level-1 3x3x3 16->16 convolutions of the residual blocks, 3 = the same
 * with two more events bound to the kernel launch itself (gf_dev_unet_probe_read2), 2 = every
 * convolution (+ a counting kernel per table for the number of rules).  State is per host thread.
 * gf_dev_unet_probe_read waits for the events and returns the number of records (at most max_records) and clears
 * them: meta[9*i..] = level (0-based), kind (0 input conv, 1 / 2 first / second conv of a block, 3 identity 1x1x1,
 * 4 strided, 5 inverse), K, Cin, Cout, M_in, M_out, residual epilogue (0/1), rules (-1 when not counted);
 * us[i] = microseconds between the two events. */
int gf_dev_unet_probe(int mode);
/* Nanoseconds the calling host thread has spent blocked inside gf_unet_fwd's own waits (the two voxel-count read-backs)
 * since the last reset: bench.py subtracts them (and the Python-side waits) from the loop's wall time -> host_busy_ms. */
unsigned long long gf_dev_host_wait_ns(int reset);
int gf_dev_unet_probe_read(int max_records, int* meta, float* us);
/* The same plus, in mode 1, the launch's duration by two events BOUND TO THE KERNEL (hipExtLaunchKernelGGL: the
 * dispatch's own begin / end timestamps -- what a profiler's kernel trace reports), -1 where the launch did not take
 * the pipelined level-1 kernel.  Events recorded before / after a launch on its stream (us[]) add the command
 * processor's handling of the two event packets, ~1.3 us on a 20 us kernel. */
int gf_dev_unet_probe_read2(int max_records, int* meta, float* us, float* us_kernel);
/* Two caller-owned hipEvent_t that the next launch of the pipelined level-1 kernel on this host thread binds to itself
 * (NULL, NULL: none); ..._taken: 1 if the last launch consumed them. */
int gf_dev_conv_kernel_events(void* start, void* stop);
int gf_dev_conv_kernel_events_taken(void);

/* Two caller-owned hipEvent_t that the NEXT launch of operator `op`'s main kernel (from any host thread: backward
 * kernels are launched by the framework's autograd thread) binds to itself
 * (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps -- what a profiler's kernel trace reports for that
 * kernel; no host time and no neighbouring launch between them); (NULL, NULL) disarms.  op: 0 geodesic BFS
 * (k_geodesic_bfs_lds), 1 decoder cross-attention (k_decoder_cross_attn), 2 mask head (k_mask_head), 3 furthest point
 * sampling (k_fps), 4 cross-attention backward, 5 / 6 mask-head backward (feature / parameter kernel), 7 the mask-driven
 * weight gradient (k_conv_wgrad_t).  ..._taken: 1 once a launch has consumed them.  bench.py's operator rooflines. */
int gf_dev_op_kernel_events(int op, void* start, void* stop);
int gf_dev_op_kernel_events_taken(int op);
/* hipEvent_t (timing enabled) for the hook above, owned by the caller; elapsed waits for `stop`. */
void* gf_dev_event_create(void);
int gf_dev_event_destroy(void* event);
int gf_dev_event_elapsed_us(void* start, void* stop, float* us);

/* Cross-attention kernel of the 16-wave shape: 1 = the three 64 x 64 products as fp32-accurate bf16 MFMAs over the exact
 * three-piece split (k_decoder_cross_attn_bf3, default), 0 = fp32 MFMAs (k_decoder_cross_attn<16>), -1 = default /
 * GF_CROSS_ATTN_BF3. */
int gf_dev_cross_attn_bf3(int on);

/* Upper bound of the BFS kernels' LDS queue capacity (entries per level, >= 64; 0 = what the LDS share allows): tests
 * set it so that scene-sized graphs exercise the queues' overflow into global memory. */
int gf_dev_bfs_qcap_max(int qcap);

#ifdef __cplusplus
}
#endif
#endif
