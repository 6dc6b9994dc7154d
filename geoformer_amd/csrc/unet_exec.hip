// The sparse-voxel U-Net of the eval forward as ONE native call (include/geoformer_hip.h: gf_unet_fwd).
//
// Reference structure: GeoFormer.input_conv -> UBlock x7 -> output_layer (model/geoformer/geoformer.py:42-53,398-401;
// UBlock / ResidualBlock: model/geoformer/geoformer_modules.py:10-35,52-129), all BatchNorm in eval mode.
//
// Why it exists: the 13 rulebooks and 71 convolutions are ~150 launches of 5-25 us; issued from Python (one ctypes
// crossing + a torch allocation per block) the host needs 1.7 ms for them while the device needs ~1.3 ms -- the
// stretch was bound by the host's launch rate.  Here the same entry points (gf_index_build, gf_rules_subm3,
// gf_rules_down2_chain, gf_conv_fwd, gf_resblock_fwd, gf_backbone_transformer -- so results are identical to the
// per-module path) are issued from C++ out of one caller-owned workspace, with
//   * the down-sampling rulebook chain and the submanifold tables of levels >= 2 on a SIDE stream beside the
//     level-1 convolutions (small latency-bound kernels next to the only HBM-heavy ones),
//   * one host wait for the chain's voxel counts (an event behind an async copy into pinned memory), taken after
//     the level-1 work has been queued,
//   * the output BatchNorm + ReLU in the epilogue of the last convolution.
#include <time.h>
#include <vector>

#include "common.h"
#include "geoformer_hip_dev.h"

namespace {

__global__ void k_pad_channels(const float* __restrict__ in, int M, int cin, int cout, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * cout) return;
    const int r = i / cout, c = i - r * cout;
    out[i] = c < cin ? in[(size_t)r * cin + c] : 0.f;
}

// out[r] = (a[r], b[r]): the skip concatenation of UBlock.forward (geoformer_modules.py:116), 16-byte pieces
__global__ void k_concat2(const float4* __restrict__ a, const float4* __restrict__ b, int M, int c4, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * 2 * c4) return;
    const int r = i / (2 * c4), c = i - r * 2 * c4;
    out[i] = c < c4 ? a[(size_t)r * c4 + c] : b[(size_t)r * c4 + (c - c4)];
}

// The skip concatenation AND the tail block's 1x1x1 identity branch in one launch (round 6): cat[r] = (a[r], b[r]) and
// idn[r] = W_i^T cat[r], W_i in the packed K = 1 layout of gf_conv_pack_weights (2C -> C).  One wave per (16-row group,
// 16-column block): it reads the group's rows of a and b once -- 2C / 16 float4 per lane, all in flight together with the
// weight operands --, the wave of column block 0 also writes them out as the concatenated rows, and the product runs
// transposed like every convolution here (A = weights, B = rows: lane (r, q) ends with four consecutive output channels of
// row r).  Replaces k_concat2 + a K = 1 gf_conv_fwd launch: one dependent launch less per tail block (six per forward).
#define IDC_MAXCH 16  // 2C / 16 <= 16: widths up to C = 128
__global__ __launch_bounds__(256) void k_concat2_idn(const float* __restrict__ a, const float* __restrict__ b, int M, int C,
                                                     const float4* __restrict__ Wp, float* __restrict__ cat,
                                                     float* __restrict__ idn) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int ncb = C >> 4, nch = C >> 3, nc1 = C >> 4;  // column blocks, input chunks (2C / 16), chunks that come from a
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ngroups = (M + 15) >> 4;
    if (item >= ngroups * ncb) return;
    const int g = item / ncb, cb = item - g * ncb;
    const int row = g * 16 + r;
    const int rc = row < M ? row : M - 1;
    float4 x[IDC_MAXCH], w[IDC_MAXCH];
#pragma unroll
    for (int ch = 0; ch < IDC_MAXCH; ch++) {
        if (ch < nch) {
            const float* src = ch < nc1 ? a + (size_t)rc * C + ch * 16 + 4 * q : b + (size_t)rc * C + (ch - nc1) * 16 + 4 * q;
            x[ch] = *reinterpret_cast<const float4*>(src);
            w[ch] = Wp[(size_t)(ch * ncb + cb) * 64 + lane];
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < IDC_MAXCH; ch++) {
        if (ch < nch) {
            if (cb == 0 && row < M) *reinterpret_cast<float4*>(cat + (size_t)row * 2 * C + ch * 16 + 4 * q) = x[ch];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ch].x, x[ch].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ch].y, x[ch].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ch].z, x[ch].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ch].w, x[ch].w, acc, 0, 0, 0);
        }
    }
    if (row < M) *reinterpret_cast<float4*>(idn + (size_t)row * C + cb * 16 + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// scene_offsets[b] = first row of batch b (rows in batch-major order: every level below the first)
__global__ void k_scene_offsets(const int32_t* __restrict__ coords, int M, int B, int32_t* __restrict__ offs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    const int prev = i > 0 ? coords[(size_t)(i - 1) * 4] : -1;
    const int cur = i < M ? coords[(size_t)i * 4] : B;
    for (int b = prev + 1; b <= cur && b <= B; b++) offs[b] = i;
}

struct Bump {
    unsigned char* base;
    size_t off = 0, cap;
    Bump(void* p, size_t c) : base((unsigned char*)p), cap(c) {}
    template <class T>
    T* take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

inline int r16(long long n) { return (int)((n < 16 ? 16 : n + 15) / 16 * 16); }

struct LevelTables {  // submanifold table of one level
    int32_t* nbr = nullptr;
    uint32_t* gmask = nullptr;
    int32_t* steps = nullptr;
    int32_t* flat = nullptr;  // flat step table (gf_rules_flat_steps) of the levels the LDS-weight conv kernel serves
    int ld = 0;
};

// events of the side-stream fork/join and the two host waits: per host thread, created once
struct EvPair {
    hipEvent_t fork = nullptr, chain = nullptr, chain2 = nullptr, rules = nullptr, tbl0 = nullptr, flat0 = nullptr, tbl1 = nullptr, flat1 = nullptr;
    hipEvent_t ws_done = nullptr;  // end of this thread's last call on its main stream: the tables may be rewritten
    bool ws_done_recorded = false;
};
thread_local EvPair t_ev;
// nanoseconds this host thread spent blocked in the executor's own waits (gf_dev_host_wait_ns: bench.py's host_busy figure)
thread_local unsigned long long t_wait_ns = 0;
struct WaitClock {
    timespec t0;
    WaitClock() { clock_gettime(CLOCK_MONOTONIC, &t0); }
    ~WaitClock() {
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        t_wait_ns += (unsigned long long)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec));
    }
};

constexpr int kStepsMinRows = 6000 * 16;  // gf_conv_fwd takes the counted-loop kernel from 6000 groups up
constexpr int kFlatMinRows = 1500 * 16;   // gf_conv_fwd_flat takes the LDS-weight kernel from 1500 groups up (spconv_lw.hip)

struct LevelBufs {  // feature buffers of one level, [rows, C] each (cat: [rows, 2C])
    float *x, *tmp, *idn, *o0, *o1, *o2, *up, *tr, *cat, *a0, *a1;
    void* tr_scratch;
    int32_t* tr_offs;
};
void carve_level(Bump& a, LevelBufs& b, size_t rows, size_t C, bool transformer, int B) {
    b.x = a.take<float>(rows * C);
    b.tmp = a.take<float>(rows * C);
    b.idn = a.take<float>(rows * C);
    b.o0 = a.take<float>(rows * C);
    b.o1 = a.take<float>(rows * C);
    b.o2 = a.take<float>(rows * C);
    b.up = a.take<float>(rows * C);
    b.tr = a.take<float>(rows * C);
    b.cat = a.take<float>(rows * 2 * C);
    b.a0 = a.take<float>(rows * C);  // activated copies (dual-output convolutions of the first level)
    b.a1 = a.take<float>(rows * C);
    b.tr_scratch = nullptr;
    b.tr_offs = nullptr;
    if (transformer) {
        b.tr_scratch = a.take<unsigned char>(gf_backbone_transformer_scratch_bytes((int)rows));
        b.tr_offs = a.take<int32_t>(B + 1);
    }
}

// ---- measurement hook (include/geoformer_hip_dev.h): events around the convolution launches ----
__global__ void k_count_rules(const int32_t* __restrict__ tbl, int K, int ld, int M, int* __restrict__ out) {
    int c = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)K * M; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / M), o = (int)(i - (long long)k * M);
        c += tbl[(size_t)k * ld + o] >= 0;
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
struct ProbeRec {
    int level, kind, K, Cin, Cout, M_in, M_out, res, slot;
    hipEvent_t a, b;    // recorded on the stream before / after the launch
    hipEvent_t ka, kb;  // bound to the kernel itself (null: the launch took another kernel than the pipelined one)
    int share = 1;      // layers of ONE persistent chain launch share its two events: each is booked elapsed / share
};
struct Probe {
    int mode = 0;
    std::vector<ProbeRec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    int* d_counts = nullptr;  // rule counts of the tables seen in "every conv" mode
    int nslots = 0;
    hipEvent_t ev() {
        if (used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[used++];
    }
};
thread_local Probe t_probe;
constexpr int kProbeSlots = 4096;
}  // namespace

extern "C" unsigned long long gf_dev_host_wait_ns(int reset) {
    const unsigned long long v = t_wait_ns;
    if (reset) t_wait_ns = 0;
    return v;
}

extern "C" int gf_dev_unet_probe(int mode) {
    GF_CHECK_ARG(mode >= 0 && mode <= 3, "gf_dev_unet_probe: mode %d (0 off, 1 / 3 level-1 block convs, 2 every conv)", mode);
    t_probe.mode = mode;
    if (mode == 2 && !t_probe.d_counts) GF_TRY(hipMalloc(&t_probe.d_counts, kProbeSlots * sizeof(int)));
    return GF_OK;
}

// meta: max_records x 9 ints (level, kind, K, Cin, Cout, M_in, M_out, residual, rules or -1); us: max_records floats.
// Waits for the recorded events; returns the number of records (clears them), or a negative status.
extern "C" int gf_dev_unet_probe_read2(int max_records, int* meta, float* us, float* us_kernel);
extern "C" int gf_dev_unet_probe_read(int max_records, int* meta, float* us) {
    return gf_dev_unet_probe_read2(max_records, meta, us, nullptr);
}
// us_kernel (optional): the launch's duration by the events bound to the kernel itself, -1 where there are none
extern "C" int gf_dev_unet_probe_read2(int max_records, int* meta, float* us, float* us_kernel) {
    Probe& pb = t_probe;
    std::vector<int> counts;
    if (pb.nslots > 0) {
        counts.resize(pb.nslots);
        GF_TRY(hipDeviceSynchronize());
        GF_TRY(hipMemcpy(counts.data(), pb.d_counts, sizeof(int) * pb.nslots, hipMemcpyDeviceToHost));
    }
    int n = 0;
    for (const ProbeRec& r : pb.recs) {
        if (n >= max_records) break;
        GF_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        GF_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        int* m = meta + (size_t)n * 9;
        m[0] = r.level; m[1] = r.kind; m[2] = r.K; m[3] = r.Cin; m[4] = r.Cout; m[5] = r.M_in; m[6] = r.M_out; m[7] = r.res;
        m[8] = r.slot >= 0 ? counts[r.slot] : -1;
        if (us_kernel) {
            us_kernel[n] = -1.f;
            if (r.ka) {
                float km = 0.f;
                GF_TRY(hipEventSynchronize(r.kb));
                GF_TRY(hipEventElapsedTime(&km, r.ka, r.kb));
                us_kernel[n] = km * 1e3f;
            }
        }
        us[n++] = ms * 1e3f / (float)(r.share > 0 ? r.share : 1);
    }
    pb.recs.clear();
    pb.used = 0;
    pb.nslots = 0;
    return n;
}

// int32/byte size of everything gf_unet_fwd carves out of its workspace (capacity bounds, no device access)
static size_t unet_layout(const GfUnetParams* P, int M0, int B, int X, int Y, int Z, long long* offs, int* caps,
                          int* shapes, int* nl_out, long long* chain_elems) {
    int nl = 0;
    long long total = 0;
    gf_rules_down2_chain_plan(M0, B, X, Y, Z, P->nlevels - 1, offs, caps, shapes, &total, &nl);
    *nl_out = nl;
    *chain_elems = total;
    Bump a(nullptr, 0);
    const size_t words0 = gf_index_words(B, X, Y, Z);
    a.take<uint32_t>(words0);
    a.take<int32_t>(words0);
    a.take<int32_t>(M0 > 0 ? M0 : 1);
    a.take<unsigned char>(gf_index_scratch_bytes(words0));
    const int ld0 = r16(M0);
    a.take<int32_t>((size_t)27 * ld0);
    a.take<uint32_t>(ld0 / 16);
    a.take<int32_t>(gf_rules_steps_words(ld0));
    if (ld0 >= kFlatMinRows) a.take<int32_t>(gf_rules_flat_words(27, ld0));
    a.take<int32_t>((size_t)total);
    a.take<int32_t>(GF_UNET_MAX_LEVELS + 1);
    for (int l = 1; l <= nl; l++) {
        a.take<int32_t>((size_t)27 * caps[l]);
        a.take<uint32_t>(caps[l] / 16);
        if (caps[l] >= kFlatMinRows) a.take<int32_t>(gf_rules_flat_words(27, caps[l]));
    }
    // features: per level at its capacity (the call itself carves them at the real row counts, which are smaller)
    for (int l = 0; l <= nl; l++) {
        LevelBufs b;
        carve_level(a, b, l ? (size_t)caps[l] : (size_t)ld0, (size_t)P->level[l].C, P->level[l].tr_layers > 0, B);
    }
    a.take<float>((size_t)ld0 * 16);  // zero-padded input rows
    return a.off;
}

extern "C" size_t gf_unet_ws_bytes(const GfUnetParams* P, int M0, int B, int X, int Y, int Z) {
    if (!P || P->nlevels < 1 || P->nlevels > GF_UNET_MAX_LEVELS || M0 < 0 || B < 1) return 0;
    long long offs[GF_UNET_MAX_LEVELS * 10];
    int caps[GF_UNET_MAX_LEVELS + 1], shapes[3 * (GF_UNET_MAX_LEVELS + 1)], nl = 0;
    long long chain = 0;
    return unet_layout(P, M0, B, X, Y, Z, offs, caps, shapes, &nl, &chain);
}

extern "C" int gf_unet_fwd(const GfUnetParams* P, const float* feats, const int32_t* coords, int M0, int B, int X, int Y,
                           int Z, void* ws, size_t ws_bytes, int32_t* host_counts, float* out, void* stream,
                           void* side_stream) {
    return gf_unet_fwd_phased(P, feats, coords, M0, B, X, Y, Z, ws, ws_bytes, host_counts, out, stream, side_stream, nullptr,
                              0, nullptr, nullptr);
}

static int unet_fwd_impl(const GfUnetParams* P, const float* feats, const int32_t* coords, int M0, int B, int X, int Y, int Z,
                         void* ws, size_t ws_bytes, int32_t* host_counts, float* out, void* stream, void* side_stream,
                         void* const* gate_events, int n_gate, GfUnetBetween between, void* user,
                         void* const* input_events, int n_input);

extern "C" int gf_unet_fwd_phased(const GfUnetParams* P, const float* feats, const int32_t* coords, int M0, int B, int X,
                                  int Y, int Z, void* ws, size_t ws_bytes, int32_t* host_counts, float* out, void* stream,
                                  void* side_stream, void* const* gate_events, int n_gate, GfUnetBetween between,
                                  void* user) {
    return unet_fwd_impl(P, feats, coords, M0, B, X, Y, Z, ws, ws_bytes, host_counts, out, stream, side_stream, gate_events,
                         n_gate, between, user, nullptr, -1);
}

// The rulebooks AHEAD of the caller's stream (round 6).  Index, tables and the down-sampling chain read the voxel
// coordinates and nothing else, and they are ~0.15 ms at the head of a backbone whose convolutions wait for them.  When the
// caller can say what the COORDINATES wait for -- input_events: n_input recorded events, none for coordinates that have
// been resident all along -- every rulebook launch goes to the side stream behind those events (and behind the end of
// this thread's previous call, which read the same workspace), NOT behind whatever `stream` still has queued: in a loop
// of forwards they run under the previous scene's latency-bound sampling / BFS stretch and this scene's convolutions
// start on finished tables.  `feats` is read there too (the zero-padded input rows): same events, or made on the side
// stream.  Same launches, same results as gf_unet_fwd; needs a side stream (else: gf_unet_fwd).
extern "C" int gf_unet_fwd_ahead(const GfUnetParams* P, const float* feats, const int32_t* coords, int M0, int B, int X, int Y,
                                 int Z, void* ws, size_t ws_bytes, int32_t* host_counts, float* out, void* stream,
                                 void* side_stream, void* const* input_events, int n_input) {
    GF_CHECK_ARG(n_input >= 0 && (n_input == 0 || input_events != nullptr), "gf_unet_fwd_ahead: %d input events without a list",
                 n_input);
    return unet_fwd_impl(P, feats, coords, M0, B, X, Y, Z, ws, ws_bytes, host_counts, out, stream, side_stream, nullptr, 0,
                         nullptr, nullptr, input_events, side_stream ? n_input : -1);
}

static int unet_fwd_impl(const GfUnetParams* P, const float* feats, const int32_t* coords, int M0, int B, int X, int Y, int Z,
                         void* ws, size_t ws_bytes, int32_t* host_counts, float* out, void* stream, void* side_stream,
                         void* const* gate_events, int n_gate, GfUnetBetween between, void* user,
                         void* const* input_events, int n_input) {
    GF_CHECK_ARG(P && feats && coords && ws && host_counts && out, "gf_unet_fwd: null argument");
    GF_CHECK_ARG(n_gate >= 0 && (n_gate == 0 || gate_events != nullptr), "gf_unet_fwd_phased: %d gate events without a list", n_gate);
    // the hand-over between the phases happens exactly once on every path that gets past the argument checks below
    bool handed = false;
    auto hand_over = [&](hipStream_t s_) -> int {
        if (handed) return GF_OK;
        handed = true;
        if (!between) return GF_OK;
        void* evs[8];
        const int n = between(user, evs, 8);
        if (n < 0 || n > 8) {
            gf_set_error("gf_unet_fwd_phased: the hand-over callback failed (%d)", n);
            return GF_ERR_CALLBACK;
        }
        for (int e = 0; e < n; e++) GF_TRY(hipStreamWaitEvent(s_, (hipEvent_t)evs[e], 0));
        return GF_OK;
    };
    GF_CHECK_ARG(P->nlevels >= 1 && P->nlevels <= GF_UNET_MAX_LEVELS, "gf_unet_fwd: %d levels (1..%d)", P->nlevels,
                 GF_UNET_MAX_LEVELS);
    GF_CHECK_ARG(P->cin >= 1 && P->cin <= 16 && P->level[0].C == 16,
                 "gf_unet_fwd: input conv implemented for <= 16 input channels and 16 output channels (got %d -> %d)",
                 P->cin, P->level[0].C);
    GF_CHECK_ARG(M0 >= 0 && B >= 1, "gf_unet_fwd: bad sizes");
    if (M0 == 0) return hand_over((hipStream_t)stream);
    for (int l = 0; l < P->nlevels; l++)
        GF_CHECK_ARG(P->level[l].C > 0 && P->level[l].C % 16 == 0, "gf_unet_fwd: level %d width %d (multiples of 16)", l,
                     P->level[l].C);
    long long offs[GF_UNET_MAX_LEVELS * 10];
    int caps[GF_UNET_MAX_LEVELS + 1], shapes[3 * (GF_UNET_MAX_LEVELS + 1)], nl = 0;
    long long chain_elems = 0;
    const size_t need = unet_layout(P, M0, B, X, Y, Z, offs, caps, shapes, &nl, &chain_elems);
    GF_CHECK_ARG(nl == P->nlevels - 1, "gf_unet_fwd: the %dx%dx%d grid supports %d of the %d down-samplings", X, Y, Z, nl,
                 P->nlevels - 1);
    GF_CHECK_ARG(ws_bytes >= need, "gf_unet_fwd: workspace of %zu bytes, need %zu (gf_unet_ws_bytes)", ws_bytes, need);
    GF_CHECK_ARG(((uintptr_t)ws % 256) == 0, "gf_unet_fwd: workspace must be 256-byte aligned");

    hipStream_t st = (hipStream_t)stream;
    hipStream_t ss = side_stream ? (hipStream_t)side_stream : st;
    const bool forked = ss != st;
    if (forked && !t_ev.fork) {
        GF_TRY(hipEventCreateWithFlags(&t_ev.fork, hipEventDisableTiming));
        GF_TRY(hipEventCreateWithFlags(&t_ev.rules, hipEventDisableTiming));
        GF_TRY(hipEventCreateWithFlags(&t_ev.tbl0, hipEventDisableTiming));
        GF_TRY(hipEventCreateWithFlags(&t_ev.flat0, hipEventDisableTiming));
        GF_TRY(hipEventCreateWithFlags(&t_ev.tbl1, hipEventDisableTiming));
        GF_TRY(hipEventCreateWithFlags(&t_ev.flat1, hipEventDisableTiming));
        GF_TRY(hipEventCreateWithFlags(&t_ev.ws_done, hipEventDisableTiming));
    }
    // rulebooks ahead of `st`: the side stream waits for the coordinates' own events and for the previous call's readers
    const bool ahead = forked && n_input >= 0;
    auto fork_side = [&]() -> int {
        if (!forked) return GF_OK;
        if (!ahead) {
            GF_TRY(hipStreamWaitEvent(ss, t_ev.fork, 0));
            return GF_OK;
        }
        for (int e = 0; e < n_input; e++) GF_TRY(hipStreamWaitEvent(ss, (hipEvent_t)input_events[e], 0));
        if (t_ev.ws_done_recorded) GF_TRY(hipStreamWaitEvent(ss, t_ev.ws_done, 0));
        return GF_OK;
    };
    hipStream_t s_rules = ahead ? ss : st;  // where the first two levels' index / tables are built
    if (!t_ev.chain) {
        GF_TRY(hipEventCreateWithFlags(&t_ev.chain, hipEventDisableTiming));
        GF_TRY(hipEventCreateWithFlags(&t_ev.chain2, hipEventDisableTiming));
    }

    Bump a(ws, ws_bytes);
    const size_t words0 = gf_index_words(B, X, Y, Z);
    uint32_t* bitmap0 = a.take<uint32_t>(words0);
    int32_t* prefix0 = a.take<int32_t>(words0);
    int32_t* perm0 = a.take<int32_t>(M0);
    void* iscratch = a.take<unsigned char>(gf_index_scratch_bytes(words0));
    LevelTables T[GF_UNET_MAX_LEVELS];
    const int ld0 = r16(M0);
    T[0].ld = ld0;
    T[0].nbr = a.take<int32_t>((size_t)27 * ld0);
    T[0].gmask = a.take<uint32_t>(ld0 / 16);
    {
        int32_t* s = a.take<int32_t>(gf_rules_steps_words(ld0));
        T[0].steps = ld0 >= kStepsMinRows ? s : nullptr;
    }
    if (ld0 >= kFlatMinRows) T[0].flat = a.take<int32_t>(gf_rules_flat_words(27, ld0));
    int32_t* cws = a.take<int32_t>((size_t)chain_elems);
    int32_t* d_counts = a.take<int32_t>(GF_UNET_MAX_LEVELS + 1);
    for (int l = 1; l <= nl; l++) {
        T[l].ld = caps[l];
        T[l].nbr = a.take<int32_t>((size_t)27 * caps[l]);
        T[l].gmask = a.take<uint32_t>(caps[l] / 16);
        if (caps[l] >= kFlatMinRows) T[l].flat = a.take<int32_t>(gf_rules_flat_words(27, caps[l]));
    }

    // feature buffers are carved after the counts are known for the levels below the first; level 1 now
    int M[GF_UNET_MAX_LEVELS];
    M[0] = M0;
    LevelBufs Bf[GF_UNET_MAX_LEVELS];
    auto carve = [&](int l) {
        carve_level(a, Bf[l], (size_t)r16(M[l]), (size_t)P->level[l].C, P->level[l].tr_layers > 0, B);
    };
    carve(0);
    float* x16 = a.take<float>((size_t)ld0 * 16);

    int rc;
#define UN_TRY(call)                  \
    do {                              \
        rc = (call);                  \
        if (rc != GF_OK) return rc;   \
    } while (0)

    // ---- main stream: level-1 index and table, input conv, first two blocks ----
    if (forked) GF_TRY(hipEventRecord(t_ev.fork, st));
    // the level-parallel rulebook chain is eight launches: queued FIRST, it runs beside the level-1 index and table
    // (~110 us of small integer kernels) and is over before the first convolution starts -- queued behind the level-1
    // convolutions (where the 36-launch serial chain has to go, below) its two fat kernels overlap the first two of
    // them (rocprofv3: 20.5 instead of 18.7 us per level-1 launch)
    const bool chain_first = nl > 0 && gf_rules_level_parallel();
    if (ahead) {
        // the first level's index and table FIRST on the side stream -- they are what the first convolution waits for --, the
        // chain behind them: with nothing of a previous scene to hide under (a loop that waits for every scene's results) the
        // chain's ~0.11 ms would otherwise sit in front of the first convolution instead of beside the first level's
        UN_TRY(fork_side());
        // (the zero-padded input rows too: the voxel features were made on this stream -- GeoFormer._inputs_ahead)
        hipLaunchKernelGGL(k_pad_channels, dim3(gf_div_up(M0 * 16, 256)), dim3(256), 0, ss, feats, M0, P->cin, 16, x16);
        UN_TRY(gf_index_build(coords, M0, nullptr, B, X, Y, Z, bitmap0, prefix0, perm0, iscratch, ss));
        UN_TRY(gf_rules_subm3(coords, M0, nullptr, X, Y, Z, bitmap0, prefix0, perm0, T[0].nbr, ld0, T[0].gmask, T[0].steps, ss));
        GF_TRY(hipEventRecord(t_ev.tbl0, ss));
        GF_TRY(hipStreamWaitEvent(st, t_ev.tbl0, 0));  // (the first level's convolutions read its table)
    }
    if (chain_first) {
        if (!ahead) UN_TRY(fork_side());
        UN_TRY(gf_rules_down2_chain_all(coords, M0, B, X, Y, Z, nl, cws, d_counts, ss));
        GF_TRY(hipMemcpyAsync(host_counts + 1, d_counts + 1, sizeof(int32_t) * nl, hipMemcpyDeviceToHost, ss));
        GF_TRY(hipEventRecord(t_ev.chain, ss));
        if (nl > 1) GF_TRY(hipEventRecord(t_ev.chain2, ss));
    }
    if (!ahead) {
        UN_TRY(gf_index_build(coords, M0, nullptr, B, X, Y, Z, bitmap0, prefix0, perm0, iscratch, st));
        UN_TRY(gf_rules_subm3(coords, M0, nullptr, X, Y, Z, bitmap0, prefix0, perm0, T[0].nbr, ld0, T[0].gmask, T[0].steps, st));
    }
    // the first level's flat step table (its one reader is the up pass's 32 -> 16 convolution, the last block of the call):
    // built on the side stream BEHIND the deeper levels' tables, i.e. beside the second level's convolutions -- beside the
    // first level's own it cost them 18 -> 31 us per launch (profiles/r6_conv_lw_notes.md)
    if (T[0].flat && M0 < kFlatMinRows) T[0].flat = nullptr;
    bool flat0_pending = false, flat1_pending = false;
    auto build_flat0 = [&]() -> int {
        if (!T[0].flat) return GF_OK;
        if (forked) {
            GF_TRY(hipStreamWaitEvent(ss, t_ev.tbl0, 0));  // (recorded below, right behind gf_rules_subm3 of the first level)
            const int r_ = gf_rules_flat_steps(T[0].nbr, T[0].gmask, 27, M0, ld0, 0, T[0].flat, ss);
            if (r_ != GF_OK) return r_;
            GF_TRY(hipEventRecord(t_ev.flat0, ss));
            flat0_pending = true;
            return GF_OK;
        }
        return gf_rules_flat_steps(T[0].nbr, T[0].gmask, 27, M0, ld0, 0, T[0].flat, st);
    };
    if (forked && !ahead) GF_TRY(hipEventRecord(t_ev.tbl0, st));


    // every convolution of the call goes through here (kind: 0 input, 1 / 2 first / second conv of a block, 3 its
    // 1x1x1 identity branch, 4 strided, 5 inverse); the dev probe, when armed, brackets the launch with two events.
    auto conv = [&](int l, int kind, const float* in, const float* wp, const int32_t* nbr, const uint32_t* gmask,
                    const int32_t* steps, int K, int M_in, int M_out, int ld, int Cin, int Cout, const float* sc,
                    const float* sh, const float* res, const float* osc, const float* osh, float* outp,
                    float* out_act = nullptr) -> int {
        Probe& pb = t_probe;
        const bool rec = pb.mode == 2 || ((pb.mode == 1 || pb.mode == 3) && l == 0 && (kind == 1 || kind == 2) && Cin == 16 && Cout == 16);
        // (a submanifold convolution of a level with a flat step table hands it over: gf_conv_fwd_flat)
        const int32_t* fl = (K == 27 && nbr != nullptr && nbr == T[l].nbr) ? T[l].flat : nullptr;
        if (fl && !gf_conv_lw_supported(K, M_in, M_out, Cin, Cout, true, nullptr)) fl = nullptr;  // (no wait for a table nobody reads)
        if (fl && l == 0 && flat0_pending) {
            GF_TRY(hipStreamWaitEvent(st, t_ev.flat0, 0));
            flat0_pending = false;
        }
        if (fl && l == 1 && flat1_pending) {
            GF_TRY(hipStreamWaitEvent(st, t_ev.flat1, 0));
            flat1_pending = false;
        }
        auto launch = [&]() -> int {
            if (fl)
                return gf_conv_fwd_flat(in, wp, nbr, gmask, steps, fl, K, M_in, M_out, ld, Cin, Cout, sc, sh, res, osc, osh, outp,
                                        out_act, st);
            if (out_act)
                return gf_conv_fwd_dual(in, wp, nbr, gmask, steps, K, M_in, M_out, ld, Cin, Cout, sc, sh, res, osc, osh, outp,
                                        out_act, st);
            return gf_conv_fwd(in, wp, nbr, gmask, steps, K, M_in, M_out, ld, Cin, Cout, sc, sh, res, osc, osh, outp, st);
        };
        if (!rec) return launch();
        ProbeRec r{l, kind, K, Cin, Cout, M_in, M_out, res != nullptr, -1, pb.ev(), pb.ev(), nullptr, nullptr};
        GF_CHECK_ARG(r.a && r.b, "gf_unet_fwd: probe events");
        if (pb.mode == 3) {
            r.ka = pb.ev();
            r.kb = pb.ev();
            GF_CHECK_ARG(r.ka && r.kb, "gf_unet_fwd: probe events");
            gf_dev_conv_kernel_events(r.ka, r.kb);
        }
        if (pb.mode == 2 && nbr && pb.nslots < kProbeSlots && M_out > 0) {
            r.slot = pb.nslots++;
            GF_TRY(hipMemsetAsync(pb.d_counts + r.slot, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_count_rules, dim3(256), dim3(256), 0, st, nbr, K, ld, M_out, pb.d_counts + r.slot);
        }
        GF_TRY(hipEventRecord(r.a, st));
        const int rc_ = launch();
        GF_TRY(hipEventRecord(r.b, st));
        if (r.ka && !gf_dev_conv_kernel_events_taken()) {
            gf_dev_conv_kernel_events(nullptr, nullptr);
            r.ka = r.kb = nullptr;
        }
        pb.recs.push_back(r);
        return rc_;
    };
    // pre-activation residual block = the launches of gf_resblock_fwd (spconv_conv.hip): 1x1x1 identity branch when the
    // widths differ, first conv with bn0+ReLU on its input and bn1+ReLU on its output, second conv + residual; the
    // block whose output feeds the output layer carries that BatchNorm + ReLU in its last epilogue (osc, osh)
    // x_act: the block's input already through its bn0 + ReLU (written by the producer's second output): the first
    // convolution then gathers activated rows and has no prologue.  next: the block after this one wants the same --
    // (scale, shift) of ITS bn0 and the buffer for the activated copy of this block's output
    struct NextAct {
        const float *s, *t;
        float* buf;
    };
    bool idn_made = false;  // the next resblock's identity branch is in Bf[l].idn already (k_concat2_idn)
    auto resblock = [&](const GfResBlockParams& rb, int l, int cin, const float* x, float* outp, const float* osc,
                        const float* osh, const float* x_act = nullptr, const NextAct* next = nullptr) -> int {
        const int C = P->level[l].C;
        const LevelTables& t = T[l];
        GF_CHECK_ARG(rb.wp0 && rb.wp1 && rb.s0 && rb.t0 && rb.s1 && rb.t1, "gf_unet_fwd: level %d: block parameters missing", l);
        GF_CHECK_ARG((rb.wpi != nullptr) == (cin != C), "gf_unet_fwd: level %d: identity-branch weights must exist iff the widths differ", l);
        GF_CHECK_ARG(!(next && osc), "gf_unet_fwd: a block has one epilogue activation");
        int r;
        if (rb.wpi && !idn_made) {
            r = conv(l, 3, x, rb.wpi, nullptr, nullptr, nullptr, 1, M[l], M[l], 0, cin, C, nullptr, nullptr, nullptr, nullptr,
                     nullptr, Bf[l].idn);
            if (r != GF_OK) return r;
        }
        idn_made = false;
        if (x_act)
            r = conv(l, 1, x_act, rb.wp0, t.nbr, t.gmask, t.steps, 27, M[l], M[l], t.ld, cin, C, nullptr, nullptr, nullptr, rb.s1,
                     rb.t1, Bf[l].tmp);
        else
            r = conv(l, 1, x, rb.wp0, t.nbr, t.gmask, t.steps, 27, M[l], M[l], t.ld, cin, C, rb.s0, rb.t0, nullptr, rb.s1, rb.t1,
                     Bf[l].tmp);
        if (r != GF_OK) return r;
        if (next)
            return conv(l, 2, Bf[l].tmp, rb.wp1, t.nbr, t.gmask, t.steps, 27, M[l], M[l], t.ld, C, C, nullptr, nullptr,
                        rb.wpi ? Bf[l].idn : x, next->s, next->t, outp, next->buf);
        return conv(l, 2, Bf[l].tmp, rb.wp1, t.nbr, t.gmask, t.steps, 27, M[l], M[l], t.ld, C, C, nullptr, nullptr,
                    rb.wpi ? Bf[l].idn : x, osc, osh, outp);
    };
    // the first level's 16 -> 16 convolutions run the pipelined counted-loop kernel, which can write both forms
    const bool dual0 = gf_conv_dual_supported(M0, ld0, 16, 16, T[0].steps != nullptr) != 0;

    // phased call: every convolution sits behind the caller's events; the level-1 index / table above and the rulebook
    // chain on the side stream (forked from a point in front of the waits) do not
    for (int e = 0; e < n_gate; e++) GF_TRY(hipStreamWaitEvent(st, (hipEvent_t)gate_events[e], 0));
    {
        const int n = M0 * 16;
        if (!ahead) hipLaunchKernelGGL(k_pad_channels, dim3(gf_div_up(n, 256)), dim3(256), 0, st, feats, M0, P->cin, 16, x16);
        const GfResBlockParams& b0 = P->level[0].blocks[0];
        if (dual0)
            UN_TRY(conv(0, 0, x16, P->input_wp, T[0].nbr, T[0].gmask, T[0].steps, 27, M0, M0, ld0, 16, 16, nullptr, nullptr, nullptr,
                        b0.s0, b0.t0, Bf[0].x, Bf[0].a0));
        else
            UN_TRY(conv(0, 0, x16, P->input_wp, T[0].nbr, T[0].gmask, T[0].steps, 27, M0, M0, ld0, 16, 16, nullptr, nullptr, nullptr,
                        nullptr, nullptr, Bf[0].x));
    }
    const bool single = P->nlevels == 1;
    {
        const GfResBlockParams& b1 = P->level[0].blocks[1];
        const NextAct n1{b1.s0, b1.t0, Bf[0].a1};
        UN_TRY(resblock(P->level[0].blocks[0], 0, 16, Bf[0].x, Bf[0].o0, nullptr, nullptr, dual0 ? Bf[0].a0 : nullptr,
                        dual0 ? &n1 : nullptr));
        const bool last = single && P->level[0].tr_layers == 0;
        UN_TRY(resblock(b1, 0, 16, Bf[0].o0, last ? out : Bf[0].o1, last ? P->out_s : nullptr, last ? P->out_t : nullptr,
                        dual0 ? Bf[0].a1 : nullptr, nullptr));
    }

    // ---- side stream: the chain of down-sampling rulebooks, queued AFTER the level-1 work (its ~35 launches are
    // 5 us of dispatch each and little else: the device runs them beside the level-1 convolutions; queued first they
    // would hold everything back for 0.2 ms on an idle device).  First its level 0 alone: the second level's voxel
    // count and tables are what the main stream needs next ----
    const int32_t* lcoords[GF_UNET_MAX_LEVELS];
    lcoords[0] = coords;
    host_counts[0] = M0;
    auto subm_tables = [&](int l, hipStream_t s_) -> int {
        if (M[l] == 0) return GF_OK;
        const long long* o = offs + (l - 1) * 10;
        const int r_ = gf_rules_subm3(lcoords[l], M[l], nullptr, shapes[3 * l], shapes[3 * l + 1], shapes[3 * l + 2],
                                      (const uint32_t*)(cws + o[0]), cws + o[1], nullptr, T[l].nbr, T[l].ld, T[l].gmask, nullptr, s_);
        if (r_ != GF_OK) return r_;
        if (T[l].flat && M[l] < kFlatMinRows) T[l].flat = nullptr;
        if (!T[l].flat) return GF_OK;
        if (l == 1 && ahead) {
            // (ahead of the main stream: table and flat table on the side stream, the main stream waits for both)
            const int r2 = gf_rules_flat_steps(T[l].nbr, T[l].gmask, 27, M[l], T[l].ld, 0, T[l].flat, s_);
            if (r2 != GF_OK) return r2;
            return GF_OK;
        }
        if (l == 1 && forked && s_ == st) {
            // the second level's flat table beside its strided convolution (which does not read it), not in front of it
            GF_TRY(hipEventRecord(t_ev.tbl1, st));
            GF_TRY(hipStreamWaitEvent(ss, t_ev.tbl1, 0));
            const int r2 = gf_rules_flat_steps(T[l].nbr, T[l].gmask, 27, M[l], T[l].ld, 0, T[l].flat, ss);
            if (r2 != GF_OK) return r2;
            GF_TRY(hipEventRecord(t_ev.flat1, ss));
            flat1_pending = true;
            return GF_OK;
        }
        return gf_rules_flat_steps(T[l].nbr, T[l].gmask, 27, M[l], T[l].ld, 0, T[l].flat, s_);
    };
    auto take_counts = [&](int l0, int l1) -> int {
        for (int l = l0; l <= l1; l++) {
            M[l] = host_counts[l];
            GF_CHECK_ARG(M[l] >= 0 && M[l] <= caps[l], "gf_unet_fwd: level %d reports %d voxels (capacity %d)", l + 1, M[l],
                         caps[l]);
            lcoords[l] = cws + offs[(l - 1) * 10 + 3];
        }
        return GF_OK;
    };
    bool blocks_done[GF_UNET_MAX_LEVELS] = {true};
    bool tr_tables[GF_UNET_MAX_LEVELS + 1] = {false};
    auto down_conv = [&](int l) -> int {
        const GfUnetLevelParams& L = P->level[l];
        const long long* o = offs + l * 10;
        GF_CHECK_ARG(L.down_wp && L.down_s && L.down_t && L.up_wp && L.up_s && L.up_t,
                     "gf_unet_fwd: level %d: strided / inverse conv parameters missing", l);
        // BN + ReLU + SparseConv3d(k=2, s=2): child table of the chain
        return conv(l, 4, Bf[l].o1, L.down_wp, cws + o[4], (const uint32_t*)(cws + o[8]), nullptr, 8, M[l], M[l + 1],
                    caps[l + 1], L.C, P->level[l + 1].C, L.down_s, L.down_t, nullptr, nullptr, nullptr, Bf[l + 1].x);
    };
    // a level whose C -> C convolutions run the LDS-weight kernel (both outputs): a block's second convolution also writes
    // the next block's activated input, whose first convolution then has no prologue (as the first level does, dual0)
    auto lw_dual = [&](int l) -> bool {
        return l > 0 && T[l].flat != nullptr && gf_conv_lw_supported(27, M[l], M[l], P->level[l].C, P->level[l].C, true, nullptr) != 0;
    };
    auto two_blocks = [&](int l) -> int {
        const GfUnetLevelParams& L = P->level[l];
        const bool dual = lw_dual(l);
        const NextAct n1{L.blocks[1].s0, L.blocks[1].t0, Bf[l].a1};
        int r = resblock(L.blocks[0], l, L.C, Bf[l].x, Bf[l].o0, nullptr, nullptr, nullptr, dual ? &n1 : nullptr);
        if (r != GF_OK) return r;
        blocks_done[l] = true;
        return resblock(L.blocks[1], l, L.C, Bf[l].o0, Bf[l].o1, nullptr, nullptr, dual ? Bf[l].a1 : nullptr, nullptr);
    };
    if (nl > 0) {
        // the whole chain is queued at once (it carries its counts on the device); two events mark the points the
        // host waits for
        if (!ahead) UN_TRY(fork_side());
        if (chain_first) {
            // (queued at the top of the call: every stage one launch over all levels, all counts and tables together)
        } else {
            UN_TRY(gf_rules_down2_chain_range(coords, M0, B, X, Y, Z, nl, 0, 1, cws, d_counts, ss));
            GF_TRY(hipMemcpyAsync(host_counts + 1, d_counts + 1, sizeof(int32_t), hipMemcpyDeviceToHost, ss));
            GF_TRY(hipEventRecord(t_ev.chain, ss));
            if (nl > 1) {
                UN_TRY(gf_rules_down2_chain_range(coords, M0, B, X, Y, Z, nl, 1, nl, cws, d_counts, ss));
                GF_TRY(hipMemcpyAsync(host_counts + 2, d_counts + 2, sizeof(int32_t) * (nl - 1), hipMemcpyDeviceToHost, ss));
                GF_TRY(hipEventRecord(t_ev.chain2, ss));
            }
        }
        // ---- end of phase A (level 1 and the whole rulebook chain queued): the caller's hand-over BEFORE this call's
        // first host wait -- the hand-over serves another scene's read-back, which must not sit behind a wait for THIS
        // scene's rulebooks (they crawl beside that scene's convolutions) -- then phase B behind the events it returns
        UN_TRY(hand_over(st));
        {
            WaitClock wc;
            GF_TRY(hipEventSynchronize(t_ev.chain));  // host wait 1: voxel count of the second level
        }
        UN_TRY(take_counts(1, 1));
        // main stream: second level's table, the first strided conv and that level's two blocks -- ~0.18 ms of device
        // work that needs nothing from the rest of the chain
        if (forked) GF_TRY(hipStreamWaitEvent(st, t_ev.chain, 0));
        UN_TRY(subm_tables(1, s_rules));
        if (ahead) {
            GF_TRY(hipEventRecord(t_ev.tbl1, ss));
            GF_TRY(hipStreamWaitEvent(st, t_ev.tbl1, 0));
        }
        carve(1);
        UN_TRY(down_conv(0));
        UN_TRY(two_blocks(1));
    }
    UN_TRY(hand_over(st));  // (a single-level net: here)
    if (nl > 1) {
        {
            WaitClock wc;
            GF_TRY(hipEventSynchronize(t_ev.chain2));  // host wait 2: the levels below (the device is busy with level 2)
        }
        UN_TRY(take_counts(2, nl));
        for (int l = 2; l <= nl; l++) UN_TRY(subm_tables(l, ss));
        for (int l = 2; l <= nl; l++) carve(l);
        // the voxel transformers' scene offsets and tile tables: functions of the levels' coordinates, built here with the
        // rulebooks instead of as two small launches in front of every transformer on the main stream
        for (int l = 2; l <= nl; l++)
            if (P->level[l].tr_layers > 0 && M[l] > 0 && Bf[l].tr_offs) {
                hipLaunchKernelGGL(k_scene_offsets, dim3(gf_div_up(M[l] + 1, 256)), dim3(256), 0, ss, lcoords[l], M[l], B,
                                   Bf[l].tr_offs);
                UN_TRY(gf_backbone_transformer_tables(Bf[l].tr_offs, B, M[l], Bf[l].tr_scratch, ss));
                tr_tables[l] = true;
            }
        if (forked) {
            GF_TRY(hipEventRecord(t_ev.rules, ss));
            GF_TRY(hipStreamWaitEvent(st, t_ev.rules, 0));
        }
        UN_TRY(build_flat0());
    }

    if (nl <= 1) UN_TRY(build_flat0());
    // ---- down pass ----
    for (int l = 1; l < nl; l++) {
        if (!blocks_done[l]) UN_TRY(two_blocks(l));
        UN_TRY(down_conv(l));
    }
    // ---- deepest level, then the up pass ----
    for (int l = nl; l >= 0; l--) {
        const GfUnetLevelParams& L = P->level[l];
        const float* cur;
        const bool last_tail_is_output = l == 0 && L.tr_layers == 0;
        if (l == nl) {
            if (!blocks_done[l]) UN_TRY(two_blocks(l));
            cur = Bf[l].o1;
            if (l == 0 && L.tr_layers == 0) cur = out;  // single-level net: already written with the output activation
        } else {
            const long long* o = offs + l * 10;
            // BN + ReLU + SparseInverseConv3d: the one-hot `up` table of the chain; rows without a coarse cell stay zero
            UN_TRY(conv(l, 5, Bf[l + 1].o2, L.up_wp, cws + o[7], (const uint32_t*)(cws + o[9]), nullptr, 8, M[l + 1], M[l],
                        caps[l], P->level[l + 1].C, L.C, L.up_s, L.up_t, nullptr, nullptr, nullptr, Bf[l].up));
            if (M[l] > 0 && L.tail[0].wpi && L.C % 16 == 0 && L.C / 8 <= IDC_MAXCH) {
                // concatenation + the tail block's identity branch in one launch; the probe books it as that convolution
                Probe& pb = t_probe;
                const bool rec = pb.mode == 2;
                ProbeRec pr{l, 3, 1, 2 * L.C, L.C, M[l], M[l], 0, -1, rec ? pb.ev() : nullptr, rec ? pb.ev() : nullptr, nullptr,
                            nullptr};
                if (rec) GF_TRY(hipEventRecord(pr.a, st));
                const int items = ((M[l] + 15) / 16) * (L.C / 16);
                hipLaunchKernelGGL(k_concat2_idn, dim3(gf_div_up(items, 4)), dim3(256), 0, st, Bf[l].o1, Bf[l].up, M[l], L.C,
                                   (const float4*)L.tail[0].wpi, Bf[l].cat, Bf[l].idn);
                if (rec) {
                    GF_TRY(hipEventRecord(pr.b, st));
                    pb.recs.push_back(pr);
                }
                idn_made = true;
            } else if (M[l] > 0) {
                const int c4 = L.C / 4, n = M[l] * 2 * c4;
                hipLaunchKernelGGL(k_concat2, dim3(gf_div_up(n, 256)), dim3(256), 0, st, (const float4*)Bf[l].o1,
                                   (const float4*)Bf[l].up, M[l], c4, (float4*)Bf[l].cat);
            }
            const bool dual = (l == 0 && dual0) || lw_dual(l);  // tail[0]'s second conv writes tail[1]'s activated input as well
            const NextAct nt{L.tail[1].s0, L.tail[1].t0, Bf[l].a1};
            UN_TRY(resblock(L.tail[0], l, 2 * L.C, Bf[l].cat, Bf[l].o0, nullptr, nullptr, nullptr, dual ? &nt : nullptr));
            float* dst = last_tail_is_output ? out : Bf[l].o2;
            UN_TRY(resblock(L.tail[1], l, L.C, Bf[l].o0, dst, last_tail_is_output ? P->out_s : nullptr,
                            last_tail_is_output ? P->out_t : nullptr, dual ? Bf[l].a1 : nullptr, nullptr));
            cur = dst;
        }
        if (L.tr_layers > 0) {
            GF_CHECK_ARG(l > 0, "gf_unet_fwd: the voxel transformer needs batch-major rows (levels below the first)");
            GF_CHECK_ARG(L.tr_params != nullptr, "gf_unet_fwd: level %d: transformer parameters missing", l);
            if (M[l] > 0 && tr_tables[l]) {
                UN_TRY(gf_backbone_transformer_prepared(cur, lcoords[l], Bf[l].tr_offs, B, M[l], L.C, L.tr_layers, L.tr_params,
                                                        Bf[l].tr_scratch, Bf[l].tr, st));
                cur = Bf[l].tr;
            } else if (M[l] > 0) {
                hipLaunchKernelGGL(k_scene_offsets, dim3(gf_div_up(M[l] + 1, 256)), dim3(256), 0, st, lcoords[l], M[l], B,
                                   Bf[l].tr_offs);
                UN_TRY(gf_backbone_transformer(cur, lcoords[l], Bf[l].tr_offs, B, M[l], L.C, L.tr_layers, L.tr_params,
                                               Bf[l].tr_scratch, Bf[l].tr, st));
                cur = Bf[l].tr;  // (a level without voxels has no rows to hand on: `tr` would be unwritten)
            }
        }
        if (l > 0 && cur != Bf[l].o2) {
            // hand the level's result to the inverse conv above under one name
            Bf[l].o2 = const_cast<float*>(cur);
        }
    }
    // a flat table nobody read (the dev knob switched its kernel off): its build on the side stream still has to be over
    // before the caller reuses the workspace
    if (flat0_pending) GF_TRY(hipStreamWaitEvent(st, t_ev.flat0, 0));
    if (flat1_pending) GF_TRY(hipStreamWaitEvent(st, t_ev.flat1, 0));
    if (forked) {  // from here on nothing of this call reads the tables: a later call may build its own AHEAD of `st`
        GF_TRY(hipEventRecord(t_ev.ws_done, st));
        t_ev.ws_done_recorded = true;
    }
#undef UN_TRY
    GF_CHECK_LAUNCH("gf_unet_fwd");
    return GF_OK;
}
