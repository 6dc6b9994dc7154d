// Natives that GeoFormer inherits from PointGroup / PointNet++ but never executes.
// Implemented for binding completeness with the reference's semantics; all are small segment / copy /
// brute-force kernels (HBM- or latency-bound, integer or copy work).
//   sec_mean/min/max   <- lib/pointgroup_ops/src/sec_mean/sec_mean.cu:12-86
//   roipool fp/bp      <- lib/pointgroup_ops/src/roipool/roipool.cu:12-57
//   get_iou            <- lib/pointgroup_ops/src/get_iou/get_iou.cu:12-38
//   bfs_cluster (host) <- lib/pointgroup_ops/src/bfs_cluster/bfs_cluster.cpp:28-111
//   three_nn / three_interpolate(+grad) <- lib/pointnet2/_ext_src/src/interpolate_gpu.cu:12-157
#include <queue>
#include <vector>

#include "common.h"

__global__ void k_sec_op(int kind, const float* __restrict__ inp, const int32_t* __restrict__ offsets, int nP, int C,
                         float* __restrict__ out) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nP * C) return;
    const int p = (int)(t / C), c = (int)(t - (long long)p * C);
    const int s = offsets[p], e = offsets[p + 1];
    if (kind == 0) {
        const float cnt = (float)(e - s);
        float mean = 0.f;
        for (int i = s; i < e; i++) mean = __fadd_rn(mean, __fdiv_rn(inp[(size_t)i * C + c], cnt));  // sec_mean.cu:22
        out[t] = mean;
    } else if (kind == 1) {
        float v = __builtin_inff();  // 1e50 -> +inf in fp32
        for (int i = s; i < e; i++) v = inp[(size_t)i * C + c] < v ? inp[(size_t)i * C + c] : v;
        out[t] = v;
    } else {
        float v = -__builtin_inff();
        for (int i = s; i < e; i++) v = inp[(size_t)i * C + c] > v ? inp[(size_t)i * C + c] : v;
        out[t] = v;
    }
}
extern "C" int gf_sec_op(int kind, const float* inp, const int32_t* offsets, int nProposal, int C, float* out,
                         void* stream) {
    GF_CHECK_ARG(kind >= 0 && kind <= 2 && nProposal >= 0 && C >= 1, "gf_sec_op: bad arguments");
    if (nProposal == 0) return GF_OK;
    hipLaunchKernelGGL(k_sec_op, dim3(gf_div_up((long long)nProposal * C, 256)), dim3(256), 0, (hipStream_t)stream, kind,
                       inp, offsets, nProposal, C, out);
    GF_CHECK_LAUNCH("gf_sec_op");
    return GF_OK;
}

__global__ void k_roipool_fp(const float* __restrict__ feats, const int32_t* __restrict__ offsets, int nP, int C,
                             float* __restrict__ out, int32_t* __restrict__ maxidx) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nP * C) return;
    const int p = (int)(t / C), c = (int)(t - (long long)p * C);
    int arg = -1;
    float v = -__builtin_inff();
    for (int i = offsets[p]; i < offsets[p + 1]; i++) {
        const float x = feats[(size_t)i * C + c];
        if (x > v) {
            v = x;
            arg = i;
        }
    }
    out[t] = v;
    maxidx[t] = arg;
}
__global__ void k_roipool_bp(const float* __restrict__ d_out, const int32_t* __restrict__ maxidx, int nP, int C,
                             float* __restrict__ d_feats) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nP * C) return;
    const int c = (int)(t % C);
    const int arg = maxidx[t];
    if (arg >= 0) atomicAdd(&d_feats[(size_t)arg * C + c], d_out[t]);
}
extern "C" int gf_roipool_fp(const float* feats, const int32_t* offsets, int nProposal, int C, float* out,
                             int32_t* maxidx, void* stream) {
    if (nProposal <= 0) return GF_OK;
    hipLaunchKernelGGL(k_roipool_fp, dim3(gf_div_up((long long)nProposal * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       feats, offsets, nProposal, C, out, maxidx);
    GF_CHECK_LAUNCH("gf_roipool_fp");
    return GF_OK;
}
extern "C" int gf_roipool_bp(const float* d_out, const int32_t* maxidx, int nProposal, int C, float* d_feats,
                             void* stream) {
    if (nProposal <= 0) return GF_OK;
    hipLaunchKernelGGL(k_roipool_bp, dim3(gf_div_up((long long)nProposal * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_out, maxidx, nProposal, C, d_feats);
    GF_CHECK_LAUNCH("gf_roipool_bp");
    return GF_OK;
}

__global__ void k_get_iou(const int32_t* __restrict__ pidx, const int32_t* __restrict__ poff,
                          const long long* __restrict__ inst_labels, const int32_t* __restrict__ inst_pointnum, int nInst,
                          int nP, float* __restrict__ iou) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nP * nInst) return;
    const int p = (int)(t / nInst), inst = (int)(t - (long long)p * nInst);
    const int s = poff[p], e = poff[p + 1];
    int inter = 0;
    for (int i = s; i < e; i++) inter += ((int)inst_labels[pidx[i]] == inst) ? 1 : 0;
    const float denom = __fadd_rn((float)((e - s) + inst_pointnum[inst] - inter), 1e-5f);
    iou[t] = __fdiv_rn((float)inter, denom);
}
extern "C" int gf_get_iou(const int32_t* proposals_idx, const int32_t* proposals_offset, const long long* instance_labels,
                          const int32_t* instance_pointnum, int nInstance, int nProposal, float* iou, void* stream) {
    if (nProposal <= 0 || nInstance <= 0) return GF_OK;
    hipLaunchKernelGGL(k_get_iou, dim3(gf_div_up((long long)nProposal * nInstance, 256)), dim3(256), 0,
                       (hipStream_t)stream, proposals_idx, proposals_offset, instance_labels, instance_pointnum, nInstance,
                       nProposal, iou);
    GF_CHECK_LAUNCH("gf_get_iou");
    return GF_OK;
}

// ---- bfs_cluster: HOST routine (the reference runs it on CPU tensors, bfs_cluster.cpp:28-111) ----
extern "C" int gf_bfs_cluster_host(const int32_t* h_semantic_label, const int32_t* h_ball_query_idxs,
                                   const int32_t* h_start_len, int N, int threshold, int32_t* h_cluster_idxs,
                                   int32_t* h_cluster_offsets, int32_t* h_nCluster, int32_t* h_sumNPoint) {
    GF_CHECK_ARG(N >= 0, "gf_bfs_cluster_host: bad N");
    std::vector<char> visited((size_t)N, 0);
    int nCluster = 0, sum = 0;
    h_cluster_offsets[0] = 0;
    std::vector<int32_t> cc;
    for (int i = 0; i < N; i++) {
        if (visited[i]) continue;
        cc.clear();
        cc.push_back(i);
        visited[i] = 1;
        std::queue<int32_t> Q;
        Q.push(i);
        while (!Q.empty()) {
            const int cur = Q.front();
            Q.pop();
            const int start = h_start_len[cur * 2], len = h_start_len[cur * 2 + 1];
            const int label = h_semantic_label[cur];
            for (int j = start; j < start + len; j++) {
                const int v = h_ball_query_idxs[j];
                if (h_semantic_label[v] != label || visited[v]) continue;
                cc.push_back(v);
                visited[v] = 1;
                Q.push(v);
            }
        }
        if ((int)cc.size() >= threshold) {
            for (size_t j = 0; j < cc.size(); j++) {
                h_cluster_idxs[(size_t)(sum + j) * 2 + 0] = nCluster;
                h_cluster_idxs[(size_t)(sum + j) * 2 + 1] = cc[j];
            }
            sum += (int)cc.size();
            nCluster++;
            h_cluster_offsets[nCluster] = sum;
        }
    }
    *h_nCluster = nCluster;
    *h_sumNPoint = sum;
    return GF_OK;
}

// ---- three_nn / three_interpolate ----
__global__ void k_three_nn(const float* __restrict__ unknown, const float* __restrict__ known, int n, int m,
                           float* __restrict__ dist2, int32_t* __restrict__ idx) {
    const int bi = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    unknown += (size_t)bi * n * 3;
    known += (size_t)bi * m * 3;
    const float ux = unknown[j * 3 + 0], uy = unknown[j * 3 + 1], uz = unknown[j * 3 + 2];
    double b1 = 1e40, b2 = 1e40, b3 = 1e40;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < m; k++) {
        const float dx = ux - known[k * 3 + 0], dy = uy - known[k * 3 + 1], dz = uz - known[k * 3 + 2];
        const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        if (d < b1) {
            b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
        } else if (d < b2) {
            b3 = b2; i3 = i2; b2 = d; i2 = k;
        } else if (d < b3) {
            b3 = d; i3 = k;
        }
    }
    float* d2 = dist2 + ((size_t)bi * n + j) * 3;
    int32_t* ix = idx + ((size_t)bi * n + j) * 3;
    d2[0] = (float)b1; d2[1] = (float)b2; d2[2] = (float)b3;
    ix[0] = i1; ix[1] = i2; ix[2] = i3;
}
extern "C" int gf_three_nn(const float* unknown, const float* known, int b, int n, int m, float* dist2, int32_t* idx,
                           void* stream) {
    if (b <= 0 || n <= 0) return GF_OK;
    hipLaunchKernelGGL(k_three_nn, dim3(gf_div_up(n, 256), b), dim3(256), 0, (hipStream_t)stream, unknown, known, n, m,
                       dist2, idx);
    GF_CHECK_LAUNCH("gf_three_nn");
    return GF_OK;
}
__global__ void k_three_interpolate(const float* __restrict__ points, const int32_t* __restrict__ idx,
                                    const float* __restrict__ weight, int c, int m, int n, float* __restrict__ out,
                                    long long total) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % n);
    const long long bc = t / n;
    const int bi = (int)(bc / c);
    const float* w = weight + ((size_t)bi * n + j) * 3;
    const int32_t* ix = idx + ((size_t)bi * n + j) * 3;
    const float* p = points + bc * m;
    out[t] = fmaf(p[ix[2]], w[2], fmaf(p[ix[1]], w[1], p[ix[0]] * w[0]));
}
__global__ void k_three_interpolate_grad(const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
                                         const float* __restrict__ weight, int c, int n, int m,
                                         float* __restrict__ grad_points, long long total) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % n);
    const long long bc = t / n;
    const int bi = (int)(bc / c);
    const float* w = weight + ((size_t)bi * n + j) * 3;
    const int32_t* ix = idx + ((size_t)bi * n + j) * 3;
    const float g = grad_out[t];
    float* gp = grad_points + bc * m;
    atomicAdd(&gp[ix[0]], g * w[0]);
    atomicAdd(&gp[ix[1]], g * w[1]);
    atomicAdd(&gp[ix[2]], g * w[2]);
}
extern "C" int gf_three_interpolate(const float* points, const int32_t* idx, const float* weight, int b, int c, int m,
                                    int n, float* out, void* stream) {
    const long long total = (long long)b * c * n;
    if (total <= 0) return GF_OK;
    hipLaunchKernelGGL(k_three_interpolate, dim3(gf_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, points, idx,
                       weight, c, m, n, out, total);
    GF_CHECK_LAUNCH("gf_three_interpolate");
    return GF_OK;
}
extern "C" int gf_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight, int b, int c,
                                         int n, int m, float* grad_points, void* stream) {
    const long long total = (long long)b * c * n;
    if (total <= 0) return GF_OK;
    hipLaunchKernelGGL(k_three_interpolate_grad, dim3(gf_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, grad_out,
                       idx, weight, c, n, m, grad_points, total);
    GF_CHECK_LAUNCH("gf_three_interpolate_grad");
    return GF_OK;
}
