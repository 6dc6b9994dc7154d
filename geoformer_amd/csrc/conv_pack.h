// Element functions of the MFMA B-operand weight packing (gf_conv_pack_weights / gf_conv_pack_weights_t), shared by the
// per-convolution kernels (spconv_conv.hip) and the training executor's one-launch-per-step form (unet_train.hip).
#pragma once
#include <hip/hip_runtime.h>

// float4 t of the packed stream of W [K,Cin,Cout]:
//   Wp[(((k*NCH + ch)*NCB + cb)*64 + lane)*4 + kk] = W[k][ch*16 + 4*(lane>>4) + kk][cb*16 + (lane&15)]
__device__ __forceinline__ float4 gf_pack_weights_elem(const float* __restrict__ W, int Cin, int Cout, int NCH, int NCB,
                                                       size_t t) {
    const int lane = (int)(t & 63);
    size_t u = t >> 6;
    const int cb = (int)(u % NCB);
    u /= NCB;
    const int ch = (int)(u % NCH);
    const int k = (int)(u / NCH);
    const int r = lane & 15, q = lane >> 4;
    const int col = cb * 16 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int row = ch * 16 + 4 * q + kk;
        if (row < Cin && col < Cout) pv[kk] = W[((size_t)k * Cin + row) * Cout + col];
    }
    return v;
}

// float4 t of the packed stream of W' [K,Cout,Cin], W'[k] = W[flip ? K-1-k : k]^T (the input gradient's weights);
// NCH / NCB are those of W': ceil(Cout/16) / ceil(Cin/16)
__device__ __forceinline__ float4 gf_pack_weights_t_elem(const float* __restrict__ W, int K, int Cin, int Cout, int NCH,
                                                         int NCB, int flip, size_t t) {
    const int lane = (int)(t & 63);
    size_t u = t >> 6;
    const int cb = (int)(u % NCB);
    u /= NCB;
    const int ch = (int)(u % NCH);
    const int k = (int)(u / NCH);
    const int ks = flip ? K - 1 - k : k;
    const int r = lane & 15, q = lane >> 4;
    const int col = cb * 16 + r;  // column of W' = input channel of W
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int row = ch * 16 + 4 * q + kk;  // row of W' = output channel of W
        if (row < Cout && col < Cin) pv[kk] = W[((size_t)ks * Cin + col) * Cout + row];
    }
    return v;
}
