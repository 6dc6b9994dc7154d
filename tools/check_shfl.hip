// Dev tool: gf_shfl_xor<D> / gf_wave_sum (csrc/common.h) against __shfl_xor on the device.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -Igeoformer_amd/csrc tools/check_shfl.hip -o /tmp/check_shfl && /tmp/check_shfl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "common.h"
__global__ void k(const float* in, float* out) {
    const float v = in[threadIdx.x];
    float* o = out + threadIdx.x * 8;
    o[0] = gf_shfl_xor<32>(v) - __shfl_xor(v, 32, 64);
    o[1] = gf_shfl_xor<16>(v) - __shfl_xor(v, 16, 64);
    o[2] = gf_shfl_xor<8>(v) - __shfl_xor(v, 8, 64);
    o[3] = gf_shfl_xor<4>(v) - __shfl_xor(v, 4, 64);
    o[4] = gf_shfl_xor<2>(v) - __shfl_xor(v, 2, 64);
    o[5] = gf_shfl_xor<1>(v) - __shfl_xor(v, 1, 64);
    float s = v;
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    o[6] = gf_wave_sum(v) - s;
    o[7] = 0.f;
}
int main() {
    float h[128], *d_in, *d_out, r[128 * 8];
    for (int i = 0; i < 128; i++) h[i] = 1.0f + 0.37f * i + 1e-3f * i * i;
    hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, sizeof(r));
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d_in, d_out);
    hipMemcpy(r, d_out, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128 * 8; i++) if (r[i] != 0.f) { if (bad < 10) printf("lane %d field %d: %g\n", i / 8, i % 8, r[i]); bad++; }
    printf(bad ? "MISMATCH (%d)\n" : "gf_shfl_xor / gf_wave_sum: identical to __shfl_xor (%d)\n", bad);
    return bad != 0;
}
