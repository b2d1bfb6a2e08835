// standalone check of wave_sum_dpp_f32 (device_common.h) on a gfx950: hipcc --offload-arch=gfx950 -O3 -I blues_amd/csrc scripts/devtests/dpp_f32_test.hip -o scripts/devtests/dpp_f32_test
#include "device_common.h"
#include <cstdio>
__global__ void k(const float* in, float* out) { out[threadIdx.x] = wave_sum_dpp_f32(in[threadIdx.x]); }
int main() {
    float h[64], r[64], *d, *o; double s = 0;
    for (int i = 0; i < 64; i++) { h[i] = (float)(i * i % 17) - 3.5f; s += h[i]; }
    hipMalloc(&d, 256); hipMalloc(&o, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o); hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    printf("expected %.3f got lane0 %.3f lane63 %.3f lane17 %.3f\n", s, r[0], r[63], r[17]);
    return 0;
}
