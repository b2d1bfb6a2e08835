#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../blues_amd/csrc/device_common.h"
__global__ void k(const double* in, double* out, int* cnt) {
    const int lane = threadIdx.x;
    double v = in[lane];
    out[lane] = wave_sum_dpp(v);
    out[64 + lane] = wave_sum(v);
    const bool pass = (lane % 3) == 0;
    const unsigned long long bal = __ballot(pass);
    const unsigned blo = (unsigned)bal, bhi = (unsigned)(bal >> 32);
    int base = 5;
    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0u));
    int c1, c2;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(c1) : "s"(blo), "v"(base));
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(c2) : "s"(bhi), "v"(c1));
    cnt[lane] = pos; cnt[64 + lane] = c2;
}
int main() {
    double h[64], o[128]; int c[128];
    for (int i = 0; i < 64; i++) h[i] = 1.0 + 0.001 * i * i;
    double *din, *dout; int* dc;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o); hipMalloc(&dc, sizeof c);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dc);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost); hipMemcpy(c, dc, sizeof c, hipMemcpyDeviceToHost);
    double ref = 0; for (int i = 0; i < 64; i++) ref += h[i];
    printf("ref %.12f dpp[0] %.12f dpp[37] %.12f shfl[0] %.12f\n", ref, o[0], o[37], o[64]);
    int bad = 0; for (int i = 0; i < 64; i++) if (o[i] != o[0]) bad++;
    printf("lanes differing from lane 0: %d\n", bad);
    printf("pos[0..6] %d %d %d %d %d %d %d  total %d (expect 5+22=27)\n", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[64]);
    return 0;
}
