// Developer probe: effective shader clock and dependent-fp64 latency under a sparse stream of tiny kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chain(double* out, int n, long long* cyc, long long* rt) {
    long long c0 = clock64(), r0 = wall_clock64();
    double a = out[threadIdx.x], b = 1.0000001;
    for (int i = 0; i < n; i++) a = fma(a, b, 1e-9);
    long long c1 = clock64(), r1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; rt[blockIdx.x] = r1 - r0; }
}
__global__ void chain4(double* out, int n, long long* cyc, long long* rt) {
    long long c0 = clock64(), r0 = wall_clock64();
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001;
    for (int i = 0; i < n; i++) { a0 = fma(a0, b, 1e-9); a1 = fma(a1, b, 1e-9); a2 = fma(a2, b, 1e-9); a3 = fma(a3, b, 1e-9); }
    long long c1 = clock64(), r1 = wall_clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; rt[blockIdx.x] = r1 - r0; }
}
int main() {
    double* d; long long *cyc, *rt; hipMalloc(&d, 8 * 1024); hipMalloc(&cyc, 8 * 1024); hipMalloc(&rt, 8 * 1024); hipMemset(d, 0, 8 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 2, 256, 2048}) for (int mode = 0; mode < 2; mode++) {
        const int n = 4000, reps = 300;
        for (int w = 0; w < 20; w++) { if (mode) chain4<<<blocks, 64>>>(d, n, cyc, rt); else chain<<<blocks, 64>>>(d, n, cyc, rt); }
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) { if (mode) chain4<<<blocks, 64>>>(d, n, cyc, rt); else chain<<<blocks, 64>>>(d, n, cyc, rt); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc, hr; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hr, rt, 8, hipMemcpyDeviceToHost);
        printf("blocks %4d %s: %.2f us/launch, %lld cycles, %.2f us in-kernel, clock %.0f MHz, %.1f cycles per %s\n", blocks, mode ? "4 chains" : "1 chain ",
               1e3 * ms / reps, hc, hr / 100.0, hc / (hr / 100.0), (double)hc / n, mode ? "4 fma" : "fma");
    }
    return 0;
}
