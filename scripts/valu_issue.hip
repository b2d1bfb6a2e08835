// valu_issue.hip -- what does a gfx950 SIMD issue per clock?  (settles VERDICT r01 weak #4)
//
// K independent v_fma_f32 chains per lane, W waves per SIMD, every CU busy.  Reports wave-instructions per second over the
// whole chip and the cycles one SIMD spends per wave64 VALU instruction, for plain fp32 FMA, packed fp32 FMA
// (v_pk_fma_f32: two fp32 FMAs per lane per instruction), a transcendental (v_exp_f32) and fp64 FMA.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue scripts/valu_issue.hip && ./valu_issue
//
// Reading: MI355X_MICROARCH.md says "v_fma_f32 (wave64) 2 cyc; one wave alone: 4".  If that holds, the VALU ceiling of a
// kernel with >= 2 waves per SIMD resident is 1024 SIMDs x clock / 2 wave-instr/s, twice what DESIGN.md r01 assumed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, int MODE>
__global__ void __launch_bounds__(256) k_issue(float* out, int iters, float seed) {
    float a[K]; double d[K];
#pragma unroll
    for (int k = 0; k < K; k++) { a[k] = seed + k + threadIdx.x * 1e-3f; d[k] = a[k]; }
    const float b = 0.999f, c = 1e-3f; const double bd = 0.999, cd = 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                else if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
                else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(bd), "v"(cd));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) s += a[k] + (float)d[k];
    if (s == 12345.678f) out[0] = s;
}

// packed fp32: a register PAIR per chain (two floats per lane), one v_pk_fma_f32 updates both
template <int K>
__global__ void __launch_bounds__(256) k_issue_pk(float* out, int iters, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[K];
#pragma unroll
    for (int k = 0; k < K; k++) { a[k].x = seed + k + threadIdx.x * 1e-3f; a[k].y = a[k].x * 0.5f; }
    const f2 b = {0.999f, 0.998f}, c = {1e-3f, 2e-3f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int k = 0; k < K; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) s += a[k].x + a[k].y;
    if (s == 12345.678f) out[0] = s;
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate * 1e-6;
    printf("device: %s, %d CUs, %d SIMDs, max clock %.2f GHz\n", prop.gcnArchName, cus, cus * 4, clock_ghz);
    float* out; CHECK(hipMalloc(&out, 64));
    const int iters = 4096;
    printf("%-14s %6s %10s %14s %16s %12s\n", "instruction", "chains", "waves/SIMD", "T wave-instr/s", "cyc/instr/SIMD", "ms");
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_fma_f64"};
    for (int mode = 0; mode < 4; mode++) {
        for (int K : {1, 2, 4, 8}) {
            for (int w : {1, 2, 4, 8}) {
                const int blocks = cus * w;   // 256-thread blocks: 4 waves, one per SIMD -> w waves per SIMD when all are resident
                auto launch = [&]() {
#define L(KK) do { if (mode == 1) hipLaunchKernelGGL(k_issue_pk<KK>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); \
                   else if (mode == 0) hipLaunchKernelGGL((k_issue<KK, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); \
                   else if (mode == 2) hipLaunchKernelGGL((k_issue<KK, 2>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); \
                   else hipLaunchKernelGGL((k_issue<KK, 3>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); } while (0)
                    if (K == 1) L(1); else if (K == 2) L(2); else if (K == 4) L(4); else L(8);
#undef L
                };
                const double ms = time_ms(launch, 5);
                const double insts = (double)blocks * 4 * iters * 8 * K;          // wave-instructions
                const double rate = insts / (ms * 1e-3) / 1e12;
                const double cyc = (double)cus * 4 * clock_ghz * 1e9 / (rate * 1e12);
                printf("%-14s %6d %10d %14.3f %16.2f %12.3f\n", names[mode], K, w, rate, cyc, ms);
            }
        }
    }
    CHECK(hipFree(out));
    return 0;
}
