// pk_f32.hip -- what packed fp32 vector instructions cost on an MI355X beside their scalar-per-lane forms (round 5).
// The fragment-list nonbonded kernel is bound by vector-instruction issue (295 per chunk of 64 fragment pairs): if v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 issue at the cost of ONE v_fma_f32, a lane that evaluates two atom pairs side by side halves the
// arithmetic part of its instruction stream.  Each kernel runs `iters` rounds of 16 independent chains per lane (no dependency stalls),
// on every SIMD of the chip, at 1, 2 and 4 waves per SIMD; reported: cycles per wave-instruction per SIMD (clock64 inside the kernel).
//   hipcc --offload-arch=gfx950 -O3 -o pk_f32 scripts/pk_f32.hip && ./pk_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND> __global__ void __launch_bounds__(256) k_loop(float* out, long long* cyc, int iters, float seed) {
    float2v a[16];
    for (int q = 0; q < 16; q++) a[q] = float2v{seed + q, seed - q};
    const float2v m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[q].x) : "v"(m.x), "v"(c.x));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(m), "v"(c));
            if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[q].x) : "v"(m.x));
            if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[q]) : "v"(m));
            if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[q].x) : "v"(c.x));
            if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[q]) : "v"(c));
            if (KIND == 6) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[q].x) : "v"(m.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[q].y) : "v"(m.y), "v"(c.y)); }
            if (KIND == 7) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[q].x));
            if (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q].x) : "v"(m.x));
            if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(a[q].x) : "v"(m.x));
            if (KIND == 10) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(a[q]) : "v"(m));
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int q = 0; q < 16; q++) s += a[q].x + a[q].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> static void run(const char* name, int per_iter, float* out, long long* cyc, long long* h) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    printf("%-28s", name);
    for (int wps : {1, 2, 4}) {   // waves per SIMD: a 256-thread workgroup is one wave on each SIMD of a CU
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k_loop<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10, 1.0f);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_loop<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
        double mean = 0; for (int b = 0; b < blocks; b++) mean += (double)h[b]; mean /= blocks;
        // clock64 = s_memtime: a 100 MHz counter on this part; wall time gives cycles at the 2.4 GHz the guide quotes
        const double n_instr = (double)iters * 16 * per_iter * wps;   // wave-instructions per SIMD
        printf("  %dw: %6.2f cyc/instr (wall %.0f us)", wps, ms * 1e-3 * 2.4e9 / n_instr, ms * 1e3);
        (void)mean;
    }
    printf("\n");
}

int main() {
    float* out; long long* cyc; CHECK(hipMalloc(&out, 4 * 256 * 1024)); CHECK(hipMalloc(&cyc, 8 * 1024));
    long long* h = (long long*)malloc(8 * 1024);
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs, clock %d kHz; cycles at 2.4 GHz per wave-instruction and SIMD (16 independent chains per lane)\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("v_fma_f32", 1, out, cyc, h);
    run<1>("v_pk_fma_f32", 1, out, cyc, h);
    run<6>("2 x v_fma_f32", 2, out, cyc, h);
    run<2>("v_mul_f32", 1, out, cyc, h);
    run<3>("v_pk_mul_f32", 1, out, cyc, h);
    run<4>("v_add_f32", 1, out, cyc, h);
    run<5>("v_pk_add_f32", 1, out, cyc, h);
    run<7>("v_rsq_f32", 1, out, cyc, h);
    run<9>("v_mov_b32", 1, out, cyc, h);
    run<10>("v_pk_mov_b32", 1, out, cyc, h);
    return 0;
}
