// valu_mix.hip -- issue cost of the NON-FMA instructions of the nonbonded pair body on a gfx950 SIMD (round 4).
//
// scripts/valu_issue.hip settled v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_fma_f64.  The pair body of k_nonbonded_atom also
// spends instructions on integer subtracts (fixed-point minimum image), int->float conversions, selects, shifts/ands for the
// LDS address and DPP adds for the wave sum.  If those issue at half the fp32 rate, they weigh twice in the kernel's budget.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_mix scripts/valu_mix.hip && ./valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define K 8
template <int MODE>
__global__ void __launch_bounds__(256) k_mix(float* out, int iters, float seed, int iseed) {
    float a[K]; unsigned u[K];
#pragma unroll
    for (int k = 0; k < K; k++) { a[k] = seed + k + threadIdx.x * 1e-3f; u[k] = iseed + k * 77 + threadIdx.x; }
    const float b = 0.999f, c = 1e-3f; const unsigned ub = 0x01010101u + iseed;
    unsigned long long mask = 0x5555555555555555ull + iseed;
    if (MODE == 18) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(b) : "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                else if (MODE == 1) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[k]) : "v"(ub));
                else if (MODE == 2) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[k]) : "v"(u[k]));
                else if (MODE == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
                else if (MODE == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[k]) : "v"(ub));
                else if (MODE == 5) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(u[k]));
                else if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if (MODE == 7) asm volatile("v_add_f32 %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
                else if (MODE == 8) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
                else if (MODE == 9) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "s"(b));
                else if (MODE == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
                else if (MODE == 11) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(u[k]) : "v"(ub));
                else if (MODE == 12) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if (MODE == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(a[(k + 1) % K]));
                else if (MODE == 14) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(c));
                else if (MODE == 15) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "s"(b));
                else if (MODE == 16) asm volatile("v_fmamk_f32 %0, %0, 0x41400000, %1" : "+v"(a[k]) : "v"(c));
                else if (MODE == 17) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(mask));
                else if (MODE == 18) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));
                else if (MODE == 19) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(a[k]), "v"(b));
                else if (MODE == 20) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[k]) : "v"(b), "v"(c));
                else if (MODE == 21) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(u[k]) : "s"(ub));
                else if (MODE == 22) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[k]) : "v"(u[k]));
                else if (MODE == 23) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "s"(b));
                else if (MODE == 24) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(a[k]));
                else if (MODE == 25) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[k]));
                else if (MODE == 26) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if (MODE == 27) asm volatile("v_mad_u32_u24 %0, %0, 16, %1" : "+v"(u[k]) : "v"(ub));
                else if (MODE == 28) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(ub));
                else if (MODE == 29) asm volatile("v_add_lshl_u32 %0, %0, %1, 4" : "+v"(u[k]) : "v"(ub));
                else if (MODE == 30) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[k]) : "v"(ub));
                else if (MODE == 31) asm volatile("v_bfe_u32 %0, %0, 0, 15" : "+v"(u[k]));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) s += a[k] + (float)u[k];
    if (s == 12345.678f || mask == 77ull) out[0] = s;
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
    printf("device: %s, %d CUs, max clock %.2f GHz; %d independent chains per lane\n", prop.gcnArchName, cus, clock_ghz, K);
    float* out; CHECK(hipMalloc(&out, 64));
    const int iters = 4096;
    const char* names[32] = {"v_fma_f32", "v_sub_u32", "v_cvt_f32_i32", "v_cndmask_b32", "v_and_b32", "v_lshlrev_b32", "v_mul_f32", "v_add_f32 dpp",
                             "v_rsq_f32", "v_mul_f32 sgpr", "v_cmp_lt_f32", "v_lshl_add_u32", "v_sub_f32", "v_mov_b32",
                             "v_fma v,v,s", "v_fma v,s,s", "v_fmamk literal", "v_cndmask e64 sgpr", "v_cndmask vcc set", "v_cmp e64 ->sgpr", "v_fma clamp", "v_sub_u32 sgpr", "v_cvt_f32_u32", "v_add_f32 sgpr", "v_mul literal", "v_mul inline 2.0", "v_max_f32", "v_mad_u32_u24", "v_add_u32", "v_add_lshl_u32", "v_mul_u32_u24", "v_bfe_u32"};
    printf("%-16s %10s %14s %16s\n", "instruction", "waves/SIMD", "T wave-instr/s", "cyc/instr/SIMD");
    for (int mode = (getenv("MIX_FROM") ? atoi(getenv("MIX_FROM")) : 0); mode < 32; mode++) {
        for (int w : {4}) {
            const int blocks = cus * w;
            auto launch = [&]() {
                switch (mode) {
#define L(M) case M: hipLaunchKernelGGL(k_mix<M>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 3); break;
                    L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31)
#undef L
                }
            };
            const double ms = time_ms(launch, 5);
            const double insts = (double)blocks * 4 * iters * 8 * K;
            const double rate = insts / (ms * 1e-3) / 1e12;
            const double cyc = (double)cus * 4 * clock_ghz * 1e9 / (rate * 1e12);
            printf("%-16s %10d %14.3f %16.2f\n", names[mode], w, rate, cyc);
        }
    }
    CHECK(hipFree(out));
    return 0;
}
