// grid_barrier.hip -- what a grid-wide barrier costs on an MI355X (round 5): the building block of a persistent one-launch step for a
// lone chain.  W workgroups of 256 threads run N rounds of { a little work; barrier }: a monotone arrival counter in
// global memory (agent-scope atomics, release / acquire fences as the guide's inter-workgroup visibility rules ask), every workgroup
// spins on it.  Workgroup ids are dealt to the 8 XCDs round-robin, so W <= 8 puts every workgroup on its own XCD (the worst case:
// the counter lives in one L2 / memory channel and the others reach it through the fabric).
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier scripts/grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_rounds(unsigned* counter, double* data, int rounds, int W, int payload) {
    const int tid = threadIdx.x, b = blockIdx.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; r++) {
        // "work": every workgroup writes `payload` doubles the others read after the barrier (what a force phase hands to the step phase)
        for (int q = tid; q < payload; q += 256) data[(size_t)b * payload + q] = (double)(r + q);
        __threadfence();   // release: this workgroup's writes are visible device-wide before it arrives
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * (unsigned)W;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __threadfence();   // acquire side for the other threads of the workgroup
        for (int q = tid; q < payload; q += 256) acc += __builtin_nontemporal_load(&data[(size_t)((b + 1) % W) * payload + q]);
    }
    if (acc == -1.0) data[0] = acc;
}
__global__ void k_empty() {}

int main() {
    unsigned* counter; double* data;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&data, sizeof(double) * 64 * 4096));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    {   // a launch that does nothing, back to back: what the four-launch step pays four times
        for (int w = 0; w < 100; w++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        for (int w = 0; w < 2000; w++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty launches back to back: %.2f us each\n", 1000.0 * ms / 2000);
    }
    printf("%4s %8s %14s\n", "W", "payload", "us per round");
    for (int payload : {0, 1024}) for (int W : {1, 2, 4, 8, 10, 16, 32, 64}) {
        const int rounds = 2000;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipMemset(counter, 0, 4));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_rounds, dim3(W), dim3(256), 0, 0, counter, data, rounds, W, payload);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) printf("%4d %8d %14.2f\n", W, payload, 1000.0 * ms / rounds);
        }
    }
    return 0;
}
