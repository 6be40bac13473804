// development microbenchmark: cost of a device-wide barrier inside one kernel (release fence, one
// device-scope atomic per block, spin on it, acquire fence) against the ~4.7 us that a dependent
// near-empty kernel launch costs in the frame timeline.  Bounded spin: gives up after ~2 s.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_barrier.hip -o tools/ubench_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ bool grid_barrier(unsigned *bar, unsigned target, unsigned *gave_up)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(bar, 1u);
        unsigned tries = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++tries > 4000000u) {
                atomicExch(gave_up, 1u);
                ok = false;
                break;
            }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_barriers(unsigned *bar, unsigned *gave_up, int rounds, float *data, int work)
{
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int w = 0; w < work; ++w) // a little global traffic between barriers, like a sparse level
            acc += data[(blockIdx.x * 256 + threadIdx.x + w * 65536) & 0xFFFFF];
        if (work)
            data[(blockIdx.x * 256 + threadIdx.x + r * 4096) & 0xFFFFF] = acc;
        if (!grid_barrier(bar, (unsigned)(r + 1) * gridDim.x, gave_up))
            return;
    }
    if (acc == 12345.f)
        data[0] = acc;
}

__global__ void k_empty(float *data) { if (data == nullptr) data[0] = 0; }

int main()
{
    unsigned *bar, *gave;
    float *data;
    hipMalloc(&bar, 8);
    hipMalloc(&data, (1 << 20) * sizeof(float));
    hipMemset(data, 0, (1 << 20) * sizeof(float));
    gave = bar + 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {64, 256, 512, 1024})
        for (int work : {0, 4}) {
            const int rounds = 200;
            hipMemset(bar, 0, 8);
            k_barriers<<<blocks, 256>>>(bar, gave, 2, data, work);
            hipDeviceSynchronize();
            hipMemset(bar, 0, 8);
            hipEventRecord(e0);
            k_barriers<<<blocks, 256>>>(bar, gave, rounds, data, work);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned h[2];
            hipMemcpy(h, bar, 8, hipMemcpyDeviceToHost);
            printf("%4d blocks, work %d: %7.3f us per barrier round%s\n", blocks, work, ms * 1e3 / rounds,
                   h[1] ? "  (GAVE UP)" : "");
        }
    // dependent empty launches for comparison
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i)
        k_empty<<<512, 256>>>(data);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent empty launches: %7.3f us each\n", ms * 1e3 / 200);
    return 0;
}
