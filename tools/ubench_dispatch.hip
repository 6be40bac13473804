// development microbenchmark: how fast does the GPU dispatch the blocks of a launch whose blocks do (almost)
// nothing - per block or per wave?  Same number of waves as k_primary at 1080p (32 640) in blocks of 64 .. 1024
// threads, with and without the LDS request of the trace kernels (21 KB per 256 threads).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_dispatch.hip -o tools/ubench_dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>

extern __shared__ float lds[];
__global__ void k_nothing(float *out, int touch_lds)
{
    if (touch_lds && threadIdx.x == 0)
        lds[0] = 1.0f;
    if (out && blockIdx.x == 0x7FFFFFFF)
        out[threadIdx.x] = lds[0];
}

int main()
{
    float *d;
    hipMalloc(&d, 4096);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int waves = 32640;
    for (int lds_per_256 = 0; lds_per_256 <= 21504; lds_per_256 += 21504)
        for (int threads = 64; threads <= 1024; threads *= 2) {
            const int blocks = waves * 64 / threads;
            const size_t lds_bytes = (size_t)lds_per_256 * threads / 256;
            if (lds_bytes > 65536)
                continue;
            for (int i = 0; i < 5; ++i)
                hipLaunchKernelGGL(k_nothing, dim3(blocks), dim3(threads), lds_bytes, 0, d, lds_per_256 != 0);
            hipDeviceSynchronize();
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(a, 0);
                hipLaunchKernelGGL(k_nothing, dim3(blocks), dim3(threads), lds_bytes, 0, d, lds_per_256 != 0);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            printf("lds/256thr %5d  block %4d  blocks %6d: %7.2f us  = %6.2f ns per block, %5.2f ns per wave\n", lds_per_256,
                   threads, blocks, best * 1e3f, best * 1e6f / blocks, best * 1e6f / waves);
        }
    return 0;
}
