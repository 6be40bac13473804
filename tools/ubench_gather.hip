// development: what a CU's vector memory pipe charges for the walk's chunk fetches.  Every lane of a wave reads NL x 16 bytes of its own
// 64-byte chunk (random chunks of a table that stays in the caches), 20 waves a CU like k_trace.  Prints ns per wave-level load instruction per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int NL>
__global__ __launch_bounds__(256) void k(const uint4 *tab, uint32_t nchunks, uint32_t iters, uint32_t *out)
{
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    uint4 cur[NL];
    uint32_t c = x % nchunks;
#pragma unroll
    for (int u = 0; u < NL; ++u) cur[u] = tab[4u * c + u];
    for (uint32_t i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t nc = (x >> 8) % nchunks;
        uint4 nx[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) nx[u] = tab[4u * nc + u];
        // some arithmetic on the chunk in hand (the tests), dependent on its data
#pragma unroll
        for (int u = 0; u < NL; ++u) acc += cur[u].x * 3u + cur[u].y + (cur[u].z ^ cur[u].w);
        x += acc & 1u;
#pragma unroll
        for (int u = 0; u < NL; ++u) cur[u] = nx[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int NL> static void run(const uint4 *tab, uint32_t nchunks, uint32_t *out, int blocks)
{
    const uint32_t iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NL><<<blocks, 256>>>(tab, nchunks, 10, out);
    hipEventRecord(a);
    k<NL><<<blocks, 256>>>(tab, nchunks, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves_per_cu = blocks * 4.0 / 256.0;
    const double instr_per_cu = waves_per_cu * iters * NL;
    printf("chunks %7u  loads/iter %d  %.3f ms  %.1f ns per wave-load per CU (%.0f cycles at 2.4 GHz), %.1f ns per wave-iteration per CU\n", nchunks, NL, ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4, ms * 1e6 / (waves_per_cu * iters));
}
int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 1280; // 5 blocks of 4 waves a CU
    for (uint32_t nchunks : {64u, 128u, 256u, 512u, 1024u, 2496u, 55771u, 1u << 20}) {
        uint4 *tab; uint32_t *out;
        hipMalloc(&tab, (size_t)nchunks * 64); hipMalloc(&out, 64);
        hipMemset(tab, 1, (size_t)nchunks * 64);
        run<1>(tab, nchunks, out, blocks); run<2>(tab, nchunks, out, blocks); run<3>(tab, nchunks, out, blocks); run<4>(tab, nchunks, out, blocks);
        hipFree(tab); hipFree(out);
    }
    return 0;
}
