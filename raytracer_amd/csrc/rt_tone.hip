/*
 * rt_tone.hip - the tone-reproduction step of the reference (apply_tone / ward_tone / reinhard_tone,
 * raytrace.c:792-871) on a device-resident radiance frame.
 *
 * The reference keeps this step between its pixel loop and the 8-bit pack (the call is commented out
 * at raytrace.c:917): scale the frame to scene luminance, take the mean and the log-mean luminance
 * of ALL pixels, scale every pixel by an operator-dependent factor, go back to the display range.
 * It is the one place of the path with a real exchange step: the two sums are global, so N GPUs
 * that hold slices of a frame all-reduce them (3 doubles) between the two kernels.
 *
 *   k_tone_sums    grid-strided, HBM-bound (12 B read per pixel): per block the sums of L and of
 *                  log(1e-5 + L), L = luminance of the lMax-scaled pixel, in double.
 *   k_tone_finish  one block: the partials in a fixed order (the result does not depend on timing)
 *                  -> sums[0..1]; sums[2] = luminance of the key pixel (Reinhard, raytrace.c:816-819).
 *   k_tone_apply   per pixel (12 B read, 12 + 4 B written): the operator and the 8-bit pack.
 *
 * Difference to the reference, on purpose: its two sums are float running sums in pixel order, which
 * lose ~3e-4 of their value at 2 M pixels; these are double tree sums.  Per-pixel arithmetic is the
 * reference's float arithmetic.  tests/test_gpu_parity.py compares with the CPU restatement of the
 * reference (bit-exact to it on the fixtures) at the north star's 1e-4.
 */
#include "rt_device.h"
#include "rt_hip.h"
#include "rt_internal.h"

#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>

constexpr uint32_t kToneBlocks = 1024; /* partial sums per launch */
constexpr size_t kScratchDoubles = 2 * kToneBlocks + 4; /* 2 * kToneBlocks partials, then 3 sums (+1 pad) */

/*
 * rt_hip_tone_static: the tone step's scratch block (partial sums + the three sums) as a file-scope
 * device array - 16 KiB + 32 B on every GPU the library runs on, held until the process ends - for
 * callers that cannot allocate.  In host code `scratch` is only a handle for hipGetSymbolAddress, never
 * a device address: the address is resolved once per device, with that device current
 * (rt_tone_static_init), and kept in g_static_ptr[device]; kernels, and anything else that needs the
 * memory, get that pointer.
 */
static __device__ __align__(256) double scratch[kScratchDoubles];

namespace {

__device__ __forceinline__ float luminance(float r, float g, float b)
{
    return 0.27f * r + 0.67f * g + 0.06f * b; /* get_luminance color.c:168-173 */
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(RT_BLOCK) void k_tone_sums(const float *radiance, uint64_t n, float lMax,
                                                         double *partial)
{
    double sl = 0.0, sg = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * RT_BLOCK) {
        /* raytrace.c:853-858 */
        const float r = radiance[3 * i + 0] * lMax, g = radiance[3 * i + 1] * lMax, b = radiance[3 * i + 2] * lMax;
        const float l = luminance(r, g, b);
        sl += (double)l;
        sg += (double)logf(.00001f + l);
    }
    __shared__ double red[2][RT_BLOCK / 64];
    sl = wave_sum(sl);
    sg = wave_sum(sg);
    if ((threadIdx.x & 63u) == 0) {
        red[0][threadIdx.x >> 6] = sl;
        red[1][threadIdx.x >> 6] = sg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, c = 0.0;
        for (uint32_t w = 0; w < RT_BLOCK / 64; ++w) {
            a += red[0][w];
            c += red[1][w];
        }
        partial[2 * blockIdx.x + 0] = a;
        partial[2 * blockIdx.x + 1] = c;
    }
}

__global__ __launch_bounds__(RT_BLOCK) void k_tone_finish(const double *partial, uint32_t nparts,
                                                           const float *radiance, float lMax, int64_t key_pixel,
                                                           double *sums)
{
    double a = 0.0, c = 0.0;
    for (uint32_t k = threadIdx.x; k < nparts; k += RT_BLOCK) {
        a += partial[2 * k + 0];
        c += partial[2 * k + 1];
    }
    __shared__ double red[2][RT_BLOCK / 64];
    a = wave_sum(a);
    c = wave_sum(c);
    if ((threadIdx.x & 63u) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double x = 0.0, y = 0.0;
        for (uint32_t w = 0; w < RT_BLOCK / 64; ++w) {
            x += red[0][w];
            y += red[1][w];
        }
        sums[0] = x;
        sums[1] = y;
        double key = 0.0;
        if (key_pixel >= 0) {
            const float r = radiance[3 * key_pixel + 0] * lMax, g = radiance[3 * key_pixel + 1] * lMax,
                        b = radiance[3 * key_pixel + 2] * lMax;
            key = (double)luminance(r, g, b);
        }
        sums[2] = key;
    }
}

__global__ __launch_bounds__(RT_BLOCK) void k_tone_apply(float *radiance, uint32_t *argb, uint64_t n, int op,
                                                          float ldMax, float lMax, const double *sums,
                                                          uint64_t total_pixels, int use_key)
{
    /* the frame-wide factor, from the (all-reduced) sums: a dozen operations, every thread its own */
    const float totalLogLum = (float)sums[1];
    const float logAvgLum = expf((1.0f / (float)total_pixels) * totalLogLum); /* raytrace.c:795,811 */
    float factor;
    if (op == RT_TONE_REINHARD) {
        factor = 0.18f / logAvgLum; /* :814 */
        if (use_key)
            factor = (float)(0.18 / (double)(float)sums[2]); /* :818, a double quotient stored to float */
    } else {
        factor = powf((1.219f + powf(ldMax / 2.0f, 0.4f)) / (1.219f + powf(logAvgLum, 0.4f)), 2.5f); /* :797-798 */
    }
    const float ldMaxInv = 1.0f / ldMax; /* :838 */
    for (uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * RT_BLOCK) {
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = radiance[3 * i + k] * lMax; /* :853 */
            v = v * factor;                       /* :804 / :823 */
            if (op == RT_TONE_REINHARD)
                v = (v * ldMax) / (1.0f + v);     /* :824-829 */
            v = v * ldMaxInv;                     /* :867, clamped (color.c:44-54) */
            c[k] = v > 1.0f ? 1.0f : v;
            radiance[3 * i + k] = c[k];
        }
        if (argb) { /* write_color_32 raytrace.c:783-790 */
            const uint32_t R = (uint32_t)(int)(255.0f * c[0]) & 0xFFu;
            const uint32_t G = (uint32_t)(int)(255.0f * c[1]) & 0xFFu;
            const uint32_t B = (uint32_t)(int)(255.0f * c[2]) & 0xFFu;
            argb[i] = 0xFF000000u | (R << 16) | (G << 8) | B;
        }
    }
}

/* per-device scratch of the convenience entry points: partial sums + the three sums */
struct ToneScratch {
    double *partial = nullptr; /* hipMalloc'd on first use: rt_hip_tone, rt_hip_tone_sums */
};
std::mutex g_tone_mu;
ToneScratch g_tone[64];

std::atomic<double *> g_static_ptr[64]; /* device address of `scratch` per device (rt_tone_static_init writes it under g_tone_mu;
                                           rt_hip_tone_static reads it without the lock) */

int current_device(int *dev)
{
    const hipError_t e = hipGetDevice(dev);
    if (e != hipSuccess) {
        rt_set_err("hipGetDevice failed: %s", hipGetErrorString(e));
        return -(int)e;
    }
    if (*dev < 0 || *dev >= 64) {
        rt_set_err("device %d outside the 64 this library keeps tone scratch for", *dev);
        return RT_E_INVALID;
    }
    return 0;
}

int tone_scratch(double **partial, double **sums)
{
    int dev = 0;
    const int rc = current_device(&dev);
    if (rc)
        return rc;
    std::lock_guard<std::mutex> lk(g_tone_mu);
    if (!g_tone[dev].partial) {
        const hipError_t e = hipMalloc((void **)&g_tone[dev].partial, kScratchDoubles * sizeof(double));
        if (e != hipSuccess) {
            rt_set_err("allocating the tone step's partial sums failed: %s", hipGetErrorString(e));
            return -(int)e;
        }
    }
    *partial = g_tone[dev].partial;
    *sums = g_tone[dev].partial + 2 * kToneBlocks;
    return 0;
}

uint32_t tone_grid(uint64_t n)
{
    const uint64_t want = (n + RT_BLOCK - 1) / RT_BLOCK;
    return (uint32_t)(want < 1 ? 1 : want > kToneBlocks ? kToneBlocks : want);
}

int launched(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        return 0;
    rt_set_err("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return -(int)e;
}

/* both halves on `st`, partial sums and the three sums in the caller's scratch block */
int tone_with(double *partial, double *sums, float *d_radiance, uint32_t *d_argb, uint64_t npixels, int op, float ldMax,
              float lMax, int64_t key_pixel, hipStream_t st)
{
    if (op != RT_TONE_REINHARD)
        key_pixel = -1; /* Ward's operator has no key pixel (raytrace.c:792-806) */
    const uint32_t grid = tone_grid(npixels);
    k_tone_sums<<<grid, RT_BLOCK, 0, st>>>(d_radiance, npixels, lMax, partial);
    k_tone_finish<<<1, RT_BLOCK, 0, st>>>(partial, grid, d_radiance, lMax, key_pixel, sums);
    k_tone_apply<<<grid * 4u, RT_BLOCK, 0, st>>>(d_radiance, d_argb, npixels, op, ldMax, lMax, sums, npixels,
                                                  key_pixel >= 0);
    return launched("rt_hip_tone");
}

int check_tone_args(const float *d_radiance, uint64_t npixels, int op, int64_t key_pixel)
{
    if (!d_radiance || (op != RT_TONE_WARD && op != RT_TONE_REINHARD)) {
        rt_set_err("tone step: null frame or operator %d (RT_TONE_WARD, RT_TONE_REINHARD)", op);
        return RT_E_INVALID;
    }
    if (op == RT_TONE_REINHARD && key_pixel >= (int64_t)npixels) { /* (Ward's operator has no key pixel: whatever the caller left there is ignored) */
        rt_set_err("tone step: key pixel %lld outside the %llu pixels", (long long)key_pixel, (unsigned long long)npixels);
        return RT_E_INVALID;
    }
    return 0;
}

} // namespace

int rt_tone_static_init(void)
{
    int dev = 0;
    const int rc = current_device(&dev);
    if (rc)
        return rc;
    std::lock_guard<std::mutex> lk(g_tone_mu);
    if (!g_static_ptr[dev].load()) {
        double *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            rt_set_err("hipGetSymbolAddress(tone scratch) on device %d failed: %s", dev, hipGetErrorString(e));
            return e != hipSuccess ? -(int)e : RT_E_INVALID;
        }
        g_static_ptr[dev].store(ptr);
    }
    return 0;
}

extern "C" {

int rt_hip_tone_sums(const float *d_radiance, uint64_t npixels, float lMax, int64_t key_pixel, double *d_sums,
                     void *stream)
{
    if (!d_sums) {
        rt_set_err("rt_hip_tone_sums: null sums");
        return RT_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    if (npixels == 0) { /* a rank that holds no pixel of the frame contributes zeros (and still joins the all-reduce) */
        const hipError_t e = hipMemsetAsync(d_sums, 0, 3 * sizeof(double), st);
        if (e != hipSuccess) {
            rt_set_err("rt_hip_tone_sums: hipMemsetAsync failed: %s", hipGetErrorString(e));
            return -(int)e;
        }
        return 0;
    }
    int rc = check_tone_args(d_radiance, npixels, RT_TONE_REINHARD, key_pixel); /* (the sums carry the key pixel's luminance) */
    if (rc)
        return rc;
    double *partial, *own;
    rc = tone_scratch(&partial, &own);
    if (rc)
        return rc;
    const uint32_t grid = tone_grid(npixels);
    k_tone_sums<<<grid, RT_BLOCK, 0, st>>>(d_radiance, npixels, lMax, partial);
    k_tone_finish<<<1, RT_BLOCK, 0, st>>>(partial, grid, d_radiance, lMax, key_pixel, d_sums);
    return launched("rt_hip_tone_sums");
}

int rt_hip_tone_apply(float *d_radiance, uint32_t *d_argb, uint64_t npixels, int op, float ldMax, float lMax,
                      const double *d_sums, uint64_t total_pixels, int use_key, void *stream)
{
    if (npixels == 0)
        return 0; /* nothing of the frame on this rank */
    if (!d_radiance || !d_sums || total_pixels == 0 || (op != RT_TONE_WARD && op != RT_TONE_REINHARD)) {
        rt_set_err("rt_hip_tone_apply: null argument, empty frame or operator %d", op);
        return RT_E_INVALID;
    }
    k_tone_apply<<<tone_grid(npixels) * 4u, RT_BLOCK, 0, (hipStream_t)stream>>>(d_radiance, d_argb, npixels, op, ldMax,
                                                                                  lMax, d_sums, total_pixels, use_key);
    return launched("rt_hip_tone_apply");
}

int rt_hip_tone(float *d_radiance, uint32_t *d_argb, uint64_t npixels, int op, float ldMax, float lMax,
                int64_t key_pixel, void *stream)
{
    if (npixels == 0)
        return 0;
    int rc = check_tone_args(d_radiance, npixels, op, key_pixel);
    if (rc)
        return rc;
    double *partial, *sums;
    rc = tone_scratch(&partial, &sums);
    if (rc)
        return rc;
    return tone_with(partial, sums, d_radiance, d_argb, npixels, op, ldMax, lMax, key_pixel, (hipStream_t)stream);
}

int rt_hip_tone_static(float *d_radiance, uint32_t *d_argb, uint64_t npixels, int op, float ldMax, float lMax,
                       int64_t key_pixel, void *stream)
{
    if (npixels > RT_TONE_STATIC_MAX_PIXELS) {
        rt_set_err("rt_hip_tone_static: %llu pixels exceed the %llu this entry point takes (RT_TONE_STATIC_MAX_PIXELS)",
                   (unsigned long long)npixels, (unsigned long long)RT_TONE_STATIC_MAX_PIXELS);
        return RT_E_INVALID;
    }
    if (npixels == 0)
        return 0;
    int rc = check_tone_args(d_radiance, npixels, op, key_pixel);
    if (rc)
        return rc;
    int dev = 0;
    if ((rc = current_device(&dev)))
        return rc;
    if (!g_static_ptr[dev].load() && (rc = rt_tone_static_init())) /* first call on this device (no context made yet) */
        return rc;
    double *const block = g_static_ptr[dev].load();
    /* the kernels write every partial and every sum they later read: nothing of the last call's
     * contents needs clearing */
    return tone_with(block, block + 2 * kToneBlocks, d_radiance, d_argb, npixels, op, ldMax, lMax, key_pixel,
                     (hipStream_t)stream);
}

} /* extern "C" */
