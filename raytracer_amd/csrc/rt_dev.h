/*
 * rt_dev.h - development instrumentation of the trace kernels, kept out of rt_kernels.hip: phase clocks
 * and timing ablations.  In a product build (none of the RT_ABL_* / RT_PHASE_CLOCKS macros defined: what
 * csrc/Makefile builds) everything here compiles to nothing.  Ablation builds render WRONG pictures on
 * purpose: they price one part of a kernel (make EXTRA=-DRT_ABL_NO_LIGHTS B=build_b OUT=../librt_hip_b.so,
 * tools/ab.sh); profiles/r02_pmc_notes.md lists what they measured.
 */
#pragma once

#ifdef RT_ABL_NO_ANGLESUM /* price of the polygon inside test (the reference scene's floor as a box test) */
#define RT_DEV_NO_ANGLESUM(px, pz) return (px) > -15.0f && (px) < 7.0f && (pz) > -100.0f && (pz) < 0.0f
#else
#define RT_DEV_NO_ANGLESUM(px, pz) ((void)0)
#endif
#ifdef RT_ABL_NO_BAND /* price of stages 2-3 of the inside test (wrong pixels along polygon edges) */
#define RT_DEV_NO_BAND(stats) do { if (!(stats)) return false; } while (0)
#else
#define RT_DEV_NO_BAND(stats) ((void)0)
#endif
#ifdef RT_ABL_NO_LIGHTS /* price of shadow rays + Phong */
#define RT_DEV_LIGHTS(n) 0u
#else
#define RT_DEV_LIGHTS(n) (n)
#endif
#ifdef RT_ABL_NO_SPAWN /* price of everything behind the local colour */
#define RT_DEV_NO_SPAWN(hit, leaf, r, g, b, sr, st) \
    do { sr = st = false; if (hit) { leaf[0] = clamp1(r); leaf[1] = clamp1(g); leaf[2] = clamp1(b); } return false; } while (0)
#else
#define RT_DEV_NO_SPAWN(hit, leaf, r, g, b, sr, st) ((void)0)
#endif
#ifdef RT_ABL_NO_CHILD /* price of the child searches (children never hit) */
#define RT_DEV_CHILD(want) false
#else
#define RT_DEV_CHILD(want) (want)
#endif
#ifdef RT_ABL_NO_ATOMIC /* price of the reservation's atomic round trip */
#define RT_DEV_ATOMIC(cond) false
#else
#define RT_DEV_ATOMIC(cond) (cond)
#endif

#ifdef RT_ABL_NO_LITTILE /* tiles of one lit polygon take the general sample loop */
#define RT_DEV_LITTILE(cond) false
#else
#define RT_DEV_LITTILE(cond) (cond)
#endif

/* -DRT_PHASE_CLOCKS (tools/phase_probe.py, tools/phase_probe_k0.py): wave cycles per phase of k_bounce,
 * read with s_memtime, summed into the statistic slots the
 * plain build leaves unused (ST_SPH_TESTS + phase).  Compiles to nothing otherwise. */
struct PhaseClock {
#ifdef RT_PHASE_CLOCKS
    unsigned long long last = 0, acc[10] = {};
    /* the clock read waits for outstanding LDS / scalar loads only: vector-memory stalls are charged
     * to the phase whose instructions actually wait for them, as in the plain build */
    __device__ __forceinline__ void start() { last = clock64(); }
    __device__ __forceinline__ void mark(int k)
    {
        const unsigned long long t = clock64();
        acc[k] += t - last;
        last = t;
    }
    __device__ __forceinline__ void flush(uint64_t *st)
    {
        if (__lane_id() == 0)
            for (int k = 0; k < 10; ++k)
                atomicAdd((unsigned long long *)&st[ST_SPH_TESTS + k], acc[k]);
    }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(uint64_t *) {}
#endif
};


/* -DRT_TILE_CLOCKS (tools/tile_probe.py): waves and wave cycles of k_primary per class of tile, in the
 * statistic slots the plain build leaves unused (ST_SPH_TESTS + 2 * class: waves, + 1: cycles). */
#ifdef RT_TILE_TRACE /* (tools/tile_trace.py: start / end of every k_primary wave on the 100 MHz wall clock) */
static __device__ unsigned long long rt_tile_trace[65536 * 4];
#define RT_DEV_TILE_START() const unsigned long long tile_t0_ = wall_clock64()
#define RT_DEV_TILE_END(st, cls) do { if (__lane_id() == 0) { \
    const unsigned w_ = (ty * nbx + bx) * 4u + (threadIdx.x >> 6); \
    if (w_ < 65536u) { rt_tile_trace[w_ * 4u] = tile_t0_; rt_tile_trace[w_ * 4u + 1u] = wall_clock64(); \
        rt_tile_trace[w_ * 4u + 2u] = (cls); \
        rt_tile_trace[w_ * 4u + 3u] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) /* HW_ID */ \
            | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); } } } while (0)
#elif defined(RT_TILE_CLOCKS)
#define RT_DEV_TILE_START() const unsigned long long tile_t0_ = clock64()
#define RT_DEV_TILE_END(st, cls) do { if (__lane_id() == 0) { \
    atomicAdd((unsigned long long *)&(st)[ST_SPH_TESTS + 2 * (cls)], 1ull); \
    atomicAdd((unsigned long long *)&(st)[ST_SPH_TESTS + 2 * (cls) + 1], clock64() - tile_t0_); } } while (0)
#else
#define RT_DEV_TILE_START() ((void)0)
#define RT_DEV_TILE_END(st, cls) ((void)0)
#endif
