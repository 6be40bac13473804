/*
 * rt_kernels.hip - hand-written HIP kernels (gfx950 / CDNA4, wave64) for the Whitted
 * trace-and-shade loop of davidoguns/raytracer (raytrace.c:160-564, 688-790).
 *
 * Pipeline per image batch:
 *   k_primary  : one lane per pixel (a wave = an 8x8 tile), loops the sqrtSpp x sqrtSpp sample grid.  Tiles
 *                are classified first from the cones their samples span (sky / no spheres / one polygon /
 *                every light visible): sky tiles are finished without tracing.  Primary rays are generated
 *                and traced in registers (never touch HBM), depth-0 shade nodes are lit (shadow rays any-hit
 *                inline), the child rays are traced by the lane that spawns them; childless samples
 *                accumulate in registers, samples whose children hit something allocate a tree node and
 *                queue the HITS for level 1 with wave ballot / prefix-sum compaction.
 *   k_bounce(d): one lane per queued hit of level d: shade, spawn, trace the children, queue their hits.
 *   k_subtree(d): the sparse deep end of the tree in one launch: a lane finishes the whole subtree of a
 *                hit queued for level d, depth first, and writes its colour into the parent's child slot.
 *   k_fold(d)  : deepest level first, node colour = clamp(local + kr*refl + kt*trans) written
 *                into the parent's child slot (the reference clamps at every recursion level,
 *                raytrace.c:562, so the tree cannot be flattened into a throughput product);
 *                k_fold_deep folds several sparse levels in one launch.
 *   k_resolve  : per pixel, fold the depth-0 roots, sum in sample order, scale by 1/spp^2, pack ARGB8.
 *   k_publish  : the batch's counters to the host's mapped mirror, and zeroed.
 *
 * Small scenes keep spheres, polygon data and the shading tables in LDS (per-lane indexed reads) and read
 * what every lane reads from the same address (a search's spheres and planes) through the scalar cache;
 * big scenes walk a uniform grid in HBM.
 *
 * Numerics (the exact build; the tolerance build: rt_arith.h): compiled with
 * -ffp-contract=off; every float op is individually rounded in the
 * order the reference's C expressions evaluate, sqrt/div are IEEE-correct (hipcc default), and
 * the reference's double-precision islands (sphere roots ray.c:217-218, refraction
 * raytrace.c:517-518, angle sum ray.c:98,169) are evaluated in double.  acos is taken in float
 * first and redone in double only next to the acceptance threshold; powf with an integral
 * exponent is a double square-and-multiply chain (see polygon_inside / phong_pow).
 */
#include "rt_device.h"
#include "rt_arith.h"
#include "rt_dev.h"

#ifndef RT_JIT /* (RT_JIT: this file compiled at run time by hiprtc for one scene shape, rt_jit.cpp - device code only) */
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <hip/hip_ext.h>
#include "rt_internal.h"
#endif

#ifndef RT_SPHERES_PER_TRIP
#define RT_SPHERES_PER_TRIP 8u
#endif
#ifndef RT_WPE
#ifndef RT_WPE_SHAPE1
#define RT_WPE_SHAPE1 7 /* 72 VGPRs and six of them spilled, still 3 % faster than 6 waves without spills */
#endif
#ifndef RT_WPE_BOUNCE1
#define RT_WPE_BOUNCE1 1
#endif
#define RT_WPE 1 /* min waves per SIMD asked of the register allocator (launch_bounds 2nd arg) */
#endif

/* This file is compiled twice (csrc/Makefile): as it stands - every operation rounded like the reference's
 * x86 build - and with -DRT_FAST -ffp-contract=fast, the tolerance build (rt_arith.h).  The two sets
 * of kernels live in namespaces of their own so that profiles tell them apart (rt_exact::k_primary,
 * rt_fast::k_primary); the fast set has no instrumented and no phong-blinn variants. */
#ifdef RT_FAST
#define RT_NSNAME rt_fast
constexpr bool kFast = true;
#else
#define RT_NSNAME rt_exact
constexpr bool kFast = false;
#endif

namespace RT_NSNAME {

struct Ray {
    float ox, oy, oz;
    float dx, dy, dz;
    float mag;
};

/* per-lane work tally of the instrumented (STATS) build; empty otherwise */
template <bool STATS> struct Tally;
template <> struct Tally<false> {
    __device__ __forceinline__ void sph_test() {}
    __device__ __forceinline__ void sph_hit() {}
    __device__ __forceinline__ void poly_test() {}
    __device__ __forceinline__ void poly_par() {}
    __device__ __forceinline__ void poly_behind() {}
    __device__ __forceinline__ void poly_full() {}
    __device__ __forceinline__ void poly_hit() {}
    __device__ __forceinline__ void lit() {}
    __device__ __forceinline__ void node(bool) {}
    __device__ __forceinline__ void screen_mismatch() {}
    __device__ __forceinline__ void screen_unsure() {}
};
template <> struct Tally<true> {
    uint32_t n_sph_test = 0, n_sph_hit = 0, n_poly_test = 0, n_poly_par = 0, n_poly_behind = 0,
             n_poly_full = 0, n_poly_hit = 0, n_lit = 0, n_node_sph = 0, n_node_poly = 0, n_mismatch = 0,
             n_unsure = 0;
    __device__ __forceinline__ void sph_test() { ++n_sph_test; }
    __device__ __forceinline__ void sph_hit() { ++n_sph_hit; }
    __device__ __forceinline__ void poly_test() { ++n_poly_test; }
    __device__ __forceinline__ void poly_par() { ++n_poly_par; }
    __device__ __forceinline__ void poly_behind() { ++n_poly_behind; }
    __device__ __forceinline__ void poly_full() { ++n_poly_full; }
    __device__ __forceinline__ void poly_hit() { ++n_poly_hit; }
    __device__ __forceinline__ void lit() { ++n_lit; }
    __device__ __forceinline__ void node(bool sphere) { if (sphere) ++n_node_sph; else ++n_node_poly; }
    __device__ __forceinline__ void screen_mismatch() { ++n_mismatch; }
    __device__ __forceinline__ void screen_unsure() { ++n_unsure; }
};

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

/* sum a per-lane counter over the wave and add it to a 64-bit global statistic */
__device__ __forceinline__ void wave_add_stat(uint64_t *stats, int slot, uint32_t v)
{
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if (lane_id() == 0 && s != 0)
        atomicAdd((unsigned long long *)&stats[slot], s);
}

/* vec4_magnitude vector4.c:219-233: (float)sqrt((double)float_sum) == correctly rounded sqrtf */
/* a float4 every lane reads from the same read-only address: through the scalar cache into scalar registers */
__device__ __forceinline__ float4 const_load4(const float4 *p)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = *((const __attribute__((address_space(4))) f4v *)(uintptr_t)p);
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t const_load1(const uint32_t *p)
{
    return *((const __attribute__((address_space(4))) uint32_t *)(uintptr_t)p);
}

/* where a ray meets what it hit, d along it: origin + direction * d, a product and a sum each rounded (ray.c:121-122,
 * 236-237) in BOTH builds - the hit queues carry (ray, distance), and every kernel that takes a hit from them, or
 * goes down into a child itself, must form the same point */
__device__ __forceinline__ void hit_point(const Ray &r, float d, float &px, float &py, float &pz)
{
    px = __fadd_rn(r.ox, __fmul_rn(r.dx, d));
    py = __fadd_rn(r.oy, __fmul_rn(r.dy, d));
    pz = __fadd_rn(r.oz, __fmul_rn(r.dz, d));
}

__device__ __forceinline__ float mag3(float x, float y, float z)
{
    return sqrt_rn((x * x + y * y) + z * z);
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz; /* vec4_dot vector4.c:141-143, left to right */
}

/* vec4_normalize vector4.c:240-262: every component DIVIDED by the length (not multiplied by a
 * reciprocal); the three quotients share one refined reciprocal (rt_arith.h) */
__device__ __forceinline__ void normalize3(float &x, float &y, float &z)
{
#ifdef RT_FAST
    const float rm = rsq_approx((x * x + y * y) + z * z);
    x = x * rm; y = y * rm; z = z * rm;
#else
    const float m = mag3(x, y, z);
    const float rm = rcp_refined(m);
    x = div_with(x, m, rm);
    y = div_with(y, m, rm);
    z = div_with(z, m, rm);
#endif
}

/* ... for a direction of ANY length: a transmission ray's is the incoming direction over the material's index
 * (raytrace.c:455-457), and an index of 1e-30 makes its squares overflow - the reference then divides by an infinite
 * length and pushes the origin by zero.  div_with alone gives NaN there (and for a length of zero): the hardware's
 * fix-up restores the IEEE quotients, three instructions that only these rays pay. */
__device__ __forceinline__ void normalize3_any(float &x, float &y, float &z)
{
#ifdef RT_FAST
    normalize3(x, y, z);
#else
    const float m = mag3(x, y, z);
    const float rm = rcp_refined(m);
    x = __builtin_amdgcn_div_fixupf(div_with(x, m, rm), m, x);
    y = __builtin_amdgcn_div_fixupf(div_with(y, m, rm), m, y);
    z = __builtin_amdgcn_div_fixupf(div_with(z, m, rm), m, z);
#endif
}

/* ray_create's length and its reciprocal (ray.c:34-37): mag = |d|, inv = 1.0f / mag */
__device__ __forceinline__ void length_and_reciprocal(float dx, float dy, float dz, float &mag, float &inv)
{
#ifdef RT_FAST
    const float l2 = (dx * dx + dy * dy) + dz * dz;
    inv = rsq_approx(l2);
    mag = l2 * inv;
#else
    mag = mag3(dx, dy, dz);
    inv = div_with(1.0f, mag, rcp_refined(mag));
#endif
}

/* 1.0f / m as ray_create computes it (ray.c:36-37) */
__device__ __forceinline__ float recip_rn(float m) { return div_with(1.0f, m, rcp_refined(m)); }

/* ray_tinypush ray.c:72-86: origin += normalize(direction) * 0.001f (normalize divides) */
/* UNIT: the caller knows the direction to be of unit length (to rounding); the tolerance build then
 * skips the normalisation, the exact build never does (the reference divides whatever it gets) */
template <bool UNIT = false>
__device__ __forceinline__ void tinypush(Ray &r)
{
    float nx = r.dx, ny = r.dy, nz = r.dz;
    if constexpr (UNIT) {
        if constexpr (!kFast)
            normalize3(nx, ny, nz);
    } else {
        normalize3_any(nx, ny, nz);
    }
    r.ox = r.ox + nx * 0.001f;
    r.oy = r.oy + ny * 0.001f;
    r.oz = r.oz + nz * 0.001f;
}

/* Distance returned by ray_intersect_sphere (ray.c:209-240) once det >= 0 is known.
 * wOne = (float)((-B - sqrt(det)) / 2) with the expression in double.  The reference's second
 * root never wins: wTwo >= wOne always, so `wTwo > 0 && wTwo < w` is false both when
 * w = wOne > 0 and when w = 0.  Hence w = wOne if det == 0, else max-like (wOne > 0 ? wOne : 0).
 * B >= 0 gives wOne <= 0 without needing the square root (w = 0, or -B/2 when det == 0). */
__device__ __forceinline__ float sphere_distance(float B, float det)
{
#ifdef RT_FAST
    /* the same selection in float: wOne = (-B - sqrt(det)) / 2, clamped at 0 unless det == 0 */
    const float wOne = (-B - sqrt_rn(det)) * 0.5f;
    return (det == 0.0f || wOne > 0.0f) ? wOne : 0.0f;
#else
    if (det == 0.0f)
        return (float)((double)(-B) * 0.5);
    if (!(B < 0.0f))
        return 0.0f;
    const float wOne = (float)(((double)(-B) - sqrt_rn_f64_pos((double)det)) * 0.5); /* det > 0 here */
    return wOne > 0.0f ? wOne : 0.0f;
#endif
}

/*
 * Inside test of ray_intersect_polygon (ray.c:128-176): accept iff
 *     sum_i acos(costheta(v_i - pt, v_{i+1} - pt)) + .001 > 2*pi      (sum in double).
 * Three stages, each deciding only what it can decide exactly like the reference:
 *   1. edge_screen      signed distances from the polygon's edges (planar convex polygons): deep
 *                       inside -> accept, far outside -> reject; ~36 instructions for a quad
 *   2. float angle sum  the reference's own float cosines (vec4_costheta: IEEE sqrt and divide) with
 *                       a polynomial acos and a float sum, which is within nv * 1e-6 of the
 *                       reference's double sum: decides unless within 1e-4 (nv <= 25) of the threshold
 *   3. rotation         the sign of the double sum minus the threshold, without acos (below)
 * The instrumented build runs stages 2-3 for every test and counts stage-1 verdicts that contradict
 * them (ST_POLY_SCREEN_MISMATCH must stay 0: asserted over 20 M tests of config 2 in the GPU suite).
 */
/* acos(x) for |x| <= 1 by Abramowitz & Stegun 4.4.46 (|error| <= 2e-8 before float rounding, ~1e-6
 * after it): acos(x) = sqrt(1-x) * (a0 + a1 x + ... + a7 x^7) for x >= 0, pi - acos(-x) otherwise.
 * The square root is the 1-ulp hardware approximation; |x| > 1 gives NaN, like acos. */
__device__ __forceinline__ float acos_screen(float x)
{
    const float ax = fabsf(x);
    float p = -0.0012624911f;
    p = __builtin_fmaf(p, ax, 0.0066700901f);
    p = __builtin_fmaf(p, ax, -0.0170881256f);
    p = __builtin_fmaf(p, ax, 0.0308918810f);
    p = __builtin_fmaf(p, ax, -0.0501743046f);
    p = __builtin_fmaf(p, ax, 0.0889789874f);
    p = __builtin_fmaf(p, ax, -0.2145988016f);
    p = __builtin_fmaf(p, ax, 1.5707963050f);
    const float r = __builtin_amdgcn_sqrtf(1.0f - ax) * p;
    return x < 0.0f ? 3.14159265358979f - r : r;
}

/* sum of the angles pt subtends with consecutive vertices.  (v_i - pt) and its length are shared by the
 * two edges that use them; the values are identical to the reference's recomputation. */
template <typename T>
__device__ __forceinline__ T polygon_angle_sum(const float4 *vtx, uint32_t nv, float px, float py,
                                               float pz)
{
    const float4 v0 = vtx[0];
    const float a0x = v0.x - px, a0y = v0.y - py, a0z = v0.z - pz;
    const float m0 = mag3(a0x, a0y, a0z);
    float ax = a0x, ay = a0y, az = a0z, ma = m0;
    T total = 0;
    for (uint32_t i = 1; i < nv; ++i) {
        const float4 v = vtx[i];
        const float bx = v.x - px, by = v.y - py, bz = v.z - pz;
        const float mb = mag3(bx, by, bz);
        const float c = div_rn(dot3(ax, ay, az, bx, by, bz), ma * mb);
        if constexpr (sizeof(T) == 8)
            total += acos((double)c);
        else
            total += acos_screen(c);
        ax = bx; ay = by; az = bz; ma = mb;
    }
    const float c = div_rn(dot3(ax, ay, az, a0x, a0y, a0z), ma * m0);
    if constexpr (sizeof(T) == 8)
        total += acos((double)c);
    else
        total += acos_screen(c);
    return total;
}

/*
 * Exact-path tie-break.  When the float angle sum lands within 1e-4 of T = 2*pi - .001, the
 * reference's verdict hinges on its double-precision sum S = sum_i acos(c_i).  Instead of four
 * double acos calls (95 instructions and ~30 registers each) the sign of S - T is read off the
 * rotation e^{i(S-T)} = e^{-iT} * prod_i (c_i + i*sqrt(1 - c_i^2)): |S - T| < 2e-4 here, so
 * sign(S - T) = sign(Im), all in double multiply-adds and square roots (relative error ~1e-15,
 * the same as the reference's own rounding noise; the two can differ only if |S - T| < ~1e-14,
 * which needs a sample placed to 1e-14 on the acceptance contour).  A cosine outside [-1,1]
 * gives NaN and "outside", as acos does in the reference.
 */
struct Rot {
    double re, im;
};
__device__ __forceinline__ Rot rot_start()
{
    /* e^{-iT}, T = 2*pi - .001:  cos(T) = cos(.001), sin(T) = -sin(.001) */
    Rot r;
    r.re = 0.99999950000004166667;
    r.im = 0.00099999983333334167;
    return r;
}
__device__ __forceinline__ void rot_by_cos(Rot &z, float c)
{
    const double cd = (double)c;
    const double sd = sqrt(1.0 - cd * cd);
    const double re = z.re * cd - z.im * sd;
    const double im = z.re * sd + z.im * cd;
    z.re = re;
    z.im = im;
}

__device__ __forceinline__ bool polygon_inside_exact(const float4 *vtx, uint32_t nv, float px, float py,
                                                     float pz)
{
    const float twopi = 6.2831853071795864769f;
    const float sf = polygon_angle_sum<float>(vtx, nv, px, py, pz);
    const float margin = (sf + 0.001f) - twopi;
    if (kFast || fabsf(margin) > fmaxf(1e-4f, 4e-6f * (float)nv)) /* NaN (a cosine outside [-1,1]) falls through */
        return margin > 0.0f; /* (the tolerance build takes the float sum's word for it) */
    /* The same float cosines again, folded into the rotation.  Both edge vectors are rebuilt per
     * edge (same expressions, same values) so that nothing but the rotation lives across the
     * double-precision square root: this path runs for a few samples per frame, but its register
     * peak would otherwise set the occupancy of the whole kernel. */
    Rot z = rot_start();
#pragma unroll 1
    for (uint32_t i = 0; i < nv; ++i) {
        const float4 va = vtx[i];
        const float4 vb = vtx[(i + 1u == nv) ? 0u : i + 1u];
        const float ax = va.x - px, ay = va.y - py, az = va.z - pz;
        const float bx = vb.x - px, by = vb.y - py, bz = vb.z - pz;
        rot_by_cos(z, div_rn(dot3(ax, ay, az, bx, by, bz), mag3(ax, ay, az) * mag3(bx, by, bz)));
    }
    return z.im > 0.0;
}

/* ---- quads: the four vertices are fetched up front and the four edges are evaluated side by
 * side, so that the sqrt / divide / acos chains of the edges overlap instead of running as one
 * dependent chain behind an LDS read per vertex.  Same operations and, for the exact sum, the
 * same order of additions as the generic loops above (edges 01, 12, 23, 30). ---- */
struct Quad {
    float ax[4], ay[4], az[4];
};

__device__ __forceinline__ Quad quad_edges(const float4 *vtx, float px, float py, float pz)
{
    const float4 v0 = vtx[0], v1 = vtx[1], v2 = vtx[2], v3 = vtx[3];
    Quad q;
    q.ax[0] = v0.x - px; q.ay[0] = v0.y - py; q.az[0] = v0.z - pz;
    q.ax[1] = v1.x - px; q.ay[1] = v1.y - py; q.az[1] = v1.z - pz;
    q.ax[2] = v2.x - px; q.ay[2] = v2.y - py; q.az[2] = v2.z - pz;
    q.ax[3] = v3.x - px; q.ay[3] = v3.y - py; q.az[3] = v3.z - pz;
    return q;
}

__device__ __forceinline__ bool quad_inside_exact(const Quad &q)
{
    float m[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        m[i] = mag3(q.ax[i], q.ay[i], q.az[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) & 3;
        c[i] = div_rn(dot3(q.ax[i], q.ay[i], q.az[i], q.ax[j], q.ay[j], q.az[j]), m[i] * m[j]);
    }
    const float twopi = 6.2831853071795864769f;
    const float sf = ((acos_screen(c[0]) + acos_screen(c[1])) + acos_screen(c[2])) + acos_screen(c[3]);
    const float margin = (sf + 0.001f) - twopi;
    if (kFast || fabsf(margin) > 1e-4f)
        return margin > 0.0f;
    Rot z = rot_start();
#pragma unroll 1
    for (int i = 0; i < 4; ++i)
        rot_by_cos(z, c[i]);
    return z.im > 0.0;
}

/*
 * First screening stage, for planar convex polygons (the host decides: rt_api.hip, edge_screen_data).
 * With m_i the unit in-plane normal of edge i pointing inwards, g_i = m_i.p + k_i is the signed
 * distance of the hit point from the edge's line; edg[i] holds the equation shifted and scaled,
 * G_i = (g_i - hin_i) / (hin_i + hout), so that the two verdicts are min G >= 0 and min G <= -1:
 *   every g_i >= hin_i                     => deep inside: the exact angle sum is 2*pi and the
 *       reference's rounding (cosines within 8e-7, every sin(theta_i) = L_i g_i / (a_i b_i) >= nv*1e-3)
 *       moves its sum by less than 8e-4: accepted there and here.
 *   some  g_i <= -hout                      => at least hout away from the polygon: the exact sum is
 *       below 2*pi - 4*asin(hout/(hout+D)), further below the acceptance threshold than the
 *       reference's worst-case rounding (nv * 1.3e-3): rejected there and here.
 * 18 instructions and four LDS reads for a quad against ~170 for the float angle sum, which now only sees the band
 * around the edges.  Polygons that are not planar and convex carry the constant G = -0.5 (undecided).
 */
/* (not one of the reference's operations: fused multiply-adds are fine here, and three instead of six) */
__device__ __forceinline__ float edge_value(const float4 e, float px, float py, float pz)
{
    return __builtin_fmaf(e.x, px, __builtin_fmaf(e.y, py, __builtin_fmaf(e.z, pz, e.w)));
}
__device__ __forceinline__ int edge_screen(const float4 *edg, uint32_t nv, float px, float py, float pz)
{
    float gmin = 3.0e38f;
    for (uint32_t i = 0; i < nv; ++i) {
        const float4 e = edg[i];
        gmin = fminf(gmin, edge_value(e, px, py, pz));
    }
    return gmin >= 0.0f ? 1 : (gmin <= -1.0f ? -1 : 0);
}
__device__ __forceinline__ int edge_screen_quad(const float4 *edg, float px, float py, float pz)
{
    const float4 e0 = edg[0], e1 = edg[1], e2 = edg[2], e3 = edg[3];
    const float g0 = edge_value(e0, px, py, pz), g1 = edge_value(e1, px, py, pz);
    const float g2 = edge_value(e2, px, py, pz), g3 = edge_value(e3, px, py, pz);
    const float gmin = fminf(fminf(g0, g1), fminf(g2, g3));
    return gmin >= 0.0f ? 1 : (gmin <= -1.0f ? -1 : 0);
}

/* QUAD: use the side-by-side quad evaluation (more registers, less latency).  The closest-hit
 * searches use it; the any-hit shadow search, which runs with all the shading state live, keeps
 * the register-light loops so that the kernel's register peak (and with it the number of waves a
 * SIMD can hold) is not set by a path that rarely runs. */
template <bool STATS, bool QUAD>
__device__ __forceinline__ bool polygon_inside(const float4 *vtx, const float4 *edg, uint32_t nv,
                                               float px, float py, float pz, Tally<STATS> &tl)
{
    RT_DEV_NO_ANGLESUM(px, pz);
    const bool quad = QUAD && nv == 4u;
    const int s = quad ? edge_screen_quad(edg, px, py, pz) : edge_screen(edg, nv, px, py, pz);
    if constexpr (!STATS) {
        if (s != 0)
            return s > 0;
    }
    RT_DEV_NO_BAND(STATS);
    const bool e = quad ? quad_inside_exact(quad_edges(vtx, px, py, pz)) : polygon_inside_exact(vtx, nv, px, py, pz);
    if constexpr (STATS) {
        if (s != 0 && (s > 0) != e)
            tl.screen_mismatch();
        if (s == 0)
            tl.screen_unsure();
    }
    return e;
}

/* powf(s, ke) of raytrace.c:371.  For the usual integral shininess (ke = 2, 20, ...) the power is a
 * square-and-multiply chain in double (relative error < 2^-49) rounded once to float: the CORRECTLY
 * rounded value.  glibc 2.35's powf is only faithfully rounded - it returns the neighbouring float
 * for ~0.07 % of arguments - and that is the one last-bit difference left between these kernels and
 * the reference (tests: test_radiance_is_bit_identical_but_for_libm_powf).  Other exponents use the
 * device powf (<= 1 ulp). */
__device__ __forceinline__ float phong_pow(float s, float ke)
{
#ifdef RT_FAST
    /* s >= 0: s^ke = 2^(ke * log2 s), v_log_f32 and v_exp_f32 (1 ulp each; the product's error is
     * ~ke * |log2 s| * 1e-7 relative).  log2(0) = -inf gives 0 for ke > 0; ke == 0 gives 1 like powf. */
    return ke == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(ke * __builtin_amdgcn_logf(s));
#else
    const int e = (int)ke;
    if ((float)e == ke && e >= 0 && e <= 4096) {
        double r = 1.0, b = (double)s;
        for (int k = e; k; k >>= 1) {
            if (k & 1)
                r *= b;
            b *= b;
        }
        return (float)r;
    }
    return powf(s, ke);
#endif
}

/* Cooperative copy of n float4 from HBM to LDS by the whole block. */
/* float4 slots of the sphere region of LDS: none for grid scenes, else all spheres followed by their
 * object indices (sph_obj), so that the closest-hit search never waits on a global load */
__host__ __device__ __forceinline__ uint32_t sphere_slots(uint32_t nSph)
{
    return nSph > RT_TILE_SPHERES ? 0u : nSph + (nSph + 3u) / 4u; /* (big scenes walk a grid in HBM) */
}

/* Who stages a block's LDS: all its threads together (the default), or - kernels whose sky waves leave before
 * the staging (k_primary of the reference's scene shape) - every wave the whole of it on its own: the waves
 * that are left cannot know which others are, the data is a few hundred bytes, and identical values written
 * twice to the same address are harmless. */
struct Stagers {
    uint32_t tid, stride;
    __device__ __forceinline__ static Stagers block() { return {threadIdx.x, RT_BLOCK}; }
    __device__ __forceinline__ static Stagers wave() { return {threadIdx.x & 63u, 64u}; }
};
__device__ __forceinline__ void block_copy(float4 *dst, const float4 *src, uint32_t n, Stagers st = Stagers::block())
{
    for (uint32_t i = st.tid; i < n; i += st.stride)
        dst[i] = src[i];
}

/*
 * Sphere search of a big scene (more than RT_TILE_SPHERES spheres): walk the cells of a uniform grid
 * along the ray (3-D DDA) and test only the spheres listed there, instead of all of them - the same
 * winner as the reference's loop over every sphere (smallest accepted distance, lowest index on a
 * tie), because
 *   - a sphere is listed in every cell its slightly grown bounding box overlaps, so every sphere the
 *     ray's line touches is met in some visited cell (testing a sphere twice changes nothing);
 *   - the unfiltered search (primary rays) walks the WHOLE line inside the grid, also behind the
 *     origin: the reference gives distance 0 to a sphere whose centre is behind the ray but whose
 *     line it crosses (ray.c:209-231), and such a sphere then wins;
 *   - the walk stops early only when the best distance so far is smaller than the parameter at
 *     which the ray leaves the current cell (no unvisited cell can hold a nearer crossing), or, for
 *     shadow rays, at the first accepted hit.
 * The grid lives in HBM (L2-resident: 16 384 spheres need ~0.6 MB); every lane walks its own cells.
 */
template <bool FILTERED, bool ANYHIT, bool STATS, bool EARLY, class Disc, class Cand>
__device__ __forceinline__ void grid_walk(const DevScene &S, const Ray &r, bool &found, float &d,
                                          Disc &discriminant, Cand &candidate, Tally<STATS> &tl,
                                          uint32_t exc_sph)
{
    const GridHeader *G = (const GridHeader *)S.grid;
    const uint32_t d0 = G->dim[0], d1 = G->dim[1], d2 = G->dim[2];
    const uint32_t *cellStart = S.grid + G->words;
    const uint32_t *items = cellStart + (d0 * d1 * d2 + 1u);
    const float o[3] = {r.ox, r.oy, r.oz}, dir[3] = {r.dx, r.dy, r.dz};
    float t0 = FILTERED ? 0.0f : -3.0e38f, t1 = FILTERED ? r.mag : 3.0e38f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { /* the part of the ray inside the grid's box */
        if (dir[a] != 0.0f) {
            const float inv = 1.0f / dir[a];
            const float ta = (G->gmin[a] - o[a]) * inv, tb = (G->gmax[a] - o[a]) * inv;
            t0 = fmaxf(t0, fminf(ta, tb));
            t1 = fminf(t1, fmaxf(ta, tb));
        } else if (o[a] < G->gmin[a] || o[a] > G->gmax[a]) {
            return;
        }
    }
    if (!(t0 <= t1))
        return;
    int c[3], step[3];
    float tmax[3], tdel[3];
    const uint32_t dim[3] = {d0, d1, d2};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float p = o[a] + dir[a] * t0;
        int ci = (int)floorf((p - G->gmin[a]) * G->inv_h[a]);
        ci = ci < 0 ? 0 : (ci >= (int)dim[a] ? (int)dim[a] - 1 : ci);
        c[a] = ci;
        if (dir[a] > 0.0f) {
            step[a] = 1;
            tmax[a] = (G->gmin[a] + (float)(ci + 1) * G->h[a] - o[a]) / dir[a];
            tdel[a] = G->h[a] / dir[a];
        } else if (dir[a] < 0.0f) {
            step[a] = -1;
            tmax[a] = (G->gmin[a] + (float)ci * G->h[a] - o[a]) / dir[a];
            tdel[a] = -G->h[a] / dir[a];
        } else {
            step[a] = 0;
            tmax[a] = 3.0e38f;
            tdel[a] = 3.0e38f;
        }
    }
    /* at most one cell per step along some axis: the count bounds the loop whatever the arithmetic does */
    for (uint32_t left = d0 + d1 + d2 + 3u; left; --left) {
        const uint32_t cell = ((uint32_t)c[2] * d1 + (uint32_t)c[1]) * d0 + (uint32_t)c[0];
        const uint32_t beg = cellStart[cell], end = cellStart[cell + 1u];
        for (uint32_t j = beg; j < end; ++j) {
            const uint32_t k = items[j];
            if (FILTERED && k == exc_sph)
                continue;
            tl.sph_test();
            float B;
            const float det = discriminant(S.sph[k], B);
            if (!(det < 0.0f))
                candidate(k, B, det, false);
        }
        const float texit = fminf(tmax[0], fminf(tmax[1], tmax[2]));
        if (found && !STATS) {
            if (ANYHIT)
                break;
            if (EARLY && texit > 0.0f && d < texit - (1e-5f + 1e-5f * texit))
                break;
        }
        if (!(texit < t1))
            break; /* the ray leaves the box (or ends) inside this cell */
        const int a = (tmax[0] <= tmax[1]) ? ((tmax[0] <= tmax[2]) ? 0 : 2) : ((tmax[1] <= tmax[2]) ? 1 : 2);
        /* (selects instead of run-time indexed arrays: those would live in scratch memory) */
        const int nc = (a == 0 ? c[0] : a == 1 ? c[1] : c[2]) + (a == 0 ? step[0] : a == 1 ? step[1] : step[2]);
        if (nc < 0 || nc >= (int)(a == 0 ? d0 : a == 1 ? d1 : d2))
            break;
        if (a == 0) { c[0] = nc; tmax[0] += tdel[0]; }
        else if (a == 1) { c[1] = nc; tmax[1] += tdel[1]; }
        else { c[2] = nc; tmax[2] += tdel[2]; }
    }
}

/*
 * The reference's loop over EVERY sphere, for the lanes whose ray cannot use the grid, done by the
 * whole wave: the lanes take such rays one at a time, each lane tests every 64th sphere against the
 * broadcast ray (coalesced loads), and a butterfly reduction picks the smallest accepted distance,
 * the lowest index on a tie - the loop's winner (raytrace.c:188-272).  Must be reached by all 64 lanes.
 */
template <bool FILTERED, bool STATS>
__device__ __forceinline__ void wave_loop_all(const DevScene &S, bool need, const Ray &r, uint32_t exc_sph,
                                              bool &found, float &d, uint32_t &best_k, Tally<STATS> &tl)
{
    unsigned long long todo = __ballot(need);
    const uint32_t lane = lane_id();
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const float ox = __shfl(r.ox, src, 64), oy = __shfl(r.oy, src, 64), oz = __shfl(r.oz, src, 64);
        const float dx = __shfl(r.dx, src, 64), dy = __shfl(r.dy, src, 64), dz = __shfl(r.dz, src, 64);
        const float mag = __shfl(r.mag, src, 64);
        const uint32_t exc = __shfl(exc_sph, src, 64);
        bool f = false;
        float bw = 0.0f;
        uint32_t bk = RT_NONE;
        for (uint32_t k = lane; k < S.nSph; k += 64u) {
            if (FILTERED && k == exc)
                continue;
            tl.sph_test();
            const float4 s = S.sph[k];
            const float ex = ox - s.x, ey = oy - s.y, ez = oz - s.z;
            const float B = 2.0f * ((dx * ex + dy * ey) + dz * ez);
            const float Cq = ((ex * ex + ey * ey) + ez * ez) - s.w;
            const float det = (B * B) - (4.0f * Cq);
            if (det < 0.0f)
                continue;
            tl.sph_hit();
            if (FILTERED && !(B < 0.0f))
                continue;
            const float w = sphere_distance(B, det);
            if (FILTERED && !(w < mag && w > 0.0f))
                continue;
            if (!f || w < bw) { /* ascending k within a lane: the first of equal distances stays */
                bw = w;
                bk = k;
                f = true;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const bool of = __shfl_xor((int)f, off, 64) != 0;
            const float ow = __shfl_xor(bw, off, 64);
            const uint32_t ok = __shfl_xor(bk, off, 64);
            if (of && (!f || ow < bw || (ow == bw && ok < bk))) {
                f = true;
                bw = ow;
                bk = ok;
            }
        }
        if ((int)lane == src) {
            found = f;
            d = bw;
            best_k = bk;
        }
    }
}

/*
 * Intersect-all: get_object3d_intersect (raytrace.c:188-222, FILTERED=false) and
 * get_object3d_intersect_excl (raytrace.c:231-272, FILTERED=true: skips the excluded object,
 * accepts only 0 < d < ray.magnitude).  Objects are visited as "all spheres, then all
 * polygons"; the reference's rule (strictly smaller distance replaces, so the lowest object
 * index wins a tie) is kept by comparing (distance, object index) when a polygon ties.
 *
 * ANYHIT (shadow rays): the reference only uses "was anything hit" (raytrace.c:334-340), so the
 * lane stops at its first accepted hit.  STATS disables that early exit and counts every test
 * the reference would make.
 *
 * GRID: more spheres than LDS holds (RT_TILE_SPHERES): the spheres are searched through a grid in
 * HBM (grid_walk) instead of a loop over all of them.
 */
template <bool GRID, bool FILTERED, bool ANYHIT, bool STATS, bool BIG, bool EYE = false>
__device__ __forceinline__ void intersect_all(const DevScene &S, const float4 *lds_poly,
                                              float4 *lds_sph, bool active, const Ray &r,
                                              uint32_t exc_sph, uint32_t exc_poly, float &out_d,
                                              uint32_t &out_obj, Tally<STATS> &tl,
                                              const float4 *eye_terms = nullptr, bool no_spheres = false)
{
    /* no_spheres (wave-uniform): the caller has shown that no lane's ray can meet a sphere (classify_pixel) */
    /* EYE: the ray starts at the eye, like every primary ray, so the terms of the tests that depend on
     * the origin alone were evaluated once per block by stage_eye_terms (same expressions, same
     * rounding): (o - c, 4*(|o - c|^2 - r^2)) per sphere, -(n.o + F) per polygon. */
    static_assert(!EYE || (!GRID && !BIG && !FILTERED), "eye terms: small scenes, primary rays");
    float d = 0.0f;
    uint32_t best_k = RT_NONE; /* winning sphere-list index */
    bool found = false;

    /* one sphere against this lane's ray: ray.c:185-195 up to the discriminant */
    auto discriminant = [&](const float4 s, float &B) -> float {
        if (EYE) { /* s = (ex, ey, ez, 4*Cq) */
            B = 2.0f * ((r.dx * s.x + r.dy * s.y) + r.dz * s.z);
            return (B * B) - s.w;
        }
        const float ex = r.ox - s.x, ey = r.oy - s.y, ez = r.oz - s.z;
        B = 2.0f * ((r.dx * ex + r.dy * ey) + r.dz * ez);
        const float Cq = ((ex * ex + ey * ey) + ez * ez) - s.w;
        return (B * B) - (4.0f * Cq);
    };
    /* a sphere whose discriminant is not negative: ray.c:209-240 and the caller's filter.  ORDERED:
     * candidates arrive in list order (the reference's loop: the first of equal distances wins);
     * otherwise (grid walk) ties go to the lower index explicitly. */
    auto candidate = [&](uint32_t k, float B, float det, bool ordered) {
        tl.sph_hit();
        if (FILTERED) {
            /* only 0 < w < magnitude can be accepted; w > 0 needs B < 0 */
            if (B < 0.0f) {
                const float w = sphere_distance(B, det);
                if (w < r.mag && w > 0.0f && (!found || w < d || (!ordered && w == d && k < best_k))) {
                    d = w;
                    best_k = k;
                    found = true;
                }
            }
        } else {
            const float w = sphere_distance(B, det);
            if (!found || w < d || (!ordered && w == d && k < best_k)) {
                d = w;
                best_k = k;
                found = true;
            }
        }
    };
    if (GRID && !no_spheres) {
        /* the grid is only exact for rays the reference's test treats (almost) geometrically: a
         * direction of unit length (shorter ones are fine too: the test then accepts LESS than
         * geometry, but the reported distance is off the sphere, so they walk without early exit) and
         * an origin within the grid's reach (GridHeader).  The reference's refraction formula
         * (raytrace.c:517-525) makes rays LONGER than unit when they leave a sphere with n < 1
         * (|d|^2 = 1 + (1 - cos^2)(1/n^2 - n^2)): for those the reference's test accepts spheres in a
         * cone around the ray, and they take the reference's own loop over every sphere. */
        const GridHeader *G = (const GridHeader *)S.grid;
        const float l2 = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz;
        const float qx = r.ox - G->centre[0], qy = r.oy - G->centre[1], qz = r.oz - G->centre[2];
        const bool walk = active && (l2 - 1.0f <= RT_GRID_UNIT) && ((qx * qx + qy * qy) + qz * qz <= G->origin_reach2);
        if (walk) {
            if (l2 - 1.0f >= -RT_GRID_UNIT)
                grid_walk<FILTERED, ANYHIT, STATS, true>(S, r, found, d, discriminant, candidate, tl, exc_sph);
            else
                grid_walk<FILTERED, ANYHIT, STATS, false>(S, r, found, d, discriminant, candidate, tl, exc_sph);
        }
        wave_loop_all<FILTERED, STATS>(S, active && !walk, r, exc_sph, found, d, best_k, tl);
    } else if (!GRID && active && !no_spheres) {
        const uint32_t cnt = S.nSph;
        uint32_t j = 0;
        if (BIG) {
            /* Scenes with hundreds of spheres: RT_SPHERES_PER_TRIP spheres per trip.  The LDS reads are
             * in flight together and the loop's scalar bookkeeping is shared; with one sphere
             * per trip every test waited out its own LDS round trip at 2-3 waves per SIMD (the
             * sphere tile leaves room for no more).  Candidates are still taken in list order. */
            constexpr uint32_t U = RT_SPHERES_PER_TRIP;
            for (; j + U <= cnt; j += U) {
                float Bq[U], detq[U];
                float4 sq[U];
#pragma unroll
                for (uint32_t u = 0; u < U; ++u)
                    sq[u] = lds_sph[j + u];
                bool any = false;
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {
                    detq[u] = discriminant(sq[u], Bq[u]);
                    if (FILTERED && (j + u) == exc_sph)
                        detq[u] = -1.0f; /* the excluded sphere: never a candidate, never counted */
                    else
                        tl.sph_test();
                    any = any || !(detq[u] < 0.0f);
                }
                if (any) {
#pragma unroll
                    for (uint32_t u = 0; u < U; ++u)
                        if (!(detq[u] < 0.0f))
                            candidate(j + u, Bq[u], detq[u], true);
                    if (ANYHIT && !STATS && found)
                        break;
                }
            }
        }
        if (!(BIG && ANYHIT && !STATS && found)) {
            /* Four spheres per trip while four are left: their scalar loads (or LDS broadcasts) are in flight together
             * and the tests of one trip share the loop's bookkeeping - with one sphere per trip a 64-sphere scene waited
             * out a scalar-cache round trip per test (the tolerance build, half the arithmetic, was 9 % faster).
             * Candidates are still taken in list order; an any-hit search may test up to three spheres more. */
            constexpr uint32_t V = 4u;
            for (; j + V <= cnt; j += V) {
                float4 sq[V];
                float Bq[V], detq[V];
#pragma unroll
                for (uint32_t u = 0; u < V; ++u)
                    sq[u] = EYE ? eye_terms[j + u] : const_load4(S.sph + j + u);
                bool any = false;
#pragma unroll
                for (uint32_t u = 0; u < V; ++u) {
                    detq[u] = discriminant(sq[u], Bq[u]);
                    if (FILTERED && (j + u) == exc_sph)
                        detq[u] = -1.0f; /* the excluded sphere: never a candidate, never counted */
                    else
                        tl.sph_test();
                    any = any || !(detq[u] < 0.0f);
                }
                if (any) {
#pragma unroll
                    for (uint32_t u = 0; u < V; ++u)
                        if (!(detq[u] < 0.0f))
                            candidate(j + u, Bq[u], detq[u], true);
                }
                if (ANYHIT && !STATS && __all(found)) /* (a wave-uniform exit: the loads of a trip are scalar) */
                    break;
            }
            if (!(ANYHIT && !STATS && found))
            for (; j < cnt; ++j) {
                /* (same address in all lanes: the eye terms as an LDS broadcast, the spheres themselves through
                 * the scalar cache - a wave-wide 16-byte LDS read occupies the CU's LDS pipeline for 8 cycles) */
                const float4 s = EYE ? eye_terms[j] : const_load4(S.sph + j);
                if (FILTERED && j == exc_sph)
                    continue;
                tl.sph_test();
                float B;
                const float det = discriminant(s, B);
                if (!(det < 0.0f)) {
                    candidate(j, B, det, true);
                    if (ANYHIT && !STATS && found)
                        break;
                }
            }
        }
    }

    uint32_t best_obj = RT_NONE;
    if (!ANYHIT && found) {
        if (GRID)
            best_obj = S.sph_obj[best_k];
        else
            best_obj = ((const uint32_t *)(lds_sph + S.nSph))[best_k];
    }

    if (active && !(ANYHIT && !STATS && found)) {
        for (uint32_t p = 0; p < S.nPoly; ++p) {
            if (FILTERED && p == exc_poly)
                continue;
            tl.poly_test();
            const float4 pl = const_load4(S.poly + p);
            /* ray.c:92-117 */
            const float num = EYE ? eye_terms[S.nSph + p].x
                                  : -1.0f * (dot3(pl.x, pl.y, pl.z, r.ox, r.oy, r.oz) + pl.w);
            const float den = dot3(pl.x, pl.y, pl.z, r.dx, r.dy, r.dz);
            if (den == 0.0f) {
                tl.poly_par();
                continue;
            }
            const float w = div_rn(num, den);
            if (w < 0.0f) {
                tl.poly_behind();
                continue;
            }
            tl.poly_full();
            const float4 info = const_load4(S.poly + S.nPoly + p);
            const uint32_t pobj = __float_as_uint(info.z);
            /* would this hit be taken at all?  (skips the angle sum when it cannot matter) */
            bool wins;
            if (FILTERED)
                wins = (w < r.mag && w > 0.0f) && (!found || ANYHIT || w < d || (w == d && pobj < best_obj));
            else
                wins = !found || w < d || (w == d && pobj < best_obj);
            if (!wins && !STATS)
                continue;
            const float px = r.ox + r.dx * w, py = r.oy + r.dy * w, pz = r.oz + r.dz * w;
            const float4 *const vtx = lds_poly + 2 * S.nPoly + __float_as_uint(info.x);
            /* the edge equations follow the vertex pool, in the same order */
            const bool inside = polygon_inside<STATS, !ANYHIT>(vtx, vtx + ((S.nPolySlots - 2u * S.nPoly) >> 1),
                                                      __float_as_uint(info.y), px, py, pz, tl);
            if (inside) {
                tl.poly_hit();
                if (wins) {
                    d = w;
                    best_obj = pobj;
                    found = true;
                    if (ANYHIT && !STATS)
                        break;
                }
            }
        }
    }
    out_d = d;
    out_obj = ANYHIT ? (found ? 0u : RT_NONE) : (found ? best_obj : RT_NONE);
}

/* number of LDS words that hold the light-visibility bits of all but the last 32 lights */
__host__ __device__ __forceinline__ uint32_t vis_words(uint32_t nLights)
{
    return nLights > 32u ? ((nLights - 1u) / 32u) * RT_BLOCK : 0u;
}

/* Light and object tables: staged in LDS behind the visibility words when they are small
 * (DevScene::tabInLds, kernels instantiated with TABLDS), read from HBM / L2 otherwise.  The
 * choice is a template parameter so that the LDS case compiles to ds_read: through one generic
 * pointer every table read is a flat_load, whose wait is vmcnt(0) - it sits out every global
 * store and prefetch in flight (measured: that serialised k_bounce). */
template <bool TABLDS> struct Tables {
    const float4 *p;
    __device__ __forceinline__ float4 operator[](uint32_t i) const { return p[i]; }
};
/* callers follow up with __syncthreads() */
template <bool TABLDS>
__device__ __forceinline__ Tables<TABLDS> stage_tables(const DevScene &S, uint32_t *lds_vis, Stagers st = Stagers::block())
{
    if constexpr (TABLDS) {
        float4 *lds_tab = (float4 *)(lds_vis + vis_words(S.nLights));
        block_copy(lds_tab, S.tab, S.tabSlots, st);
        return Tables<TABLDS>{lds_tab};
    } else {
        return Tables<TABLDS>{S.tab};
    }
}

/* Scene colours every shade needs (background, ambient light), in LDS rather than in scalar registers (the trace
 * kernels run out of them and spill into VGPR lanes) */
enum { KC_BG = 0, KC_AMBIENT = 3, KC_WORDS = 8 };
__device__ __forceinline__ void stage_colours(const DevScene &S, float *kc, Stagers st = Stagers::block())
{
    if (st.tid == 0)
        for (int k = 0; k < 3; ++k) {
            kc[KC_BG + k] = S.bg[k];
            kc[KC_AMBIENT + k] = S.ambient[k];
        }
}

/* The origin-only terms of the ray-object tests for rays that start at the eye (intersect_all<EYE>):
 * ray.c:185-195 and ray.c:92-95 with origin = eye.  Needs lds_sph / lds_poly staged (barrier before),
 * callers follow up with __syncthreads(). */
__device__ __forceinline__ void stage_eye_terms(const DevScene &S, const DevCamera &C, const float4 *lds_poly,
                                                const float4 *lds_sph, float4 *eye_terms, Stagers st = Stagers::block())
{
    const float ox = C.eye[0], oy = C.eye[1], oz = C.eye[2];
    for (uint32_t k = st.tid; k < S.nSph + S.nPoly; k += st.stride) {
        if (k < S.nSph) {
            const float4 s = lds_sph[k];
            const float ex = ox - s.x, ey = oy - s.y, ez = oz - s.z;
            const float Cq = ((ex * ex + ey * ey) + ez * ez) - s.w;
            eye_terms[k] = make_float4(ex, ey, ez, 4.0f * Cq);
        } else {
            const float4 pl = lds_poly[k - S.nSph];
            eye_terms[k] = make_float4(-1.0f * (dot3(pl.x, pl.y, pl.z, ox, oy, oz) + pl.w), 0.f, 0.f, 0.f);
        }
    }
}

/* The shard a tile's wave feeds.  A multiplicative hash of the tile's index: with the index modulo the shards, a
 * compact object - the glass sphere of the reference's scene, a few dozen tiles wide and high - filled a few shards
 * 3.4 times fuller than the average (a row of a 1920-pixel frame is 240 tiles, 48 modulo 64: a column of tiles
 * meets only four shards), and the fullest shard is what a context's queues are sized for. */
#ifdef RT_DEV_SHARD_MODULO /* development: the old rule */
__device__ __forceinline__ uint32_t tile_shard(uint32_t tile) { return tile % RT_SHARDS; }
#else
__device__ __forceinline__ uint32_t tile_shard(uint32_t tile) { return (tile * 2654435761u) >> 26; }
#endif
static_assert(RT_SHARDS == 64, "tile_shard keeps the top six bits");

/* counter line of (level, shard): low word = nodes made at `level`, high word = rays queued
 * for level+1 (rt_device.h) */
__device__ __forceinline__ unsigned long long *level_line(uint32_t *counters, uint32_t level,
                                                          uint32_t shard)
{
    return (unsigned long long *)(counters + CNT_LEVELS + (level * RT_SHARDS + shard) * RT_LINE_WORDS);
}

/* What a shard has made so far, in ONE memory round trip: lane l of the calling wave loads the counter
 * line of level l (every lane must be active), the sums are taken over lanes.  The loop of scalar
 * loads it replaces, one per level and each waited for, cost 3 us per frame (0.4068 -> 0.4038 ms).
 *   used  = nodes made on levels < level (first node index of `level` in the shard's range)
 *   nodes = nodes made on `level` itself (final once the kernel that processes `level` has ended)
 *   hits  = hits queued for `level` (by the kernel that processed level - 1; 0 for level 0)          */
struct LevelCounts { uint32_t used, nodes, hits; };
__device__ __forceinline__ LevelCounts level_counts(uint32_t *counters, uint32_t level, uint32_t shard)
{
    const unsigned long long line = *level_line(counters, min(lane_id(), level), shard);
    const uint32_t lo = (uint32_t)line, hi = (uint32_t)(line >> 32);
    LevelCounts c;
    c.used = 0;
    for (uint32_t l = 0; l < level; ++l)
        c.used += (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)l);
    c.nodes = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)level);
    c.hits = level ? (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)(level - 1u)) : 0u;
    return c;
}

/* Did any kernel of this batch overflow a queue or a node range?  The batch is abandoned then (the
 * host renders it again in smaller pieces) and every later kernel returns at once: its inputs hold
 * unwritten slots.  The kernels that could have raised the flag have finished, so the answer is the
 * same for every thread of the launch (block-uniform).  One word for the whole batch: the per-level
 * words this replaced cost every thread of every fold launch 64 loads. */
__device__ __forceinline__ bool overflowed_before(const uint32_t *counters)
{
    return counters[CNT_OVERFLOW] != 0u;
}

/*
 * Wave-aggregated reservation of one tree node per calling lane plus ray-queue slots for the
 * lanes that spawn a reflection and/or a transmission ray: ballot the requests, ONE 64-bit
 * atomicAdd per wave on the shard's counter line, exclusive prefixes by popcount of the lower
 * lanes (v_mbcnt).  Must be called by exactly the lanes that want a node.  Reflection rays of
 * the wave occupy [q, q+nr), transmission rays [q+nr, q+nr+nt).  A reservation past the room
 * left sets the overflow flag and yields RT_NONE (the host re-renders the batch in smaller pieces).
 */
__device__ __forceinline__ void wave_reserve(unsigned long long *line, bool want_r, bool want_t,
                                             uint32_t node_room, uint32_t ray_room,
                                             uint32_t *overflow, uint32_t &node_local,
                                             uint32_t &slot_r, uint32_t &slot_t)
{
    const unsigned long long m = __ballot(1);
    const unsigned long long mr = __ballot(want_r);
    const unsigned long long mt = __ballot(want_t);
    const uint32_t lane = lane_id();
    const unsigned long long below = (1ull << lane) - 1ull;
    const int leader = __ffsll((long long)m) - 1;
    const uint32_t nr = __popcll(mr), nt = __popcll(mt);
    unsigned long long old = 0;
    if (RT_DEV_ATOMIC((int)lane == leader))
        old = atomicAdd(line, (unsigned long long)__popcll(m) | ((unsigned long long)(nr + nt) << 32));
    const uint32_t old_lo = __shfl((uint32_t)old, leader, 64);
    const uint32_t old_hi = __shfl((uint32_t)(old >> 32), leader, 64);
    node_local = old_lo + __popcll(m & below);
    slot_r = old_hi + __popcll(mr & below);
    slot_t = old_hi + nr + __popcll(mt & below);
    bool bad = node_local >= node_room;
    if (bad)
        node_local = RT_NONE;
    if (!want_r || bad || slot_r >= ray_room) {
        bad = bad || (want_r && slot_r >= ray_room);
        slot_r = RT_NONE;
    }
    if (!want_t || node_local == RT_NONE || slot_t >= ray_room) {
        bad = bad || (want_t && slot_t >= ray_room);
        slot_t = RT_NONE;
    }
    if (bad)
        atomicOr(overflow, 1u);
}

__device__ __forceinline__ float clamp1(float v) { return v > 1.0f ? 1.0f : v; } /* color.c:44-54 */

/* Wave totals of shade nodes and spawned rays.  They are ballot popcounts taken where the whole
 * wave has reconverged (never inside a branch some lanes skip: a lane that sits out would miss
 * the increment), so they live in scalar registers instead of one VGPR per lane. */
struct SpawnCounts {
    uint32_t shade_nodes = 0, refl = 0, trans = 0;
    __device__ __forceinline__ void add(bool node, bool r, bool t)
    {
        shade_nodes += (uint32_t)__popcll(__ballot(node));
        refl += (uint32_t)__popcll(__ballot(r));
        trans += (uint32_t)__popcll(__ballot(t));
    }
};
__device__ __forceinline__ void wave_add_total(uint64_t *stats, int slot, uint32_t total)
{
    if (lane_id() == 0 && total != 0)
        atomicAdd((unsigned long long *)&stats[slot], (unsigned long long)total);
}

/*
 * get_shade_color_phong (raytrace.c:282-564) for one hit, minus the recursion: computes the
 * local colour (ambient + per-light diffuse/specular with inline any-hit shadow rays), then
 * either returns a finished leaf colour or allocates a tree node and queues the reflection /
 * transmission rays for the next level.
 *
 * eye_src      : gives back the origin / direction of the ray that found the hit (pushed
 *                origin, possibly unnormalised direction) when they are needed again after the
 *                shadow rays: re-read from the queue or rebuilt from the sample position, so
 *                that they do not occupy registers during the any-hit searches
 * hit, obj     : whether this lane has a hit and which object
 * px,py,pz     : the hit point, origin + direction * distance (ray.c:121-122, 236-237)
 * level        : recursion depth of this node
 * link         : value stored in the node's link word (parent<<3|slot for level >= 1,
 *                next-root<<3 for level 0); flags are OR-ed in here
 * spawn_r/t    : whether the node asked for a reflection / transmission ray (for the counts)
 * Returns true when the lane produced a tree node (index in node_idx); otherwise `leaf` holds
 * the clamped colour of a childless node.
 */
/* what shade_and_spawn hands back in its DFS mode (k_subtree) instead of making a node and queueing the
 * children's hits: the node's own colour and coefficients, which children it asked for and which of them hit
 * something, and the transmission child's ray and hit (the reflection child's are parked in LDS rows 7-16) */
struct DfsNode {
    float r, g, b, kr, kt;
    bool want_r, want_t, hit_r, hit_t;
    Ray t_ray;
    float t_dist;
    uint32_t t_obj;
};
template <bool GRID, bool STATS, bool TABLDS, bool BLINN = false, bool DFS = false, typename EyeSource>
__device__ __forceinline__ bool shade_and_spawn(const DevScene &S, const DevCamera &C,
                                                const DevWork &Wk, const float4 *lds_poly,
                                                float4 *lds_sph, uint32_t *lds_vis, float *lds_park,
                                                const float *kc, const Tables<TABLDS> &tab, bool hit,
                                                const EyeSource &eye_src, float px, float py, float pz,
                                                uint32_t obj, uint32_t level,
                                                uint32_t link, uint32_t shard, uint32_t node_base,
                                                uint32_t node_room, float leaf[3], uint32_t &node_idx,
                                                bool &spawn_r, bool &spawn_t, Tally<STATS> &tl,
                                                PhaseClock &pc, bool all_lit = false, DfsNode *dfs = nullptr)
{
    /* all_lit (wave-uniform): the caller has shown that no shadow ray of this wave's hits can meet anything
     * (classify_pixel: one polygon in the scene, every sphere clear of every line from a hit to a light) */
    node_idx = RT_NONE;
    spawn_r = spawn_t = false;
    if (!GRID && !hit)
        return false;

    /*
     * Two passes over the lights, so that the any-hit searches (the register-hungry part: sphere
     * roots in double, polygon angle sums) run while only the hit point and the exclusion ids are
     * live, and the material / normal / view vector are loaded and built only afterwards:
     *   pass A  shadow rays of all lights -> visibility bits
     *   pass B  Phong terms of the visible lights, in light order (the sums are order-sensitive)
     * The shadow direction is rebuilt in pass B by the same expressions, hence bit-identical.
     */
    const uint32_t rec = 2u * S.nLights + (hit ? obj : 0u) * (uint32_t)RT_OBJ_FIELDS; /* this object's record */
    uint32_t exc_sph = RT_NONE, exc_poly = RT_NONE;
    if (hit) {
        const float4 g = tab[rec + RT_OBJ_GEOM];
        const bool sph = __float_as_uint(g.y) == 1u; /* GEOMETRY_SPHERE */
        exc_sph = sph ? __float_as_uint(g.z) : RT_NONE;
        exc_poly = sph ? RT_NONE : __float_as_uint(g.z);
        tl.node(sph);
    }

    float dfr = 0.f, dfg = 0.f, dfb = 0.f, spr = 0.f, spg = 0.f, spb = 0.f;
    const uint32_t n_lights = RT_DEV_LIGHTS(S.nLights);
    /* pass A (raytrace.c:327-347).  Visibility words of all but the last 32 lights go to LDS. */
    uint32_t visible = 0;
    if (all_lit) /* (at most 32 lights: the caller's condition) */
        visible = hit ? (n_lights >= 32u ? 0xFFFFFFFFu : (1u << n_lights) - 1u) : 0u;
    for (uint32_t l = 0; l < (all_lit ? 0u : n_lights); ++l) {
        const float4 lp = tab[2u * l];
        /* ray_create(pt, light) ray.c:29-41 */
        Ray sh;
        sh.ox = px; sh.oy = py; sh.oz = pz;
        sh.dx = lp.x - px; sh.dy = lp.y - py; sh.dz = lp.z - pz;
        float inv;
        length_and_reciprocal(sh.dx, sh.dy, sh.dz, sh.mag, inv);
        sh.dx = sh.dx * inv; sh.dy = sh.dy * inv; sh.dz = sh.dz * inv;
        if constexpr (!BLINN) /* raytrace.c:613-616: the older routine does not push its shadow rays */
            tinypush<true>(sh);
        float sd;
        uint32_t sobj;
        intersect_all<GRID, true, true, STATS, !TABLDS>(S, lds_poly, lds_sph, hit, sh, exc_sph, exc_poly, sd,
                                                sobj, tl);
        if (hit && sobj == RT_NONE)
            visible |= 1u << (l & 31u);
        if (n_lights == 1u) { /* one light: its direction waits in LDS for pass B (rows 7-9 are free until
                                 the child searches) instead of being rebuilt there */
            float *const keep = lds_park + threadIdx.x;
            keep[7 * RT_BLOCK] = sh.dx; keep[8 * RT_BLOCK] = sh.dy; keep[9 * RT_BLOCK] = sh.dz;
        }
        if ((l & 31u) == 31u && l + 1u < n_lights) {
            lds_vis[(l >> 5) * RT_BLOCK + threadIdx.x] = visible; /* own slot: no barrier needed */
            visible = 0;
        }
    }
    pc.mark(2);
    /* the eye ray's origin, back from where it is kept (cannot be merged with the first read) */
    float eox, eoy, eoz;
    eye_src.origin(eox, eoy, eoz);

    /* the tolerance build has registers to spare (63 of the 72 that seven waves per SIMD allow): normal and
     * view vector are built once and stay; the exact build builds them again for the spawn (below) */
    float Nx = 0.f, Ny = 0.f, Nz = 0.f, Vx = 0.f, Vy = 0.f, Vz = 0.f;
    if (kFast && hit) {
        const float4 gD = tab[rec + RT_OBJ_GEOM];
        const float4 gE = tab[rec + RT_OBJ_SHAPE];
        if (__float_as_uint(gD.y) == 1u) {
            Nx = px - gE.x; Ny = py - gE.y; Nz = pz - gE.z;
            normalize3(Nx, Ny, Nz);
        } else {
            Nx = gE.x; Ny = gE.y; Nz = gE.z;
        }
        Vx = eox - px; Vy = eoy - py; Vz = eoz - pz;
        normalize3(Vx, Vy, Vz);
    }
    /* pass B (raytrace.c:349-384) */
    if (hit && n_lights != 0) {
        const float ke = tab[rec + RT_OBJ_PHONG].y;
        if constexpr (!kFast) {
            const float4 gD = tab[rec + RT_OBJ_GEOM];
            const float4 gE = tab[rec + RT_OBJ_SHAPE];
            /* normal: geometry.c:37-57 */
            if (__float_as_uint(gD.y) == 1u) {
                const float nx = px - gE.x, ny = py - gE.y, nz = pz - gE.z;
                Nx = nx; Ny = ny; Nz = nz;
                normalize3(Nx, Ny, Nz);
            } else {
                Nx = gE.x; Ny = gE.y; Nz = gE.z;
            }
            /* V = normalize(eye.origin - pt) raytrace.c:318-320 */
            Vx = eox - px; Vy = eoy - py; Vz = eoz - pz;
            normalize3(Vx, Vy, Vz);
        }
        const uint32_t last = (n_lights - 1u) >> 5;
        for (uint32_t c = 0; c <= last; ++c) {
            uint32_t m = (c == last) ? visible : lds_vis[c * RT_BLOCK + threadIdx.x];
            while (m) {
                const uint32_t l = c * 32u + (uint32_t)__ffs((int)m) - 1u;
                m &= m - 1u;
                const float4 lp = tab[2u * l];
                const float4 lc = tab[2u * l + 1u];
                float sx, sy, sz;
                if (n_lights == 1u && !all_lit) { /* parked by pass A: the same expressions, hence the same values */
                    uint32_t t = threadIdx.x;
                    asm volatile("" : "+v"(t));
                    const float *back = lds_park + t;
                    sx = back[7 * RT_BLOCK]; sy = back[8 * RT_BLOCK]; sz = back[9 * RT_BLOCK];
                } else {
                    sx = lp.x - px; sy = lp.y - py; sz = lp.z - pz;
                    float len, inv;
                    length_and_reciprocal(sx, sy, sz, len, inv);
                    sx = sx * inv; sy = sy * inv; sz = sz * inv;
                }
                const float t = dot3(sx, sy, sz, Nx, Ny, Nz);
                if (!(t <= 0.0f)) {
                    tl.lit();
                    dfr += lc.x * t; dfg += lc.y * t; dfb += lc.z * t;
                    float s;
                    if constexpr (BLINN) { /* halfway vector raytrace.c:632-634,647; the double quotient
                                              1.0 / |V+S| stored to float is the rounded float quotient */
                        const float hx = Vx + sx, hy = Vy + sy, hz = Vz + sz;
                        const float inv = recip_rn(mag3(hx, hy, hz));
                        s = dot3(hx * inv, hy * inv, hz * inv, Nx, Ny, Nz);
                    } else {
                        const float t2 = 2.0f * t;
                        const float Rx = Nx * t2 - sx, Ry = Ny * t2 - sy, Rz = Nz * t2 - sz;
                        s = dot3(Rx, Ry, Rz, Vx, Vy, Vz);
                    }
                    if (s < 0.0f)
                        s = 0.0f;
                    s = phong_pow(s, ke);
                    spr += lc.x * s; spg += lc.y * s; spb += lc.z * s;
                }
            }
        }
    }
    pc.mark(3);
    if (!GRID && !hit)
        return false; /* (grid scenes: every lane must reach the child searches below - wave_loop_all needs whole waves) */

    /* material, object colour, normal and view vector for the combine and the spawn */
    const float4 mA = tab[rec + RT_OBJ_DIFFUSE];
    const float4 mB = tab[rec + RT_OBJ_SPECULAR];
    const float4 mC = tab[rec + RT_OBJ_PHONG];
    const float4 mD = tab[rec + RT_OBJ_GEOM];
    const bool is_sphere = __float_as_uint(mD.y) == 1u;

    /* get_object_color object3d.c:64-117 */
    float ocr, ocg, ocb;
    if (BLINN || is_sphere) { /* raytrace.c:589,669-680: the older routine takes the material's colour for polygons too */
        ocr = mA.x; ocg = mA.y; ocb = mA.z;
    } else {
        /* get_2dproject_position object3d.c:30-39; quotients by constants: any reciprocal within
         * 1 ulp serves the correction steps of div_with */
        const float u = div_with(px + 15.0f, 22.0f, 1.0f / 22.0f);
        const float v = div_with(pz + 100.0f, 100.0f, 1.0f / 100.0f);
        const int xt = (int)(11.0f * u);
        const int yt = (int)(50.0f * v);
        ocr = 1.0f; ocb = 0.0f;
        ocg = ((xt % 2) == (yt % 2)) ? 0.0f : 1.0f;
    }
    /* ambient raytrace.c:303-307 */
    float outr = (ocr * kc[KC_AMBIENT + 0]) * mA.w;
    float outg = (ocg * kc[KC_AMBIENT + 1]) * mA.w;
    float outb = (ocb * kc[KC_AMBIENT + 2]) * mA.w;

    /* raytrace.c:390-401 */
    outr += (mC.x * dfr) * ocr + (mB.w * spr) * mB.x;
    outg += (mC.x * dfg) * ocg + (mB.w * spg) * mB.y;
    outb += (mC.x * dfb) * ocb + (mB.w * spb) * mB.z;

    const float kr = mC.z, kt = mC.w;
    const bool want_r = !BLINN && hit && (kr != 0.0f) && (level != C.depth); /* raytrace.c:405 */
    const bool want_t = !BLINN && hit && (kt != 0.0f) && (level != C.depth); /* raytrace.c:452 */
    RT_DEV_NO_SPAWN(hit, leaf, outr, outg, outb, spawn_r, spawn_t);
    spawn_r = want_r;
    spawn_t = want_t;
    if ((BLINN || !GRID) && !(want_r || want_t)) { /* (the older routine never spawns: raytrace.c:683-684) */
        if (!hit)
            return false;
        leaf[0] = clamp1(outr); /* raytrace.c:562 */
        leaf[1] = clamp1(outg);
        leaf[2] = clamp1(outb);
        return false;
    }

    /* normal (geometry.c:37-57) and V = normalize(eye.origin - pt) (raytrace.c:318-320) again */
    if constexpr (!kFast) {
        const float4 gE = tab[rec + RT_OBJ_SHAPE];
        if (is_sphere) {
            const float nx = px - gE.x, ny = py - gE.y, nz = pz - gE.z;
            Nx = nx; Ny = ny; Nz = nz;
            normalize3(Nx, Ny, Nz);
        } else {
            Nx = gE.x; Ny = gE.y; Nz = gE.z;
        }
        Vx = eox - px; Vy = eoy - py; Vz = eoz - pz;
        normalize3(Vx, Vy, Vz);
    }

    /*
     * The child rays are traced HERE, by the lane that spawns them (get_ray_color's closest-hit
     * search, raytrace.c:231-272, of the recursive calls at :427 and :539): a child that hits
     * nothing contributes the background colour at once (raytrace.c:446,509,553) and never touches
     * HBM - 71 % of the level-1 rays of the Whitted scene - and a node is made only when a child
     * did hit something.  The queues carry hits (point, object, the ray that found it), so the
     * consumer starts at the shading.
     */
    /* Both child rays are built first; the transmission ray, the node's own colour and its
     * coefficients then sit in LDS while the searches run (rows of RT_PARK_WORDS):
     *   0-6   transmission ray (o, d, magnitude) - over the eye ray k_bounce parked there, which
     *         is not needed once both rays exist
     *   7-16  the reflection child's queue entry (point, object, ray origin, ray direction)
     *   17-19 local colour (kr, kt are read again from the object's record afterwards)      */
    float *const park = lds_park + threadIdx.x;
    Ray cr; /* the child ray being traced: the reflection ray first */
    cr.ox = cr.oy = cr.oz = cr.dx = cr.dy = cr.dz = cr.mag = 0.0f;
    {
        Ray tr;
        tr.ox = tr.oy = tr.oz = tr.dx = tr.dy = tr.dz = tr.mag = 0.0f;
        if (want_t) { /* transmission ray raytrace.c:455-537 */
#ifdef RT_FAST
            const float inv_n = rcp_refined(mD.x);
#else
            const float inv_n = 1.0f / mD.x;
#endif
            float nit = inv_n;
            float edx, edy, edz;
            eye_src.direction(edx, edy, edz);
            const float Tx = edx * nit, Ty = edy * nit, Tz = edz * nit;
            const float cs = dot3(Vx, Vy, Vz, Nx, Ny, Nz);
            float Fx = Nx, Fy = Ny, Fz = Nz;
            if (cs < 0.0f) { /* inside the object moving out */
                nit = kFast ? mD.x : 1.0f / nit;
                Fx = Nx * -1.0f; Fy = Ny * -1.0f; Fz = Nz * -1.0f;
            }
            const float det = 1.0f + (nit * nit) * (cs * cs - 1.0f);
            tr.ox = px; tr.oy = py; tr.oz = pz;
            if (det < 0.0f) { /* total internal reflection :470-489 */
                const float t2x = 2.0f * dot3(Vx, Vy, Vz, Fx, Fy, Fz);
                tr.dx = Fx * t2x - Vx; tr.dy = Fy * t2x - Vy; tr.dz = Fz * t2x - Vz;
                tr.mag = 999999.0f;
            } else { /* :517-525: 1/n (not nit) times cos, minus a double sqrt */
#ifdef RT_FAST
                const float k = inv_n * cs - sqrt_rn(det);
#else
                const float k = (float)((double)(inv_n * cs) - sqrt((double)det));
#endif
                tr.dx = Tx + Fx * k; tr.dy = Ty + Fy * k; tr.dz = Tz + Fz * k;
                tr.mag = 99999.9f;
            }
            tinypush(tr);
        }
        if (want_r) { /* reflection ray raytrace.c:408-425 */
            const float t2x = 2.0f * dot3(Vx, Vy, Vz, Nx, Ny, Nz);
            cr.ox = px; cr.oy = py; cr.oz = pz;
            cr.dx = Nx * t2x - Vx; cr.dy = Ny * t2x - Vy; cr.dz = Nz * t2x - Vz;
            cr.mag = 9999999.9f;
            tinypush<true>(cr); /* |2 (V.N) N - V| = 1 for unit V, N */
        }
        park[0 * RT_BLOCK] = tr.ox; park[1 * RT_BLOCK] = tr.oy; park[2 * RT_BLOCK] = tr.oz;
        park[3 * RT_BLOCK] = tr.dx; park[4 * RT_BLOCK] = tr.dy; park[5 * RT_BLOCK] = tr.dz;
        park[6 * RT_BLOCK] = tr.mag;
        park[17 * RT_BLOCK] = outr; park[18 * RT_BLOCK] = outg; park[19 * RT_BLOCK] = outb;
    }
    pc.mark(8);
    bool hit_r = false, hit_t = false;
    float cd = 0.0f;
    uint32_t cobj = RT_NONE;
#pragma unroll 1
    for (uint32_t child = 0; child < 2u; ++child) {
        const bool want = child ? want_t : want_r;
        if (child != 0u) { /* the transmission ray, back from LDS (opaque index: no register forwarding) */
            uint32_t t = threadIdx.x;
            asm volatile("" : "+v"(t));
            const float *back = lds_park + t;
            cr.ox = back[0 * RT_BLOCK]; cr.oy = back[1 * RT_BLOCK]; cr.oz = back[2 * RT_BLOCK];
            cr.dx = back[3 * RT_BLOCK]; cr.dy = back[4 * RT_BLOCK]; cr.dz = back[5 * RT_BLOCK];
            cr.mag = back[6 * RT_BLOCK];
        }
        intersect_all<GRID, true, false, STATS, !TABLDS>(S, lds_poly, lds_sph, RT_DEV_CHILD(want), cr, RT_NONE, RT_NONE, cd, cobj, tl);
        const bool chit = want && (cobj != RT_NONE);
        if (child == 0u) {
            hit_r = chit;
            /* the child's hit: distance and object (the hit point is formed from the ray where it is needed,
             * hit_point() below, as the child's get_ray_color would form it: ray.c:121-122, 236-237) */
            park[7 * RT_BLOCK] = cd;
            park[10 * RT_BLOCK] = __uint_as_float(cobj);
            park[11 * RT_BLOCK] = cr.ox; park[12 * RT_BLOCK] = cr.oy; park[13 * RT_BLOCK] = cr.oz;
            park[14 * RT_BLOCK] = cr.dx; park[15 * RT_BLOCK] = cr.dy; park[16 * RT_BLOCK] = cr.dz;
        } else {
            hit_t = chit;
        }
    }
    pc.mark(9);
    if (!hit)
        return false;
    float kr2, kt2;
    {
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t)); /* opaque index: the compiler must not keep the parked values in registers */
        const float *back = lds_park + t;
        outr = back[17 * RT_BLOCK]; outg = back[18 * RT_BLOCK]; outb = back[19 * RT_BLOCK];
        uint32_t rec2 = rec; /* kr, kt: read again from the object's record rather than parked */
        asm volatile("" : "+v"(rec2));
        const float4 mC2 = tab[rec2 + RT_OBJ_PHONG];
        kr2 = mC2.z; kt2 = mC2.w;
    }
    if (!(hit_r || hit_t)) {
        /* every child (if any) left the scene: the node is finished.  Same expression order as
         * fold_values: local + kr*child, then + kt*child, then the clamp (raytrace.c:440-442,547-549,562) */
        if (want_r) { outr += kr2 * kc[KC_BG + 0]; outg += kr2 * kc[KC_BG + 1]; outb += kr2 * kc[KC_BG + 2]; }
        if (want_t) { outr += kt2 * kc[KC_BG + 0]; outg += kt2 * kc[KC_BG + 1]; outb += kt2 * kc[KC_BG + 2]; }
        leaf[0] = clamp1(outr);
        leaf[1] = clamp1(outg);
        leaf[2] = clamp1(outb);
        return false;
    }

    pc.mark(4);
    if constexpr (DFS) { /* the caller goes on with the children itself: no node, no queue entries */
        dfs->r = outr; dfs->g = outg; dfs->b = outb;
        dfs->kr = kr2; dfs->kt = kt2;
        dfs->want_r = want_r; dfs->want_t = want_t;
        dfs->hit_r = hit_r; dfs->hit_t = hit_t;
        dfs->t_ray = cr;
        dfs->t_dist = cd;
        dfs->t_obj = cobj;
        return true;
    }
    /* tree node + queue slots for the children that hit: one reservation per wave */
    uint32_t *cnt = Wk.counters;
    uint32_t local, slot_r, slot_t;
    wave_reserve(level_line(cnt, level, shard), hit_r, hit_t, node_room, Wk.capS, &cnt[CNT_OVERFLOW],
                 local, slot_r, slot_t);
    pc.mark(5);
    if (local == RT_NONE) { /* overflow: the batch will be re-rendered in smaller pieces */
        leaf[0] = leaf[1] = leaf[2] = 0.0f;
        return false;
    }
    const uint32_t node = node_base + local;
    node_idx = node;
    Wk.nodeA()[node] = make_float4(outr, outg, outb, kr2);
    Wk.nodeK()[node] = make_float2(kt2, __uint_as_float(link | (want_r ? 2u : 0u) | (want_t ? 4u : 0u)));
    const float4 bg4 = make_float4(kc[KC_BG + 0], kc[KC_BG + 1], kc[KC_BG + 2], 0.0f);
    if (want_r && !hit_r)
        Wk.childR()[node] = bg4;
    if (want_t && !hit_t)
        Wk.childT()[node] = bg4;

    /* queue of the next level: picked with selects - indexing the kernel-argument pointer array
     * with a run-time value makes the compiler re-load the pointer from memory at every use */
    const uint32_t odd_next = (level + 1) & 1u;
    float4 *const outA = Wk.qA(odd_next);
    float4 *const outB = Wk.qB(odd_next);
    uint32_t *const outO = Wk.qObj(odd_next);
    const uint32_t qbase = shard * Wk.capS;
    if (slot_r != RT_NONE) { /* 36 bytes a hit: (ray origin, distance), (ray direction, parent), object */
        outA[qbase + slot_r] = make_float4(park[11 * RT_BLOCK], park[12 * RT_BLOCK], park[13 * RT_BLOCK],
                                           park[7 * RT_BLOCK]);
        outB[qbase + slot_r] = make_float4(park[14 * RT_BLOCK], park[15 * RT_BLOCK], park[16 * RT_BLOCK],
                                           __uint_as_float((node << 1) | 0u));
        outO[qbase + slot_r] = __float_as_uint(park[10 * RT_BLOCK]);
    }
    if (slot_t != RT_NONE) {
        outA[qbase + slot_t] = make_float4(cr.ox, cr.oy, cr.oz, cd);
        outB[qbase + slot_t] = make_float4(cr.dx, cr.dy, cr.dz, __uint_as_float((node << 1) | 1u));
        outO[qbase + slot_t] = cobj;
    }
    pc.mark(6);
    return true;
}

/* Camera constants of the sample loop, kept in LDS (every lane reads the same words: broadcast)
 * rather than in ~16 scalar registers for the whole kernel: the trace kernels run out of SGPRs and
 * spill them into VGPR lanes. */
enum { CAM_EYE = 0, CAM_U = 3, CAM_V = 6, CAM_SHIFTN = 9, CAM_SPPHALF = 12, CAM_SPPW = 13, CAM_WORDS = 16 };
__device__ __forceinline__ void stage_camera(const DevCamera &C, float *cam, Stagers st = Stagers::block())
{
    if (st.tid == 0) {
        for (int k = 0; k < 3; ++k) {
            cam[CAM_EYE + k] = C.eye[k];
            cam[CAM_U + k] = C.U[k];
            cam[CAM_V + k] = C.V[k];
            cam[CAM_SHIFTN + k] = C.shiftN[k];
        }
        cam[CAM_SPPHALF] = C.sppHalf;
        cam[CAM_SPPW] = C.sppW;
    }
}

/* get_pixel_color's ray of sub-sample (i,j) (raytrace.c:751-771): shift = N*vd + U*fu + V*fv;
 * target = eye + shift; ray_create(eye, target) */
__device__ __forceinline__ void primary_ray(const float *cam, float fu0, float fv0, uint32_t i,
                                            uint32_t j, Ray &r)
{
    const float fv = fv0 + (cam[CAM_SPPHALF] + (float)j * cam[CAM_SPPW]);
    const float fu = fu0 + (cam[CAM_SPPHALF] + (float)i * cam[CAM_SPPW]);
    const float sx = (cam[CAM_SHIFTN + 0] + cam[CAM_U + 0] * fu) + cam[CAM_V + 0] * fv;
    const float sy = (cam[CAM_SHIFTN + 1] + cam[CAM_U + 1] * fu) + cam[CAM_V + 1] * fv;
    const float sz = (cam[CAM_SHIFTN + 2] + cam[CAM_U + 2] * fu) + cam[CAM_V + 2] * fv;
    r.ox = cam[CAM_EYE + 0]; r.oy = cam[CAM_EYE + 1]; r.oz = cam[CAM_EYE + 2];
    if constexpr (kFast) { /* (eye + shift) - eye = shift up to rounding */
        r.dx = sx; r.dy = sy; r.dz = sz;
    } else {
        const float tx = r.ox + sx, ty = r.oy + sy, tz = r.oz + sz;
        r.dx = tx - r.ox; r.dy = ty - r.oy; r.dz = tz - r.oz;
    }
    float inv;
    length_and_reciprocal(r.dx, r.dy, r.dz, r.mag, inv);
    r.dx = r.dx * inv; r.dy = r.dy * inv; r.dz = r.dz * inv;
}

/*
 * What the samples of one pixel can hit, decided conservatively from the pixel's cone of directions.  The
 * samples' un-normalised directions lie within (|U| + |V|) * half a pixel of the pixel centre's
 * (raytrace.c:751-771), so the unit directions d lie within eta of the centre's dc.
 *   sphere  (ray.c:186-206) misses when B^2 - 4C < 0, i.e. when the centre's distance |e x d| to the line
 *           exceeds r; |e x d| >= |e x dc| - |e| eta, and a margin of 1e-4 |e|^2 in the squares covers the
 *           float error of either side (1e-6 relative).  Big scenes test the ball around their grid instead.
 *   polygon (ray.c:92-117) is rejected when w = -(n.o + F) / (n.d) < 0: numerator and denominator of
 *           opposite, exactly known signs - n.d keeps the sign of n.dc when |n.dc| > |n| (eta + 1e-4),
 *           and magnitudes far from underflow keep the quotient's sign.  When the signs agree and
 *           |n.dc| > 4 |n| eta, every sample hits the plane within
 *               rho = 1.5 * t_c * eta * (1 + |n| / |n.dc|)
 *           of the centre ray's hit (p(d) - p_c = s [delta - dc (n.delta) / (n.dc)], s <= 4/3 t_c), and the
 *           polygon's scaled edge equations G_i (edge_screen: G >= 0 deep inside, G <= -1 far outside,
 *           both with the reference's verdict guaranteed) move by at most gmax * rho, gmax = max_i |grad G_i|
 *           (host: rt_api.hip).  min G_i(p_c) - gmax rho >= 1e-3: every sample is a hit, accepted by the
 *           screening stage of the inside test; min G_i(p_c) + gmax rho <= -1 - 1e-3: every sample is rejected by it.
 * `false` / "unknown" just means "trace it".  Precision of this function itself is irrelevant (hardware
 * reciprocals and square roots): its margins are a thousand times their error.
 */
struct PixelView {
    bool spheres;     /* some sphere may be hit by some sample                                          */
    bool unsure;      /* some polygon may or may not be hit                                             */
    uint32_t inside;  /* polygon that EVERY sample hits, inside for sure (RT_NONE: none; two such: unsure) */
    bool lit;         /* (with `inside`) no shadow ray from any of those hits can meet a sphere: a sphere at
                         more than its radius + the footprint's radius from the line (centre hit, light) is
                         missed by every line (hit, light) - with the margin of the sphere test above        */
};
/* where classify_pixel reads the scene from: the block's LDS copy, or - before anything is staged - the kernel
 * arguments and the scene blob through the scalar cache (every address is the same in all lanes) */
template <bool TABLDS> struct LdsSource {
    const float *cam_;
    const float4 *poly_, *sph_;
    const Tables<TABLDS> &tab_;
    const float *kc_;
    __device__ __forceinline__ float kc(uint32_t i) const { return kc_[i]; }
    __device__ __forceinline__ float cam(uint32_t i) const { return cam_[i]; }
    __device__ __forceinline__ float4 poly(uint32_t i) const { return poly_[i]; }
    __device__ __forceinline__ float4 sph(uint32_t i) const { return sph_[i]; }
    __device__ __forceinline__ float4 tab(uint32_t i) const { return tab_[i]; }
};
struct ArgSource {
    const DevScene &S;
    const DevCamera &C;
    __device__ __forceinline__ float cam(uint32_t i) const
    {
        return i < CAM_U ? C.eye[i - CAM_EYE] : i < CAM_V ? C.U[i - CAM_U] : i < CAM_SHIFTN ? C.V[i - CAM_V]
               : i < CAM_SPPHALF ? C.shiftN[i - CAM_SHIFTN] : i == CAM_SPPHALF ? C.sppHalf : C.sppW;
    }
    __device__ __forceinline__ float kc(uint32_t i) const { return i < KC_AMBIENT ? S.bg[i - KC_BG] : S.ambient[i - KC_AMBIENT]; }
    __device__ __forceinline__ float4 poly(uint32_t i) const { return const_load4(S.poly + i); }
    __device__ __forceinline__ float4 sph(uint32_t i) const { return const_load4(S.sph + i); }
    __device__ __forceinline__ float4 tab(uint32_t i) const { return const_load4(S.tab + i); }
};
template <bool GRID, class Source>
__device__ __forceinline__ PixelView classify_pixel(const DevScene &S, const Source &src, float fu0,
                                                    float fv0, uint32_t sqrtSpp)
{
    PixelView pv;
    pv.spheres = true; pv.unsure = true; pv.inside = RT_NONE; pv.lit = false;
    const float half = 0.5f * (float)sqrtSpp * src.cam(CAM_SPPW);
    const float fu = fu0 + half, fv = fv0 + half;
    const float Dx = (src.cam(CAM_SHIFTN + 0) + src.cam(CAM_U + 0) * fu) + src.cam(CAM_V + 0) * fv;
    const float Dy = (src.cam(CAM_SHIFTN + 1) + src.cam(CAM_U + 1) * fu) + src.cam(CAM_V + 1) * fv;
    const float Dz = (src.cam(CAM_SHIFTN + 2) + src.cam(CAM_U + 2) * fu) + src.cam(CAM_V + 2) * fv;
    const float D2 = dot3(Dx, Dy, Dz, Dx, Dy, Dz);
    if (!(D2 > 1e-20f))
        return pv;
    const float invD = __builtin_amdgcn_rsqf(D2);
    const float dcx = Dx * invD, dcy = Dy * invD, dcz = Dz * invD;
    const float lenU = __builtin_amdgcn_sqrtf(dot3(src.cam(CAM_U + 0), src.cam(CAM_U + 1), src.cam(CAM_U + 2), src.cam(CAM_U + 0), src.cam(CAM_U + 1), src.cam(CAM_U + 2)));
    const float lenV = __builtin_amdgcn_sqrtf(dot3(src.cam(CAM_V + 0), src.cam(CAM_V + 1), src.cam(CAM_V + 2), src.cam(CAM_V + 0), src.cam(CAM_V + 1), src.cam(CAM_V + 2)));
    /* |d - dc| <= |D - Dc| / |Dc| for unit vectors made from D, Dc (times 1 / (1 - x/2) <= 1.13 for the x < 0.22 admitted below) */
    const float eta = 1.15f * ((lenU + lenV) * fabsf(half)) * invD + 1e-6f;
    if (!(eta < 0.25f))
        return pv;
    const float ox = src.cam(CAM_EYE + 0), oy = src.cam(CAM_EYE + 1), oz = src.cam(CAM_EYE + 2);
    bool spheres = false;
    if constexpr (GRID) {
        /* Big scenes: one test against the ball around the grid's box, which holds every sphere whole.  A
         * line that misses the ball by delta misses sphere k by delta as well, but for a small sphere
         * rho^2 - r^2 >= delta^2 is all that is left of the margin: delta >= 5e-3 (|e| + R) keeps it at
         * 2.5e-5 |e_k|^2, six times the float error of the reference's discriminant. */
        const GridHeader &G = *(const GridHeader *)S.grid;
        const float bx = G.gmax[0] - G.gmin[0], by = G.gmax[1] - G.gmin[1], bz = G.gmax[2] - G.gmin[2];
        const float R = 0.5f * __builtin_amdgcn_sqrtf(dot3(bx, by, bz, bx, by, bz)) * 1.001f;
        const float ex = ox - G.centre[0], ey = oy - G.centre[1], ez = oz - G.centre[2];
        const float e2 = dot3(ex, ey, ez, ex, ey, ez);
        const float t = dot3(ex, ey, ez, dcx, dcy, dcz);
        const float E = __builtin_amdgcn_sqrtf(e2);
        const float m = __builtin_amdgcn_sqrtf(fmaxf(e2 - t * t, 0.0f)) - E * eta;
        spheres = !(m > R + 5e-3f * (E + R));
    }
    for (uint32_t k = 0; k < (GRID ? 0u : S.nSph); ++k) {
        const float4 sp = src.sph(k); /* (centre, r^2) */
        const float ex = ox - sp.x, ey = oy - sp.y, ez = oz - sp.z;
        const float e2 = dot3(ex, ey, ez, ex, ey, ez);
        const float t = dot3(ex, ey, ez, dcx, dcy, dcz);
        const float rho = __builtin_amdgcn_sqrtf(fmaxf(e2 - t * t, 0.0f));
        const float m = rho - __builtin_amdgcn_sqrtf(e2) * eta * 1.0001f;
        spheres = spheres || !(m > 0.0f && m * m > sp.w + 1e-4f * e2 + 1e-30f);
    }
    bool unsure = false, lit = false;
    uint32_t inside = RT_NONE;
    for (uint32_t k = 0; k < S.nPoly; ++k) {
        const float4 pl = src.poly(k); /* (A, B, C, F) */
        const float num = -1.0f * (dot3(pl.x, pl.y, pl.z, ox, oy, oz) + pl.w);
        const float q = dot3(pl.x, pl.y, pl.z, dcx, dcy, dcz);
        const float nlen = __builtin_amdgcn_sqrtf(dot3(pl.x, pl.y, pl.z, pl.x, pl.y, pl.z)) * 1.0001f;
        const bool sane = fabsf(q) < 1e10f && fabsf(num) > 1e-20f && fabsf(num) < 1e10f;
        const bool opposite = (num > 0.0f) != (q > 0.0f);
        if (sane && opposite && fabsf(q) > nlen * (eta + 1e-4f))
            continue; /* behind the eye for every sample */
        int verdict = 0; /* +1: every sample inside, -1: every sample outside */
        if (sane && !opposite && fabsf(q) > nlen * (4.0f * eta + 1e-4f)) {
            const float4 info = src.poly(S.nPoly + k);
            const uint32_t nv = __float_as_uint(info.y);
            const uint32_t edg = 2u * S.nPoly + __float_as_uint(info.x) + ((S.nPolySlots - 2u * S.nPoly) >> 1); /* first edge equation */
            const float tc = num * __builtin_amdgcn_rcpf(q);
            const float pcx = ox + dcx * tc, pcy = oy + dcy * tc, pcz = oz + dcz * tc;
            float gmin = 3.0e38f;
            for (uint32_t i = 0; i < nv; ++i)
                gmin = fminf(gmin, edge_value(src.poly(edg + i), pcx, pcy, pcz));
            const float rho = 1.5f * tc * eta * (1.0f + nlen * __builtin_amdgcn_rcpf(fabsf(q)));
            const float move = info.w * rho + 1e-3f; /* info.w = gmax */
            verdict = (gmin - move >= 0.0f) ? 1 : ((gmin + move <= -1.0f) ? -1 : 0);
            if (!GRID && verdict > 0 && inside == RT_NONE) { /* shadow rays of this polygon's hits */
                lit = true;
                for (uint32_t l = 0; l < S.nLights; ++l) {
                    const float4 lp = src.tab(2u * l);
                    const float lx = lp.x - pcx, ly = lp.y - pcy, lz = lp.z - pcz;
                    const float il = __builtin_amdgcn_rsqf(dot3(lx, ly, lz, lx, ly, lz));
                    const float ux = lx * il, uy = ly * il, uz = lz * il;
                    for (uint32_t j = 0; j < S.nSph; ++j) {
                        const float4 sp = src.sph(j);
                        const float ex = sp.x - pcx, ey = sp.y - pcy, ez = sp.z - pcz;
                        const float e2 = dot3(ex, ey, ez, ex, ey, ez);
                        const float t = dot3(ex, ey, ez, ux, uy, uz);
                        const float m = __builtin_amdgcn_sqrtf(fmaxf(e2 - t * t, 0.0f)) - (1.01f * rho + 2e-3f);
                        lit = lit && (m > 0.0f && m * m > sp.w + 1e-4f * e2 + 1e-30f);
                    }
                }
            }
        }
        if (verdict < 0)
            continue;
        if (verdict > 0 && inside == RT_NONE)
            inside = k;
        else
            unsure = true;
    }
    pv.spheres = spheres; pv.unsure = unsure; pv.inside = inside; pv.lit = lit;
    return pv;
}

/* Where shade_and_spawn gets the eye ray back from after the shadow rays (see there).  The empty
 * asm makes the input opaque, so the compiler cannot fold the rebuild / re-read into the first
 * one and keep the values in registers across the any-hit searches after all. */
struct PrimaryEye {
    const float *cam;
    float fu0, fv0;
    uint32_t i, j;
    __device__ __forceinline__ void origin(float &x, float &y, float &z) const
    {
        x = cam[CAM_EYE + 0]; y = cam[CAM_EYE + 1]; z = cam[CAM_EYE + 2];
    }
    __device__ __forceinline__ void direction(float &x, float &y, float &z) const
    {
        float f = fu0;
        asm volatile("" : "+v"(f));
        Ray r;
        primary_ray(cam, f, fv0, i, j, r);
        x = r.dx; y = r.dy; z = r.dz;
    }
};
/* k_bounce parks the ray in LDS (one column per thread, SoA rows: conflict-free) rather than
 * re-reading the queue: on gfx950 a global load issued after the previous iteration's stores waits
 * for their write acknowledgements as well (one in-order counter), an LDS read does not. */
struct StashedEye {
    const float *lds_eye;
    __device__ __forceinline__ static void park(float *lds_eye, const Ray &r)
    {
        float *c = lds_eye + threadIdx.x;
        c[0 * RT_BLOCK] = r.ox; c[1 * RT_BLOCK] = r.oy; c[2 * RT_BLOCK] = r.oz;
        c[3 * RT_BLOCK] = r.dx; c[4 * RT_BLOCK] = r.dy; c[5 * RT_BLOCK] = r.dz;
    }
    __device__ __forceinline__ void origin(float &x, float &y, float &z) const
    {
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t)); /* opaque index (the pointer stays an LDS pointer): no forwarding
                                       of the parked registers */
        x = lds_eye[t + 0 * RT_BLOCK]; y = lds_eye[t + 1 * RT_BLOCK]; z = lds_eye[t + 2 * RT_BLOCK];
    }
    __device__ __forceinline__ void direction(float &x, float &y, float &z) const
    {
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));
        x = lds_eye[t + 3 * RT_BLOCK]; y = lds_eye[t + 4 * RT_BLOCK]; z = lds_eye[t + 5 * RT_BLOCK];
    }
};

#ifdef RT_FAST
/*
 * The sample loop of a tile whose every sample hits ONE polygon, inside, sees every light from there and
 * spawns nothing (classify_pixel: tile_polygon + tile_lit; the reference scene's floor outside the spheres'
 * shadows, a third of its tiles) - the tolerance build only.  The object, its material, the plane, the
 * normal and the lights are the same for all lanes and all samples: they are read once, into scalar
 * registers, and what is left per sample is straight-line vector code with the expressions of the general
 * path (hit point from the un-normalised direction, get_shade_color_phong raytrace.c:282-401 with every
 * shadow ray known to be free, get_object_color object3d.c:64-117, clamp :562) - no search, no exec-mask
 * bookkeeping, no LDS traffic.  Returns the sum of the samples' colours in the reference's order.
 */
template <class Source>
__device__ __forceinline__ void lit_polygon_tile(const DevScene &S, const DevCamera &C, const Source &src,
                                                 uint32_t polygon, uint32_t rec, float fu0, float fv0,
                                                 float &accr, float &accg, float &accb)
{
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const float4 pl4 = src.poly(polygon);
    const float plx = uni(pl4.x), ply = uni(pl4.y), plz = uni(pl4.z), plw = uni(pl4.w);
    const float ox = uni(src.cam(CAM_EYE + 0)), oy = uni(src.cam(CAM_EYE + 1)), oz = uni(src.cam(CAM_EYE + 2));
    const float num = -1.0f * (dot3(plx, ply, plz, ox, oy, oz) + plw);
    const float4 mA = src.tab(rec + RT_OBJ_DIFFUSE), mB = src.tab(rec + RT_OBJ_SPECULAR), mC = src.tab(rec + RT_OBJ_PHONG);
    const float4 gE = src.tab(rec + RT_OBJ_SHAPE);
    const float ka = uni(mA.w), kd = uni(mC.x), ke = uni(mC.y), ks = uni(mB.w);
    const float sr = uni(mB.x), sg = uni(mB.y), sb = uni(mB.z);
    const float Nx = uni(gE.x), Ny = uni(gE.y), Nz = uni(gE.z);
    const float ar = uni(src.kc(KC_AMBIENT + 0)), ag = uni(src.kc(KC_AMBIENT + 1)), ab = uni(src.kc(KC_AMBIENT + 2));
    const float Ux = uni(src.cam(CAM_U + 0)), Uy = uni(src.cam(CAM_U + 1)), Uz = uni(src.cam(CAM_U + 2));
    const float Vx_ = uni(src.cam(CAM_V + 0)), Vy_ = uni(src.cam(CAM_V + 1)), Vz_ = uni(src.cam(CAM_V + 2));
    const float Sx = uni(src.cam(CAM_SHIFTN + 0)), Sy = uni(src.cam(CAM_SHIFTN + 1)), Sz = uni(src.cam(CAM_SHIFTN + 2));
    const float spph = uni(src.cam(CAM_SPPHALF)), sppw = uni(src.cam(CAM_SPPW));
    for (uint32_t j = 0; j < C.sqrtSpp; ++j) {
        const float fv = fv0 + (spph + (float)j * sppw);
        for (uint32_t i = 0; i < C.sqrtSpp; ++i) {
            const float fu = fu0 + (spph + (float)i * sppw);
            const float Dx = (Sx + Ux * fu) + Vx_ * fv;
            const float Dy = (Sy + Uy * fu) + Vy_ * fv;
            const float Dz = (Sz + Uz * fu) + Vz_ * fv;
            const float t = num * rcp_refined(dot3(plx, ply, plz, Dx, Dy, Dz));
            const float px = ox + Dx * t, py = oy + Dy * t, pz = oz + Dz * t;
            float Vx = ox - px, Vy = oy - py, Vz = oz - pz; /* raytrace.c:318-320 */
            normalize3(Vx, Vy, Vz);
            float dfr = 0.f, dfg = 0.f, dfb = 0.f, spr = 0.f, spg = 0.f, spb = 0.f;
            for (uint32_t l = 0; l < S.nLights; ++l) { /* raytrace.c:349-384, every light visible */
                const float4 lp = src.tab(2u * l);
                const float4 lc = src.tab(2u * l + 1u);
                float sx = lp.x - px, sy = lp.y - py, sz = lp.z - pz;
                float len, inv;
                length_and_reciprocal(sx, sy, sz, len, inv);
                sx = sx * inv; sy = sy * inv; sz = sz * inv;
                const float c = dot3(sx, sy, sz, Nx, Ny, Nz);
                const bool dark = c <= 0.0f;
                const float cp = dark ? 0.0f : c;
                dfr += lc.x * cp; dfg += lc.y * cp; dfb += lc.z * cp;
                const float c2 = 2.0f * c;
                const float Rx = Nx * c2 - sx, Ry = Ny * c2 - sy, Rz = Nz * c2 - sz;
                float h = dot3(Rx, Ry, Rz, Vx, Vy, Vz);
                if (h < 0.0f)
                    h = 0.0f;
                h = dark ? 0.0f : phong_pow(h, ke);
                spr += lc.x * h; spg += lc.y * h; spb += lc.z * h;
            }
            /* get_object_color object3d.c:64-117: the checkerboard every polygon gets */
            const float u = div_with(px + 15.0f, 22.0f, 1.0f / 22.0f);
            const float v = div_with(pz + 100.0f, 100.0f, 1.0f / 100.0f);
            const int xt = (int)(11.0f * u);
            const int yt = (int)(50.0f * v);
            const float ocg = ((xt % 2) == (yt % 2)) ? 0.0f : 1.0f; /* (red 1, blue 0) */
            float outr = ar * ka, outg = (ocg * ag) * ka, outb = (0.0f * ab) * ka; /* raytrace.c:303-307 */
            outr += kd * dfr + (ks * spr) * sr; /* raytrace.c:390-401 */
            outg += (kd * dfg) * ocg + (ks * spg) * sg;
            outb += (kd * dfb) * 0.0f + (ks * spb) * sb;
            accr = accr + clamp1(outr); /* raytrace.c:562, 774 */
            accg = accg + clamp1(outg);
            accb = accb + clamp1(outb);
        }
    }
}
#endif

template <bool STATS>
__device__ __forceinline__ void flush_tally(uint64_t *st, Tally<STATS> &tl, bool k0)
{
    if constexpr (STATS) {
        wave_add_stat(st, ST_SPH_TESTS, tl.n_sph_test);
        wave_add_stat(st, ST_SPH_HITS, tl.n_sph_hit);
        wave_add_stat(st, ST_POLY_TESTS, tl.n_poly_test);
        wave_add_stat(st, ST_POLY_PAR, tl.n_poly_par);
        wave_add_stat(st, ST_POLY_BEHIND, tl.n_poly_behind);
        wave_add_stat(st, ST_POLY_FULL, tl.n_poly_full);
        wave_add_stat(st, ST_POLY_HITS, tl.n_poly_hit);
        wave_add_stat(st, ST_LIT, tl.n_lit);
        wave_add_stat(st, ST_NODES_SPH, tl.n_node_sph);
        wave_add_stat(st, ST_NODES_POLY, tl.n_node_poly);
        wave_add_stat(st, ST_POLY_SCREEN_MISMATCH, tl.n_mismatch);
        wave_add_stat(st, ST_POLY_SCREEN_UNSURE, tl.n_unsure);
        if (k0) {
            wave_add_stat(st, ST_K0_SPH_TESTS, tl.n_sph_test);
            wave_add_stat(st, ST_K0_SPH_HITS, tl.n_sph_hit);
            wave_add_stat(st, ST_K0_POLY_TESTS, tl.n_poly_test);
            wave_add_stat(st, ST_K0_POLY_PAR, tl.n_poly_par);
            wave_add_stat(st, ST_K0_POLY_BEHIND, tl.n_poly_behind);
            wave_add_stat(st, ST_K0_POLY_FULL, tl.n_poly_full);
            wave_add_stat(st, ST_K0_LIT, tl.n_lit);
            wave_add_stat(st, ST_K0_NODES_SPH, tl.n_node_sph);
            wave_add_stat(st, ST_K0_NODES_POLY, tl.n_node_poly);
        }
    }
}

/*
 * Scene shapes the trace kernels are also compiled for with every count a compile-time constant:
 * the sphere / polygon / light loops unroll, their branches and loop counters go away and the LDS
 * offsets fold (measured on the reference's scene: k_primary 0.348 -> 0.283 ms; the loops run 2, 1
 * and 1 times there, so their bookkeeping was a fifth of the kernel).  SHAPE 0 is any scene.
 *   SHAPE 1 = 2 spheres, 1 quad (10 polygon slots), 1 light, tables in LDS: the scene the
 *             reference's parse_scene always returns (scene.c:86-258), i.e. every run of the reference.
 * rt_scene_shape() is the host-side match; the kernels overwrite their by-value copy of the counts.
 */
template <int SHAPE> __device__ __forceinline__ void fix_shape(DevScene &) {}
template <> __device__ __forceinline__ void fix_shape<1>(DevScene &S)
{
    S.nSph = 2u; S.nPoly = 1u; S.nPolySlots = 10u; S.nObj = 3u; S.nLights = 1u; S.tabSlots = 17u;
}
#ifdef RT_JIT
/* SHAPE 2 = the shape this translation unit is being compiled for at run time (rt_jit.cpp: hiprtc at
 * rt_hip_ctx_specialize, counts as macros): any scene of the small class gets what SHAPE 1 gives the reference's */
template <> __device__ __forceinline__ void fix_shape<2>(DevScene &S)
{
    S.nSph = RT_JIT_NSPH; S.nPoly = RT_JIT_NPOLY; S.nPolySlots = RT_JIT_NPOLYSLOTS; S.nObj = RT_JIT_NOBJ;
    S.nLights = RT_JIT_NLIGHTS; S.tabSlots = RT_JIT_TABSLOTS;
}
#endif

/* a pixel's sum of sample colours -> the frame: x 1/spp^2 (raytrace.c:779), radiance, 8-bit pack (raytrace.c:783-790).
 * p: pixel of the batch.  k_resolve's last step, and k_primary's for the tiles without roots. */
__device__ __forceinline__ uint32_t store_pixel(const DevCamera &C, const DevRows &R, size_t p, float r, float g, float b,
                                                uint32_t *argb, float *radiance)
{
    r = r * C.scale; g = g * C.scale; b = b * C.scale;
    const size_t o = (size_t)R.out_first * C.W + p;
    if (radiance) {
        radiance[3 * o + 0] = r; radiance[3 * o + 1] = g; radiance[3 * o + 2] = b;
    }
    /* (unsigned char)(255 * c): truncation, low byte of the int conversion */
    const uint32_t R8 = (uint32_t)(int)(255.0f * r) & 0xFFu;
    const uint32_t G8 = (uint32_t)(int)(255.0f * g) & 0xFFu;
    const uint32_t B8 = (uint32_t)(int)(255.0f * b) & 0xFFu;
    const uint32_t packed = 0xFF000000u | (R8 << 16) | (G8 << 8) | B8;
    argb[o] = packed;
    return packed;
}

/* ------------------------------------------------------------------------------------------ */
/* k_primary: get_pixel_color + get_ray_color + depth-0 get_shade_color_phong, fused           */
/* ------------------------------------------------------------------------------------------ */
template <bool GRID, bool STATS, bool TABLDS, int SHAPE, bool BLINN = false>
__global__ __launch_bounds__(RT_BLOCK, (SHAPE != 0 ? RT_WPE_SHAPE1 : TABLDS ? 6 : RT_WPE)) void k_primary(DevScene S, DevCamera C, DevWork Wk,
                                                      DevRows R)
{
    fix_shape<SHAPE>(S);
    RT_DEV_TILE_START();
    /* a wave owns an 8x8 pixel tile; the block is 4 tiles side by side (32x8 pixels) */
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    /* Which block of the batch this is: the raster top-down, or - once the context has timed a frame of this
     * shape - the launch plan's order, slowest blocks first (rt_api.hip: the launch is as long as the waves
     * that start last and run longest; the raster started the floor's edge and shadow tiles at 80 us of 122). */
    uint32_t bx = blockIdx.x, ty = blockIdx.y;
    const uint32_t nbx = (C.W + 31u) / 32u;
    if (R.plan) { /* (a 1-D grid of plan_len blocks) */
        const uint32_t e = const_load1(R.plan + blockIdx.x);
        bx = e & 0x7fffu;
        ty = (e >> 15) & 0x7fffu;
    }
    const unsigned long long wave_t0 = R.wave_ticks ? wall_clock64() : 0ull;
    auto wave_done = [&]() { /* (every exit of the kernel) */
        if (R.wave_ticks && lane == 0u)
            R.wave_ticks[(ty * nbx + bx) * (RT_BLOCK / 64u) + wv] = (uint32_t)(wall_clock64() - wave_t0);
    };
    /* the wave's tile has no roots: its pixels are final (wave-uniform; DevRows::tile_done) */
    auto tile_is_final = [&](bool final) {
        if (R.tile_done && lane == 0u)
            R.tile_done[(ty * nbx + bx) * (RT_BLOCK / 64u) + wv] = final ? 1u : 0u;
        return final && R.tile_done != nullptr;
    };
    const uint32_t x = bx * 32u + wv * 8u + (lane & 7u);
    const uint32_t y = R.y_first + ty * R.y_step + (lane >> 3);
    const bool valid = (x < C.W) && (y < R.y_limit);

    /* raytrace.c:755-763: the pixel-dependent part of the U and V shifts */
    const float fu0 = (((float)x - C.hfbW) / C.hfbW) * C.halfW;
    const float fv0 = ((C.hfbH - (float)y) / C.hfbH) * C.halfH;

    /* every wave feeds its own shard: even a single tile row spreads over all of them */
    const uint32_t shard = tile_shard((ty * nbx + bx) * (RT_BLOCK / 64u) +
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
    const uint32_t node_base = shard * Wk.capS; /* level 0 opens every shard's node range */

    float accr = 0.0f, accg = 0.0f, accb = 0.0f; /* color_init raytrace.c:908 */
    uint32_t nroots = 0; /* samples of this pixel that became tree roots */
    const size_t pix = (size_t)(ty * 8u + (lane >> 3)) * C.W + x; /* pixel of the batch */
    const size_t pix_stride = (size_t)R.ntiles * 8u * C.W;
    SpawnCounts sc;
    Tally<STATS> tl;
    PhaseClock pc; /* (development builds: cycles per phase of a sample, tools/phase_probe_k0.py) */

    /*
     * What the wave's tile needs, decided from the cones of directions its pixels' samples span
     * (classify_pixel; the instrumented kernels trace everything: they count the reference's tests).
     * A tile none of whose pixels can see anything (sky: 59 % of the tiles of the reference's scene) is
     * finished without tracing: spp^2 times the background colour, added in the reference's order; the
     * tolerance build also finishes the tiles of one lit polygon here (lit_polygon_tile).  Returns true when
     * the wave is done.
     */
    bool tile_no_spheres = false;    /* wave-uniform: no sample of the tile can meet a sphere */
    uint32_t tile_polygon = RT_NONE; /* wave-uniform: every sample of the tile hits this polygon and nothing else */
    bool tile_lit = false;           /* wave-uniform: ... and no shadow ray from those hits can meet anything   */
    auto classify_tile = [&](const auto &src) -> bool {
        const PixelView pv = classify_pixel<GRID>(S, src, fu0, fv0, C.sqrtSpp);
        const bool nothing = !valid || (!pv.spheres && !pv.unsure && pv.inside == RT_NONE);
        tile_no_spheres = __all(!valid || !pv.spheres);
        if constexpr (!BLINN) {
            /* (the exact build uses tile_polygon for tile_lit alone: only the tolerance build forms the hit
             * point without normalising the direction) */
            const unsigned long long live = __ballot(valid);
            const uint32_t first = live ? (uint32_t)__builtin_amdgcn_readlane((int)pv.inside, __ffsll((long long)live) - 1) : RT_NONE;
            if (tile_no_spheres && first != RT_NONE && __all(!valid || (pv.inside == first && !pv.unsure))) {
                tile_polygon = first;
                tile_lit = S.nPoly == 1u && S.nLights <= 32u && __all(!valid || pv.lit);
            }
        }
        uint64_t *const stats = (uint64_t *)(Wk.counters + CNT_STATS + shard * CNT_STATS_STRIDE);
        if (__all(nothing)) { /* wave-uniform */
            for (uint32_t k = 0; k < C.sqrtSpp * C.sqrtSpp; ++k) {
                accr = accr + src.kc(KC_BG + 0);
                accg = accg + src.kc(KC_BG + 1);
                accb = accb + src.kc(KC_BG + 2);
            }
            if (tile_is_final(true)) {
                if (valid)
                    store_pixel(C, R, pix, accr, accg, accb, R.argb, R.radiance);
            } else if (valid) {
                Wk.pixAcc[pix] = make_float4(accr, accg, accb, __uint_as_float(0u));
            }
            wave_add_total(stats, ST_K0_CULLED, (uint32_t)__popcll(__ballot(valid)) * C.sqrtSpp * C.sqrtSpp);
            RT_DEV_TILE_END(stats, 0);
            wave_done();
            return true;
        }
#ifdef RT_FAST
        if constexpr (!BLINN) {
            if (RT_DEV_LITTILE(tile_polygon != RT_NONE && tile_lit)) { /* wave-uniform */
                const uint32_t pobj = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(src.poly(S.nPoly + tile_polygon).z));
                const uint32_t prec = 2u * S.nLights + pobj * (uint32_t)RT_OBJ_FIELDS;
                const float4 mC = src.tab(prec + RT_OBJ_PHONG);
                const bool spawns = __builtin_amdgcn_readfirstlane((int)((mC.z != 0.0f || mC.w != 0.0f) && C.depth != 0u)) != 0;
                if (!spawns) { /* raytrace.c:405,452 */
                    lit_polygon_tile(S, C, src, tile_polygon, prec, fu0, fv0, accr, accg, accb);
                    if (tile_is_final(true)) {
                        if (valid)
                            store_pixel(C, R, pix, accr, accg, accb, R.argb, R.radiance);
                    } else if (valid) {
                        Wk.pixAcc[pix] = make_float4(accr, accg, accb, __uint_as_float(0u));
                    }
                    wave_add_total(stats, ST_K0_SHADE_NODES, (uint32_t)__popcll(__ballot(valid)) * C.sqrtSpp * C.sqrtSpp);
                    RT_DEV_TILE_END(stats, 1);
                    wave_done();
                    return true;
                }
            }
        }
#endif
        return false;
    };
    /* The kernel of the reference's scene shape classifies BEFORE anything is staged, from the kernel
     * arguments and the scene blob through the scalar cache: the waves of the tiles finished above - two
     * thirds of them - end without a global load waited for, an LDS word written or a barrier (a barrier
     * waits for the waves of its block that are still alive only).  What is left of the block then stages
     * wave by wave (Stagers). */
    constexpr bool EARLY = SHAPE != 0 && !STATS;
    if constexpr (EARLY) {
        if (classify_tile(ArgSource{S, C}))
            return;
    }
    const Stagers stagers = EARLY ? Stagers::wave() : Stagers::block();

    extern __shared__ __attribute__((aligned(16))) float4 lds[];
    float *lds_park = (float *)lds; /* RT_PARK_FLOATS, ahead of the scene data */
    float4 *lds_poly = lds + RT_PARK_FLOATS / 4u;
    float4 *lds_sph = lds_poly + S.nPolySlots;
    uint32_t *lds_vis = (uint32_t *)(lds_sph + sphere_slots(S.nSph));
    const Tables<TABLDS> tab = stage_tables<TABLDS>(S, lds_vis, stagers);
    __shared__ float cam[CAM_WORDS];
    __shared__ float kc[KC_WORDS];
    stage_camera(C, cam, stagers);
    stage_colours(S, kc, stagers);
    block_copy(lds_poly, S.poly, S.nPolySlots, stagers);
    if (!GRID) {
        block_copy(lds_sph, S.sph, S.nSph, stagers);
        for (uint32_t k = stagers.tid; k < S.nSph; k += stagers.stride)
            ((uint32_t *)(lds_sph + S.nSph))[k] = S.sph_obj[k];
    }
    __syncthreads();
    float4 *eye_terms = nullptr;
    if constexpr (TABLDS) { /* small scenes: behind the shading tables */
        eye_terms = (float4 *)(lds_vis + vis_words(S.nLights)) + S.tabSlots;
        stage_eye_terms(S, C, lds_poly, lds_sph, eye_terms, stagers);
        __syncthreads();
    }
    if constexpr (!EARLY && !STATS) {
        if (classify_tile(LdsSource<TABLDS>{cam, lds_poly, lds_sph, tab, kc}))
            return; /* wave-uniform; no barrier follows */
    }

    for (uint32_t j = 0; j < C.sqrtSpp; ++j) {
        for (uint32_t i = 0; i < C.sqrtSpp; ++i) {
            float px, py, pz;
            uint32_t obj;
            pc.start();
            if (kFast && tile_polygon != RT_NONE) {
                /* every sample of this tile hits polygon tile_polygon, inside, and nothing else: the hit
                 * point is eye + D * (num / n.D) with the un-normalised direction D (raytrace.c:751-768,
                 * ray.c:92-122 without the detour over the unit vector) */
                const float fv = fv0 + (cam[CAM_SPPHALF] + (float)j * cam[CAM_SPPW]);
                const float fu = fu0 + (cam[CAM_SPPHALF] + (float)i * cam[CAM_SPPW]);
                const float Dx = (cam[CAM_SHIFTN + 0] + cam[CAM_U + 0] * fu) + cam[CAM_V + 0] * fv;
                const float Dy = (cam[CAM_SHIFTN + 1] + cam[CAM_U + 1] * fu) + cam[CAM_V + 1] * fv;
                const float Dz = (cam[CAM_SHIFTN + 2] + cam[CAM_U + 2] * fu) + cam[CAM_V + 2] * fv;
                const float4 pl = lds_poly[tile_polygon];
                const float ox = cam[CAM_EYE + 0], oy = cam[CAM_EYE + 1], oz = cam[CAM_EYE + 2];
                const float num = -1.0f * (dot3(pl.x, pl.y, pl.z, ox, oy, oz) + pl.w);
                const float t = num * rcp_refined(dot3(pl.x, pl.y, pl.z, Dx, Dy, Dz));
                px = ox + Dx * t; py = oy + Dy * t; pz = oz + Dz * t;
                obj = __float_as_uint(lds_poly[S.nPoly + tile_polygon].z);
            } else {
                Ray r;
                primary_ray(cam, fu0, fv0, i, j, r);
                float d;
                intersect_all<GRID, false, false, STATS, !TABLDS, TABLDS>(S, lds_poly, lds_sph, valid, r, RT_NONE,
                                                                  RT_NONE, d, obj, tl, eye_terms, tile_no_spheres);
                px = r.ox + r.dx * d; py = r.oy + r.dy * d; pz = r.oz + r.dz * d;
            }
            pc.mark(0);
            const bool hit = valid && (obj != RT_NONE);
            float leaf[3] = {kc[KC_BG + 0], kc[KC_BG + 1], kc[KC_BG + 2]}; /* miss: raytrace.c:700-703 */
            uint32_t node;
            bool sr, st_;
            const PrimaryEye eye_src{cam, fu0, fv0, i, j};
            bool deferred = shade_and_spawn<GRID, STATS, TABLDS, BLINN>(S, C, Wk, lds_poly, lds_sph, lds_vis, lds_park, kc, tab, hit,
                                                          eye_src, px, py, pz, obj, 0u, RT_LINK_NONE << 3, shard,
                                                          node_base, Wk.capS, leaf, node, sr, st_, tl, pc, tile_lit);
            sc.add(hit, sr, st_);
            /* The reference adds a pixel's samples in order (raytrace.c:774) and float addition does
             * not commute: a finished sample that FOLLOWS a root of its pixel must be added after
             * that root, so it becomes a childless root itself (silhouette pixels only: rare). */
            if (valid && !deferred && nroots != 0u) {
                uint32_t local, no_r, no_t;
                wave_reserve(level_line(Wk.counters, 0u, shard), false, false, Wk.capS, Wk.capS,
                             &Wk.counters[CNT_OVERFLOW], local, no_r, no_t);
                if (local != RT_NONE) {
                    node = node_base + local;
                    Wk.nodeA()[node] = make_float4(leaf[0], leaf[1], leaf[2], 0.0f);
                    Wk.nodeK()[node] = make_float2(0.0f, __uint_as_float(RT_LINK_NONE << 3));
                    deferred = true;
                }
            }
            if (deferred) {
                Wk.roots[(size_t)nroots * pix_stride + pix] = node; /* resolved by k_resolve */
                ++nroots;
            } else {
                accr = accr + leaf[0]; /* color_add raytrace.c:774 */
                accg = accg + leaf[1];
                accb = accb + leaf[2];
            }
            pc.mark(7);
        }
    }
    if (tile_is_final(__all(!valid || nroots == 0u))) {
        if (valid)
            store_pixel(C, R, pix, accr, accg, accb, R.argb, R.radiance);
    } else if (valid) {
        Wk.pixAcc[pix] = make_float4(accr, accg, accb, __uint_as_float(nroots));
    }

    uint64_t *st = (uint64_t *)(Wk.counters + CNT_STATS + shard * CNT_STATS_STRIDE);
#ifndef RT_PHASE_NO_K0 /* (development: the phase clocks of the bounce kernels alone) */
    pc.flush(st);
#endif
    wave_add_total(st, ST_K0_SHADE_NODES, sc.shade_nodes);
    wave_add_total(st, ST_K0_REFL, sc.refl);
    wave_add_total(st, ST_K0_TRANS, sc.trans);
    flush_tally<STATS>(st, tl, true);
    RT_DEV_TILE_END(st, tile_polygon != RT_NONE ? 2 : tile_no_spheres ? 3 : 4);
    wave_done();
}

/* ------------------------------------------------------------------------------------------ */
/* k_bounce: one queued secondary ray per lane (reflection / transmission of level-1 nodes)    */
/* ------------------------------------------------------------------------------------------ */
template <bool GRID, bool STATS, bool TABLDS, int SHAPE>
__global__ __launch_bounds__(RT_BLOCK, (SHAPE != 0 ? RT_WPE_BOUNCE1 : RT_WPE)) void k_bounce(DevScene S, DevCamera C, DevWork Wk,
                                                     uint32_t level)
{
    fix_shape<SHAPE>(S);
    extern __shared__ __attribute__((aligned(16))) float4 lds[];
    float *lds_park = (float *)lds; /* RT_PARK_FLOATS, ahead of the scene data */
    float *lds_eye = lds_park;
    float4 *lds_poly = lds + RT_PARK_FLOATS / 4u;
    float4 *lds_sph = lds_poly + S.nPolySlots;
    uint32_t *lds_vis = (uint32_t *)(lds_sph + sphere_slots(S.nSph));

    /* block b consumes shard b % RT_SHARDS, as slice b / RT_SHARDS of gridDim.x / RT_SHARDS */
    const uint32_t shard = blockIdx.x % RT_SHARDS;
    const uint32_t slice = blockIdx.x / RT_SHARDS, nslices = gridDim.x / RT_SHARDS;
    uint32_t *cnt = Wk.counters;
    const LevelCounts lc = level_counts(cnt, level, shard); /* (issued before the flag is waited for) */
    if (overflowed_before(cnt))
        return; /* the batch is being abandoned: its queues hold unwritten slots */
    const uint32_t n = min(lc.hits, Wk.capS);
    if (slice * RT_BLOCK >= n)
        return; /* block-uniform */

    const Tables<TABLDS> tab = stage_tables<TABLDS>(S, lds_vis);
    __shared__ float kc[KC_WORDS];
    stage_colours(S, kc);
    block_copy(lds_poly, S.poly, S.nPolySlots);
    if (!GRID) {
        block_copy(lds_sph, S.sph, S.nSph);
        for (uint32_t k = threadIdx.x; k < S.nSph; k += RT_BLOCK)
            ((uint32_t *)(lds_sph + S.nSph))[k] = S.sph_obj[k];
    }
    __syncthreads();

    /* first node index of this level in this shard = nodes made by all shallower levels */
    const uint32_t used = min(lc.used, Wk.capS);
    const uint32_t node_base = shard * Wk.capS + used;
    const uint32_t node_room = Wk.capS - used;
    const uint32_t qbase = shard * Wk.capS;

    SpawnCounts sc;
    Tally<STATS> tl;
    PhaseClock pc;
    /*
     * The queue entry of the NEXT iteration is fetched before this iteration's work and is made to
     * land before this iteration's stores are issued (the empty asm below "uses" it there).  Waits
     * on gfx950 count loads and stores together and in order, so without that the next load's wait
     * would also sit out the write latency of every store issued before it: measured, entry load
     * and store drain were 64 % of an iteration's ~13k cycles, the trace-and-shade work itself 36 %.
     */
    const float4 *const inA = Wk.qA(level & 1u);
    const float4 *const inB = Wk.qB(level & 1u);
    const uint32_t *const inO = Wk.qObj(level & 1u);
    /* (loads are unconditional, on an index clamped into the shard's entries: a load under a branch
     * makes the compiler merge old and new values and wait right behind the load) */
    float4 na, nb;
    uint32_t nc;
    {
        const uint32_t first = min(slice * RT_BLOCK + threadIdx.x, n - 1u);
        na = inA[qbase + first];
        nb = inB[qbase + first];
        nc = inO[qbase + first];
    }
#define RT_LANDED(P_, Q_, R_) \
    asm volatile("" : "+v"(P_.x), "+v"(P_.y), "+v"(P_.z), "+v"(P_.w), "+v"(Q_.x), "+v"(Q_.y), "+v"(Q_.z), \
                      "+v"(Q_.w), "+v"(R_))
    RT_LANDED(na, nb, nc);
    for (uint32_t base = slice * RT_BLOCK; base < n; base += nslices * RT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < n;
        pc.start();
        const float4 a = na, b = nb;
        const uint32_t c = nc;
        {
            const uint32_t inext = min(i + nslices * RT_BLOCK, n - 1u);
            na = inA[qbase + inext];
            nb = inB[qbase + inext];
            nc = inO[qbase + inext];
        }
        /* a queued hit: (origin of the ray that found it, distance), (its direction, parent), object */
        Ray r;
        r.ox = a.x; r.oy = a.y; r.oz = a.z;
        r.dx = b.x; r.dy = b.y; r.dz = b.z;
        r.mag = 0.0f;
        float px, py, pz;
        hit_point(r, a.w, px, py, pz);
        const uint32_t obj = active ? c : 0u;
        const uint32_t parent = active ? __float_as_uint(b.w) : 0u;
        pc.mark(0);
        pc.mark(1);
        const bool hit = active;
        float leaf[3] = {kc[KC_BG + 0], kc[KC_BG + 1], kc[KC_BG + 2]};
        uint32_t node;
        bool sr, st_;
        StashedEye::park(lds_eye, r);
        const StashedEye eye_src{lds_eye};
        const bool deferred = shade_and_spawn<GRID, STATS, TABLDS>(S, C, Wk, lds_poly, lds_sph, lds_vis, lds_park, kc, tab, hit, eye_src,
                                                            px, py, pz, obj, level,
                                                            ((parent >> 1) << 3) | (parent & 1u), shard, node_base,
                                                            node_room, leaf, node, sr, st_, tl, pc);
        sc.add(hit, sr, st_);
        RT_LANDED(na, nb, nc); /* next entry has landed */
        if (active && !deferred && (parent >> 1) < Wk.cap) {
            /* finished child colour goes straight into the parent's slot */
            const uint32_t pn = parent >> 1;
            float4 *dst = (parent & 1u) ? Wk.childT() : Wk.childR();
            dst[pn] = make_float4(leaf[0], leaf[1], leaf[2], 0.0f);
        }
        pc.mark(7);
    }
    uint64_t *st = (uint64_t *)(Wk.counters + CNT_STATS + shard * CNT_STATS_STRIDE);
    pc.flush(st);
    wave_add_total(st, ST_KB_SHADE_NODES, sc.shade_nodes);
    wave_add_total(st, ST_KB_REFL, sc.refl);
    wave_add_total(st, ST_KB_TRANS, sc.trans);
    flush_tally<STATS>(st, tl, false);
}

/* ------------------------------------------------------------------------------------------ */
/* k_subtree: the sparse deep end of the ray tree in ONE launch                                  */
/* ------------------------------------------------------------------------------------------ */
/*
 * The levels from `level` on hold a few per cent of the rays of the reference's scene and used to cost a
 * launch each (bounce) plus their folds: ~60 us of launch and per-hit latency.  Here a lane takes one hit
 * queued for `level` and finishes its WHOLE subtree itself, depth first: shade the hit (shade_and_spawn in
 * its DFS mode: local colour, child rays built and traced, nothing stored), go down into the reflection
 * child if it hit something, then into the transmission child, and on the way up form each node's colour
 * exactly as fold_values does - local + kr * reflection, then + kt * transmission, then the clamp
 * (raytrace.c:440-442,547-549,562) - until the subtree's colour goes into the child slot of the level -
 * 1 node that queued the hit.  No nodes, no queue entries and no folds below `level`.  The pending
 * ancestors (colour so far, coefficients, the transmission child still to do: its ray, distance and
 * object, from which the hit point is formed as the spawning lane forms it) wait on a per-lane stack of
 * RT_DFS_FRAMES frames in scratch memory, touched once per node.  The lanes of a wave run the shade
 * step together, each on its own node; a lane whose subtree is done waits for the others of its wave.
 * The host uses this kernel only when the levels are sparse (rt_api.hip) and depth - level < RT_DFS_FRAMES.
 */
#define RT_DFS_FRAMES RT_SUBTREE_LEVELS
template <bool GRID, bool TABLDS, int SHAPE>
__global__ __launch_bounds__(RT_BLOCK) void k_subtree(DevScene S, DevCamera C, DevWork Wk, uint32_t level)
{
    fix_shape<SHAPE>(S);
    extern __shared__ __attribute__((aligned(16))) float4 lds[];
    float *lds_park = (float *)lds; /* RT_PARK_FLOATS, ahead of the scene data */
    float *lds_eye = lds_park;
    float4 *lds_poly = lds + RT_PARK_FLOATS / 4u;
    float4 *lds_sph = lds_poly + S.nPolySlots;
    uint32_t *lds_vis = (uint32_t *)(lds_sph + sphere_slots(S.nSph));

    const uint32_t shard = blockIdx.x % RT_SHARDS;
    const uint32_t slice = blockIdx.x / RT_SHARDS, nslices = gridDim.x / RT_SHARDS;
    uint32_t *cnt = Wk.counters;
    const LevelCounts lc = level_counts(cnt, level, shard);
    if (overflowed_before(cnt))
        return; /* the batch is being abandoned: its queues hold unwritten slots */
    const uint32_t n = min(lc.hits, Wk.capS);
    if (slice * RT_BLOCK >= n)
        return; /* block-uniform */

    const Tables<TABLDS> tab = stage_tables<TABLDS>(S, lds_vis);
    __shared__ float kc[KC_WORDS];
    stage_colours(S, kc);
    block_copy(lds_poly, S.poly, S.nPolySlots);
    if (!GRID) {
        block_copy(lds_sph, S.sph, S.nSph);
        for (uint32_t k = threadIdx.x; k < S.nSph; k += RT_BLOCK)
            ((uint32_t *)(lds_sph + S.nSph))[k] = S.sph_obj[k];
    }
    __syncthreads();

    const uint32_t qbase = shard * Wk.capS;
    const float4 *const inA = Wk.qA(level & 1u);
    const float4 *const inB = Wk.qB(level & 1u);
    const uint32_t *const inO = Wk.qObj(level & 1u);
    SpawnCounts sc;
    Tally<false> tl;
    PhaseClock pc;
    const float bgr = kc[KC_BG + 0], bgg = kc[KC_BG + 1], bgb = kc[KC_BG + 2];
    /* colour += k * child, a product and a sum each rounded - what k_fold does (fold_values, compiled once without
     * contraction for both builds): a frame of the tolerance build must not depend on which of the two folded it */
    auto fold_child = [](auto &f, float k, float cr_, float cg_, float cb_) {
        f.r = __fadd_rn(f.r, __fmul_rn(k, cr_));
        f.g = __fadd_rn(f.g, __fmul_rn(k, cg_));
        f.b = __fadd_rn(f.b, __fmul_rn(k, cb_));
    };

    struct Frame { /* a node whose children are not all known yet */
        float r, g, b, kr, kt;   /* colour so far, coefficients                                   */
        uint32_t flags;          /* 1: waiting for the transmission child (else the reflection child);
                                    2: a transmission child was asked for; 4: ... and it hit: t_*  */
        float ox, oy, oz, dx, dy, dz, dist;
        uint32_t obj;
    };
    Frame stack[RT_DFS_FRAMES];

    for (uint32_t base = slice * RT_BLOCK; base < n; base += nslices * RT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        bool active = i < n;
        const uint32_t at = qbase + min(i, n - 1u);
        const float4 a = inA[at], b = inB[at];
        /* the node being shaded: hit point, object, the ray that found it, its level */
        uint32_t obj = active ? inO[at] : 0u;
        const uint32_t parent = active ? __float_as_uint(b.w) : 0u;
        Ray r;
        r.ox = a.x; r.oy = a.y; r.oz = a.z;
        r.dx = b.x; r.dy = b.y; r.dz = b.z;
        r.mag = 0.0f;
        float px, py, pz;
        hit_point(r, a.w, px, py, pz);
        uint32_t lv = level, sp = 0;
        while (__any(active)) {
            float leaf[3] = {bgr, bgg, bgb};
            uint32_t node;
            bool sr, st_;
            DfsNode nd;
            StashedEye::park(lds_eye, r);
            const StashedEye eye_src{lds_eye};
            const bool inner = shade_and_spawn<GRID, false, TABLDS, false, true>(S, C, Wk, lds_poly, lds_sph, lds_vis, lds_park, kc,
                                                                                 tab, active, eye_src, px, py, pz, obj, lv, 0u, shard,
                                                                                 0u, 0u, leaf, node, sr, st_, tl, pc, false, &nd);
            sc.add(active, sr, st_);
            if (!active)
                continue;
            float vr = leaf[0], vg = leaf[1], vb = leaf[2]; /* colour of a finished subtree, on its way up */
            bool up = !inner;
            if (inner) { /* at least one child hit something: this node waits on the stack */
                Frame f;
                f.r = nd.r; f.g = nd.g; f.b = nd.b; f.kr = nd.kr; f.kt = nd.kt;
                f.flags = (nd.want_t ? 2u : 0u) | (nd.hit_t ? 4u : 0u);
                f.ox = nd.t_ray.ox; f.oy = nd.t_ray.oy; f.oz = nd.t_ray.oz;
                f.dx = nd.t_ray.dx; f.dy = nd.t_ray.dy; f.dz = nd.t_ray.dz;
                f.dist = nd.t_dist;
                f.obj = nd.t_obj;
                if (nd.hit_r) { /* down into the reflection child: its hit is parked in rows 7-16 */
                    const float *park = lds_park + threadIdx.x;
                    obj = __float_as_uint(park[10 * RT_BLOCK]);
                    r.ox = park[11 * RT_BLOCK]; r.oy = park[12 * RT_BLOCK]; r.oz = park[13 * RT_BLOCK];
                    r.dx = park[14 * RT_BLOCK]; r.dy = park[15 * RT_BLOCK]; r.dz = park[16 * RT_BLOCK];
                    hit_point(r, park[7 * RT_BLOCK], px, py, pz);
                } else { /* only the transmission child hit: the reflection ray, if any, left the scene */
                    if (nd.want_r) fold_child(f, f.kr, bgr, bgg, bgb);
                    f.flags |= 1u;
                    r = nd.t_ray;
                    hit_point(r, nd.t_dist, px, py, pz);
                    obj = nd.t_obj;
                }
                stack[sp] = f;
                ++sp;
                ++lv;
            }
            while (up) { /* a subtree is finished: into its parent, which may finish in turn */
                if (sp == 0u) {
                    if ((parent >> 1) < Wk.cap) {
                        float4 *dst = (parent & 1u) ? Wk.childT() : Wk.childR();
                        dst[parent >> 1] = make_float4(vr, vg, vb, 0.0f);
                    }
                    active = false;
                    break;
                }
                Frame f = stack[sp - 1u];
                if (!(f.flags & 1u)) { /* it was the reflection child */
                    fold_child(f, f.kr, vr, vg, vb);
                    if (f.flags & 4u) { /* now the transmission child: down again */
                        f.flags |= 1u;
                        stack[sp - 1u] = f;
                        r.ox = f.ox; r.oy = f.oy; r.oz = f.oz;
                        r.dx = f.dx; r.dy = f.dy; r.dz = f.dz;
                        hit_point(r, f.dist, px, py, pz);
                        obj = f.obj;
                        break; /* (lv is this frame's level + 1 already) */
                    }
                    if (f.flags & 2u) fold_child(f, f.kt, bgr, bgg, bgb);
                } else { /* it was the transmission child */
                    fold_child(f, f.kt, vr, vg, vb);
                }
                vr = clamp1(f.r); vg = clamp1(f.g); vb = clamp1(f.b);
                --sp;
                --lv;
            }
        }
    }
    uint64_t *st = (uint64_t *)(Wk.counters + CNT_STATS + shard * CNT_STATS_STRIDE);
    wave_add_total(st, ST_KB_SHADE_NODES, sc.shade_nodes);
    wave_add_total(st, ST_KB_REFL, sc.refl);
    wave_add_total(st, ST_KB_TRANS, sc.trans);
}

#ifndef RT_JIT
#include "rt_jobs.inc" /* the pipeline of big scenes: rays as jobs of persistent lanes */
#endif

#if !defined(RT_FAST) && !defined(RT_JIT) /* the kernels below do no arithmetic a tolerance could speed up: one copy, shared by both builds */
/* colour of a tree node once both children are known: raytrace.c:440-442, 547-549, 562 */
__device__ __forceinline__ uint32_t fold_values(const float4 A, const float2 K, const float4 cr,
                                                const float4 ct, float out[3])
{
    const uint32_t link = __float_as_uint(K.y);
    float r = A.x, g = A.y, b = A.z;
    if (link & 2u) { r += A.w * cr.x; g += A.w * cr.y; b += A.w * cr.z; }
    if (link & 4u) { r += K.x * ct.x; g += K.x * ct.y; b += K.x * ct.z; }
    out[0] = clamp1(r); out[1] = clamp1(g); out[2] = clamp1(b);
    return link;
}
/* all four loads are issued together (the child slots unconditionally: one round trip, not two) */
__device__ __forceinline__ uint32_t fold_node(const DevWork &Wk, uint32_t i, float out[3])
{
    return fold_values(Wk.nodeA()[i], Wk.nodeK()[i], Wk.childR()[i], Wk.childT()[i], out);
}

/* k_fold: nodes of `level` (>= 1) -> their parent's child slot */
__global__ __launch_bounds__(RT_BLOCK) void k_fold(DevWork Wk, uint32_t level)
{
    uint32_t *cnt = Wk.counters;
    const uint32_t shard = blockIdx.x % RT_SHARDS;
    const uint32_t slice = blockIdx.x / RT_SHARDS, nslices = gridDim.x / RT_SHARDS;
    const LevelCounts lc = level_counts(cnt, level, shard);
    if (overflowed_before(cnt))
        return;
    const uint32_t used = min(lc.used, Wk.capS);
    const uint32_t start = shard * Wk.capS + used;
    const uint32_t end = start + min(lc.nodes, Wk.capS - used);
    for (uint32_t i = start + slice * RT_BLOCK + threadIdx.x; i < end; i += nslices * RT_BLOCK) {
        float c[3];
        const uint32_t link = fold_node(Wk, i, c);
        const uint32_t pn = link >> 3;
        if (pn >= Wk.cap)
            continue;
        float4 *dst = (link & 1u) ? Wk.childT() : Wk.childR();
        dst[pn] = make_float4(c[0], c[1], c[2], 0.0f);
    }
}

/*
 * k_fold_deep: the folds of the sparse deep levels, top_level down to bottom_level, in ONE launch.
 * A node's children are queued to, and made in, the shard of their parent, so a shard's trees never
 * touch another shard's nodes: one block per shard folds level after level with nothing but a block
 * barrier in between (what it reads was written by an earlier kernel, or by itself on its own CU).
 * Each of those folds was a launch of ~5 us, nearly all of it latency.
 */
__global__ __launch_bounds__(RT_BLOCK * RT_DEEP_SUB) void k_fold_deep(DevWork Wk, uint32_t top_level, uint32_t bottom_level)
{
    uint32_t *cnt = Wk.counters;
    if (overflowed_before(cnt))
        return; /* raised by an earlier kernel: the same answer for every thread */
    const uint32_t shard = blockIdx.x; /* one block per shard */
    /* used_at[l] = nodes of this shard made on levels < l: one load per lane of the first wave, summed by thread 0 */
    __shared__ uint32_t used_at[RT_MAX_LEVELS + 1];
    if (threadIdx.x < RT_MAX_LEVELS)
        used_at[threadIdx.x + 1u] = threadIdx.x <= top_level ? (uint32_t)*level_line(cnt, threadIdx.x, shard) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        used_at[0] = 0;
        for (uint32_t l = 1; l <= top_level + 1u; ++l) {
            sum += used_at[l];
            used_at[l] = min(sum, Wk.capS);
        }
    }
    __syncthreads();
    for (uint32_t level = top_level + 1u; level-- > bottom_level;) {
        const uint32_t end = shard * Wk.capS + used_at[level + 1u];
        for (uint32_t i = shard * Wk.capS + used_at[level] + threadIdx.x; i < end; i += RT_BLOCK * RT_DEEP_SUB) {
            float c[3];
            const uint32_t link = fold_node(Wk, i, c);
            const uint32_t pn = link >> 3;
            if (pn >= Wk.cap)
                continue;
            float4 *dst = (link & 1u) ? Wk.childT() : Wk.childR();
            dst[pn] = make_float4(c[0], c[1], c[2], 0.0f);
        }
        __syncthreads(); /* the parents' slots are written before the level above reads them */
    }
}

/* k_resolve: per pixel - fold this pixel's depth-0 roots, add them to the in-register sum of
 * the childless samples, scale (raytrace.c:779), store radiance, pack (raytrace.c:783-790). */
#ifndef RT_RESOLVE_GRID
#define RT_RESOLVE_GRID 4096u
#endif
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(DevCamera C, DevWork Wk, DevRows R,
                                                      uint32_t *argb, float *radiance, uint32_t *argb_host)
{
    if (overflowed_before(Wk.counters))
        return; /* rows of an abandoned batch are rendered again by the host in smaller batches */
    const size_t npix = (size_t)R.ntiles * 8u * C.W;
    /* A wave takes the 8x8 pixel tile a wave of k_primary rendered, a block the same 32x8 pixels: the nodes of
     * the tile's k-th roots were reserved together and lie side by side, so a wave's 64 gathers per root are
     * whole lines that no other wave touches (pixel rows across the grid read every line of node data from up
     * to eight blocks, on as many XCDs). */
    const uint32_t nbx = (C.W + 31u) / 32u, lane = threadIdx.x & 63u;
    for (uint32_t blk = blockIdx.x; blk < nbx * R.ntiles; blk += gridDim.x) {
        if (R.tile_done && R.tile_done[blk * (RT_BLOCK / 64u) + (threadIdx.x >> 6)])
            continue; /* (wave-uniform) k_primary has finished this tile's pixels */
        const uint32_t x = (blk % nbx) * 32u + (threadIdx.x >> 6) * 8u + (lane & 7u);
        const uint32_t row = (blk / nbx) * 8u + (lane >> 3); /* row of the batch: tile row*8 + line */
        if (x >= C.W || R.y_first + (row >> 3) * R.y_step + (row & 7u) >= R.y_limit)
            continue; /* the last tile row of a slab may hang over its end */
        const size_t p = (size_t)row * C.W + x;
        const float4 acc = Wk.pixAcc[p];
        float r = acc.x, g = acc.y, b = acc.z;
        const uint32_t nroots = __float_as_uint(acc.w); /* roots of the pixel, in sample order: slots 0 .. n-1 */
        /* sixteen root ids at a time (one round trip for a whole 4x4-sample pixel), then their node
         * words four roots at a time, all loads of a step independent; the sums in root order */
        for (uint32_t k0 = 0; k0 < nroots; k0 += 16u) {
            uint32_t id[16];
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u) {
                const uint32_t k = min(k0 + u, nroots - 1u);
                id[u] = Wk.roots[(size_t)k * npix + p];
            }
#pragma unroll
            for (uint32_t g4 = 0; g4 < 16u; g4 += 4u) {
                if (k0 + g4 >= nroots)
                    break;
                float4 A[4], cr[4], ct[4];
                float2 K[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t j = id[g4 + u] < Wk.cap ? id[g4 + u] : 0u;
                    A[u] = Wk.nodeA()[j]; K[u] = Wk.nodeK()[j]; cr[u] = Wk.childR()[j]; ct[u] = Wk.childT()[j];
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    if (k0 + g4 + u < nroots && id[g4 + u] < Wk.cap) {
                        float c[3];
                        fold_values(A[u], K[u], cr[u], ct[u], c);
                        r = r + c[0]; g = g + c[1]; b = b + c[2];
                    }
                }
            }
        }
        const uint32_t packed = store_pixel(C, R, p, r, g, b, argb, radiance);
        /* the blocking host path: k_primary's pixels went to the host in a copy beside the levels; these follow
         * them straight into its mapped buffer (rt_api.hip, "early copy") */
        if (argb_host)
            argb_host[(size_t)R.out_first * C.W + p] = packed;
    }
}

/* k_publish: end of a batch - its counter block goes to the host's pinned mirror (mapped memory:
 * plain stores over the link) and is zeroed for its next use.  One small kernel in place of a
 * device-to-host copy plus a memset command, each of which cost ~10 us of queue latency per frame. */
__global__ __launch_bounds__(RT_BLOCK) void k_publish(uint4 *counters, uint4 *mirror, uint32_t n16)
{
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n16; i += gridDim.x * RT_BLOCK) {
        mirror[i] = counters[i];
        counters[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

/* rt_hip_deinterleave: frame row y = block b*8 + line; b = k*nphases + r lives at gathered row
 * (r*blocks_per_rank + k)*8 + line.  One thread per 16 bytes when the width allows, else per pixel. */
template <typename T>
__global__ __launch_bounds__(RT_BLOCK) void k_deinterleave(const T *gathered, T *frame, uint32_t row_words,
                                                           uint32_t height, uint32_t nphases,
                                                           uint32_t blocks_per_rank)
{
    const size_t total = (size_t)height * row_words;
    for (size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < total; i += (size_t)gridDim.x * RT_BLOCK) {
        const uint32_t y = (uint32_t)(i / row_words), x = (uint32_t)(i % row_words);
        const uint32_t b = y >> 3, line = y & 7u;
        const uint32_t r = b % nphases, k = b / nphases;
        frame[i] = gathered[((size_t)(r * blocks_per_rank + k) * 8u + line) * row_words + x];
    }
}

/* rt_hip_selftest_arith: the helpers of rt_arith.h as the kernels use them */
__global__ __launch_bounds__(RT_BLOCK) void k_selftest_arith(const float *a, const float *b, const float *c,
                                                             uint32_t n, float *out, double *out64)
{
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const float x = a[i], y = b[i], z = c[i];
    float nx = x, ny = y, nz = z;
    normalize3(nx, ny, nz);
    float *o = out + (size_t)i * 7u;
    o[0] = sqrt_rn(x);
    o[1] = div_rn(x, y);
    o[2] = div_with(x, y, rcp_refined(y));
    o[3] = nx; o[4] = ny; o[5] = nz;
    o[6] = recip_rn(y);
    out64[i] = sqrt_rn_f64_pos((double)x);
}

/* rt_hip_selftest_grid: for n rays (7 floats each: origin, direction, magnitude) the sphere search of a
 * big scene twice - through the grid, and as the reference's loop over every sphere - so that a test
 * can demand identical winners.  out[4*i..] = object, distance bits (grid), object, distance bits (loop). */
template <bool FILTERED>
__global__ __launch_bounds__(RT_BLOCK) void k_selftest_grid(DevScene S, const float *rays, uint32_t n, uint32_t *out)
{
    const uint32_t i0 = blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool act = i0 < n; /* (no early return: the search needs whole waves) */
    const uint32_t i = act ? i0 : n - 1u;
    S.nPoly = 0u; /* spheres only */
    Ray r;
    r.ox = rays[7 * i + 0]; r.oy = rays[7 * i + 1]; r.oz = rays[7 * i + 2];
    r.dx = rays[7 * i + 3]; r.dy = rays[7 * i + 4]; r.dz = rays[7 * i + 5];
    r.mag = rays[7 * i + 6];
    Tally<false> tl;
    float dg = 0.0f;
    uint32_t og = RT_NONE;
    intersect_all<true, FILTERED, false, false, true>(S, nullptr, nullptr, act, r, RT_NONE, RT_NONE, dg, og, tl);
    if (!act)
        return;
    bool found = false;
    float d = 0.0f;
    uint32_t best = RT_NONE;
    for (uint32_t k = 0; k < S.nSph; ++k) { /* raytrace.c:188-272 over the sphere list */
        const float4 s = S.sph[k];
        const float ex = r.ox - s.x, ey = r.oy - s.y, ez = r.oz - s.z;
        const float B = 2.0f * ((r.dx * ex + r.dy * ey) + r.dz * ez);
        const float Cq = ((ex * ex + ey * ey) + ez * ez) - s.w;
        const float det = (B * B) - (4.0f * Cq);
        if (det < 0.0f)
            continue;
        if (FILTERED && !(B < 0.0f))
            continue;
        const float w = sphere_distance(B, det);
        if (FILTERED && !(w < r.mag && w > 0.0f))
            continue;
        if (!found || w < d) {
            d = w;
            best = k;
            found = true;
        }
    }
    out[4 * i + 0] = og;
    out[4 * i + 1] = __float_as_uint(dg);
    out[4 * i + 2] = found ? S.sph_obj[best] : RT_NONE;
    out[4 * i + 3] = __float_as_uint(found ? d : 0.0f);
}

#endif /* !RT_FAST */

} // namespace RT_NSNAME
#ifndef RT_JIT
using namespace RT_NSNAME;

/* ------------------------------------------------------------------------------------------ */
/* launch wrappers                                                                            */
/* ------------------------------------------------------------------------------------------ */

/* a kernel compiled at run time for the scene's shape (rt_jit.cpp): the same arguments, through the module API */
template <class A, class B, class C4, class D>
static void launch_module(void *fn, dim3 grid, size_t lds, hipStream_t stream, const A &a, const B &b, const C4 &c, const D &d,
                          hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr)
{
    void *args[] = {(void *)&a, (void *)&b, (void *)&c, (void *)&d};
    if (ev0) /* (the "ext" launch counts threads, not blocks) */
        (void)hipExtModuleLaunchKernel((hipFunction_t)fn, grid.x * RT_BLOCK, grid.y, grid.z, RT_BLOCK, 1, 1, lds, stream, args, nullptr, ev0, ev1, 0);
    else
        (void)hipModuleLaunchKernel((hipFunction_t)fn, grid.x, grid.y, grid.z, RT_BLOCK, 1, 1, (unsigned int)lds, stream, args, nullptr);
}
/* a kernel of the library, with events that take its own start and end when the caller times it: two events recorded
 * around the launch bracket the dispatch packet and read 6-9 us long beside a busy tail stream */
static void launch_primary(void (*kernel)(DevScene, DevCamera, DevWork, DevRows), dim3 grid, size_t lds, hipStream_t stream,
                           const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R, hipEvent_t ev0, hipEvent_t ev1)
{
    if (ev0)
        hipExtLaunchKernelGGL(kernel, grid, dim3(RT_BLOCK), (unsigned int)lds, stream, ev0, ev1, 0, S, C, Wk, R);
    else
        kernel<<<grid, RT_BLOCK, lds, stream>>>(S, C, Wk, R);
}

#ifndef RT_FAST
int rt_scene_shape(const DevScene &S)
{
    static const bool off = getenv("RT_HIP_NO_SHAPES") != nullptr; /* development: generic kernels only */
    if (off || S.noShape)
        return 0;
    if (S.nSph == 2u && S.nPoly == 1u && S.nPolySlots == 10u && S.nObj == 3u && S.nLights == 1u &&
        S.tabSlots == 17u && S.tabInLds)
        return 1;
    return 0;
}

size_t rt_scene_lds_bytes(const DevScene &S)
{
    const uint32_t sph = sphere_slots(S.nSph);
    /* + one visibility word per thread for every full group of 32 lights before the last group,
     * + the light / object tables when they are staged */
    return RT_PARK_FLOATS * sizeof(float) + (size_t)(S.nPolySlots + sph) * sizeof(float4) +
           (size_t)vis_words(S.nLights) * sizeof(uint32_t) +
           (S.tabInLds ? (size_t)(S.tabSlots + S.nSph + S.nPoly) * sizeof(float4) : 0); /* + eye terms */
}

#endif /* !RT_FAST */

#ifdef RT_FAST
/* the tolerance build's kernels: plain Phong renders only (the instrumented and phong-blinn variants
 * exist in the exact build alone; rt_api.hip routes those there) */
void rt_launch_primary_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                            hipStream_t stream, const RtJitKernels *jit, hipEvent_t ev0, hipEvent_t ev1)
{
    const dim3 grid((C.W + 31u) / 32u, R.ntiles);
    const size_t lds = rt_scene_lds_bytes(S);
    if (jit) {
        launch_module(jit->primary, R.plan ? dim3(R.plan_len) : grid, lds, stream, S, C, Wk, R, ev0, ev1);
        return;
    }
    const bool grid_scene = S.nSph > RT_TILE_SPHERES;
    void (*kernel)(DevScene, DevCamera, DevWork, DevRows) =
        grid_scene ? k_primary<true, false, false, 0>
        : rt_scene_shape(S) == 1 ? k_primary<false, false, true, 1>
        : S.tabInLds ? k_primary<false, false, true, 0> : k_primary<false, false, false, 0>;
    static const size_t pad_to = [] { const char *e = getenv("RT_HIP_PRIMARY_LDS"); return e ? (size_t)atol(e) : 0; }(); /* as in rt_launch_primary */
    /* (a plan is a list of blocks) */
    launch_primary(kernel, R.plan ? dim3(R.plan_len) : grid, lds < pad_to && pad_to <= 65536 ? pad_to : lds, stream, S, C, Wk, R, ev0, ev1);
}
#else
void rt_launch_primary(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                       bool stats, bool blinn, hipStream_t stream, const RtJitKernels *jit, hipEvent_t ev0, hipEvent_t ev1)
{
    const dim3 grid((C.W + 31u) / 32u, R.ntiles);
    const size_t lds = rt_scene_lds_bytes(S);
    if (jit && !stats && !blinn) {
        launch_module(jit->primary, R.plan ? dim3(R.plan_len) : grid, lds, stream, S, C, Wk, R, ev0, ev1);
        return;
    }
    const bool grid_scene = S.nSph > RT_TILE_SPHERES;
    if (blinn) { /* get_shade_color_phongblinn (raytrace.c:574-685): the generic kernels only */
        void (*kernel)(DevScene, DevCamera, DevWork, DevRows) =
            grid_scene ? (stats ? k_primary<true, true, false, 0, true> : k_primary<true, false, false, 0, true>)
            : S.tabInLds ? (stats ? k_primary<false, true, true, 0, true> : k_primary<false, false, true, 0, true>)
                         : (stats ? k_primary<false, true, false, 0, true> : k_primary<false, false, false, 0, true>);
        launch_primary(kernel, R.plan ? dim3(R.plan_len) : grid, lds, stream, S, C, Wk, R, ev0, ev1);
        return;
    }
    /* three scene classes: small (everything in LDS), mid (spheres in LDS, tables in HBM), grid (spheres in HBM) */
    void (*kernel)(DevScene, DevCamera, DevWork, DevRows) =
        grid_scene ? (stats ? k_primary<true, true, false, 0> : k_primary<true, false, false, 0>)
        : rt_scene_shape(S) == 1 ? (stats ? k_primary<false, true, true, 1> : k_primary<false, false, true, 1>)
        : S.tabInLds ? (stats ? k_primary<false, true, true, 0> : k_primary<false, false, true, 0>)
                     : (stats ? k_primary<false, true, false, 0> : k_primary<false, false, false, 0>);
    /* RT_HIP_PRIMARY_LDS=<bytes>: ask for at least that much LDS per block.  23600 leaves room for six
     * blocks per CU instead of seven, i.e. a free block slot (LDS, 80 VGPRs) on every CU for the kernels of
     * a tail stream, which otherwise get in only when a block of this kernel retires */
    static const size_t pad_to = [] { const char *e = getenv("RT_HIP_PRIMARY_LDS"); return e ? (size_t)atol(e) : 0; }();
    launch_primary(kernel, R.plan ? dim3(R.plan_len) : grid, lds < pad_to && pad_to <= 65536 ? pad_to : lds, stream, S, C, Wk, R, ev0, ev1);
}
#endif

/*
 * Grid of a queue-draining kernel: RT_SHARDS shards x slices, grid-strided over each shard's
 * entries.  The grid is exactly one resident set - (blocks the kernel fits per CU) x (CUs),
 * rounded down to whole shards: a few blocks more and they would run as a second round after the
 * first has drained, doubling a latency-bound kernel (measured with 2048 blocks at 7 per CU).
 */
static uint32_t queue_grid(const DevWork &Wk, const void *kernel, size_t lds, uint32_t expect, bool module_fn = false)
{
    struct Entry { const void *kernel; size_t lds; int device; uint32_t blocks; };
    constexpr int kCache = 256; /* devices x kernel variants x LDS sizes */
    static Entry cache[kCache];
    static int ncache = 0;
    static std::mutex mu; /* contexts launch from several threads */
    int device = 0;
    (void)hipGetDevice(&device);
    uint32_t resident = 0;
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < ncache; ++i)
        if (cache[i].kernel == kernel && cache[i].lds == lds && cache[i].device == device)
            resident = cache[i].blocks;
    if (!resident) {
        int per_cu = 0, cus = 0;
        const hipError_t rc = module_fn ? hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (hipFunction_t)kernel, RT_BLOCK, lds)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RT_BLOCK, lds);
        if (rc != hipSuccess || per_cu < 1)
            per_cu = 4;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1)
            cus = 256;
        resident = (uint32_t)per_cu * (uint32_t)cus;
        if (ncache < kCache)
            cache[ncache++] = {kernel, lds, device, resident};
    }
    uint32_t slices = resident / RT_SHARDS;
    uint32_t most = (Wk.capS + RT_BLOCK - 1) / RT_BLOCK;
    if (expect) { /* the deep levels hold a few thousand entries: do not launch 1500 blocks for them */
        const uint32_t want = (uint32_t)(((uint64_t)expect * 2u + RT_BLOCK - 1) / RT_BLOCK);
        if (want < most)
            most = want;
    }
    if (slices > most)
        slices = most;
    if (slices < 1)
        slices = 1;
    return slices * RT_SHARDS;
}

#ifdef RT_FAST
void rt_launch_bounce_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, uint32_t level,
                           uint32_t expect, hipStream_t stream, const RtJitKernels *jit)
{
    const size_t lds = rt_scene_lds_bytes(S);
    if (jit) {
        launch_module(jit->bounce, dim3(queue_grid(Wk, jit->bounce, lds, expect, true)), lds, stream, S, C, Wk, level);
        return;
    }
    const bool grid_scene = S.nSph > RT_TILE_SPHERES;
    void (*kernel)(DevScene, DevCamera, DevWork, uint32_t) =
        grid_scene ? k_bounce<true, false, false, 0>
        : rt_scene_shape(S) == 1 ? k_bounce<false, false, true, 1>
        : S.tabInLds ? k_bounce<false, false, true, 0> : k_bounce<false, false, false, 0>;
    const uint32_t grid = queue_grid(Wk, (const void *)kernel, lds, expect);
    kernel<<<grid, RT_BLOCK, lds, stream>>>(S, C, Wk, level);
}
void rt_launch_subtree_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, uint32_t level,
                            uint32_t expect, hipStream_t stream, const RtJitKernels *jit)
{
    const size_t lds = rt_scene_lds_bytes(S);
    if (jit) {
        launch_module(jit->subtree, dim3(queue_grid(Wk, jit->subtree, lds, expect, true)), lds, stream, S, C, Wk, level);
        return;
    }
    const bool grid_scene = S.nSph > RT_TILE_SPHERES;
    void (*kernel)(DevScene, DevCamera, DevWork, uint32_t) =
        grid_scene ? k_subtree<true, false, 0>
        : rt_scene_shape(S) == 1 ? k_subtree<false, true, 1>
        : S.tabInLds ? k_subtree<false, true, 0> : k_subtree<false, false, 0>;
    const uint32_t grid = queue_grid(Wk, (const void *)kernel, lds, expect);
    kernel<<<grid, RT_BLOCK, lds, stream>>>(S, C, Wk, level);
}
#else
void rt_launch_bounce(const DevScene &S, const DevCamera &C, const DevWork &Wk, uint32_t level,
                      bool stats, uint32_t expect, hipStream_t stream, const RtJitKernels *jit)
{
    const size_t lds = rt_scene_lds_bytes(S);
    if (jit && !stats) {
        launch_module(jit->bounce, dim3(queue_grid(Wk, jit->bounce, lds, expect, true)), lds, stream, S, C, Wk, level);
        return;
    }
    const bool grid_scene = S.nSph > RT_TILE_SPHERES;
    void (*kernel)(DevScene, DevCamera, DevWork, uint32_t) =
        grid_scene ? (stats ? k_bounce<true, true, false, 0> : k_bounce<true, false, false, 0>)
        : rt_scene_shape(S) == 1 ? (stats ? k_bounce<false, true, true, 1> : k_bounce<false, false, true, 1>)
        : S.tabInLds ? (stats ? k_bounce<false, true, true, 0> : k_bounce<false, false, true, 0>)
                     : (stats ? k_bounce<false, true, false, 0> : k_bounce<false, false, false, 0>);
    const uint32_t grid = queue_grid(Wk, (const void *)kernel, lds, expect);
    kernel<<<grid, RT_BLOCK, lds, stream>>>(S, C, Wk, level);
}

void rt_launch_subtree(const DevScene &S, const DevCamera &C, const DevWork &Wk, uint32_t level,
                       uint32_t expect, hipStream_t stream, const RtJitKernels *jit)
{
    const size_t lds = rt_scene_lds_bytes(S);
    if (jit) {
        launch_module(jit->subtree, dim3(queue_grid(Wk, jit->subtree, lds, expect, true)), lds, stream, S, C, Wk, level);
        return;
    }
    const bool grid_scene = S.nSph > RT_TILE_SPHERES;
    void (*kernel)(DevScene, DevCamera, DevWork, uint32_t) =
        grid_scene ? k_subtree<true, false, 0>
        : rt_scene_shape(S) == 1 ? k_subtree<false, true, 1>
        : S.tabInLds ? k_subtree<false, true, 0> : k_subtree<false, false, 0>;
    const uint32_t grid = queue_grid(Wk, (const void *)kernel, lds, expect);
    kernel<<<grid, RT_BLOCK, lds, stream>>>(S, C, Wk, level);
}

/* folds of levels top_level .. bottom_level (top first) in one launch, one block per shard */
void rt_launch_fold_deep(const DevWork &Wk, uint32_t top_level, uint32_t bottom_level, hipStream_t stream)
{
    k_fold_deep<<<RT_SHARDS, RT_BLOCK * RT_DEEP_SUB, 0, stream>>>(Wk, top_level, bottom_level);
}

void rt_launch_fold(const DevWork &Wk, uint32_t level, uint32_t expect, hipStream_t stream)
{
    k_fold<<<queue_grid(Wk, (const void *)k_fold, 0, expect), RT_BLOCK, 0, stream>>>(Wk, level);
}

void rt_launch_resolve(const DevCamera &C, const DevWork &Wk, const DevRows &R, uint32_t *d_argb,
                       float *d_radiance, uint32_t *d_argb_host, hipStream_t stream)
{
    size_t grid = (size_t)((C.W + 31u) / 32u) * R.ntiles; /* blocks of 32x8 pixels, as k_primary's */
    if (grid > RT_RESOLVE_GRID)
        grid = RT_RESOLVE_GRID;
    if (grid == 0)
        grid = 1;
    k_resolve<<<(uint32_t)grid, RT_BLOCK, 0, stream>>>(C, Wk, R, d_argb, d_radiance, d_argb_host);
}

void rt_launch_selftest_arith(const float *a, const float *b, const float *c, uint32_t n, float *out,
                              double *out64, hipStream_t stream)
{
    if (n)
        k_selftest_arith<<<(n + RT_BLOCK - 1) / RT_BLOCK, RT_BLOCK, 0, stream>>>(a, b, c, n, out, out64);
}

void rt_launch_selftest_grid(const DevScene &S, const float *rays, uint32_t n, bool filtered, uint32_t *out,
                             hipStream_t stream)
{
    if (!n)
        return;
    if (filtered)
        k_selftest_grid<true><<<(n + RT_BLOCK - 1) / RT_BLOCK, RT_BLOCK, 0, stream>>>(S, rays, n, out);
    else
        k_selftest_grid<false><<<(n + RT_BLOCK - 1) / RT_BLOCK, RT_BLOCK, 0, stream>>>(S, rays, n, out);
}

void rt_launch_deinterleave(const uint32_t *gathered, uint32_t *frame, uint32_t width, uint32_t height,
                            uint32_t nphases, hipStream_t stream)
{
    const uint32_t blocks_per_rank = ((height + 7u) / 8u + nphases - 1u) / nphases;
    const bool wide = (width % 4u) == 0 && ((uintptr_t)gathered % 16u) == 0 && ((uintptr_t)frame % 16u) == 0;
    const uint32_t row_words = wide ? width / 4u : width;
    const size_t total = (size_t)height * row_words;
    if (!total)
        return;
    size_t grid = (total + RT_BLOCK - 1) / RT_BLOCK;
    if (grid > 8192)
        grid = 8192;
    if (wide)
        k_deinterleave<uint4><<<(uint32_t)grid, RT_BLOCK, 0, stream>>>((const uint4 *)gathered, (uint4 *)frame,
                                                                        row_words, height, nphases, blocks_per_rank);
    else
        k_deinterleave<uint32_t><<<(uint32_t)grid, RT_BLOCK, 0, stream>>>(gathered, frame, row_words, height,
                                                                           nphases, blocks_per_rank);
}

void rt_launch_publish(uint32_t *counters, uint32_t *mirror, uint32_t words, hipStream_t stream)
{
    const uint32_t n16 = words / 4u; /* the block is a whole number of 16-byte pieces (rt_device.h) */
    k_publish<<<(n16 + RT_BLOCK - 1) / RT_BLOCK, RT_BLOCK, 0, stream>>>((uint4 *)counters, (uint4 *)mirror, n16);
}
#endif /* RT_FAST */

/* ---- the job pipeline of big scenes (rt_jobs.inc): both builds ---- */
#ifdef RT_FAST
#define RT_WRAP(name) name##_fast
#else
#define RT_WRAP(name) name
#endif
void RT_WRAP(rt_launch_jobs_hits)(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                                  const DevLevel0 &Z, hipStream_t stream)
{
    const dim3 grid((C.W + 31u) / 32u, R.ntiles);
    if (hits_with_shadows(S))
        k_hits<true><<<grid, RT_BLOCK, (size_t)S.nPolySlots * sizeof(float4), stream>>>(S, C, Wk, R, Z);
    else
        k_hits<false><<<grid, RT_BLOCK, (size_t)S.nPolySlots * sizeof(float4), stream>>>(S, C, Wk, R, Z);
}
/* `expect`: hits per shard the level is expected to hold (0: unknown, or level 0 - a full resident set).  The
 * shadow rays of the level's hits, then (spawns: level != depth, raytrace.c:405,452) their child rays. */
void RT_WRAP(rt_launch_jobs_trace)(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                                   const DevLevel0 &Z, uint32_t level, uint32_t expect, bool spawns, hipStream_t stream)
{
    const size_t lds = (size_t)S.nPolySlots * sizeof(float4);
    for (int shadow = 1; shadow >= 0; --shadow) {
        const uint32_t kinds = shadow ? S.nLights : (spawns ? 2u : 0u);
        if (!kinds || (shadow && !S.lgrid))
            continue;
        void (*kernel)(DevScene, DevCamera, DevWork, DevRows, DevLevel0, uint32_t) =
            level == 0u ? (shadow ? k_trace<true, true> : k_trace<true, false>)
                        : (shadow ? k_trace<false, true> : k_trace<false, false>);
        /* a block's four waves work through ~2 rounds of jobs at least: a sparse level gets a small grid */
        const uint32_t slices = expect ? (uint32_t)((uint64_t)expect * kinds / (2u * RT_BLOCK)) + 1u : 0u;
        const uint32_t grid = queue_grid(Wk, (const void *)kernel, lds, slices ? slices * RT_BLOCK / 2u : 0u);
        if (!(level == 0u && shadow && hits_with_shadows(S))) /* (k_hits sent them, all but those for the loop below) */
            kernel<<<grid, RT_BLOCK, lds, stream>>>(S, C, Wk, R, Z, level);
        /* ... and its rays that the grids could not take (few: a small grid), before the other kind of ray re-uses
         * the shard's loop queue */
        void (*loop)(DevScene, DevWork, DevLevel0, uint32_t) =
            level == 0u ? (shadow ? k_loop<true, true> : k_loop<true, false>) : (shadow ? k_loop<false, true> : k_loop<false, false>);
        loop<<<RT_SHARDS * 32u, RT_BLOCK, 0, stream>>>(S, Wk, Z, level); /* (a block per ray, 32 blocks a shard) */
    }
}
void RT_WRAP(rt_launch_jobs_shadefold)(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevLevel0 &Z,
                                       uint32_t level, uint32_t expect, hipStream_t stream)
{
    k_shadefold<<<queue_grid(Wk, (const void *)k_shadefold, 0, expect), RT_BLOCK, 0, stream>>>(S, C, Wk, Z, level);
}
void RT_WRAP(rt_launch_jobs_resolve)(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                                     const DevLevel0 &Z, hipStream_t stream)
{
    size_t grid = ((size_t)Z.npix + RT_BLOCK - 1) / RT_BLOCK;
    grid = grid > 16384 ? 16384 : (grid ? grid : 1);
    k_resolve_jobs<<<(uint32_t)grid, RT_BLOCK, 0, stream>>>(S, C, Wk, R, Z);
}

#if defined(RT_TILE_TRACE) && defined(RT_FAST)
extern "C" int rt_dev_tile_trace_read(void *host, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rt_tile_trace), bytes);
}
#endif
#endif /* !RT_JIT */
