/*
 * rt_arith.h - correctly rounded float divide / square root for the kernels, without the range
 * scaffolding of the compiler's generic expansions.
 *
 * The reference is plain C on x86: every `/` and every sqrt() it executes is IEEE correctly
 * rounded (vector4.c:219-285, ray.c:29-41,72-86), so the kernels must round the same way.  hipcc's
 * default expansions do that, but they pay for inputs this path never sees: a float divide is 12
 * instructions (two v_div_scale, v_div_fmas: rescaling for quotients near the ends of the exponent
 * range) and a float sqrt 18 (denormal input scaling, class fix-ups).  Divides and square roots
 * were ~45 % of the vector instructions of k_primary, which is bound by vector-instruction issue.
 *
 * The helpers below run the same refinement steps as the compiler's expansion (so results are
 * bit-identical to it wherever its scaling is inactive) and keep only what the data can reach:
 *
 *   sqrt_rn(x)      9 instructions.  v_sqrt_f32 is within 1 ulp; the neighbours s-1ulp and s+1ulp
 *                   are tested with exact FMA residuals.  Exact for every x that is not denormal
 *                   (0, -0, inf, NaN and negative x included: the residual tests are then false
 *                   or NaN and the hardware's own result, which is right for those, is kept).
 *   rcp_refined(b)  v_rcp_f32 + one Newton step: the reciprocal a divide sequence needs.
 *   div_with(a,b,y) 5 instructions per quotient once y = rcp_refined(b) exists: two FMA residual
 *                   corrections.  Three quotients by one length (a normalisation) cost 3 + 15
 *                   instead of 36.  Exact when b and a/b are normal numbers in [2^-96, 2^96];
 *                   b = 0 or inf gives NaN instead of inf / 0, and a zero numerator may lose its
 *                   sign (+0 for -0; nothing on this path divides by or takes the sign of such a
 *                   component).  Only div_rn repairs those.
 *   div_rn(a,b)     9 instructions: the above + v_div_fixup_f32, which restores the IEEE results
 *                   for zero, infinite and NaN operands.
 *   sqrt_rn_f64_pos 11 instructions, for finite normal x > 0 only (the sphere discriminant,
 *                   ray.c:217-218): the compiler's f64 expansion minus scaling and fix-ups.
 *
 * tests/test_gpu_parity.py::test_arith_helpers_are_ieee checks all of them against IEEE results
 * on 2^22 random and special operands through rt_hip_selftest_arith.
 */
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#ifdef RT_FAST
/*
 * Tolerance build (-DRT_FAST, rt_kernels.hip compiled a second time with -ffp-contract=fast): the
 * north star asks for 8-bit pixels within 1 LSB and radiance within 1e-4 relative, not for the
 * reference's last bit.  The same helper names then stand for the hardware's 1-ulp approximations:
 * one v_sqrt_f32 / v_rcp_f32 / v_rsq_f32 (two issue slots each) in place of the 9-instruction exact
 * sequences, and a quotient is numerator times reciprocal.  Every operation stays within ~2 ulp of the
 * exact build's, so results differ from the reference's by ~1e-6 relative, except for the rare sample
 * whose hit / miss / inside verdict sits within rounding of its threshold (as between two CPU builds of
 * the reference with different compiler flags).
 */
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rcp_refined(float b) { return __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float div_with(float a, float b, float y) { (void)b; return a * y; }
__device__ __forceinline__ float div_rn(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float rsq_approx(float x) { return __builtin_amdgcn_rsqf(x); }
#else

__device__ __forceinline__ float sqrt_rn(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float up = __uint_as_float(__float_as_uint(s) + 1u);
    const float below = __builtin_fmaf(-dn, s, x); /* x - (s-1ulp)*s, exact */
    const float above = __builtin_fmaf(-up, s, x); /* x - (s+1ulp)*s, exact */
    float r = (below <= 0.0f) ? dn : s;
    r = (above > 0.0f) ? up : r;
    return r;
}

__device__ __forceinline__ float rcp_refined(float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}

__device__ __forceinline__ float div_with(float a, float b, float y)
{
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}

__device__ __forceinline__ float div_rn(float a, float b)
{
    return __builtin_amdgcn_div_fixupf(div_with(a, b, rcp_refined(b)), b, a);
}

__device__ __forceinline__ double sqrt_rn_f64_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y;
    const double h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, s0, 0.5);
    const double s1 = __builtin_fma(s0, r0, s0);
    const double h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-s1, s1, x);
    const double s2 = __builtin_fma(d0, h1, s1);
    const double d1 = __builtin_fma(-s2, s2, x);
    return __builtin_fma(d1, h1, s2);
}
#endif /* RT_FAST */
