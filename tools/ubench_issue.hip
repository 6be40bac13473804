// development microbenchmark: what do scalar / branch / exec-mask instructions cost next to vector
// ones when several waves share a SIMD?  (k_primary lost 19 % of its time with 24 % fewer SALU and
// branch instructions and only 3 % fewer VALU.)
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o tools/ubench_issue ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP16(X) X X X X X X X X X X X X X X X X

#define V10                                                                                        \
    "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n"       \
    "v_fma_f32 %3, %3, %10, %11\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n"       \
    "v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n v_fma_f32 %8, %8, %10, %11\n"       \
    "v_fma_f32 %9, %9, %10, %11\n"

#define KERNEL(NAME, EXTRA)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float a, float b)           \
    {                                                                                              \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5,   \
              r6 = r0 + 6, r7 = r0 + 7, r8 = r0 + 8, r9 = r0 + 9;                                  \
        for (int i = 0; i < iters; ++i) {                                                          \
            asm volatile(REP16(V10 EXTRA)                                                          \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6),   \
                           "+v"(r7), "+v"(r8), "+v"(r9)                                            \
                         : "v"(a), "v"(b)                                                          \
                         : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "vcc", "scc");  \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9;     \
    }

KERNEL(k_v, "")
KERNEL(k_v_salu4, "s_or_b64 s[40:41], s[42:43], s[44:45]\n s_and_b64 s[46:47], s[42:43], s[44:45]\n"
                  "s_or_b64 s[40:41], s[42:43], s[44:45]\n s_and_b64 s[46:47], s[42:43], s[44:45]\n")
KERNEL(k_v_salu4dep, "s_or_b64 s[40:41], s[40:41], s[44:45]\n s_and_b64 s[40:41], s[40:41], s[44:45]\n"
                     "s_or_b64 s[40:41], s[40:41], s[44:45]\n s_and_b64 s[40:41], s[40:41], s[44:45]\n")
KERNEL(k_v_nop4, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
KERNEL(k_v_salu32, "s_add_i32 s40, s42, s44\n s_add_i32 s41, s42, s44\n s_add_i32 s46, s42, s44\n s_add_i32 s47, s42, s44\n")
// an `if` whose body nobody skips: save exec, (branch not taken), restore
KERNEL(k_v_if2, "s_and_saveexec_b64 s[40:41], exec\n s_cbranch_execz 1f\n 1: s_or_b64 exec, exec, s[40:41]\n"
                "s_and_saveexec_b64 s[42:43], exec\n s_cbranch_execz 2f\n 2: s_or_b64 exec, exec, s[42:43]\n")
// the same without the branch instruction
KERNEL(k_v_exec2, "s_and_saveexec_b64 s[40:41], exec\n s_or_b64 exec, exec, s[40:41]\n"
                  "s_and_saveexec_b64 s[42:43], exec\n s_or_b64 exec, exec, s[42:43]\n")
// vector compare feeding a scalar mask operation (VALU -> SALU hand-over)
KERNEL(k_v_cmp2, "v_cmp_lt_f32 s[40:41], %0, %1\n v_cmp_lt_f32 s[42:43], %2, %3\n"
                 "s_and_b64 s[44:45], s[40:41], s[42:43]\n")
// compare + select, all vector (what a branch-free `?:` costs)
KERNEL(k_v_sel2, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc\n"
                 "v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %6, %6, %7, vcc\n")
// taken branches (a forward jump over nothing)
KERNEL(k_v_jump2, "s_branch 1f\n 1:\n s_branch 2f\n 2:\n")
// 14 vector instructions instead of 10: the price of 4 more VALU
KERNEL(k_v14, "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n"
              "v_fma_f32 %3, %3, %10, %11\n")

// packed f32: 5 v_pk_fma_f32 (= 10 fma) in place of 10 v_fma_f32, and the same for mul / add
#define KERNEL_PK(NAME, OP)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float a, float b)           \
    {                                                                                              \
        typedef float f2 __attribute__((ext_vector_type(2)));                                      \
        f2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f; \
        f2 ka = {a, a}, kb = {b, b};                                                               \
        for (int i = 0; i < iters; ++i) {                                                          \
            asm volatile(REP16(OP " %0, %0, %5, %6\n" OP " %1, %1, %5, %6\n" OP " %2, %2, %5, %6\n" \
                               OP " %3, %3, %5, %6\n" OP " %4, %4, %5, %6\n")                      \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4)                        \
                         : "v"(ka), "v"(kb));                                                      \
        }                                                                                          \
        f2 s = r0 + r1 + r2 + r3 + r4;                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                           \
    }
KERNEL_PK(k_pkfma5, "v_pk_fma_f32")
#define KERNEL_PK2(NAME, OP)                                                                       \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float a, float b)           \
    {                                                                                              \
        typedef float f2 __attribute__((ext_vector_type(2)));                                      \
        f2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f; \
        f2 ka = {a, a};                                                                            \
        for (int i = 0; i < iters; ++i) {                                                          \
            asm volatile(REP16(OP " %0, %0, %5\n" OP " %1, %1, %5\n" OP " %2, %2, %5\n"            \
                               OP " %3, %3, %5\n" OP " %4, %4, %5\n")                              \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4)                        \
                         : "v"(ka));                                                               \
        }                                                                                          \
        f2 s = r0 + r1 + r2 + r3 + r4;                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + b;                                       \
    }
KERNEL_PK2(k_pkmul5, "v_pk_mul_f32")
KERNEL_PK2(k_pkadd5, "v_pk_add_f32")
KERNEL(k_mul10, "")

typedef void (*kern_t)(float *, int, float, float);

int main(int argc, char **argv)
{
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 6;
    const int iters = 2000;
    float *out;
    hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
    struct { const char *name; kern_t k; int extra; } ks[] = {
        {"10 v_fma", k_v, 0}, {"+4 independent s_or/s_and_b64", k_v_salu4, 4}, {"+4 dependent s_or/s_and_b64", k_v_salu4dep, 4},
        {"+4 s_nop 0", k_v_nop4, 4}, {"+4 s_add_i32", k_v_salu32, 4}, {"+2 x (saveexec, cbranch_execz, or exec)", k_v_if2, 6},
        {"+2 x (saveexec, or exec)", k_v_exec2, 4}, {"+2 v_cmp -> sgpr, 1 s_and_b64", k_v_cmp2, 3},
        {"+2 x (v_cmp vcc, v_cndmask)", k_v_sel2, 4}, {"+2 s_branch (taken)", k_v_jump2, 2}, {"+4 v_fma", k_v14, 4},
        {"5 v_pk_fma_f32 INSTEAD of the 10 v_fma", k_pkfma5, 0}, {"5 v_pk_mul_f32 instead", k_pkmul5, 0},
        {"5 v_pk_add_f32 instead", k_pkadd5, 0},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd; // 256 CUs x (4 SIMDs x waves_per_simd / 4 waves per block)
    double base = 0;
    printf("%d waves per SIMD, %d blocks of 256, %d x 16 groups per thread\n", waves_per_simd, blocks, iters);
    for (auto &k : ks) {
        k.k<<<blocks, 256>>>(out, 10, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k.k<<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // ns per group of 10 v_fma per SIMD, all waves of the SIMD together
        const double ns = ms * 1e6 / ((double)iters * 16 * waves_per_simd);
        if (!base)
            base = ns;
        printf("%-44s %8.3f ms  %7.2f ns per group per SIMD  extra %+6.2f ns = %+5.2f ns per added instruction\n", k.name, ms,
               ns, ns - base, k.extra ? (ns - base) / k.extra : 0.0);
    }
    return 0;
}
