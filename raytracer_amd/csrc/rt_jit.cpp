/*
 * rt_jit.cpp - trace kernels compiled at run time for the SHAPE of the uploaded scene.
 *
 * The fused trace kernels (k_primary, k_bounce, k_subtree in rt_kernels.hip) exist in the library for any scene
 * (SHAPE 0: sphere / polygon / light counts are run-time values) and for the one shape the reference's
 * parse_scene always returns (SHAPE 1, scene.c:86-258: 2 spheres, 1 quad, 1 light - every count a compile-time
 * constant, the loops unrolled, their counters, branches and LDS offsets folded: a quarter of the kernel's time).
 * Any OTHER small scene gets the same treatment here: the kernels' own source - embedded in the library by the
 * Makefile (build/rt_jit_src.inc) - is compiled by hiprtc with the scene's counts as macros (SHAPE 2, fix_shape<2>),
 * about a second per build of the three kernels, on a thread of its own; the context keeps rendering with the
 * generic kernels until the code object is there, then switches (rt_api.hip: render_batch).  Same source, same
 * expressions: the exact build's pixels and counters do not change (tests/test_gpu_jit.py).
 * No counterpart in the reference (its loops run over run-time counts, raytrace.c:188-272).
 */
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <stdio.h>
#include <string.h>
#include <condition_variable>
#include <list>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_device.h"
#include "rt_internal.h"
#include "rt_jit_src.inc" /* static const char rt_jit_src_<file>[]: the sources as they were when the library was built */

namespace {
struct Shape {
    uint32_t nSph, nPoly, nPolySlots, nObj, nLights, tabSlots;
    bool fast;
    bool operator==(const Shape &o) const
    {
        return nSph == o.nSph && nPoly == o.nPoly && nPolySlots == o.nPolySlots && nObj == o.nObj && nLights == o.nLights &&
               tabSlots == o.tabSlots && fast == o.fast;
    }
};
struct Loaded {
    int device;
    hipModule_t module;
    RtJitKernels k;
};
struct Entry {
    Shape sh;
    enum { COMPILING, READY, FAILED } state = COMPILING;
    std::vector<char> code;
    std::string lowered[3];
    std::string log;
    double seconds = 0.0;
    std::list<Loaded> loaded; /* per device (stable addresses: contexts keep pointers to Loaded::k) */
};
std::mutex g_mu;
std::condition_variable g_cv;
std::list<Entry> g_entries;
struct Threads {
    std::vector<std::thread> t;
    ~Threads()
    {
        for (std::thread &x : t)
            if (x.joinable())
                x.join(); /* (a compile takes about a second) */
    }
} g_threads;

void compile(Entry *e)
{
    const Shape &s = e->sh;
    char top[1024];
    /* more than a handful of objects: the seven waves per SIMD of SHAPE 1 (72 registers) would spill the unrolled loops */
    const bool roomy = s.nSph + s.nPoly > 4u || s.nLights > 2u;
    snprintf(top, sizeof top,
             "#define RT_JIT 1\n#define RT_JIT_NSPH %uu\n#define RT_JIT_NPOLY %uu\n#define RT_JIT_NPOLYSLOTS %uu\n"
             "#define RT_JIT_NOBJ %uu\n#define RT_JIT_NLIGHTS %uu\n#define RT_JIT_TABSLOTS %uu\n%s#include \"rt_kernels.hip\"\n",
             s.nSph, s.nPoly, s.nPolySlots, s.nObj, s.nLights, s.tabSlots,
             roomy ? "#define RT_WPE_SHAPE1 6\n#define RT_WPE_BOUNCE1 1\n" : "");
    const char *hsrc[] = {rt_jit_src_rt_kernels_hip, rt_jit_src_rt_device_h, rt_jit_src_rt_arith_h, rt_jit_src_rt_dev_h};
    const char *hname[] = {"rt_kernels.hip", "rt_device.h", "rt_arith.h", "rt_dev.h"};
    const std::string ns = s.fast ? "rt_fast" : "rt_exact";
    const std::string expr[3] = {"&" + ns + "::k_primary<false, false, true, 2, false>", "&" + ns + "::k_bounce<false, false, true, 2>",
                                 "&" + ns + "::k_subtree<false, true, 2>"};
    std::vector<const char *> opt = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize"}; /* csrc/Makefile's flags */
    if (s.fast) {
        opt.push_back("-ffp-contract=fast");
        opt.push_back("-DRT_FAST=1");
    } else {
        opt.push_back("-ffp-contract=off");
    }
    const auto t0 = std::chrono::steady_clock::now();
    hiprtcProgram prog = nullptr;
    bool ok = hiprtcCreateProgram(&prog, top, "rt_jit.hip", 4, hsrc, hname) == HIPRTC_SUCCESS;
    for (int i = 0; ok && i < 3; ++i)
        ok = hiprtcAddNameExpression(prog, expr[i].c_str()) == HIPRTC_SUCCESS;
    std::string log;
    std::vector<char> code;
    std::string lowered[3];
    if (ok) {
        ok = hiprtcCompileProgram(prog, (int)opt.size(), opt.data()) == HIPRTC_SUCCESS;
        size_t n = 0;
        if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) {
            log.resize(n);
            (void)hiprtcGetProgramLog(prog, &log[0]);
        }
    }
    if (ok) {
        size_t n = 0;
        ok = hiprtcGetCodeSize(prog, &n) == HIPRTC_SUCCESS && n > 0;
        if (ok) {
            code.resize(n);
            ok = hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS;
        }
    }
    for (int i = 0; ok && i < 3; ++i) {
        const char *l = nullptr;
        ok = hiprtcGetLoweredName(prog, expr[i].c_str(), &l) == HIPRTC_SUCCESS && l;
        if (ok)
            lowered[i] = l;
    }
    if (prog)
        (void)hiprtcDestroyProgram(&prog);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lk(g_mu);
    e->seconds = dt;
    e->log.swap(log);
    if (ok) {
        e->code.swap(code);
        for (int i = 0; i < 3; ++i)
            e->lowered[i].swap(lowered[i]);
        e->state = Entry::READY;
    } else {
        e->state = Entry::FAILED;
    }
    g_cv.notify_all();
}
} // namespace

/* Kernels compiled for the shape of S (the small class: no grid, tables in LDS), for the current device.
 * start: begin compiling if nobody has; wait: block until the compile has ended.  nullptr while it runs, or when it
 * failed (the generic kernels go on rendering; the compiler's log: rt_jit_status). */
const RtJitKernels *rt_jit_get(const DevScene &S, bool fast, bool start, bool wait)
{
    const Shape sh{S.nSph, S.nPoly, S.nPolySlots, S.nObj, S.nLights, S.tabSlots, fast};
    int device = 0;
    (void)hipGetDevice(&device);
    std::unique_lock<std::mutex> lk(g_mu);
    Entry *e = nullptr;
    for (Entry &x : g_entries)
        if (x.sh == sh)
            e = &x;
    if (!e) {
        if (!start)
            return nullptr;
        g_entries.emplace_back();
        e = &g_entries.back();
        e->sh = sh;
        if (wait) {
            lk.unlock();
            compile(e);
            lk.lock();
        } else {
            g_threads.t.emplace_back(compile, e);
        }
    }
    if (wait)
        g_cv.wait(lk, [&] { return e->state != Entry::COMPILING; });
    if (e->state != Entry::READY)
        return nullptr;
    for (Loaded &l : e->loaded)
        if (l.device == device)
            return l.k.primary ? &l.k : nullptr;
    Loaded l{};
    l.device = device;
    hipFunction_t f[3] = {nullptr, nullptr, nullptr};
    bool ok = hipModuleLoadData(&l.module, e->code.data()) == hipSuccess;
    for (int i = 0; ok && i < 3; ++i)
        ok = hipModuleGetFunction(&f[i], l.module, e->lowered[i].c_str()) == hipSuccess;
    if (ok) {
        l.k.primary = (void *)f[0];
        l.k.bounce = (void *)f[1];
        l.k.subtree = (void *)f[2];
    } else {
        e->log += "\n(loading the code object on device " + std::to_string(device) + " failed)";
    }
    e->loaded.push_back(l);
    return e->loaded.back().k.primary ? &e->loaded.back().k : nullptr;
}

/* 0: no compile was asked for, 1: compiling, 2: ready, -1: failed; `log` (optional) gets the compiler's messages */
int rt_jit_status(const DevScene &S, bool fast, char *log, size_t log_bytes, double *seconds)
{
    const Shape sh{S.nSph, S.nPoly, S.nPolySlots, S.nObj, S.nLights, S.tabSlots, fast};
    std::lock_guard<std::mutex> lk(g_mu);
    for (Entry &x : g_entries)
        if (x.sh == sh) {
            if (log && log_bytes) {
                strncpy(log, x.log.c_str(), log_bytes - 1);
                log[log_bytes - 1] = 0;
            }
            if (seconds)
                *seconds = x.seconds;
            return x.state == Entry::COMPILING ? 1 : x.state == Entry::READY ? 2 : -1;
        }
    if (log && log_bytes)
        log[0] = 0;
    return 0;
}
