/*
 * rt_api.hip - C-ABI of librt_hip.so (include/rt_hip.h): render contexts, AoS -> SoA scene
 * flattening, the per-batch launch sequence, overflow recovery, and the reference-signature
 * raytrace() entry point.  Host code only; kernels live in rt_kernels.hip.
 */
#include "rt_device.h"
#include "rt_hip.h"
#include "rt_internal.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <cmath>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>

namespace {

thread_local char g_err[512] = "";

#define set_err rt_set_err

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -(int)e_;                                                                  \
        }                                                                                     \
    } while (0)

constexpr uint32_t kMaxBatches = 64;            /* counter slots per enqueue round */
constexpr uint64_t kDefaultCap = 64ull << 20;   /* entries per queue / node pool   */
constexpr uint64_t kAutoCapLimit = 256ull << 20; /* ... of a context that sizes itself (38 GiB of the 288) */

/* a run of 8-row tile rows [t0,t1) of the current job, rendered as one image batch */
struct Batch {
    uint32_t t0, t1;
    uint32_t slot;
};

/* which frame rows a job covers (see DevRows in rt_device.h) */
struct Job {
    bool interleaved = false;
    uint32_t row0 = 0, nrows = 0;    /* contiguous slab                           */
    uint32_t phase = 0, nphases = 1; /* cyclic 8-row blocks: block b -> b % nphases */
    uint32_t ntiles = 0;             /* tile rows of the whole job                */
    bool stats = false;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

void rt_set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

struct rt_hip_ctx {
    int device = 0;
    uint64_t cap = 0;
    /* scene */
    bool have_scene = false;
    DevScene S{};
    void *scene_blob = nullptr;
    std::vector<char> scene_host; /* what scene_blob holds */
    /* work buffers */
    void *work_blob = nullptr;
    DevWork W{};
    float4 *pixAcc = nullptr;
    uint8_t *tile_done = nullptr; /* one flag per 8x8 tile of a batch (DevRows::tile_done), allocated with pixAcc */
    size_t pix_cap = 0, tile_cap = 0;
    uint32_t *roots = nullptr; /* [spp^2][pixels of a batch] */
    size_t roots_cap = 0;
    /* the job pipeline of big scenes (rt_jobs.inc): per-sample arrays of a batch, allocated when first needed */
    DevLevel0 Z{};
    float4 *loopq = nullptr; /* DevWork::loopq */
    size_t z_samples = 0, z_vis = 0; /* samples / visibility bytes the arrays hold */
    uint32_t *d_counters = nullptr; /* [kMaxBatches][RT_COUNTER_WORDS] */
    uint32_t *h_counters = nullptr; /* pinned, device-mapped mirror (written by k_publish) */
    uint32_t *h_counters_dev = nullptr; /* its device address */
    bool counters_dirty = false;    /* a render was abandoned midway: blocks may not be zero */
    /* pending render */
    bool pending = false;
    bool pending_stats = false;
    rt_camera cam{};
    DevCamera C{};
    uint32_t *d_argb = nullptr;
    float *d_rad = nullptr;
    Job job;
    hipStream_t stream = nullptr;
    std::vector<Batch> batches;      /* batches in flight (enqueued, not yet checked) */
    std::vector<Batch> todo;         /* batches still to enqueue                      */
    /* options */
    int timing = 0; /* rt_hip_ctx_set_timing: 0 off, 1 events around the dispatch, 2 the kernel's own start and end */
    bool no_shape = false; /* generic kernels only (survives scene uploads) */
    /* kernels compiled at run time for the scene's shape (rt_jit.cpp, rt_hip_ctx_set_jit): the policy, the frames
     * submitted since the scene (or the policy) changed, and what this frame launches (nullptr: the library's own) */
    int jit_mode = RT_JIT_AUTO;
    uint32_t jit_frames = 0;
    const RtJitKernels *jit_cur = nullptr;
    bool no_jobs = false;  /* big scenes through the fused kernels too (rt_hip_ctx_set_pipeline) */
    int shading = RT_SHADE_PHONG;
    int numerics = RT_NUMERICS_EXACT;
    bool auto_cap = false;  /* capacity follows the jobs (rt_hip_ctx_create with 0 entries) */
    uint64_t cap_limit = 0; /* ... up to this many entries */
    /* optional second stream for the latency-bound tail of a batch (rt_hip_ctx_set_tail_stream) */
    hipStream_t tail_stream = nullptr;
    /* "early copy" (the blocking host path, rt_hip_render): a pinned, device-mapped host frame that holds the packed
     * pixels when the batch is done.  k_primary finishes most pixels of a frame (85 % at the reference's scene); one
     * copy of the batch's rows leaves for the host behind it, beside the levels and folds, and k_resolve - which waits
     * for that copy - stores the pixels it finishes both to the device frame and straight into the host's. */
    uint32_t *early_host = nullptr, *early_alias = nullptr; /* host address, device address of the same memory */
    uint8_t *early_tiles = nullptr; /* pinned host copy of tile_done, made with the early copy of a one-batch frame (or nullptr) */
    uint32_t early_tiles_n = 0;     /* flags it holds after early_done (0: none this frame) */
    hipStream_t early_stream = nullptr;
    hipEvent_t early_from = nullptr, early_done = nullptr;
    hipEvent_t head_done = nullptr, tail_done = nullptr;
    bool tail_pending = false;  /* a tail of this context was enqueued and may still run         */
    bool round_on_tail = false; /* the round in flight ends on the tail stream                    */
    uint64_t batch_samples = 0;
    std::vector<hipEvent_t> ev; /* 2 per batch when timing */
    hipEvent_t round_done = nullptr; /* recorded behind each round's counter copy: drain() waits for
                                        THIS context's work only, so two contexts can alternate
                                        on one stream and hide each other's host latency */
    size_t cwords = RT_COUNTER_WORDS; /* counter words in use per batch (depends on depth) */
    rt_hip_stats stats{};
    /* largest per-shard entry counts the last checked batch produced per level (hits queued for the
     * level, nodes made at it): they size the next batch's bounce / fold grids */
    uint32_t seen_hits[RT_MAX_LEVELS] = {0};
    uint32_t seen_nodes[RT_MAX_LEVELS] = {0};
    bool seen_valid = false;
    double seen_density = 0.0; /* entries per sample the fullest shard of the last frame's densest BATCH needed */
    double seen_need = 0.0, seen_samples = 0.0; /* ... and of the whole frame: entries its batches needed, their samples */
    uint64_t seen_key[3] = {0, 0, 0}; /* frame the counts belong to: they are dropped when it changes */
    /*
     * Launch plan of k_primary.  The kernel is as long as the waves that start last and run longest, and the
     * raster order starts some of the slowest tiles of the reference's scene (the floor's edges and its shadow)
     * when 80 of its 122 us are over.  A frame that fits one batch therefore has every wave of k_primary leave
     * the time it took (DevRows::wave_ticks); the next frames of that shape launch the blocks slowest first
     * (DevRows::plan, a permutation of the raster: any plan renders the same pixels): 0.124 -> 0.104 ms.
     * The plan is built once per view (camera, rows, scene); while the view keeps changing the last plan is
     * used as it is and rebuilt every kPlanRefresh frames.
     */
    int launch_plan = 1;           /* rt_hip_ctx_set_launch_plan: 0 off, 1 blocks slowest first */
    uint32_t plan_len = 0;         /* entries of d_plan: the blocks */
    uint32_t *d_plan = nullptr, *d_wave_ticks = nullptr;
    size_t plan_cap = 0;           /* blocks both arrays hold */
    bool plan_valid = false;       /* d_plan holds an order for plan_nbx x plan_ntiles blocks */
    uint32_t plan_nbx = 0, plan_ntiles = 0, plan_age = 0;
    std::vector<char> plan_view;   /* the view the plan's times were taken on */
    std::vector<char> plan_view_now;
    bool plan_use = false, plan_record = false; /* this frame: follow d_plan / record wave times */
    uint64_t scene_serial = 0;     /* counts the uploads that changed the scene */
    /*
     * HIP graph of a frame (rt_hip_ctx_set_graph).  A frame of the fused pipeline is a dozen launches, 50 us of host
     * time - more than the GPU needs for a rank's share of a frame split over eight GPUs.  Once the launches of a
     * frame repeat (same view, buffers, queue sizes, level counts), they are captured from the stream once and
     * replayed as ONE graph launch; anything that changes a launch parameter captures anew.
     */
    int use_graph = 0;
    bool capturing = false;  /* enqueue_batch is being recorded, not run: no timing events */
    uint32_t graph_frame = 0; /* with timing on, every 8th frame is launched the plain way and timed */
    hipGraphExec_t graph_exec = nullptr;
    std::vector<uint64_t> graph_key; /* every launch parameter of the captured frame */
};
constexpr uint32_t kPlanRefresh = 8;

namespace {

int ctx_set_device(rt_hip_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

/* raytrace.c:737,751 + the pixel-independent pieces of :755-763, with the reference's float
 * rounding (this file is compiled with -ffp-contract=off) */
void fold_camera(DevCamera &C, const rt_camera &cam)
{
    const float e[3] = {cam.eyePos.x, cam.eyePos.y, cam.eyePos.z};
    const float U[3] = {cam.U.x, cam.U.y, cam.U.z};
    const float V[3] = {cam.V.x, cam.V.y, cam.V.z};
    const float N[3] = {cam.N.x, cam.N.y, cam.N.z};
    for (int k = 0; k < 3; ++k) {
        C.eye[k] = e[k];
        C.U[k] = U[k];
        C.V[k] = V[k];
        C.shiftN[k] = N[k] * cam.viewDistance;
    }
    C.halfW = cam.viewPlaneHalfWidth;
    C.halfH = cam.viewPlaneHalfHeight;
    C.hfbW = (float)cam.frameBufferWidth / 2.0f;
    C.hfbH = (float)cam.frameBufferHeight / 2.0f;
    C.sppW = cam.sppWidth;
    C.sppHalf = cam.sppWidth / 2.0f;
    C.scale = 1.0f / (float)(cam.sqrtSpp * cam.sqrtSpp);
    C.W = cam.frameBufferWidth;
    C.H = cam.frameBufferHeight;
    C.sqrtSpp = cam.sqrtSpp;
    C.depth = cam.depth;
}

int ensure_pix(rt_hip_ctx *c, size_t npix, size_t spp2)
{
    if (npix > c->pix_cap) {
        if (c->pixAcc)
            HIP_TRY(hipFree(c->pixAcc));
        c->pixAcc = nullptr;
        c->pix_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->pixAcc, npix * sizeof(float4)));
        c->pix_cap = npix;
    }
    /* one flag per tile: the batch's tile rows x ceil(W / 32) blocks x 4 tiles */
    const size_t tiles = c->C.W ? npix / (8u * (size_t)c->C.W) * ((c->C.W + 31u) / 32u) * (RT_BLOCK / 64u) : 0;
    if (tiles > c->tile_cap) {
        if (c->tile_done)
            HIP_TRY(hipFree(c->tile_done));
        c->tile_done = nullptr;
        c->tile_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->tile_done, tiles));
        c->tile_cap = tiles;
    }
    if (npix * spp2 > c->roots_cap) {
        if (c->roots)
            HIP_TRY(hipFree(c->roots));
        c->roots = nullptr;
        c->roots_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->roots, npix * spp2 * sizeof(uint32_t)));
        c->roots_cap = npix * spp2;
    }
    return 0;
}

/* Big scenes (a sphere grid) render through the job pipeline (rt_jobs.inc) - but for the instrumented and the
 * phong-blinn renders, which the fused kernels do, and scenes with more lights than its visibility bytes hold. */
bool jobs_path(const rt_hip_ctx *c, bool stats)
{
    static const bool off = getenv("RT_HIP_NO_JOBS") != nullptr; /* development: the fused kernels for every scene */
    return !off && !c->no_jobs && c->S.grid != nullptr && (c->S.lgrid != nullptr || c->S.nLights == 0u) && !stats && c->shading == RT_SHADE_PHONG &&
           c->S.nLights <= RT_JOBS_MAX_LIGHTS && c->S.nSph <= RT_CHUNK_MAX_SPHERES && c->S.nObj <= RT_CHUNK_MAX_SPHERES; /* (the cell chunks' 21-bit object indices: build_grid) */
}

int ensure_level0(rt_hip_ctx *c, size_t npix, size_t spp2)
{
    const size_t samples = npix * spp2, vis = samples * (c->S.nLights ? c->S.nLights : 1u);
    if (samples > 0x3FFFFFFFull) { /* (a word of the level-0 spawn log: sample | RT_HIT_*) */
        set_err("a batch of %zu samples does not fit the job pipeline's 30-bit sample index", samples);
        return RT_E_INVALID;
    }
    if (samples > c->z_samples) {
        if (c->Z.hit) HIP_TRY(hipFree(c->Z.hit));
        if (c->Z.childR) HIP_TRY(hipFree(c->Z.childR));
        if (c->Z.childT) HIP_TRY(hipFree(c->Z.childT));
        c->Z.hit = nullptr; c->Z.childR = c->Z.childT = nullptr;
        c->z_samples = 0;
        HIP_TRY(hipMalloc((void **)&c->Z.hit, samples * sizeof(uint2)));
        HIP_TRY(hipMalloc((void **)&c->Z.childR, samples * sizeof(float4)));
        HIP_TRY(hipMalloc((void **)&c->Z.childT, samples * sizeof(float4)));
        c->z_samples = samples;
    }
    if (vis > c->z_vis) {
        if (c->Z.vis) HIP_TRY(hipFree(c->Z.vis));
        c->Z.vis = nullptr;
        c->z_vis = 0;
        HIP_TRY(hipMalloc((void **)&c->Z.vis, vis));
        c->z_vis = vis;
    }
    return 0;
}

DevRows rows_of(const rt_hip_ctx *c, const Batch &b)
{
    const Job &j = c->job;
    DevRows R;
    R.ntiles = b.t1 - b.t0;
    if (j.interleaved) {
        R.y_first = (b.t0 * j.nphases + j.phase) * 8u;
        R.y_step = 8u * j.nphases;
        R.y_limit = c->C.H;
        R.out_first = b.t0 * 8u;
    } else {
        R.y_first = R.out_first = j.row0 + 8u * b.t0;
        R.y_step = 8u;
        R.y_limit = j.row0 + j.nrows;
    }
    R.argb = c->d_argb;
    R.radiance = c->d_rad;
    R.tile_done = c->tile_done;
    if (b.t0 == 0 && b.t1 == j.ntiles) { /* the whole job in one batch: what a launch plan is for */
        if (c->plan_use) {
            R.plan = c->d_plan;
            R.plan_len = c->plan_len;
        }
        if (c->plan_record)
            R.wave_ticks = c->d_wave_ticks;
    }
    return R;
}

/* First level of a batch that goes to the tail stream: RT_HIP_TAIL_LEVEL in the environment, default 3
 * (levels 1 and 2 of the reference's scene still fill the GPU; 4 is 0.8 % better there, 2 and 5 worse; a
 * rule from the last batch's level sizes - "fewer entries than one block per CU" - picked worse: 0.358
 * against 0.344 ms).  depth + 1 = only the folds and the resolve. */
uint32_t tail_level(uint32_t depth)
{
    static const uint32_t fixed = [] {
        const char *e = getenv("RT_HIP_TAIL_LEVEL");
        const uint32_t t = e && *e ? (uint32_t)strtoul(e, nullptr, 10) : 3u;
        return t < 1u ? 1u : t;
    }();
    return fixed > depth + 1u ? depth + 1u : fixed;
}

/* k_fold_deep takes the levels that hold at most this many nodes per shard (RT_HIP_DEEP_MAX in the
 * environment; 0 = never) */
uint32_t deep_entries()
{
    static const uint32_t v = [] {
        const char *e = getenv("RT_HIP_DEEP_MAX");
        return e && *e ? (uint32_t)strtoul(e, nullptr, 10) : 2048u;
    }();
    return v;
}

/* k_subtree takes the levels that hold at most this many hits per shard (RT_HIP_SUBTREE_MAX in the
 * environment; 0 = never).  A hit of a scene with a grid costs ten walks through it: there a lane that
 * finishes a whole subtree alone is the slow way until the levels are nearly empty (config 5's scene at
 * 1080p, frames pipelined: 59.0 ms with 0 or 64, 60.5 with 512, 61.2 with 4096; its k_subtree took 11.9 ms
 * where the launches per level it replaced took 8). */
uint32_t subtree_entries(bool grid)
{
    static const long fixed = [] {
        const char *e = getenv("RT_HIP_SUBTREE_MAX");
        return e && *e ? (long)strtoul(e, nullptr, 10) : -1L;
    }();
    return fixed >= 0 ? (uint32_t)fixed : grid ? 64u : 4096u;
}

/* enqueue one image batch */
int enqueue_batch(rt_hip_ctx *c, const Batch &b, bool stats)
{
    const DevRows R = rows_of(c, b);
    DevWork Wk = c->W;
    Wk.pixAcc = c->pixAcc;
    Wk.roots = c->roots;
    Wk.counters = c->d_counters + (size_t)b.slot * c->cwords;
    hipStream_t st = c->stream;
    const uint32_t depth = c->C.depth;
    /* With a tail stream the batch is cut behind the wide levels: k_primary and the bounces above
     * tail_level() stay on the render stream, everything after them - the sparse levels, the folds, the
     * resolve - goes to the tail stream behind an event, so that the render stream is free for the next
     * frame's k_primary while this batch's latency-bound remainder (0.14 of its 0.43 ms) drains beside it. */
    const uint32_t cut = (c->tail_stream && c->tail_stream != st && !stats) ? tail_level(depth) : 0u;
    if (cut && c->tail_pending) /* the queues and node pool are shared with this context's previous batch */
        HIP_TRY(hipStreamWaitEvent(st, c->tail_done, 0));
    /* the counter block is zero: k_publish at the end of its last use left it so */
    const bool timed = c->timing != 0 && !stats && !c->capturing;
    if (timed) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        c->ev.push_back(e0);
        c->ev.push_back(e1);
        if (jobs_path(c, stats) || c->timing == 1) /* (the job pipeline's level 0 is several kernels: always bracketed) */
            HIP_TRY(hipEventRecord(e0, st));
    }
    /* the tolerance build has the plain Phong kernels only: instrumented and phong-blinn renders run the exact ones */
    const bool fast = c->numerics == RT_NUMERICS_FAST && !stats && c->shading == RT_SHADE_PHONG;
    if (jobs_path(c, stats)) {
        /* Big scenes: rays as jobs (rt_jobs.inc).  The samples' closest hits, then per level the shadow and child
         * rays of its hits, then the colours from the deepest level up, then the pixels; all on the render stream.
         * The events bracket the level-0 work the fused k_primary would do: the primary rays and the rays of
         * their hits. */
        DevLevel0 Z = c->Z;
        Z.npix = R.ntiles * 8u * c->C.W;
        Z.nsamples = Z.npix * c->C.sqrtSpp * c->C.sqrtSpp;
        const bool known = c->seen_valid;
        /* Every level down to the depth asked for is launched, whatever the last frame of this shape held there: its
         * sizes only choose the grids (an empty level's kernels return at once, ~30 us a level).  Launching only the
         * levels that frame had used - plus one - lost the deeper hits of the next frame when that was another view:
         * another row slab of the same height, a camera that moved (found by tools/fuzz_big.py, FUZZ_CTX=slabs). */
        const uint32_t last = depth;
        (fast ? rt_launch_jobs_hits_fast : rt_launch_jobs_hits)(c->S, c->C, Wk, R, Z, st);
        for (uint32_t d = 0; d <= last; ++d) {
            const uint32_t expect = d == 0u ? 0u : known ? (c->seen_hits[d] ? c->seen_hits[d] : 1u) : 0u;
            (fast ? rt_launch_jobs_trace_fast : rt_launch_jobs_trace)(c->S, c->C, Wk, R, Z, d, expect, d != depth, st);
            if (d == 0u && timed)
                HIP_TRY(hipEventRecord(c->ev.back(), st));
        }
        for (uint32_t d = last; d >= 1u; --d) {
            const uint32_t expect = known ? (c->seen_hits[d] ? c->seen_hits[d] : 1u) : 0u;
            (fast ? rt_launch_jobs_shadefold_fast : rt_launch_jobs_shadefold)(c->S, c->C, Wk, Z, d, expect, st);
        }
        (fast ? rt_launch_jobs_resolve_fast : rt_launch_jobs_resolve)(c->S, c->C, Wk, R, Z, st);
        if (c->early_alias && c->d_argb && !c->job.interleaved) { /* every pixel is finished by the resolve: the rows follow it */
            const size_t off = (size_t)R.out_first * c->C.W;
            const uint32_t rows = (!c->job.interleaved && R.y_first + R.ntiles * 8u > R.y_limit) ? R.y_limit - R.y_first : R.ntiles * 8u;
            HIP_TRY(hipMemcpyAsync(c->early_host + off, c->d_argb + off, (size_t)rows * c->C.W * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        }
        rt_launch_publish(Wk.counters, c->h_counters_dev + (size_t)b.slot * c->cwords, (uint32_t)c->cwords, st);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (R.plan)
        c->stats.planned_launches++;
    /* timing 2: the kernel's own start and end (hipExtLaunchKernel) instead of two events around its dispatch, which read
     * its queue latency too - but cost the host ~25 us a launch, which frames of under 0.1 ms cannot afford (timing 1) */
    const bool own = timed && c->timing == 2;
    hipEvent_t t0 = own ? c->ev[c->ev.size() - 2] : nullptr, t1 = own ? c->ev.back() : nullptr;
    if (fast)
        rt_launch_primary_fast(c->S, c->C, Wk, R, st, c->jit_cur, t0, t1);
    else
        rt_launch_primary(c->S, c->C, Wk, R, stats, c->shading == RT_SHADE_PHONGBLINN, st, c->jit_cur, t0, t1);
    if (timed && !own)
        HIP_TRY(hipEventRecord(c->ev.back(), st));
    const bool early = c->early_alias && !stats && c->d_argb && !c->job.interleaved;
    const size_t early_off = (size_t)R.out_first * c->C.W;
    const uint32_t early_rows = (!c->job.interleaved && R.y_first + R.ntiles * 8u > R.y_limit) ? R.y_limit - R.y_first : R.ntiles * 8u;
    if (early) {
        HIP_TRY(hipEventRecord(c->early_from, st));
        HIP_TRY(hipStreamWaitEvent(c->early_stream, c->early_from, 0));
        HIP_TRY(hipMemcpyAsync(c->early_host + early_off, c->d_argb + early_off, (size_t)early_rows * c->C.W * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, c->early_stream));
        /* ... and, when the batch is the whole frame, which tiles those final pixels are: the host moves the tile rows
         * that are final throughout into the caller's buffer while the levels still run (rt_hip_render) */
        c->early_tiles_n = 0;
        if (c->early_tiles && R.tile_done && R.ntiles == c->job.ntiles) {
            c->early_tiles_n = R.ntiles * ((c->C.W + 31u) / 32u) * (RT_BLOCK / 64u);
            HIP_TRY(hipMemcpyAsync(c->early_tiles, R.tile_done, c->early_tiles_n, hipMemcpyDeviceToHost, c->early_stream));
        }
        HIP_TRY(hipEventRecord(c->early_done, c->early_stream));
    }
    auto to_tail = [&]() -> int {
        HIP_TRY(hipEventRecord(c->head_done, st));
        st = c->tail_stream;
        HIP_TRY(hipStreamWaitEvent(st, c->head_done, 0));
        return 0;
    };
    /* The sparse deep end of the tree - from the first level on that the last batch of this shape left
     * with few enough hits per shard - is ONE launch in which every lane finishes the whole subtree of a
     * hit (k_subtree): no launch per level, no nodes and no folds there.  The first render of a shape, the
     * instrumented kernels and trees deeper than the kernel's stack take one launch per level. */
    uint32_t sub = depth + 1u; /* first level k_subtree takes (depth + 1: none) */
    const uint32_t sub_max = subtree_entries(c->S.grid != nullptr);
    if (c->seen_valid && !stats && c->shading == RT_SHADE_PHONG && sub_max != 0u) {
        uint32_t l = depth;
        while (l > 2u && c->seen_hits[l - 1u] <= sub_max && depth - (l - 1u) < RT_SUBTREE_LEVELS)
            --l;
        if (l < depth && c->seen_hits[l] <= sub_max && depth - l < RT_SUBTREE_LEVELS)
            sub = l;
    }
    bool moved = false;
    for (uint32_t d = 1; d <= depth && d <= sub; ++d) {
        if (d == cut) {
            const int rc = to_tail();
            if (rc)
                return rc;
            moved = true;
        }
        const uint32_t expect = c->seen_valid ? (c->seen_hits[d] ? c->seen_hits[d] : 1u) : 0u;
        if (d == sub) {
            if (fast)
                rt_launch_subtree_fast(c->S, c->C, Wk, d, expect, st, c->jit_cur);
            else
                rt_launch_subtree(c->S, c->C, Wk, d, expect, st, c->jit_cur);
        } else if (fast) {
            rt_launch_bounce_fast(c->S, c->C, Wk, d, expect, st, c->jit_cur);
        } else {
            rt_launch_bounce(c->S, c->C, Wk, d, stats, expect, st, c->jit_cur);
        }
    }
    if (cut && !moved) { /* (a cut below the last kernel launched: the folds and the resolve go to the tail) */
        const int rc = to_tail();
        if (rc)
            return rc;
    }
    /* Nodes exist on levels 0..depth-1 (0..sub-1 with k_subtree); level 0 folds in resolve.  The sparse
     * deep levels - from the first level on that the last batch of this shape left with few enough nodes
     * per shard - fold in one launch, the others in one launch each. */
    const uint32_t node_levels = sub <= depth ? sub : depth; /* levels 0 .. node_levels - 1 hold nodes */
    uint32_t fold_from = node_levels;
    if (c->seen_valid && deep_entries() != 0u && node_levels >= 3u) {
        uint32_t bottom = node_levels;
        while (bottom > 1u && c->seen_nodes[bottom - 1u] <= deep_entries())
            --bottom;
        if (bottom + 1u < node_levels) { /* at least two levels */
            rt_launch_fold_deep(Wk, node_levels - 1u, bottom, st);
            fold_from = bottom;
        }
    }
    for (uint32_t d = fold_from; d-- > 1;)
        rt_launch_fold(Wk, d, c->seen_valid ? (c->seen_nodes[d] ? c->seen_nodes[d] : 1u) : 0u, st);
    if (early)
        HIP_TRY(hipStreamWaitEvent(st, c->early_done, 0));
    rt_launch_resolve(c->C, Wk, R, c->d_argb, c->d_rad, early ? c->early_alias : nullptr, st);
    if (c->early_alias && !early && c->d_argb && !c->job.interleaved) /* (instrumented renders: the rows follow the resolve) */
        HIP_TRY(hipMemcpyAsync(c->early_host + early_off, c->d_argb + early_off, (size_t)early_rows * c->C.W * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
    rt_launch_publish(Wk.counters, c->h_counters_dev + (size_t)b.slot * c->cwords, (uint32_t)c->cwords, st);
    if (cut) {
        HIP_TRY(hipEventRecord(c->tail_done, st));
        c->tail_pending = true;
        c->round_on_tail = true;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

uint32_t tiles_per_batch(const rt_hip_ctx *c, uint64_t max_samples)
{
    const uint64_t per_tile = 8ull * c->C.W * c->C.sqrtSpp * c->C.sqrtSpp;
    uint64_t t = per_tile ? max_samples / per_tile : 1;
    if (t < 1)
        t = 1;
    return (uint32_t)(t > 0x0FFFFFFFull ? 0x0FFFFFFFull : t);
}

bool batch_overflowed(const uint32_t *cnt) { return cnt[CNT_OVERFLOW] != 0u; } /* one flag per batch: rt_device.h */

/* remember how full the level queues / node ranges of a batch got (grid sizing of the next one) */
void note_levels(rt_hip_ctx *c, const uint32_t *cnt, uint32_t levels, bool first, uint64_t samples)
{
    if (first) {
        memset(c->seen_hits, 0, sizeof c->seen_hits);
        memset(c->seen_nodes, 0, sizeof c->seen_nodes);
        c->seen_density = 0.0;
        c->seen_need = c->seen_samples = 0.0;
    }
    /* entries the fullest shard needed per sample of this batch: its node range holds the nodes
     * of all levels, a queue the hits of one level (the next frame's batches are planned with it) */
    uint32_t need = 0;
    for (uint32_t sh = 0; sh < RT_SHARDS; ++sh) {
        uint32_t nodes = 0;
        for (uint32_t l = 0; l < levels; ++l) {
            const uint64_t v = *(const uint64_t *)(cnt + CNT_LEVELS + (l * RT_SHARDS + sh) * RT_LINE_WORDS);
            /* (the job pipeline's level-0 word counts the entries of k_hits' spawn log, which has slots of its own
             * and as many entries as samples at most) */
            nodes += (l == 0u && jobs_path(c, c->job.stats)) ? 0u : (uint32_t)v;
            if ((uint32_t)(v >> 32) > need)
                need = (uint32_t)(v >> 32);
        }
        if (nodes > need)
            need = nodes;
    }
    if (samples) {
        const double density = (double)need * RT_SHARDS / (double)samples;
        if (density > c->seen_density)
            c->seen_density = density;
        c->seen_need += (double)need * RT_SHARDS; /* (a frame in one batch needs at most the sum of its batches' needs) */
        c->seen_samples += (double)samples;
    }
    for (uint32_t l = 0; l < levels && l + 1 < RT_MAX_LEVELS; ++l)
        for (uint32_t sh = 0; sh < RT_SHARDS; ++sh) {
            const uint64_t v = *(const uint64_t *)(cnt + CNT_LEVELS + (l * RT_SHARDS + sh) * RT_LINE_WORDS);
            const uint32_t nodes = (uint32_t)v, hits = (uint32_t)(v >> 32);
            if (nodes > c->seen_nodes[l])
                c->seen_nodes[l] = nodes;
            if (hits > c->seen_hits[l + 1])
                c->seen_hits[l + 1] = hits;
        }
    c->seen_valid = true;
}

void accumulate(rt_hip_stats &s, const uint32_t *cnt, uint32_t levels)
{
    /* (the old single-counter layout is gone: see rt_device.h "Counter block") */
    for (uint32_t l = 0; l < levels; ++l)
        for (uint32_t sh = 0; sh < RT_SHARDS; ++sh) {
            const uint64_t v = *(const uint64_t *)(cnt + CNT_LEVELS + (l * RT_SHARDS + sh) * RT_LINE_WORDS);
            s.tree_nodes += (uint32_t)v;
            s.queued_rays += (uint32_t)(v >> 32);
        }
    uint64_t st[ST_COUNT] = {0};
    for (uint32_t sh = 0; sh < RT_SHARDS; ++sh) {
        const uint64_t *p = (const uint64_t *)(cnt + CNT_STATS + sh * CNT_STATS_STRIDE);
        for (int k = 0; k < ST_COUNT; ++k)
            st[k] += p[k];
    }
    s.shade_nodes += st[ST_K0_SHADE_NODES] + st[ST_KB_SHADE_NODES];
    s.reflection_rays += st[ST_K0_REFL] + st[ST_KB_REFL];
    s.transmission_rays += st[ST_K0_TRANS] + st[ST_KB_TRANS];
    s.k0_shade_nodes += st[ST_K0_SHADE_NODES];
    s.k0_reflection_rays += st[ST_K0_REFL];
    s.k0_transmission_rays += st[ST_K0_TRANS];
    s.sphere_tests += st[ST_SPH_TESTS];
    s.sphere_hits += st[ST_SPH_HITS];
    s.polygon_tests += st[ST_POLY_TESTS];
    s.polygon_rej_parallel += st[ST_POLY_PAR];
    s.polygon_rej_behind += st[ST_POLY_BEHIND];
    s.polygon_full += st[ST_POLY_FULL];
    s.polygon_hits += st[ST_POLY_HITS];
    s.lights_lit += st[ST_LIT];
    s.nodes_sphere += st[ST_NODES_SPH];
    s.nodes_polygon += st[ST_NODES_POLY];
    s.polygon_screen_unsure += st[ST_POLY_SCREEN_UNSURE];
    s.polygon_screen_mismatch += st[ST_POLY_SCREEN_MISMATCH];
    s.k0_sphere_tests += st[ST_K0_SPH_TESTS];
    s.k0_sphere_hits += st[ST_K0_SPH_HITS];
    s.k0_polygon_tests += st[ST_K0_POLY_TESTS];
    s.k0_polygon_rej_parallel += st[ST_K0_POLY_PAR];
    s.k0_polygon_rej_behind += st[ST_K0_POLY_BEHIND];
    s.k0_polygon_full += st[ST_K0_POLY_FULL];
    s.k0_lights_lit += st[ST_K0_LIT];
    s.k0_nodes_sphere += st[ST_K0_NODES_SPH];
    s.k0_nodes_polygon += st[ST_K0_NODES_POLY];
    s.primary_rays_culled += st[ST_K0_CULLED];
}

/* A frame that is one batch of the fused pipeline on one stream, uninstrumented, its level counts known: as a graph.
 * The key lists what the launches depend on; equal key = the captured graph is this frame's. */
int enqueue_batch_graphed(rt_hip_ctx *c, const Batch &b, bool stats)
{
    const bool whole = b.t0 == 0 && b.t1 == c->job.ntiles && b.slot == 0;
    if (!c->use_graph || !c->stream /* (the null stream cannot be captured) */ || stats || !whole || (c->timing && (c->graph_frame++ & 7u) == 0u) || !c->seen_valid || c->plan_record || c->early_alias ||
        (c->tail_stream && c->tail_stream != c->stream) || jobs_path(c, stats))
        return enqueue_batch(c, b, stats);
    std::vector<uint64_t> key;
    key.reserve(96);
    auto bytes = [&](const void *p, size_t n) {
        const size_t at = key.size();
        key.resize(at + (n + 7) / 8, 0);
        memcpy(key.data() + at, p, n);
    };
    bytes(&c->C, sizeof c->C);
    const Job &j = c->job;
    const uint64_t rest[] = {c->scene_serial, (uint64_t)c->numerics, (uint64_t)c->shading, j.interleaved ? 1u : 0u,
                             ((uint64_t)j.row0 << 32) | j.nrows, ((uint64_t)j.phase << 32) | j.nphases, j.ntiles,
                             (uint64_t)(uintptr_t)c->d_argb, (uint64_t)(uintptr_t)c->d_rad, (uint64_t)(uintptr_t)c->W.base,
                             c->W.stride, c->W.cap, (uint64_t)(uintptr_t)c->pixAcc, (uint64_t)(uintptr_t)c->roots,
                             (uint64_t)(uintptr_t)c->tile_done, (uint64_t)(uintptr_t)c->d_counters,
                             (uint64_t)(uintptr_t)c->scene_blob, (uint64_t)(uintptr_t)(c->plan_use ? c->d_plan : nullptr),
                             c->plan_use ? c->plan_len : 0u, (uint64_t)c->no_shape, (uint64_t)(uintptr_t)c->jit_cur, (uint64_t)c->cwords,
                             (uint64_t)c->stats.plan_builds};
    bytes(rest, sizeof rest);
    bytes(c->seen_hits, sizeof c->seen_hits);
    bytes(c->seen_nodes, sizeof c->seen_nodes);
    if (!c->graph_exec || key != c->graph_key) {
        if (c->graph_exec) {
            (void)hipGraphExecDestroy(c->graph_exec);
            c->graph_exec = nullptr;
        }
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        c->capturing = true;
        const int rc = enqueue_batch(c, b, stats);
        c->capturing = false;
        const hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc || e != hipSuccess || !g) {
            if (g)
                (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            if (rc)
                return rc;
            set_err("capturing the frame's launches failed: %s", hipGetErrorString(e));
            return -(int)e;
        }
        const hipError_t ei = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) {
            c->graph_exec = nullptr;
            set_err("instantiating the frame's graph failed: %s", hipGetErrorString(ei));
            return -(int)ei;
        }
        c->graph_key.swap(key);
        c->stats.graph_captures++;
    }
    HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
    c->stats.graph_launches++;
    return 0;
}

/* Enqueue up to kMaxBatches batches of the work list (one counter slot each) and the copy of
 * their counters to the pinned mirror.  Asynchronous. */
int enqueue_round(rt_hip_ctx *c)
{
    c->batches.clear();
    c->round_on_tail = false;
    while (!c->todo.empty() && c->batches.size() < kMaxBatches) {
        Batch b = c->todo.back();
        c->todo.pop_back();
        b.slot = (uint32_t)c->batches.size();
        int rc = enqueue_batch_graphed(c, b, c->job.stats);
        if (rc)
            return rc;
        c->batches.push_back(b);
    }
    if (!c->batches.empty()) { /* (each batch ended by publishing its counters to h_counters) */
        if (!c->round_done)
            HIP_TRY(hipEventCreateWithFlags(&c->round_done, hipEventDisableTiming));
        /* (tails run in batch order on one stream: the last batch's tail ends the round) */
        HIP_TRY(hipEventRecord(c->round_done, c->round_on_tail ? c->tail_stream : c->stream));
    }
    return 0;
}

/* The frame's single batch recorded its waves' times: order the blocks slowest first and put the order where
 * k_primary reads it.  Blocking; once per view. */
int rebuild_plan(rt_hip_ctx *c)
{
    const uint32_t nbx = (c->C.W + 31u) / 32u, ntiles = c->job.ntiles;
    const size_t blocks = (size_t)nbx * ntiles;
    std::vector<uint32_t> ticks(blocks * (RT_BLOCK / 64u));
    HIP_TRY(hipMemcpyAsync(ticks.data(), c->d_wave_ticks, ticks.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> plan(blocks);
    rt_launch_plan_order(ticks.data(), nbx, ntiles, plan.data());
    c->plan_len = (uint32_t)plan.size();
    HIP_TRY(hipMemcpyAsync(c->d_plan, plan.data(), plan.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); /* (plan[] is this function's) */
    c->plan_valid = true;
    c->plan_nbx = nbx;
    c->plan_ntiles = ntiles;
    c->plan_age = 0;
    c->plan_view = c->plan_view_now;
    c->stats.plan_builds++;
    return 0;
}

/* Wait for the round in flight; batches whose queues overflowed go back on the work list as two
 * halves, the others add their counters to the statistics.  Repeats until the list is empty. */
int ensure_work(rt_hip_ctx *c, uint64_t cap);
int ensure_loopq(rt_hip_ctx *c);
uint64_t round_cap(uint64_t cap);

int drain(rt_hip_ctx *c)
{
    for (;;) {
        bool grow = false;
        if (c->round_done && !c->batches.empty()) {
            HIP_TRY(hipEventSynchronize(c->round_done));
        } else {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->tail_pending)
                HIP_TRY(hipStreamSynchronize(c->tail_stream));
        }
        c->tail_pending = false; /* (the round's last event covers every tail enqueued before it) */
        for (size_t i = 0; i < c->batches.size(); ++i) {
            const Batch &b = c->batches[i];
            const uint32_t *cnt = c->h_counters + i * c->cwords;
            if (batch_overflowed(cnt)) {
                const uint32_t n = b.t1 - b.t0;
                if (n <= 1 && c->auto_cap && c->cap < c->cap_limit) {
                    /* one tile row needs more than the context has grown to yet (a wide frame with many samples a
                     * pixel: 3.9 M samples a row at config 5): twice the queues, and again */
                    grow = true;
                    c->stats.overflow_retries++;
                    c->todo.push_back(b);
                    continue;
                }
                if (n <= 1) {
                    set_err("ray/node queues (%llu entries) overflow on a single 8-row tile row; raise the capacity",
                            (unsigned long long)c->cap);
                    c->batches.clear();
                    c->todo.clear();
                    return RT_E_OVERFLOW;
                }
                c->stats.overflow_retries++;
                const uint32_t mid = b.t0 + n / 2;
                c->todo.push_back({mid, b.t1, 0});
                c->todo.push_back({b.t0, mid, 0});
            } else {
                accumulate(c->stats, cnt, c->C.depth + 1);
                note_levels(c, cnt, c->C.depth + 1, c->stats.batches == 0,
                            (uint64_t)(b.t1 - b.t0) * 8u * c->C.W * c->C.sqrtSpp * c->C.sqrtSpp);
                c->stats.batches++;
                if (c->plan_record && b.t0 == 0 && b.t1 == c->job.ntiles) {
                    c->plan_record = false;
                    const int rc = rebuild_plan(c);
                    if (rc)
                        return rc;
                }
            }
        }
        c->batches.clear();
        if (c->todo.empty())
            return 0;
        if (grow) { /* (the round is over: nothing of this context is in flight) */
            const uint64_t twice = c->cap * 2u > c->cap_limit ? c->cap_limit : c->cap * 2u;
            int rc = ensure_work(c, round_cap(twice));
            if (!rc && jobs_path(c, c->job.stats))
                rc = ensure_loopq(c);
            if (rc)
                return rc;
        }
        int rc = enqueue_round(c);
        if (rc)
            return rc;
    }
}

void finish_stats(rt_hip_ctx *c)
{
    rt_hip_stats &s = c->stats;
    const Job &j = c->job;
    uint64_t rows = j.nrows;
    if (j.interleaved) {
        rows = 0;
        for (uint32_t t = 0; t < j.ntiles; ++t) {
            const uint32_t y = (t * j.nphases + j.phase) * 8u;
            rows += (c->C.H - y < 8u) ? c->C.H - y : 8u;
        }
    }
    s.primary_rays = (uint64_t)c->C.W * rows * c->C.sqrtSpp * c->C.sqrtSpp;
    s.shadow_rays = s.shade_nodes * c->S.nLights;
    s.k0_shadow_rays = s.k0_shade_nodes * c->S.nLights;
    /* device memory the context holds for its work: queues and nodes, per-pixel / per-sample arrays, loop queues */
    const uint64_t bytes = (c->work_blob ? c->W.stride * 9u + c->W.stride / 2u : 0u) + c->pix_cap * sizeof(float4) + c->tile_cap +
                           c->roots_cap * sizeof(uint32_t) + c->z_samples * (sizeof(uint2) + 2u * sizeof(float4)) + c->z_vis +
                           (uint64_t)RT_SHARDS * c->W.loop_capS * 4u * sizeof(float4);
    s.work_mib = (uint32_t)((bytes + (1u << 20) - 1u) >> 20);
}

int check_args(rt_hip_ctx *c, const rt_camera *cam, uint32_t *d_argb, uint32_t row0, uint32_t nrows)
{
    if (!c || !cam || !d_argb) {
        set_err("null argument");
        return RT_E_INVALID;
    }
    if (!c->have_scene) {
        set_err("no scene uploaded");
        return RT_E_NOSCENE;
    }
    if (cam->frameBufferWidth == 0 || cam->frameBufferHeight == 0 || cam->sqrtSpp == 0) {
        set_err("empty frame or zero samples per pixel");
        return RT_E_INVALID;
    }
    if (cam->depth > RT_MAX_LEVELS - 2) {
        set_err("depth %u exceeds the supported maximum %d", cam->depth, RT_MAX_LEVELS - 2);
        return RT_E_INVALID;
    }
    if ((uint64_t)row0 + nrows > cam->frameBufferHeight) {
        set_err("rows [%u,%u) outside the %u-row frame", row0, row0 + nrows, cam->frameBufferHeight);
        return RT_E_INVALID;
    }
    if ((uint64_t)cam->sqrtSpp * cam->sqrtSpp > 0xFFFFu) {
        set_err("sqrtSpp too large");
        return RT_E_INVALID;
    }
    return 0;
}

uint64_t default_batch_samples(const rt_hip_ctx *c)
{
    /* A sample makes at most one root node, and its children queue at most two level-1 hits; scenes
     * in which more than half the samples queue two hits are rare, and a batch that does outgrow
     * the queues is split and rendered again (drain), so the plan is one sample per entry. */
    uint64_t m = c->cap;
    /* ... unless the last frame of the same shape showed what this scene needs per sample (the reference's
     * scene 0.08 entries, mirror-heavy sphere clouds up to 2): plan for that, with 15 % to spare, instead of
     * rendering an oversized batch in vain first - or many small ones where one would do */
    if (c->seen_valid && c->seen_density > 0.0) {
        /* the whole frame if it fits as one batch (its own entries per sample), else batches sized for the densest */
        const double frame = c->seen_samples > 0.0 ? c->seen_need / c->seen_samples : c->seen_density;
        const double all = (double)c->job.ntiles * 8.0 * c->C.W * c->C.sqrtSpp * c->C.sqrtSpp;
        const double per = all * frame * 1.15 <= (double)c->cap ? all + 1.0 : (double)c->cap / (c->seen_density * 1.15);
        m = per > 1.8e19 ? ~0ull : (uint64_t)per;
    }
    if (c->batch_samples)
        m = c->batch_samples; /* the caller's plan, also when it is more than the queues hold (overflow: halved and rendered again) */
    return m ? m : 1;
}

uint64_t round_cap(uint64_t cap)
{
    if (cap < 1024)
        cap = 1024;
    if (cap > (1ull << 28))
        cap = 1ull << 28; /* node links are 29-bit, ray parents 31-bit */
    return (cap + RT_SHARDS - 1) / RT_SHARDS * RT_SHARDS; /* whole shards */
}

/* (re)allocate the queues and the node pool for `cap` entries: one blob of 6 queue arrays + 3 node arrays of
 * cap float4 each + one of cap float2.  Nothing of this context may be in flight.  The old blob is kept until the
 * new one exists when the device has room for both; when even the new one alone does not fit (a shared GPU, a
 * caching allocator holding the memory), a context that sizes itself settles for half, a quarter, ... down to
 * 1 Mi entries - its frames are then rendered in row batches - and remembers the limit. */
int ensure_work(rt_hip_ctx *c, uint64_t cap)
{
    if (c->work_blob && c->cap == cap)
        return 0;
    for (;;) {
        const size_t arr = align_up(cap * sizeof(float4), 256), bytes = arr * 9 + arr / 2;
        void *blob = nullptr;
        hipError_t e = hipMalloc(&blob, bytes);
        if (e == hipErrorOutOfMemory && c->work_blob) { /* not beside the old one: without it, then */
            (void)hipGetLastError();
            HIP_TRY(hipFree(c->work_blob)); /* (waits for the device) */
            c->work_blob = nullptr;
            c->cap = 0;
            e = hipMalloc(&blob, bytes);
        }
        if (e == hipSuccess) {
            if (c->work_blob)
                HIP_TRY(hipFree(c->work_blob));
            c->work_blob = blob;
            c->cap = cap;
            c->W.base = (char *)blob; /* slots 0..9, the last one (float2) half as long: rt_device.h */
            c->W.stride = arr;
            c->W.cap = (uint32_t)cap;
            c->W.capS = (uint32_t)(cap / RT_SHARDS);
            break;
        }
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || !c->auto_cap || cap <= (1ull << 20)) {
            c->work_blob = nullptr;
            c->cap = 0;
            set_err("allocating %zu bytes of queue memory (%llu entries) on device %d failed: %s", bytes,
                    (unsigned long long)cap, c->device, hipGetErrorString(e));
            return -(int)e;
        }
        cap = round_cap(cap / 2 < (1ull << 20) ? (1ull << 20) : cap / 2);
        c->cap_limit = cap; /* larger jobs go in batches from now on */
    }
    /* the job pipeline's loop queues (big scenes only) follow the capacity: ensure_loopq */
    if (c->loopq) {
        HIP_TRY(hipFree(c->loopq));
        c->loopq = nullptr;
    }
    c->W.loopq = nullptr;
    c->W.loop_capS = 0;
    c->seen_valid = false;
    return 0;
}

int ensure_loopq(rt_hip_ctx *c)
{
    if (c->loopq || !c->cap)
        return 0;
    const uint32_t lcap = (uint32_t)(c->cap / RT_SHARDS / 8u) < 256u ? 256u : (uint32_t)(c->cap / RT_SHARDS / 8u);
    HIP_TRY(hipMalloc((void **)&c->loopq, (size_t)RT_SHARDS * lcap * 4u * sizeof(float4)));
    c->W.loopq = c->loopq;
    c->W.loop_capS = lcap;
    return 0;
}

void drop_events(rt_hip_ctx *c)
{
    for (hipEvent_t e : c->ev)
        (void)hipEventDestroy(e);
    c->ev.clear();
}

} // namespace

extern "C" {

/* k_primary's launch order from the times its waves took (wave_ticks[(tile row * nbx + block column) * 4 + wave]):
 * the blocks slowest first - a block is as slow as its slowest wave - equal times in raster order (a stable counting
 * sort on at most 2^16 keys: times beyond that lose their low bits).  plan[i] = block column | tile row << 15. */
void rt_launch_plan_order(const uint32_t *wave_ticks, uint32_t nbx, uint32_t ntiles, uint32_t *plan)
{
    const size_t blocks = (size_t)nbx * ntiles;
    constexpr uint32_t waves = RT_BLOCK / 64u;
    std::vector<uint32_t> cost(blocks);
    uint32_t most = 0;
    for (size_t b = 0; b < blocks; ++b) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < waves; ++w)
            t = wave_ticks[b * waves + w] > t ? wave_ticks[b * waves + w] : t;
        cost[b] = t;
        most = t > most ? t : most;
    }
    uint32_t shift = 0;
    while ((most >> shift) >= (1u << 16))
        ++shift;
    std::vector<uint32_t> start((1u << 16) + 1u, 0u);
    for (size_t b = 0; b < blocks; ++b)
        start[0xffffu - (cost[b] >> shift) + 1u]++;
    for (size_t k = 1; k < start.size(); ++k)
        start[k] += start[k - 1];
    for (size_t b = 0; b < blocks; ++b)
        plan[start[0xffffu - (cost[b] >> shift)]++] = (uint32_t)(b % nbx) | ((uint32_t)(b / nbx) << 15);
}


const char *rt_hip_last_error(void) { return g_err; }
const char *rt_hip_version(void) { return "rt_hip 0.1 (gfx950)"; }

int rt_hip_deinterleave(const uint32_t *d_gathered, uint32_t *d_frame, uint32_t width, uint32_t height,
                        uint32_t nphases, void *stream)
{
    if (!d_gathered || !d_frame || !nphases) {
        set_err("null argument");
        return RT_E_INVALID;
    }
    rt_launch_deinterleave(d_gathered, d_frame, width, height, nphases, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rt_hip_selftest_arith(const float *d_a, const float *d_b, const float *d_c, uint32_t n, float *d_out,
                          double *d_out64, void *stream)
{
    if (n && (!d_a || !d_b || !d_c || !d_out || !d_out64)) {
        set_err("null argument");
        return RT_E_INVALID;
    }
    rt_launch_selftest_arith(d_a, d_b, d_c, n, d_out, d_out64, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int rt_hip_selftest_grid(rt_hip_ctx *c, const float *d_rays, uint32_t n, int filtered, uint32_t *d_out)
{
    if (!c || !c->have_scene || (n && (!d_rays || !d_out))) {
        set_err("null argument or no scene");
        return RT_E_INVALID;
    }
    if (!c->S.grid) {
        set_err("the scene has no grid (%u spheres; grids are built above %d)", c->S.nSph, RT_TILE_SPHERES);
        return RT_E_INVALID;
    }
    int rc = ctx_set_device(c);
    if (rc)
        return rc;
    rt_launch_selftest_grid(c->S, d_rays, n, filtered != 0, d_out, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int rt_hip_ctx_create(rt_hip_ctx **out, int device, uint64_t max_queue_rays)
{
    if (!out) {
        set_err("null argument");
        return RT_E_INVALID;
    }
    *out = nullptr;
    if (device < 0)
        HIP_TRY(hipGetDevice(&device));
    rt_hip_ctx *c = new rt_hip_ctx();
    c->device = device;
    uint64_t cap = max_queue_rays;
    if (!cap) { /* sized by the renders (submit): nothing is allocated for the queues yet */
        const char *env = getenv("RT_HIP_QUEUE_CAP");
        cap = env ? strtoull(env, nullptr, 0) : 0;
        c->auto_cap = true;
        c->cap_limit = round_cap(cap ? cap : kAutoCapLimit);
        cap = 0;
    } else {
        cap = round_cap(cap);
    }
    int rc = ctx_set_device(c);
    if (!rc) /* the library's per-device set-up: the address of rt_hip_tone_static's scratch array */
        rc = rt_tone_static_init();
    if (rc) {
        delete c;
        return rc;
    }
    hipError_t e = hipSuccess;
    if (cap && (rc = ensure_work(c, cap))) {
        rt_hip_ctx_destroy(c);
        return rc;
    }
    if (e == hipSuccess)
        e = hipMalloc((void **)&c->d_counters, (size_t)kMaxBatches * RT_COUNTER_WORDS * sizeof(uint32_t));
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&c->h_counters, (size_t)kMaxBatches * RT_COUNTER_WORDS * sizeof(uint32_t),
                          hipHostMallocMapped);
    if (e == hipSuccess)
        e = hipHostGetDevicePointer((void **)&c->h_counters_dev, c->h_counters, 0);
    if (e == hipSuccess)
        e = hipMemset(c->d_counters, 0, (size_t)kMaxBatches * RT_COUNTER_WORDS * sizeof(uint32_t));
    if (e != hipSuccess) {
        set_err("allocating the counter blocks failed: %s", hipGetErrorString(e));
        rt_hip_ctx_destroy(c);
        return -(int)e;
    }
    *out = c;
    return 0;
}

void rt_hip_ctx_destroy(rt_hip_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->pending) {
        (void)hipStreamSynchronize(c->stream);
        if (c->tail_pending && c->tail_stream) /* (hipFree below would wait for it anyway) */
            (void)hipStreamSynchronize(c->tail_stream);
    }
    drop_events(c);
    if (c->round_done) (void)hipEventDestroy(c->round_done);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->early_from) (void)hipEventDestroy(c->early_from);
    if (c->early_done) (void)hipEventDestroy(c->early_done);
    if (c->early_stream) (void)hipStreamDestroy(c->early_stream);
    if (c->head_done) (void)hipEventDestroy(c->head_done);
    if (c->tail_done) (void)hipEventDestroy(c->tail_done);
    if (c->scene_blob) (void)hipFree(c->scene_blob);
    if (c->work_blob) (void)hipFree(c->work_blob);
    if (c->pixAcc) (void)hipFree(c->pixAcc);
    if (c->tile_done) (void)hipFree(c->tile_done);
    if (c->roots) (void)hipFree(c->roots);
    if (c->loopq) (void)hipFree(c->loopq);
    if (c->Z.hit) (void)hipFree(c->Z.hit);
    if (c->Z.vis) (void)hipFree(c->Z.vis);
    if (c->Z.childR) (void)hipFree(c->Z.childR);
    if (c->Z.childT) (void)hipFree(c->Z.childT);
    if (c->d_plan) (void)hipFree(c->d_plan);
    if (c->d_wave_ticks) (void)hipFree(c->d_wave_ticks);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    delete c;
}

void rt_hip_ctx_set_specialization(rt_hip_ctx *c, int enable)
{
    if (c) {
        c->S.noShape = enable ? 0u : 1u;
        c->no_shape = !enable;
    }
}

int rt_hip_ctx_set_jit(rt_hip_ctx *c, int mode)
{
    if (!c || mode < RT_JIT_OFF || mode > RT_JIT_NOW_ANY) {
        set_err("rt_hip_ctx_set_jit: mode %d", mode);
        return RT_E_INVALID;
    }
    c->jit_mode = mode;
    c->jit_frames = 0;
    return 0;
}

int rt_hip_ctx_jit_status(rt_hip_ctx *c, char *log, size_t log_bytes, double *compile_seconds)
{
    if (!c || !c->have_scene) {
        set_err("rt_hip_ctx_jit_status: no scene");
        return RT_E_INVALID;
    }
    return rt_jit_status(c->S, c->numerics == RT_NUMERICS_FAST, log, log_bytes, compile_seconds);
}

int rt_hip_ctx_set_pipeline(rt_hip_ctx *c, int mode)
{
    if (!c || (mode != RT_PIPELINE_AUTO && mode != RT_PIPELINE_FUSED)) {
        set_err("pipeline %d is neither RT_PIPELINE_AUTO nor RT_PIPELINE_FUSED", mode);
        return RT_E_INVALID;
    }
    if (c->pending) { /* the render in flight was planned for the other pipeline */
        const int rc = rt_hip_finish(c, nullptr);
        if (rc)
            return rc;
    }
    if (c->no_jobs != (mode == RT_PIPELINE_FUSED))
        c->seen_valid = false; /* (the two pipelines count their tree nodes differently) */
    c->no_jobs = mode == RT_PIPELINE_FUSED;
    return 0;
}

void rt_hip_ctx_set_graph(rt_hip_ctx *c, int enable)
{
    if (c)
        c->use_graph = enable != 0;
}

void rt_hip_ctx_set_launch_plan(rt_hip_ctx *c, int enable)
{
    if (c) {
        if (c->launch_plan != (enable != 0 ? 1 : 0))
            c->plan_valid = false;
        c->launch_plan = enable != 0 ? 1 : 0;
    }
}

void rt_hip_ctx_set_timing(rt_hip_ctx *c, int enable)
{
    if (c)
        c->timing = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
}

int rt_hip_ctx_set_tail_stream(rt_hip_ctx *c, void *stream)
{
    if (!c) {
        set_err("null argument");
        return RT_E_INVALID;
    }
    int rc = ctx_set_device(c);
    if (rc)
        return rc;
    if (c->pending && (rc = rt_hip_finish(c, nullptr))) /* the render in flight ends on the old stream */
        return rc;
    if (stream && !c->head_done) {
        HIP_TRY(hipEventCreateWithFlags(&c->head_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->tail_done, hipEventDisableTiming));
    }
    c->tail_stream = (hipStream_t)stream;
    return 0;
}

int rt_hip_ctx_set_shading(rt_hip_ctx *c, int model)
{
    if (!c || (model != RT_SHADE_PHONG && model != RT_SHADE_PHONGBLINN)) {
        set_err("shading model %d is neither RT_SHADE_PHONG nor RT_SHADE_PHONGBLINN", model);
        return RT_E_INVALID;
    }
    c->shading = model;
    return 0;
}

int rt_hip_ctx_set_numerics(rt_hip_ctx *c, int mode)
{
    if (!c || (mode != RT_NUMERICS_EXACT && mode != RT_NUMERICS_FAST)) {
        set_err("numerics mode %d is neither RT_NUMERICS_EXACT nor RT_NUMERICS_FAST", mode);
        return RT_E_INVALID;
    }
    if (mode != c->numerics)
        c->seen_valid = false; /* (a handful of samples may take the other side of a threshold: level sizes can move) */
    c->numerics = mode;
    return 0;
}

void rt_hip_ctx_set_batch_samples(rt_hip_ctx *c, uint64_t max_samples)
{
    if (c)
        c->batch_samples = max_samples;
}

namespace {
/*
 * Data of the kernels' first polygon screening stage (rt_kernels.hip, edge_screen): for a planar convex
 * polygon, per edge i the unit in-plane inward normal m_i and offset k_i (g_i = m_i.p + k_i = signed
 * distance from the edge's line), a "deep inside" threshold hin_i and one "far outside" threshold hout,
 * stored as the single equation G_i = (g_i - hin_i) / (hin_i + hout): inside iff min G >= 0, outside iff
 * min G <= -1.
 * With L_i the edge length and a_i, b_i the distances from p to the edge's ends,
 * sin(theta_i) = L_i g_i / (a_i b_i), and inside the polygon a_i b_i <= ((a_i + b_i)/2)^2 <= P_i, the
 * largest such value at a vertex (a + b is convex).  The reference's float cosines (vec4_costheta:
 * three products, two sums, two rounded square roots, a product, a quotient, on correctly rounded
 * differences) are within 8e-7 of the exact ones, so its angle sum is within nv * 8e-7 / s_min of 2*pi
 * when every sin(theta_i) >= s_min; s_min = nv * 1e-3 keeps that below 8e-4 of the .001 slack of
 * ray.c:168.  Outside at distance >= h the exact sum is <= 2*pi - 4*asin(h / (h + D)), D the diameter
 * (the polygon lies in a half-plane at distance h and within h + D), while a cosine at +-1 can move
 * one angle by up to sqrt(2 * 8e-7) = 1.3e-3: hout asks for 4*asin(..) >= 1.3e-3 + nv * 1.3e-3.  Both get
 * 10 % on top and the float evaluation error of g.  Anything not planar and convex gets infinite
 * thresholds (never screened).
 */
void edge_screen_data(const rt_polygon &pg, std::vector<float4> &edges)
{
    const uint32_t nv = pg.nVerticies;
    const size_t base = edges.size();
    edges.resize(base + nv, make_float4(0.f, 0.f, 0.f, -0.5f));
    if (nv < 3)
        return;
    auto V = [&](uint32_t i, double out[3]) {
        out[0] = pg.vertex[i % nv].x; out[1] = pg.vertex[i % nv].y; out[2] = pg.vertex[i % nv].z;
    };
    double n[3] = {pg.plane.A, pg.plane.B, pg.plane.C};
    const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(nl > 0.0))
        return;
    for (double &x : n)
        x /= nl;
    double D = 0.0, maxc = 0.0, cen[3] = {0, 0, 0}, v0[3];
    V(0, v0);
    for (uint32_t i = 0; i < nv; ++i) {
        double a[3];
        V(i, a);
        for (int k = 0; k < 3; ++k) {
            cen[k] += a[k] / nv;
            maxc = fmax(maxc, fabs(a[k]));
        }
        for (uint32_t j = i + 1; j < nv; ++j) {
            double b[3];
            V(j, b);
            D = fmax(D, sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2])));
        }
    }
    if (!(D > 0.0) || !std::isfinite(D))
        return;
    /* the reference's plane is n.p + F = 0 with F forced >= 0 (geometry.c:29-32): planar means every
     * vertex on the plane through v0 with that normal, AND that plane being the reference's */
    for (uint32_t i = 0; i < nv; ++i) {
        double a[3];
        V(i, a);
        const double d0 = n[0] * (a[0] - v0[0]) + n[1] * (a[1] - v0[1]) + n[2] * (a[2] - v0[2]);
        const double dp = (pg.plane.A * a[0] + pg.plane.B * a[1] + pg.plane.C * a[2] + pg.plane.F) / nl;
        if (fabs(d0) > 1e-6 * D || fabs(dp) > 1e-5 * (D + maxc))
            return;
    }
    std::vector<double> e(4 * (size_t)nv), h(nv);
    const double s_min = nv * 1e-3, eps = 1e-5 * (maxc + D);
    for (uint32_t i = 0; i < nv; ++i) {
        double a[3], b[3];
        V(i, a);
        V(i + 1, b);
        const double d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
        const double L = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (!(L > 1e-6 * D))
            return;
        double m[3] = {n[1] * d[2] - n[2] * d[1], n[2] * d[0] - n[0] * d[2], n[0] * d[1] - n[1] * d[0]};
        const double ml = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        if (!(ml > 0.0))
            return;
        double sgn = (m[0] * (cen[0] - a[0]) + m[1] * (cen[1] - a[1]) + m[2] * (cen[2] - a[2])) >= 0.0 ? 1.0 : -1.0;
        for (double &x : m)
            x *= sgn / ml;
        for (uint32_t j = 0; j < nv; ++j) { /* convex: every vertex on the inner side of every edge */
            double c[3];
            V(j, c);
            if (m[0] * (c[0] - a[0]) + m[1] * (c[1] - a[1]) + m[2] * (c[2] - a[2]) < -1e-6 * D)
                return;
        }
        e[4 * i + 0] = m[0]; e[4 * i + 1] = m[1]; e[4 * i + 2] = m[2];
        e[4 * i + 3] = -(m[0] * a[0] + m[1] * a[1] + m[2] * a[2]);
        double P = 0.0; /* bound of a*b inside the polygon */
        for (uint32_t j = 0; j < nv; ++j) {
            double c[3];
            V(j, c);
            const double da = sqrt((c[0] - a[0]) * (c[0] - a[0]) + (c[1] - a[1]) * (c[1] - a[1]) + (c[2] - a[2]) * (c[2] - a[2]));
            const double db = sqrt((c[0] - b[0]) * (c[0] - b[0]) + (c[1] - b[1]) * (c[1] - b[1]) + (c[2] - b[2]) * (c[2] - b[2]));
            P = fmax(P, 0.25 * (da + db) * (da + db));
        }
        h[i] = 1.1 * s_min * P / L + eps;
    }
    const double q = (1.3e-3 + nv * 1.3e-3) / 4.0;
    if (!(q < 0.5))
        return; /* hundreds of vertices: no useful outside threshold */
    const double hout = 1.1 * D * sin(q) / (1.0 - sin(q)) + eps;
    for (uint32_t i = 0; i < nv; ++i) {
        const double inv = 1.0 / (h[i] + hout);
        edges[base + i] = make_float4((float)(e[4 * i + 0] * inv), (float)(e[4 * i + 1] * inv), (float)(e[4 * i + 2] * inv),
                                      (float)((e[4 * i + 3] - h[i]) * inv));
    }
}
} // namespace

namespace {
/* Uniform grid over the spheres (GridHeader in rt_device.h): about one sphere per two cells, at most 128
 * cells per axis; a sphere goes into every cell its grown bounding box overlaps.
 * Returns the words GridHeader | cellStart[cells + 1] | items. */
void build_grid(const std::vector<float4> &sph, const rt_scene *sc, const std::vector<uint32_t> &sph_obj,
                std::vector<uint32_t> &words, std::vector<double> &grown, double &reach_out)
{
    const size_t n = sph.size();
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    std::vector<double> rad(n);
    for (size_t k = 0; k < n; ++k) {
        rad[k] = fabs((double)sc->objects[sph_obj[k]].sphr_obj.radius);
        const double c[3] = {sph[k].x, sph[k].y, sph[k].z};
        for (int a = 0; a < 3; ++a) {
            lo[a] = fmin(lo[a], c[a] - rad[k]);
            hi[a] = fmax(hi[a], c[a] + rad[k]);
        }
    }
    double ext[3], vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        const double pad = 1e-3 * (hi[a] - lo[a]) + 1e-4;
        lo[a] -= pad;
        hi[a] += pad;
        ext[a] = hi[a] - lo[a];
        vol *= ext[a];
    }
    const char *dens = getenv("RT_HIP_GRID_DENSITY"); /* development: spheres per cell aimed at */
    double per_cell = dens && atof(dens) > 0.0 ? atof(dens) : 0.5; /* measured: 0.5 .. 2 within 10 % */
    /* rays whose origin is within 1.5 box diagonals of the box centre are at most `reach` = 2 diagonals
     * from any sphere: what the reference's test accepts beyond geometry at that reach (GridHeader) */
    const double diag = sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);
    const double reach = 2.0 * diag, extra2 = 2.0 * (double)RT_GRID_KAPPA * reach * reach;
    GridHeader G;
    size_t cells = 1;
    for (;;) { /* coarser until the lists stay small: big or clustered spheres overlap many fine cells */
        const double cell = cbrt(vol * per_cell / (double)n);
        cells = 1;
        double listed = 0.0; /* estimate of the list entries: cells overlapped per sphere */
        for (int a = 0; a < 3; ++a) {
            long dm = lround(ext[a] / cell);
            dm = dm < 1 ? 1 : (dm > 128 ? 128 : dm);
            G.dim[a] = (uint32_t)dm;
            G.gmin[a] = (float)lo[a];
            G.gmax[a] = (float)hi[a];
            G.h[a] = (float)(ext[a] / (double)dm);
            G.inv_h[a] = (float)((double)dm / ext[a]);
            cells *= (size_t)dm;
        }
        for (size_t k = 0; k < n; ++k) {
            double per = 1.0;
            for (int a = 0; a < 3; ++a) {
                const double span = 2.0 * (sqrt(rad[k] * rad[k] + extra2) + 0.01 * (double)G.h[a]);
                per *= fmin((double)G.dim[a], floor(span / (double)G.h[a]) + 2.0);
            }
            listed += per;
        }
        if (listed <= 64.0 * (double)n || cells == 1)
            break;
        per_cell *= 8.0;
    }
    G.words = (uint32_t)(sizeof(GridHeader) / sizeof(uint32_t));
    for (int a = 0; a < 3; ++a)
        G.centre[a] = (float)(0.5 * (lo[a] + hi[a]));
    G.origin_reach2 = (float)(1.5 * diag * 1.5 * diag);
    auto range = [&](size_t k, int a, long &c0, long &c1) {
        const double c = a == 0 ? sph[k].x : a == 1 ? sph[k].y : sph[k].z;
        const double h = (double)G.h[a], grow = sqrt(rad[k] * rad[k] + extra2) + 0.01 * h;
        c0 = (long)floor((c - grow - (double)G.gmin[a]) / h);
        c1 = (long)floor((c + grow - (double)G.gmin[a]) / h);
        c0 = c0 < 0 ? 0 : c0;
        c1 = c1 >= (long)G.dim[a] ? (long)G.dim[a] - 1 : c1;
    };
    std::vector<uint32_t> count(cells + 1, 0);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<uint32_t> fill;
        std::vector<uint32_t> *out = nullptr;
        static thread_local std::vector<uint32_t> items;
        if (pass == 1) {
            uint32_t run = 0; /* counts -> starts */
            for (size_t i = 0; i <= cells; ++i) {
                const uint32_t cnt = count[i];
                count[i] = run;
                run += cnt;
            }
            items.assign(run, 0);
            fill.assign(count.begin(), count.end());
            out = &items;
        }
        for (size_t k = 0; k < n; ++k) { /* ascending k: a cell's list comes out sorted */
            long x0, x1, y0, y1, z0, z1;
            range(k, 0, x0, x1);
            range(k, 1, y0, y1);
            range(k, 2, z0, z1);
            for (long z = z0; z <= z1; ++z)
                for (long y = y0; y <= y1; ++y)
                    for (long x = x0; x <= x1; ++x) {
                        const size_t ci = ((size_t)z * G.dim[1] + (size_t)y) * G.dim[0] + (size_t)x;
                        if (pass == 0)
                            count[ci]++;
                        else
                            (*out)[fill[ci]++] = (uint32_t)k;
                    }
        }
        if (pass == 1) {
            /* ... and the cells once more as 64-byte chunks with their spheres inline (CellChunk) */
            const size_t lists = G.words + cells + 1 + items.size();
            const bool chunked = n <= RT_CHUNK_MAX_SPHERES && sc->nObjects <= RT_CHUNK_MAX_SPHERES;
            G.rec_words = chunked ? (uint32_t)((lists + 15u) & ~(size_t)15u) : 0u;
            G.pad_[0] = G.pad_[1] = G.pad_[2] = 0u;
            std::vector<CellChunk> chunks;
            if (chunked) {
                chunks.assign(cells, CellChunk{});
                auto fill_chunk = [&](CellChunk &ck, uint32_t beg, uint32_t cnt) { /* up to three of the items from beg on */
                    uint32_t k[RT_CHUNK_SPHERES] = {0u, 0u, 0u}; /* OBJECT indices (ascending with the sphere index) */
                    for (uint32_t u = 0; u < cnt && u < RT_CHUNK_SPHERES; ++u) {
                        const uint32_t ks = items[beg + u];
                        k[u] = sph_obj[ks];
                        ck.sph[u] = sph[ks];
                        const rt_material &m = sc->objects[k[u]].material; /* raytrace.c:405,452: what a hit on it spawns */
                        ck.count |= ((m.kr != 0.0f ? 1u : 0u) | (m.kt != 0.0f ? 2u : 0u)) << (24u + 2u * u);
                    }
                    ck.idx_lo = k[0] | (k[1] << 21);
                    ck.idx_hi = (k[1] >> 11) | (k[2] << 10);
                };
                for (size_t ci = 0; ci < cells; ++ci) {
                    const uint32_t beg = count[ci], cnt = count[ci + 1] - beg;
                    fill_chunk(chunks[ci], beg, cnt);
                    chunks[ci].count |= cnt; /* (< 2^24: RT_CHUNK_MAX_SPHERES) */
                    if (cnt > RT_CHUNK_SPHERES) {
                        chunks[ci].next = (uint32_t)chunks.size();
                        for (uint32_t u = RT_CHUNK_SPHERES; u < cnt; u += RT_CHUNK_SPHERES) {
                            chunks.push_back(CellChunk{});
                            fill_chunk(chunks.back(), beg + u, cnt - u);
                        }
                    }
                }
            }
            if (getenv("RT_HIP_GRID_STATS")) { /* development: how full the cells are */
                size_t hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t ci = 0; ci < cells; ++ci) {
                    const uint32_t cnt = count[ci + 1] - count[ci];
                    hist[cnt < 7u ? cnt : 7u]++;
                }
                fprintf(stderr, "rt_hip grid: %u x %u x %u = %zu cells, %zu listed spheres, %zu chunks; cells holding 0..6, 7+ spheres:",
                        G.dim[0], G.dim[1], G.dim[2], cells, items.size(), chunks.size());
                for (int i = 0; i < 8; ++i)
                    fprintf(stderr, " %zu", hist[i]);
                fprintf(stderr, "\n");
            }
            words.assign(chunked ? (size_t)G.rec_words + chunks.size() * (sizeof(CellChunk) / sizeof(uint32_t)) : lists, 0u);
            memcpy(words.data(), &G, sizeof G);
            memcpy(words.data() + G.words, count.data(), (cells + 1) * sizeof(uint32_t));
            if (!items.empty())
                memcpy(words.data() + G.words + cells + 1, items.data(), items.size() * sizeof(uint32_t));
            if (!chunks.empty())
                memcpy(words.data() + G.rec_words, chunks.data(), chunks.size() * sizeof(CellChunk));
            /* ... and all the spheres once more in GROUPS of 64 neighbours (Morton order of the centres) behind a ball
             * around each group's centres, for k_loop (rt_jobs.inc): the rays the grid cannot take - a direction longer
             * than unit: the reference's test then accepts what lies in a cone around the ray - test the 64 spheres of a
             * group only when its ball can hold a sphere the test accepts.  GridHeader::pad_[0]: where the groups start
             * (words, a multiple of 4), pad_[1]: how many; per group two float4 (centre and radius of the ball, the
             * largest r*r), then 64 (centre, r*r) per group, then 64 sphere indices per group (RT_NONE: padding). */
            {
                std::vector<std::pair<uint32_t, uint32_t>> order(n);
                for (size_t k = 0; k < n; ++k) {
                    const double c[3] = {sph[k].x, sph[k].y, sph[k].z};
                    uint32_t code = 0;
                    for (int a = 0; a < 3; ++a) {
                        double q = (c[a] - lo[a]) / ext[a];
                        q = q < 0.0 ? 0.0 : (q > 1.0 ? 1.0 : q);
                        uint32_t v = (uint32_t)(q * 1023.0);
                        for (int b = 0; b < 10; ++b)
                            code |= ((v >> b) & 1u) << (3 * b + a);
                    }
                    order[k] = {code, (uint32_t)k};
                }
                std::sort(order.begin(), order.end());
                const size_t ngroups = (n + 63u) / 64u;
                const size_t goff = (words.size() + 3u) & ~(size_t)3u;
                words.resize(goff + ngroups * 8u + ngroups * 64u * 4u + ngroups * 64u, 0u);
                float *grp = (float *)(words.data() + goff);
                float *gsph = grp + ngroups * 8u;
                uint32_t *gidx = (uint32_t *)(gsph + ngroups * 64u * 4u);
                const bool no_cull = getenv("RT_HIP_NO_LOOP_CULL") != nullptr; /* development: every group's ball holds everything */
                for (size_t g = 0; g < ngroups; ++g) {
                    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
                    float r2max = 0.0f;
                    const size_t cnt = std::min<size_t>(64u, n - g * 64u);
                    for (size_t u = 0; u < cnt; ++u) {
                        const float4 sp = sph[order[g * 64u + u].second];
                        const double c[3] = {sp.x, sp.y, sp.z};
                        for (int a = 0; a < 3; ++a) {
                            blo[a] = fmin(blo[a], c[a]);
                            bhi[a] = fmax(bhi[a], c[a]);
                        }
                        r2max = fmaxf(r2max, sp.w);
                    }
                    const float cf[3] = {(float)(0.5 * (blo[0] + bhi[0])), (float)(0.5 * (blo[1] + bhi[1])), (float)(0.5 * (blo[2] + bhi[2]))};
                    double R = 0.0; /* (around the centre as the kernel reads it: the float) */
                    for (size_t u = 0; u < 64u; ++u) {
                        float *dst = gsph + (g * 64u + u) * 4u;
                        if (u < cnt) {
                            const uint32_t k = order[g * 64u + u].second;
                            const float4 sp = sph[k];
                            const double ddx = (double)sp.x - cf[0], ddy = (double)sp.y - cf[1], ddz = (double)sp.z - cf[2];
                            R = fmax(R, sqrt(ddx * ddx + ddy * ddy + ddz * ddz));
                            dst[0] = sp.x; dst[1] = sp.y; dst[2] = sp.z; dst[3] = sp.w;
                            gidx[g * 64u + u] = k;
                        } else {
                            dst[0] = dst[1] = dst[2] = dst[3] = 0.0f;
                            gidx[g * 64u + u] = RT_NONE;
                        }
                    }
                    grp[g * 8u + 0] = cf[0]; grp[g * 8u + 1] = cf[1]; grp[g * 8u + 2] = cf[2];
                    grp[g * 8u + 3] = no_cull ? 3.0e38f : (float)(R * (1.0 + 1e-6) + 1e-30);
                    grp[g * 8u + 4] = r2max;
                }
                G.pad_[0] = (uint32_t)goff;
                G.pad_[1] = (uint32_t)ngroups;
                memcpy(words.data(), &G, sizeof G);
            }
        }
    }
    grown.resize(n);
    for (size_t k = 0; k < n; ++k)
        grown[k] = sqrt(rad[k] * rad[k] + extra2);
    reach_out = reach;
}

/* The lights' direction grids (LightGridHeader in rt_device.h).  `grown`: the radius within which the reference's
 * test can still accept a sphere around a ray's line (build_grid).  A shadow ray's line passes its light within
 * eta: the direction is (light - point) * (1 / length), rounded per component, the pushed origin stays on the line. */
void build_light_grids(const std::vector<float4> &sph, const std::vector<double> &grown, double reach,
                       const rt_scene *sc, std::vector<uint32_t> &words)
{
    const size_t n = sph.size();
    const uint32_t nL = sc->nLights;
    words.clear();
    if (!nL || !n)
        return;
    struct Span { int f, a0, a1, b0, b1; };
    const double kPi = 3.14159265358979323846;
    uint32_t res = 256;
    std::vector<uint32_t> count, gcount(nL + 1, 0);
    std::vector<std::vector<Span>> spans; /* per (light, sphere): the cell rectangles, at most one per face */
    std::vector<uint8_t> global;
    for (;; res /= 2) {
        const size_t ncell = (size_t)6 * res * res;
        count.assign((size_t)nL * ncell + 1, 0);
        spans.assign((size_t)nL * n, {});
        global.assign((size_t)nL * n, 0);
        double listed = 0.0;
        for (uint32_t l = 0; l < nL; ++l) {
            const double L[3] = {sc->lights[l].position.x, sc->lights[l].position.y, sc->lights[l].position.z};
            const double eta = 1e-5 * (fmax(fabs(L[0]), fmax(fabs(L[1]), fabs(L[2]))) + reach + 1.0);
            for (size_t k = 0; k < n; ++k) {
                const double w[3] = {sph[k].x - L[0], sph[k].y - L[1], sph[k].z - L[2]};
                const double dist = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                const double R = (grown[k] + eta) * 1.0001;
                if (!(dist > R * 1.001 + 1e-9)) { /* the light is (nearly) inside: met by lines of every direction */
                    global[(size_t)l * n + k] = 1;
                    continue;
                }
                const double sa = R / dist;
                const double u[3] = {w[0] / dist, w[1] / dist, w[2] / dist};
                for (int m = 0; m < 3; ++m)
                    for (int neg = 0; neg < 2; ++neg) {
                        const double um = neg ? -u[m] : u[m];
                        const int ja = m == 0 ? 1 : 0, jb = m == 2 ? 1 : 2; /* the other two axes, in axis order */
                        int lohi[2][2];
                        bool touched = true;
                        for (int q = 0; q < 2 && touched; ++q) {
                            const double uj = u[q ? jb : ja];
                            const double ell = sqrt(um * um + uj * uj);
                            double lo = -kPi / 4 - 1e-4, hi = kPi / 4 + 1e-4;
                            const double al = sa < ell * 0.999999 ? asin(sa / ell) + 1e-6 : kPi;
                            if (al < kPi / 2) {
                                /* the cone between its two tangent planes that hold the third axis: angles phi -+ al in the
                                 * plane of these two axes, against [-pi/4, pi/4] (phi in (-pi, pi], al < pi/2: no wrap
                                 * reaches it); a wider cone is listed across the whole face */
                                const double phi = atan2(uj, um);
                                lo = fmax(lo, phi - al);
                                hi = fmin(hi, phi + al);
                            }
                            if (lo > hi) {
                                touched = false;
                                break;
                            }
                            const double a0 = tan(lo) - 2e-4, a1 = tan(hi) + 2e-4;
                            long c0 = (long)floor((a0 + 1.0) * 0.5 * res), c1 = (long)floor((a1 + 1.0) * 0.5 * res);
                            c0 = c0 < 0 ? 0 : (c0 > (long)res - 1 ? (long)res - 1 : c0);
                            c1 = c1 < 0 ? 0 : (c1 > (long)res - 1 ? (long)res - 1 : c1);
                            lohi[q][0] = (int)c0;
                            lohi[q][1] = (int)c1;
                        }
                        if (!touched)
                            continue;
                        spans[(size_t)l * n + k].push_back({2 * m + neg, lohi[0][0], lohi[0][1], lohi[1][0], lohi[1][1]});
                        listed += (double)(lohi[0][1] - lohi[0][0] + 1) * (lohi[1][1] - lohi[1][0] + 1);
                    }
            }
        }
        static const double budget = [] { const char *e = getenv("RT_HIP_LGRID_BUDGET"); return e && atof(e) > 0.0 ? atof(e) : 16.0; }();
        if (listed <= budget * (double)n * nL || res <= 8) /* (development: cells listed per sphere and light at most) */
            break;
    }
    const size_t ncell = (size_t)6 * res * res;
    auto each_cell = [&](uint32_t l, size_t k, auto &&fn) {
        for (const Span &sp : spans[(size_t)l * n + k])
            for (int b = sp.b0; b <= sp.b1; ++b)
                for (int a = sp.a0; a <= sp.a1; ++a)
                    fn((size_t)l * ncell + ((size_t)sp.f * res + b) * res + a);
    };
    for (uint32_t l = 0; l < nL; ++l)
        for (size_t k = 0; k < n; ++k)
            each_cell(l, k, [&](size_t ci) { count[ci]++; });
    uint32_t run = 0;
    for (size_t i = 0; i < count.size(); ++i) {
        const uint32_t c = count[i];
        count[i] = run;
        run += c;
    }
    for (uint32_t l = 0; l < nL; ++l) { /* the global lists behind the cells' */
        gcount[l] = run;
        for (size_t k = 0; k < n; ++k)
            run += global[(size_t)l * n + k];
    }
    gcount[nL] = run;
    const uint32_t nitems = run;
    LightGridHeader H;
    H.res = res;
    H.ncell = (uint32_t)ncell;
    H.off_cells = (uint32_t)(sizeof H / sizeof(uint32_t));
    H.off_gstart = H.off_cells + (uint32_t)count.size();
    H.off_idx = 0;
    /* an item: (centre, r*r), (bits of the sphere's index, key, -, -).  key = distance from the light at which the
     * grown ball begins: a shadow ray that starts `mag` from the light cannot meet a sphere whose key exceeds mag,
     * and a cell's list is sorted by key, so its scan ends there.  The global lists carry key 0. */
    std::vector<float4> item_store((size_t)nitems * 2u);
    float4 *item = item_store.data();
    std::vector<uint32_t> fill(count.begin(), count.end() - 1);
    std::vector<float> key(n);
    for (uint32_t l = 0; l < nL; ++l) {
        const double L[3] = {sc->lights[l].position.x, sc->lights[l].position.y, sc->lights[l].position.z};
        const double eta = 1e-5 * (fmax(fabs(L[0]), fmax(fabs(L[1]), fabs(L[2]))) + reach + 1.0);
        uint32_t g = gcount[l];
        for (size_t k = 0; k < n; ++k) {
            const double w[3] = {sph[k].x - L[0], sph[k].y - L[1], sph[k].z - L[2]};
            const double dist = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            key[k] = (float)fmax(0.0, (dist - (grown[k] + eta) * 1.0001) * 0.9999 - 1e-4);
            each_cell(l, k, [&](size_t ci) {
                item[2 * (size_t)fill[ci]] = sph[k];
                item[2 * (size_t)fill[ci] + 1] = make_float4(__builtin_bit_cast(float, (uint32_t)k), key[k], 0.f, 0.f);
                fill[ci]++;
            });
            if (global[(size_t)l * n + k]) {
                item[2 * (size_t)g] = sph[k];
                item[2 * (size_t)g + 1] = make_float4(__builtin_bit_cast(float, (uint32_t)k), 0.f, 0.f, 0.f);
                ++g;
            }
        }
        const size_t ncell_l = ncell;
        for (size_t ci = (size_t)l * ncell_l; ci < (size_t)(l + 1) * ncell_l; ++ci) { /* nearest the light first */
            const uint32_t b0 = count[ci], e0 = count[ci + 1];
            if (e0 - b0 < 2)
                continue;
            struct It { float4 a, b; };
            It *first = (It *)(item + 2 * (size_t)b0);
            std::stable_sort(first, first + (e0 - b0), [](const It &x, const It &y) { return x.b.y < y.b.y; });
        }
    }
    /* ... and every list as 64-byte chunks of three items (LightChunk): the tables then count chunks */
    std::vector<LightChunk> chunks;
    chunks.reserve((size_t)nitems / 2u + count.size());
    auto emit = [&](uint32_t b0, uint32_t e0) {
        for (uint32_t i = b0; i < e0; i += RT_CHUNK_SPHERES) {
            LightChunk ck{};
            uint32_t k[RT_CHUNK_SPHERES] = {0u, 0u, 0u}, kb[RT_CHUNK_SPHERES] = {0x7F80u, 0x7F80u, 0x7F80u}; /* no item: key +inf */
            for (uint32_t u = 0; u < RT_CHUNK_SPHERES && i + u < e0; ++u) {
                ck.sph[u] = item[2 * (size_t)(i + u)];
                k[u] = __builtin_bit_cast(uint32_t, item[2 * (size_t)(i + u) + 1].x);
                kb[u] = __builtin_bit_cast(uint32_t, item[2 * (size_t)(i + u) + 1].y) >> 16; /* (keys are >= 0: cut off = rounded down) */
            }
            ck.idx_lo = k[0] | (k[1] << 21);
            ck.idx_hi = (k[1] >> 11) | (k[2] << 10);
            ck.keys01 = kb[0] | (kb[1] << 16);
            ck.key2 = kb[2];
            chunks.push_back(ck);
        }
    };
    std::vector<uint32_t> cstart(count.size()), gstart(nL + 1);
    for (size_t ci = 0; ci + 1 < count.size(); ++ci) {
        cstart[ci] = (uint32_t)chunks.size();
        emit(count[ci], count[ci + 1]);
    }
    cstart[count.size() - 1] = (uint32_t)chunks.size();
    for (uint32_t l = 0; l < nL; ++l) {
        gstart[l] = (uint32_t)chunks.size();
        emit(gcount[l], gcount[l + 1]);
    }
    gstart[nL] = (uint32_t)chunks.size();
    H.off_sph = (H.off_gstart + nL + 1 + 15u) & ~15u;
    H.nitems = (uint32_t)chunks.size();
    H.pad_ = 0;
    words.assign((size_t)H.off_sph + chunks.size() * (sizeof(LightChunk) / sizeof(uint32_t)), 0u);
    memcpy(words.data(), &H, sizeof H);
    memcpy(words.data() + H.off_cells, cstart.data(), cstart.size() * sizeof(uint32_t));
    memcpy(words.data() + H.off_gstart, gstart.data(), gstart.size() * sizeof(uint32_t));
    if (!chunks.empty())
        memcpy(words.data() + H.off_sph, chunks.data(), chunks.size() * sizeof(LightChunk));
}
} // namespace

int rt_hip_scene_upload(rt_hip_ctx *c, const rt_scene *sc)
{
    if (!c || !sc || (sc->nObjects && !sc->objects) || (sc->nLights && !sc->lights)) {
        set_err("null argument");
        return RT_E_INVALID;
    }
    int rc = ctx_set_device(c);
    if (rc)
        return rc;
    /* a render still in flight reads the old scene: finish it properly - overflow retries drained, the
     * tail stream waited for, counters collected - before its blob may be replaced */
    if (c->pending && (rc = rt_hip_finish(c, nullptr)))
        return rc;

    const uint32_t nObj = sc->nObjects, nL = sc->nLights;
    std::vector<float4> sph, planes, infos, verts, edges;
    std::vector<uint32_t> sph_obj;
    std::vector<float4> tab((size_t)2 * nL + (size_t)RT_OBJ_FIELDS * nObj);
    float4 *orec = tab.data() + (size_t)2 * nL;
    for (uint32_t i = 0; i < nObj; ++i) {
        const rt_object3d &o = sc->objects[i];
        const rt_material &m = o.material;
        const rt_color &df = m.colors[RT_MATERIAL_DIFFUSECOLOR], &sp = m.colors[RT_MATERIAL_SPECULARCOLOR];
        float4 *rec = orec + (size_t)RT_OBJ_FIELDS * i;
        rec[RT_OBJ_DIFFUSE] = make_float4(df.r, df.g, df.b, m.phong_ka);
        rec[RT_OBJ_SPECULAR] = make_float4(sp.r, sp.g, sp.b, m.phong_ks);
        rec[RT_OBJ_PHONG] = make_float4(m.phong_kd, m.phong_ke, m.kr, m.kt);
        uint32_t glist = RT_NONE;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o.geometryType == RT_GEOMETRY_SPHERE) {
            const rt_sphere &s = o.sphr_obj;
            glist = (uint32_t)sph.size();
            /* radius*radius is the same float product the reference forms per test (ray.c:194) */
            sph.push_back(make_float4(s.center.x, s.center.y, s.center.z, s.radius * s.radius));
            sph_obj.push_back(i);
            g = make_float4(s.center.x, s.center.y, s.center.z, 0.f);
        } else if (o.geometryType == RT_GEOMETRY_POLYGON) {
            const rt_polygon &pg = o.poly_obj;
            if (pg.nVerticies == 0 || !pg.vertex) {
                set_err("polygon object %u has no vertices", i);
                return RT_E_INVALID;
            }
            glist = (uint32_t)planes.size();
            planes.push_back(make_float4(pg.plane.A, pg.plane.B, pg.plane.C, pg.plane.F));
            float4 info;
            info.x = __builtin_bit_cast(float, (uint32_t)verts.size());
            info.y = __builtin_bit_cast(float, (uint32_t)pg.nVerticies);
            info.z = __builtin_bit_cast(float, i);
            info.w = 0.f;
            edge_screen_data(pg, edges);
            {   /* largest gradient of the polygon's scaled edge equations: how fast the screening value can
                 * move with the hit point (classify_pixel in rt_kernels.hip); rounded up */
                float gmax = 0.f;
                for (size_t e = edges.size() - pg.nVerticies; e < edges.size(); ++e)
                    gmax = fmaxf(gmax, sqrtf(edges[e].x * edges[e].x + edges[e].y * edges[e].y + edges[e].z * edges[e].z));
                info.w = gmax * 1.0001f;
            }
            infos.push_back(info);
            for (uint32_t v = 0; v < pg.nVerticies; ++v)
                verts.push_back(make_float4(pg.vertex[v].x, pg.vertex[v].y, pg.vertex[v].z, pg.vertex[v].w));
            g = make_float4(pg.plane.A, pg.plane.B, pg.plane.C, 0.f);
        } /* any other tag never intersects (raytrace.c:179-180): it keeps its index, no geometry */
        rec[RT_OBJ_GEOM] = make_float4(m.n, __builtin_bit_cast(float, (uint32_t)o.geometryType),
                                       __builtin_bit_cast(float, glist), 0.f);
        rec[RT_OBJ_SHAPE] = g;
    }
    const uint32_t nSph = (uint32_t)sph.size(), nPoly = (uint32_t)planes.size();
    const size_t polySlots = 2 * (size_t)nPoly + verts.size() + edges.size(); /* edges.size() == verts.size() */
    if (polySlots > RT_MAX_POLY_LDS) {
        set_err("polygon data needs %zu LDS slots, limit is %d", polySlots, RT_MAX_POLY_LDS);
        return RT_E_INVALID;
    }
    std::vector<float4> poly;
    poly.insert(poly.end(), planes.begin(), planes.end());
    poly.insert(poly.end(), infos.begin(), infos.end());
    poly.insert(poly.end(), verts.begin(), verts.end());
    poly.insert(poly.end(), edges.begin(), edges.end());
    for (uint32_t l = 0; l < nL; ++l) {
        const rt_pointlight &L = sc->lights[l];
        tab[2 * l + 0] = make_float4(L.position.x, L.position.y, L.position.z, L.position.w);
        tab[2 * l + 1] = make_float4(L.color.r, L.color.g, L.color.b, 0.f);
    }

    std::vector<uint32_t> grid, lgrid;
    if (nSph > RT_TILE_SPHERES) {
        /* The grid is built from the spheres' bounds and the light grids from the lights' positions: a centre, radius or
         * position that is not a finite number (or lies beyond 1e15) has no cell (the reference's own arithmetic turns such a sphere into NaN
         * comparisons; a small scene goes through the same arithmetic here, a searched one is refused). */
        for (uint32_t k = 0; k < nSph; ++k) {
            const float r = sc->objects[sph_obj[k]].sphr_obj.radius;
            const float big = 1e15f; /* (squares of the grid's extent stay floats; the reference's own |o - c|^2 is inf at 1.8e19) */
            if (!(fabsf(sph[k].x) <= big) || !(fabsf(sph[k].y) <= big) || !(fabsf(sph[k].z) <= big) || !(fabsf(r) <= big)) {
                set_err("sphere object %u has a centre or radius that is not finite (or beyond 1e15): scenes of more than %d "
                        "spheres are searched through a grid, which needs finite bounds", sph_obj[k], (int)RT_TILE_SPHERES);
                return RT_E_INVALID;
            }
        }
        for (uint32_t l = 0; l < nL; ++l) {
            const rt_pointlight &L = sc->lights[l];
            if (!(fabsf(L.position.x) <= 1e15f) || !(fabsf(L.position.y) <= 1e15f) || !(fabsf(L.position.z) <= 1e15f)) {
                set_err("light %u has a position that is not finite (or beyond 1e15): scenes of more than %d spheres are searched through "
                        "grids around their lights", l, (int)RT_TILE_SPHERES);
                return RT_E_INVALID;
            }
        }
        std::vector<double> grown;
        double reach = 0.0;
        build_grid(sph, sc, sph_obj, grid, grown, reach);
        if (nL <= RT_JOBS_MAX_LIGHTS)
            build_light_grids(sph, grown, reach, sc, lgrid);
    }

    /* one blob, 256-B aligned arrays */
    struct Seg { const void *src; size_t bytes; size_t off; };
    Seg segs[6] = {
        {sph.data(), sph.size() * sizeof(float4), 0},   {sph_obj.data(), sph_obj.size() * sizeof(uint32_t), 0},
        {poly.data(), poly.size() * sizeof(float4), 0}, {tab.data(), tab.size() * sizeof(float4), 0},
        {grid.data(), grid.size() * sizeof(uint32_t), 0}, {lgrid.data(), lgrid.size() * sizeof(uint32_t), 0}};
    size_t total = 0;
    for (Seg &s : segs) {
        s.off = total;
        total += align_up(s.bytes ? s.bytes : 16, 256);
    }
    std::vector<char> host(total, 0);
    for (const Seg &s : segs)
        if (s.bytes)
            memcpy(host.data() + s.off, s.src, s.bytes);
    /* the blocking host path uploads the scene on every call (the caller may have edited it):
     * an unchanged scene - the common case, frame after frame - keeps the device copy */
    const bool same = c->have_scene && c->scene_blob && c->scene_host == host && c->S.nSph == nSph &&
                      c->S.nPoly == nPoly && c->S.nObj == nObj && c->S.nLights == nL;
    if (!same) {
        if (c->scene_blob) {
            HIP_TRY(hipFree(c->scene_blob));
            c->scene_blob = nullptr;
            c->have_scene = false;
        }
        HIP_TRY(hipMalloc(&c->scene_blob, total));
        HIP_TRY(hipMemcpy(c->scene_blob, host.data(), total, hipMemcpyHostToDevice));
        c->scene_host.swap(host);
    }
    char *b = (char *)c->scene_blob;
    DevScene &S = c->S;
    S.sph = (const float4 *)(b + segs[0].off);
    S.sph_obj = (const uint32_t *)(b + segs[1].off);
    S.poly = (const float4 *)(b + segs[2].off);
    S.tab = (const float4 *)(b + segs[3].off);
    S.grid = grid.empty() ? nullptr : (const uint32_t *)(b + segs[4].off);
    S.lgrid = lgrid.empty() ? nullptr : (const uint32_t *)(b + segs[5].off);
    S.nSph = nSph;
    S.nPoly = nPoly;
    S.nPolySlots = (uint32_t)polySlots;
    S.nObj = nObj;
    S.nLights = nL;
    S.tabSlots = (uint32_t)tab.size();
    S.tabInLds = 0;
    S.noShape = c->no_shape ? 1u : 0u;
    S.tabInLds = (nSph <= RT_TILE_SPHERES &&
                  rt_scene_lds_bytes(S) - RT_PARK_FLOATS * sizeof(float) +
                          (tab.size() + nSph + nPoly) * sizeof(float4) /* tables + k_primary's eye terms */ <=
                      RT_LDS_TABLE_BUDGET) ? 1u : 0u;
    S.bg[0] = sc->bgColor.r; S.bg[1] = sc->bgColor.g; S.bg[2] = sc->bgColor.b;
    S.ambient[0] = sc->ambientLightColor.r;
    S.ambient[1] = sc->ambientLightColor.g;
    S.ambient[2] = sc->ambientLightColor.b;
    if (rt_scene_lds_bytes(S) > 65536u) {
        set_err("scene needs %zu bytes of LDS per block (polygon data, sphere tile, %u lights), limit is 64 KiB",
                rt_scene_lds_bytes(S), nL);
        return RT_E_INVALID;
    }
    c->have_scene = true;
    if (!same) {
        c->seen_valid = false;
        c->scene_serial++;
        c->jit_frames = 0;
    }
    return 0;
}

/* what this frame does about k_primary's launch plan (rt_hip_ctx above): follow the one it has, record the
 * waves' times for a new one, both or neither */
static int plan_frame(rt_hip_ctx *c, bool one_batch)
{
    c->plan_use = c->plan_record = false;
    const Job &j = c->job;
    const uint32_t nbx = (c->C.W + 31u) / 32u;
    if (!c->launch_plan || j.stats || !one_batch || nbx > 0x7fffu || j.ntiles > 0x7fffu)
        return 0;
    if (c->plan_valid && (c->plan_nbx != nbx || c->plan_ntiles != j.ntiles))
        c->plan_valid = false; /* an order of other blocks */
    const size_t blocks = (size_t)nbx * j.ntiles;
    if (blocks > c->plan_cap) {
        if (c->d_plan) (void)hipFree(c->d_plan);
        if (c->d_wave_ticks) (void)hipFree(c->d_wave_ticks);
        c->d_plan = c->d_wave_ticks = nullptr;
        c->plan_cap = 0;
        c->plan_valid = false;
        HIP_TRY(hipMalloc((void **)&c->d_plan, blocks * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&c->d_wave_ticks, blocks * (RT_BLOCK / 64u) * sizeof(uint32_t)));
        c->plan_cap = blocks;
    }
    /* the view: everything the tiles' times depend on */
    const uint64_t rest[6] = {c->scene_serial, (uint64_t)c->numerics, (uint64_t)c->shading, j.interleaved ? 1u : 0u,
                              ((uint64_t)j.row0 << 32) | j.nrows, ((uint64_t)j.phase << 32) | j.nphases};
    c->plan_view_now.assign((const char *)&c->C, (const char *)&c->C + sizeof c->C);
    c->plan_view_now.insert(c->plan_view_now.end(), (const char *)rest, (const char *)rest + sizeof rest);
    const bool fresh = c->plan_valid && c->plan_view_now == c->plan_view;
    c->plan_use = c->plan_valid;
    c->plan_age++;
    c->plan_record = !fresh && (!c->plan_valid || c->plan_age >= kPlanRefresh);
    return 0;
}

/* common part of the three device-resident render entry points: plan the batches, enqueue the
 * first round */
static int submit(rt_hip_ctx *c, const rt_camera *cam, uint32_t *d_argb, float *d_rad, const Job &job,
                  void *stream)
{
    int rc = ctx_set_device(c);
    if (rc)
        return rc;
    /* get_shade_color_phongblinn spawns nothing (raytrace.c:574-685): with depth 0 the pipeline is
     * k_primary -> k_resolve, no bounce and no fold launches */
    rt_camera flat_cam;
    if (c->shading == RT_SHADE_PHONGBLINN) {
        flat_cam = *cam;
        flat_cam.depth = 0;
        cam = &flat_cam;
    }
    if (c->pending) { /* the previous render's counters were never collected: drain it */
        rc = rt_hip_finish(c, nullptr);
        if (rc)
            return rc;
    }
    if (c->counters_dirty) { /* a render died between its first kernel and its k_publish */
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(c->d_counters, 0, (size_t)kMaxBatches * RT_COUNTER_WORDS * sizeof(uint32_t)));
        c->counters_dirty = false;
    }
    c->cam = *cam;
    fold_camera(c->C, *cam);
    {
        const uint64_t key[3] = {((uint64_t)cam->frameBufferWidth << 32) | cam->frameBufferHeight,
                                 ((uint64_t)cam->sqrtSpp << 32) | cam->depth,
                                 ((uint64_t)job.ntiles << 32) | ((uint64_t)c->shading << 16) |
                                     (job.interleaved ? job.nphases : 0u)};
        if (memcmp(key, c->seen_key, sizeof key) != 0) {
            memcpy(c->seen_key, key, sizeof key);
            c->seen_valid = false;
        }
    }
    c->cwords = RT_COUNTER_WORDS_FOR(cam->depth + 2);
    c->d_argb = d_argb;
    c->d_rad = d_rad;
    c->job = job;
    c->stream = (hipStream_t)stream;
    c->batches.clear();
    c->todo.clear();
    drop_events(c);
    memset(&c->stats, 0, sizeof c->stats);
    if (c->auto_cap) {
        /* A context that sizes itself holds what its frames USE, not what they could: a frame of the reference's
         * scene needs 0.08 entries per primary sample, a sphere cloud up to 2.  The first frame of a shape gets an
         * eighth of an entry per sample and is rendered in batches (a batch whose queues do overflow is halved and
         * rendered again: drain); from the second frame on the capacity is what the last frame of the shape needed
         * per sample, a quarter on top, so that the frame is ONE batch - which k_primary's launch plan needs, and
         * which pays the latency of the tree's sparse deep levels once.  Within [1 Mi, the limit]; the blob only
         * ever grows. */
        const uint64_t samples = (uint64_t)job.ntiles * 8u * c->C.W * c->C.sqrtSpp * c->C.sqrtSpp;
        const bool known = c->seen_valid && c->seen_samples > 0.0 && c->seen_need > 0.0;
        uint64_t want = known ? (uint64_t)((double)samples * (c->seen_need / c->seen_samples) * 1.25) + 1u : samples / 8u;
        want = round_cap(want < (1ull << 20) ? (1ull << 20) : want > c->cap_limit ? c->cap_limit : want);
        if (want > c->cap) {
            const double density = c->seen_density, need = c->seen_need, of = c->seen_samples;
            const bool valid = c->seen_valid;
            uint32_t hits[RT_MAX_LEVELS], nodes[RT_MAX_LEVELS];
            memcpy(hits, c->seen_hits, sizeof hits);
            memcpy(nodes, c->seen_nodes, sizeof nodes);
            if ((rc = ensure_work(c, want)))
                return rc;
            /* (what the last frame of this shape needed is still true with more room) */
            c->seen_valid = valid;
            c->seen_density = density;
            c->seen_need = need;
            c->seen_samples = of;
            memcpy(c->seen_hits, hits, sizeof hits);
            memcpy(c->seen_nodes, nodes, sizeof nodes);
        }
    }
    c->pending = true;
    if (job.ntiles == 0)
        return 0;
    uint64_t batch_samples = default_batch_samples(c);
    if (jobs_path(c, job.stats) && batch_samples > 0x3FFFFFFFull)
        batch_samples = 0x3FFFFFFFull; /* (the job pipeline's 30-bit sample index: ensure_level0) */
    const uint32_t per = tiles_per_batch(c, batch_samples);
    const uint32_t most = per < job.ntiles ? per : job.ntiles;
    if (jobs_path(c, job.stats) && (rc = ensure_loopq(c)))
        return rc;
    rc = jobs_path(c, job.stats) ? ensure_level0(c, (size_t)most * 8u * c->C.W, (size_t)c->C.sqrtSpp * c->C.sqrtSpp)
                                 : ensure_pix(c, (size_t)most * 8u * c->C.W, (size_t)c->C.sqrtSpp * c->C.sqrtSpp);
    if (rc)
        return rc;
    rc = plan_frame(c, per >= job.ntiles && !jobs_path(c, job.stats));
    if (rc)
        return rc;
    /* A small scene of another shape than the one the library has kernels for: after RT_JIT_AFTER frames of it the
     * trace kernels are compiled for its counts, on a thread of their own, and the frames switch when they are there
     * (RT_JIT_NOW: at once, and this frame waits). */
    c->jit_cur = nullptr;
    {
        static const int env = [] { const char *e = getenv("RT_HIP_JIT"); return e ? atoi(e) : -1; }(); /* overrides every context's mode */
        const int mode = env >= 0 ? env : c->jit_mode;
        const bool any_shape = mode == RT_JIT_NOW_ANY;
        if (mode != RT_JIT_OFF && !job.stats && c->shading == RT_SHADE_PHONG && !c->S.grid && c->S.tabInLds &&
            (any_shape || (!c->no_shape && rt_scene_shape(c->S) == 0))) {
            const bool now = mode >= RT_JIT_NOW;
            c->jit_cur = rt_jit_get(c->S, c->numerics == RT_NUMERICS_FAST, now || c->jit_frames >= RT_JIT_AFTER, now);
            c->jit_frames++;
        }
    }
    for (uint32_t t = job.ntiles; t > 0;) { /* pushed back to front: the list is popped from the end */
        const uint32_t n = (t % per) ? (t % per) : per;
        c->todo.push_back({t - n, t, 0});
        t -= n;
    }
    rc = enqueue_round(c);
    if (rc)
        c->counters_dirty = true;
    return rc;
}

int rt_hip_render_rows(rt_hip_ctx *c, const rt_camera *cam, uint32_t *d_argb, float *d_rad,
                       uint32_t row0, uint32_t nrows, void *stream)
{
    int rc = check_args(c, cam, d_argb, row0, nrows);
    if (rc)
        return rc;
    Job job;
    job.row0 = row0;
    job.nrows = nrows;
    job.ntiles = (nrows + 7u) / 8u;
    return submit(c, cam, d_argb, d_rad, job, stream);
}

int rt_hip_render_interleaved(rt_hip_ctx *c, const rt_camera *cam, uint32_t *d_argb_local,
                              float *d_rad_local, uint32_t phase, uint32_t nphases, void *stream)
{
    int rc = check_args(c, cam, d_argb_local, 0, 0);
    if (rc)
        return rc;
    if (nphases == 0 || phase >= nphases) {
        set_err("phase %u of %u", phase, nphases);
        return RT_E_INVALID;
    }
    const uint32_t nblocks = (cam->frameBufferHeight + 7u) / 8u;
    Job job;
    job.interleaved = true;
    job.phase = phase;
    job.nphases = nphases;
    job.ntiles = phase < nblocks ? (nblocks - phase + nphases - 1u) / nphases : 0u;
    return submit(c, cam, d_argb_local, d_rad_local, job, stream);
}

int rt_hip_finish(rt_hip_ctx *c, rt_hip_stats *stats)
{
    if (!c) {
        set_err("null argument");
        return RT_E_INVALID;
    }
    if (c->pending) {
        int rc = ctx_set_device(c);
        if (rc)
            return rc;
        const int was_timing = c->timing;
        /* only the first round is timed; re-rendered batches are not */
        std::vector<hipEvent_t> ev;
        ev.swap(c->ev);
        c->timing = 0;
        rc = drain(c);
        if (rc && rc != RT_E_OVERFLOW)
            c->counters_dirty = true;
        c->timing = was_timing;
        c->pending = false;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float ms = 0.f;
            if (rc == 0 && hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) {
                c->stats.k0_ms += ms;
                c->stats.k0_launches++;
            }
        }
        for (hipEvent_t e : ev)
            (void)hipEventDestroy(e);
        drop_events(c);
        if (rc)
            return rc;
        finish_stats(c);
    }
    if (stats)
        *stats = c->stats;
    return 0;
}

int rt_hip_render_stats(rt_hip_ctx *c, const rt_camera *cam, uint32_t *d_argb, float *d_rad,
                        uint32_t row0, uint32_t nrows, void *stream, rt_hip_stats *stats)
{
    int rc = check_args(c, cam, d_argb, row0, nrows);
    if (rc)
        return rc;
    Job job;
    job.row0 = row0;
    job.nrows = nrows;
    job.ntiles = (nrows + 7u) / 8u;
    job.stats = true;
    rc = submit(c, cam, d_argb, d_rad, job, stream);
    if (rc)
        return rc;
    return rt_hip_finish(c, stats);
}

/* ---- host-pointer entry points ---------------------------------------------------------- */

static std::mutex g_mu;
static rt_hip_ctx *g_ctx = nullptr; /* cached context of the blocking host path */
static uint32_t *g_argb = nullptr;  /* ... and its device frame buffers (grow-only) */
static float *g_rad = nullptr;
static size_t g_argb_px = 0, g_rad_px = 0;

/* Device memory -> the caller's (pageable) buffer through pinned staging, in pieces: every piece is
 * copied asynchronously up front, and the host moves piece i into place while piece i+1 is still in
 * flight.  (A plain hipMemcpy into pageable memory took 0.56 ms for a 1080p frame; this takes 0.3.) */
namespace {
constexpr int kPieces = 16;
constexpr size_t kPieceMin = (size_t)1 << 20;
char *g_stage = nullptr;
size_t g_stage_bytes = 0;
hipEvent_t g_piece_done[kPieces] = {};

int staged_copy_back(void *dst, const void *d_src, size_t bytes)
{
    if (!bytes)
        return 0;
    if (bytes > g_stage_bytes) {
        if (g_stage)
            (void)hipHostFree(g_stage);
        g_stage = nullptr;
        g_stage_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&g_stage, bytes, hipHostMallocDefault));
        g_stage_bytes = bytes;
    }
    if (!g_piece_done[0])
        for (auto &e : g_piece_done)
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    size_t piece = (bytes + kPieces - 1) / kPieces;
    if (piece < kPieceMin)
        piece = kPieceMin;
    piece = (piece + 255) & ~(size_t)255;
    int n = 0;
    for (size_t off = 0; off < bytes; off += piece, ++n) {
        const size_t len = bytes - off < piece ? bytes - off : piece;
        HIP_TRY(hipMemcpyAsync(g_stage + off, (const char *)d_src + off, len, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipEventRecord(g_piece_done[n], nullptr));
    }
    n = 0;
    for (size_t off = 0; off < bytes; off += piece, ++n) {
        const size_t len = bytes - off < piece ? bytes - off : piece;
        HIP_TRY(hipEventSynchronize(g_piece_done[n]));
        memcpy((char *)dst + off, g_stage + off, len);
    }
    return 0;
}
} // namespace

/* The blocking host path's pixels: the kernels store the packed frame straight into pinned, device-mapped host
 * memory (the stores go over the link while the kernels still run), and a few threads move it from there into the
 * caller's pageable buffer - one memcpy of a 1080p frame is 0.3 ms, a third of a warm raytrace() call. */
namespace {
uint32_t *g_hargb = nullptr, *g_hargb_dev = nullptr; /* pinned + mapped frame, its device address */
size_t g_hargb_px = 0;
uint8_t *g_tiles = nullptr; /* pinned: k_primary's flags of the tiles it finished (rt_hip_ctx::early_tiles) */
size_t g_tiles_n = 0;

class CopyPool { /* persistent helpers: each copies a slice; they spin briefly between jobs (frames come back to back), then sleep */
public:
    static CopyPool &get()
    {
        static CopyPool p;
        return p;
    }
    void copy(void *dst, const void *src, size_t bytes)
    {
        const int n = (int)workers_.size() + 1;
        if (n == 1 || bytes < ((size_t)1 << 20)) {
            memcpy(dst, src, bytes);
            return;
        }
        const size_t slice = ((bytes / n) + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu_);
            dst_ = (char *)dst; src_ = (const char *)src; bytes_ = bytes; slice_ = slice;
            pending_ = n - 1;
            ++generation_;
        }
        cv_.notify_all();
        memcpy(dst, src, slice < bytes ? slice : bytes); /* slice 0: this thread */
        while (pending_.load(std::memory_order_acquire) != 0)
            std::this_thread::yield();
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            ++generation_;
        }
        cv_.notify_all();
        for (std::thread &t : workers_)
            t.join();
    }

private:
    CopyPool()
    {
        const char *e = getenv("RT_HIP_COPY_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        int n = e && *e ? atoi(e) : (hw >= 16 ? 8 : (hw >= 8 ? 4 : 2)); /* (a 1080p frame: 0.177 / 0.117 / 0.066 / 0.080 ms with 2 / 4 / 8 / 12) */
        n = n < 1 ? 1 : (n > 16 ? 16 : n);
        for (int i = 1; i < n; ++i)
            workers_.emplace_back([this, i] { run(i); });
    }
    void run(int index)
    {
        uint64_t seen = 0;
        for (;;) {
            /* spin for the next job for a while, then wait on the condition variable */
            bool got = false;
            for (int spin = 0; spin < 20000 && !got; ++spin) {
                if (generation_.load(std::memory_order_acquire) != seen)
                    got = true;
                else
                    __builtin_ia32_pause();
            }
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_.load() != seen; });
                seen = generation_.load();
                if (stop_)
                    return;
            }
            const size_t off = slice_ * (size_t)index;
            if (off < bytes_)
                memcpy(dst_ + off, src_ + off, bytes_ - off < slice_ ? bytes_ - off : slice_);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<uint64_t> generation_{0};
    std::atomic<int> pending_{0};
    bool stop_ = false;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0, slice_ = 0;
};
} // namespace

/* RT_HIP_SHADING=phongblinn: the host-pointer entry points shade with the reference's older routine (the
 * reference itself switches by editing the call at raytrace.c:706) */
static int shading_from_env(void)
{
    const char *e = getenv("RT_HIP_SHADING");
    if (!e || !*e || !strcmp(e, "phong"))
        return RT_SHADE_PHONG;
    if (!strcmp(e, "phongblinn") || !strcmp(e, "blinn"))
        return RT_SHADE_PHONGBLINN;
    return -1; /* rejected by rt_hip_ctx_set_shading */
}

/* RT_HIP_NUMERICS=fast: the host-pointer entry points render with the tolerance build of the kernels */
static int numerics_from_env(void)
{
    const char *e = getenv("RT_HIP_NUMERICS");
    if (!e || !*e || !strcmp(e, "exact"))
        return RT_NUMERICS_EXACT;
    if (!strcmp(e, "fast"))
        return RT_NUMERICS_FAST;
    return -1; /* rejected by rt_hip_ctx_set_numerics */
}

static uint64_t g_cap_floor = 0; /* queue entries rt_hip_render's context gets at least (raised when a frame did not fit) */

static int render_locked(const rt_scene *scene, const rt_camera *cam, rt_framebuffer *fb)
{
    if (!scene || !cam || !fb || !fb->argb) {
        set_err("null argument");
        return RT_E_INVALID;
    }
    if (fb->width != cam->frameBufferWidth || fb->height != cam->frameBufferHeight) {
        set_err("frame buffer %ux%u does not match the camera's %ux%u", fb->width, fb->height,
                cam->frameBufferWidth, cam->frameBufferHeight);
        return RT_E_INVALID;
    }
    int rc;
    /* RT_HIP_TIMES=1: where the call's time goes, to stderr (development; tools/one_shot.py) */
    static const bool times = getenv("RT_HIP_TIMES") != nullptr;
    struct timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto since = [&]() {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return (double)(t.tv_sec - ts0.tv_sec) * 1e3 + (double)(t.tv_nsec - ts0.tv_nsec) * 1e-6;
    };
    double t_ctx = 0.0, t_upload = 0.0, t_buffers = 0.0, t_render = 0.0;
    /* "the current HIP device": the cached context and frame buffers follow the caller when it has
     * switched devices since the last call */
    int current = 0;
    HIP_TRY(hipGetDevice(&current));
    if (g_ctx && g_ctx->device != current) {
        (void)hipSetDevice(g_ctx->device);
        rt_hip_ctx_destroy(g_ctx);
        g_ctx = nullptr;
        if (g_argb) (void)hipFree(g_argb);
        if (g_rad) (void)hipFree(g_rad);
        if (g_hargb) (void)hipHostFree(g_hargb);
        g_argb = nullptr; g_rad = nullptr; g_hargb = g_hargb_dev = nullptr;
        g_argb_px = g_rad_px = g_hargb_px = 0;
        HIP_TRY(hipSetDevice(current));
    }
    /* queue capacity sized to the job: 2 entries per primary sample, 1 Mi .. default */
    const uint64_t samples = (uint64_t)fb->width * fb->nrows * cam->sqrtSpp * cam->sqrtSpp;
    uint64_t want = samples * 2 < (1ull << 20) ? (1ull << 20) : samples * 2;
    if (want < g_cap_floor)
        want = g_cap_floor;
    if (want > kDefaultCap)
        want = kDefaultCap;
    if (g_ctx && g_ctx->cap < want) {
        rt_hip_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
    if (!g_ctx) {
        rc = rt_hip_ctx_create(&g_ctx, -1, want);
        if (rc)
            return rc;
    }
    t_ctx = since();
    rc = rt_hip_ctx_set_shading(g_ctx, shading_from_env());
    if (!rc)
        rc = rt_hip_ctx_set_numerics(g_ctx, numerics_from_env());
    if (rc)
        return rc;
    rc = rt_hip_scene_upload(g_ctx, scene);
    if (rc)
        return rc;
    t_upload = since();
    const size_t npix = (size_t)fb->width * fb->height;
    if (npix > g_argb_px) {
        if (g_argb)
            (void)hipFree(g_argb);
        g_argb = nullptr;
        g_argb_px = 0;
        HIP_TRY(hipMalloc((void **)&g_argb, npix * sizeof(uint32_t)));
        g_argb_px = npix;
    }
    if (fb->radiance && npix > g_rad_px) {
        if (g_rad)
            (void)hipFree(g_rad);
        g_rad = nullptr;
        g_rad_px = 0;
        HIP_TRY(hipMalloc((void **)&g_rad, npix * 3 * sizeof(float)));
        g_rad_px = npix;
    }
    uint32_t *d_argb = g_argb;
    float *d_rad = fb->radiance ? g_rad : nullptr;
    /* the packed pixels reach the host in pinned, mapped memory, most of them while the frame is still being
     * finished (rt_hip_ctx::early_host); RT_HIP_NO_EARLY_COPY=1: one staged copy after the frame, as for the radiance */
    static const bool mapped = getenv("RT_HIP_NO_EARLY_COPY") == nullptr;
    if (mapped && npix > g_hargb_px) {
        if (g_hargb)
            (void)hipHostFree(g_hargb);
        g_hargb = g_hargb_dev = nullptr;
        g_hargb_px = 0;
        HIP_TRY(hipHostMalloc((void **)&g_hargb, npix * sizeof(uint32_t), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&g_hargb_dev, g_hargb, 0));
        g_hargb_px = npix;
    }
    if (mapped && !g_ctx->early_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&g_ctx->early_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g_ctx->early_from, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g_ctx->early_done, hipEventDisableTiming));
    }
    g_ctx->early_host = mapped ? g_hargb : nullptr;
    g_ctx->early_alias = mapped ? g_hargb_dev : nullptr;
    const uint32_t tile_rows = (fb->nrows + 7u) / 8u, tiles_per_row = ((fb->width + 31u) / 32u) * (RT_BLOCK / 64u);
    if (mapped && (size_t)tile_rows * tiles_per_row > g_tiles_n) {
        if (g_tiles)
            (void)hipHostFree(g_tiles);
        g_tiles = nullptr;
        g_tiles_n = 0;
        HIP_TRY(hipHostMalloc((void **)&g_tiles, (size_t)tile_rows * tiles_per_row, hipHostMallocPortable));
        g_tiles_n = (size_t)tile_rows * tiles_per_row;
    }
    g_ctx->early_tiles = mapped ? g_tiles : nullptr;
    g_ctx->early_tiles_n = 0;
    t_buffers = since();
    rc = rt_hip_render_rows(g_ctx, cam, d_argb, d_rad, fb->row0, fb->nrows, nullptr);
    /* While the levels of the ray tree still run: the tile rows whose every tile k_primary finished (their pixels
     * are in the mapped frame once the early copy is through) move into the caller's buffer now, the others after
     * the frame.  Config 2: 40 % of the rows; the copy after the frame shrinks accordingly. */
    std::vector<uint8_t> moved(tile_rows, 0);
    if (!rc && mapped && g_ctx->early_tiles_n == tile_rows * tiles_per_row && g_ctx->batches.size() == 1 &&
        hipEventSynchronize(g_ctx->early_done) == hipSuccess) {
        for (uint32_t t = 0; t < tile_rows;) {
            auto final_row = [&](uint32_t r) {
                const uint8_t *f = g_tiles + (size_t)r * tiles_per_row;
                for (uint32_t k = 0; k < tiles_per_row; ++k)
                    if (f[k] != 1u)
                        return false;
                return true;
            };
            if (!final_row(t)) {
                ++t;
                continue;
            }
            uint32_t e = t;
            while (e < tile_rows && final_row(e))
                moved[e++] = 1;
            const uint32_t r0 = fb->row0 + t * 8u, r1 = fb->row0 + (e * 8u < fb->nrows ? e * 8u : fb->nrows);
            CopyPool::get().copy(fb->argb + (size_t)r0 * fb->width, g_hargb + (size_t)r0 * fb->width, (size_t)(r1 - r0) * fb->width * sizeof(uint32_t));
            t = e;
        }
    }
    if (!rc)
        rc = rt_hip_finish(g_ctx, nullptr);
    t_render = since();
    if (!rc && fb->nrows) {
        const size_t off = (size_t)fb->row0 * fb->width, cnt = (size_t)fb->nrows * fb->width;
        if (mapped) {
            for (uint32_t t = 0; t < tile_rows;) { /* the runs of tile rows that have not been moved yet */
                if (moved[t]) {
                    ++t;
                    continue;
                }
                uint32_t e = t;
                while (e < tile_rows && !moved[e])
                    ++e;
                const uint32_t r0 = fb->row0 + t * 8u, r1 = fb->row0 + (e * 8u < fb->nrows ? e * 8u : fb->nrows);
                CopyPool::get().copy(fb->argb + (size_t)r0 * fb->width, g_hargb + (size_t)r0 * fb->width, (size_t)(r1 - r0) * fb->width * sizeof(uint32_t));
                t = e;
            }
        } else
            rc = staged_copy_back(fb->argb + off, d_argb + off, cnt * sizeof(uint32_t));
        if (!rc && fb->radiance)
            rc = staged_copy_back(fb->radiance + 3 * off, d_rad + 3 * off, cnt * 3 * sizeof(float));
    }
    if (times)
        fprintf(stderr, "rt_hip_render: context %.3f ms, scene %.3f, buffers %.3f, render %.3f, pixels back %.3f, total %.3f ms\n",
                t_ctx, t_upload - t_ctx, t_buffers - t_upload, t_render - t_buffers, since() - t_render, since());
    return rc;
}

int rt_hip_render(const rt_scene *scene, const rt_camera *cam, rt_framebuffer *fb)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (;;) {
        const int rc = render_locked(scene, cam, fb);
        /* The context holds two queue entries per sample of the frame (1 Mi at least).  One tile row of a small frame
         * whose every hit spawns two rays can need more - 2^(depth + 1) nodes a sample: four times the room then,
         * and the frame again, until it fits or the capacity is the library's default (raytrace() renders what it is
         * handed, like the reference; a caller of the context interface picks its own capacity). */
        if (rc != RT_E_OVERFLOW || !g_ctx || g_ctx->cap >= kDefaultCap)
            return rc;
        g_cap_floor = g_ctx->cap * 4u;
    }
}

/* ---- the host-pointer path on several GPUs of this process -------------------------------- */

namespace {
struct DevSlot { /* one per entry of the caller's device list (the same device may appear twice) */
    int device = -1;
    rt_hip_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    uint32_t *d_argb = nullptr;
    float *d_rad = nullptr;
    size_t argb_px = 0, rad_px = 0;
    /* pinned staging of the compact buffers: one contiguous asynchronous copy per device (a strided
     * copy straight into pageable memory took 0.9 ms longer per 1080p frame), scattered by the host */
    uint32_t *h_argb = nullptr;
    float *h_rad = nullptr;
    size_t h_argb_px = 0, h_rad_px = 0;
};
std::vector<DevSlot> g_slots;

void slot_release(DevSlot &s)
{
    if (s.device >= 0)
        (void)hipSetDevice(s.device);
    if (s.ctx) rt_hip_ctx_destroy(s.ctx);
    if (s.d_argb) (void)hipFree(s.d_argb);
    if (s.d_rad) (void)hipFree(s.d_rad);
    if (s.h_argb) (void)hipHostFree(s.h_argb);
    if (s.h_rad) (void)hipHostFree(s.h_rad);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = DevSlot();
}

/* staged compact buffer of rank r -> its 8-row blocks of the host frame */
void scatter_blocks(void *host, const void *staged, size_t px_bytes, uint32_t W, uint32_t H, uint32_t r, uint32_t n)
{
    const uint32_t nblocks = (H + 7u) / 8u;
    const size_t block_bytes = (size_t)8u * W * px_bytes;
    for (uint32_t k = 0, b = r; b < nblocks; ++k, b += n) {
        const uint32_t rows = (H - b * 8u) < 8u ? (H - b * 8u) : 8u;
        memcpy((char *)host + (size_t)b * block_bytes, (const char *)staged + (size_t)k * block_bytes,
               (size_t)rows * W * px_bytes);
    }
}
} // namespace

static int render_devices_locked(const rt_scene *scene, const rt_camera *cam, rt_framebuffer *fb, const int *devices,
                          uint32_t ndevices)
{
    if (!scene || !cam || !fb || !fb->argb || !devices || ndevices == 0 || ndevices > 64) {
        set_err("null argument or a device list of %u entries (1..64)", ndevices);
        return RT_E_INVALID;
    }
    if (fb->width != cam->frameBufferWidth || fb->height != cam->frameBufferHeight || fb->row0 != 0 ||
        fb->nrows != fb->height) {
        set_err("rt_hip_render_devices renders whole frames: buffer %ux%u rows [%u,+%u), camera %ux%u", fb->width,
                fb->height, fb->row0, fb->nrows, cam->frameBufferWidth, cam->frameBufferHeight);
        return RT_E_INVALID;
    }
    int have = 0;
    HIP_TRY(hipGetDeviceCount(&have));
    for (uint32_t r = 0; r < ndevices; ++r)
        if (devices[r] < 0 || devices[r] >= have) {
            set_err("device %d of the list does not exist (%d visible)", devices[r], have);
            return RT_E_INVALID;
        }
    int before = 0;
    HIP_TRY(hipGetDevice(&before));
    struct Restore { /* the caller's current device comes back on every exit path */
        int device;
        ~Restore() { (void)hipSetDevice(device); }
    } restore{before};
    const uint32_t W = fb->width, H = fb->height, n = ndevices;
    const uint32_t nblocks = (H + 7u) / 8u, per = (nblocks + n - 1u) / n;
    const size_t local_px = (size_t)per * 8u * W;
    const uint64_t samples = (uint64_t)local_px * cam->sqrtSpp * cam->sqrtSpp;
    uint64_t want = samples * 2 < (1ull << 20) ? (1ull << 20) : samples * 2;
    if (want < g_cap_floor)
        want = g_cap_floor;
    if (want > kDefaultCap)
        want = kDefaultCap;

    /* slots follow the list; a changed list starts over */
    bool same = g_slots.size() == n;
    for (uint32_t r = 0; same && r < n; ++r)
        same = g_slots[r].device == devices[r];
    if (!same) {
        for (DevSlot &s : g_slots)
            slot_release(s);
        g_slots.assign(n, DevSlot());
    }
    int rc = 0;
    for (uint32_t r = 0; r < n && !rc; ++r) { /* every device gets its share before anyone waits */
        DevSlot &s = g_slots[r];
        s.device = devices[r];
        HIP_TRY(hipSetDevice(s.device));
        if (s.ctx && s.ctx->cap < want) {
            rt_hip_ctx_destroy(s.ctx);
            s.ctx = nullptr;
        }
        if (!s.ctx && (rc = rt_hip_ctx_create(&s.ctx, s.device, want)))
            break;
        if (!s.stream)
            HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        if ((rc = rt_hip_ctx_set_shading(s.ctx, shading_from_env())) ||
            (rc = rt_hip_ctx_set_numerics(s.ctx, numerics_from_env())) || (rc = rt_hip_scene_upload(s.ctx, scene)))
            break;
        if (local_px > s.argb_px) {
            if (s.d_argb) (void)hipFree(s.d_argb);
            s.d_argb = nullptr; s.argb_px = 0;
            HIP_TRY(hipMalloc((void **)&s.d_argb, local_px * sizeof(uint32_t)));
            s.argb_px = local_px;
        }
        if (fb->radiance && local_px > s.rad_px) {
            if (s.d_rad) (void)hipFree(s.d_rad);
            s.d_rad = nullptr; s.rad_px = 0;
            HIP_TRY(hipMalloc((void **)&s.d_rad, local_px * 3 * sizeof(float)));
            s.rad_px = local_px;
        }
        if (local_px > s.h_argb_px) {
            if (s.h_argb) (void)hipHostFree(s.h_argb);
            s.h_argb = nullptr; s.h_argb_px = 0;
            HIP_TRY(hipHostMalloc((void **)&s.h_argb, local_px * sizeof(uint32_t), hipHostMallocDefault));
            s.h_argb_px = local_px;
        }
        if (fb->radiance && local_px > s.h_rad_px) {
            if (s.h_rad) (void)hipHostFree(s.h_rad);
            s.h_rad = nullptr; s.h_rad_px = 0;
            HIP_TRY(hipHostMalloc((void **)&s.h_rad, local_px * 3 * sizeof(float), hipHostMallocDefault));
            s.h_rad_px = local_px;
        }
        rc = rt_hip_render_interleaved(s.ctx, cam, s.d_argb, fb->radiance ? s.d_rad : nullptr, r, n, s.stream);
    }
    for (uint32_t r = 0; r < n; ++r) { /* collect in list order; later devices keep rendering meanwhile */
        DevSlot &s = g_slots[r];
        if (!s.ctx || !s.ctx->pending)
            continue;
        const int rf = rt_hip_finish(s.ctx, nullptr);
        if (!rc)
            rc = rf;
        if (rc)
            continue;
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipMemcpyAsync(s.h_argb, s.d_argb, local_px * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
        if (fb->radiance)
            HIP_TRY(hipMemcpyAsync(s.h_rad, s.d_rad, local_px * 3 * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    }
    for (uint32_t r = 0; r < n; ++r)
        if (g_slots[r].stream) {
            (void)hipSetDevice(g_slots[r].device);
            const hipError_t e = hipStreamSynchronize(g_slots[r].stream);
            if (e != hipSuccess && !rc) {
                set_err("copying pixels back from device %d: %s", g_slots[r].device, hipGetErrorString(e));
                rc = -(int)e;
            }
            if (!rc) { /* this device's blocks into the caller's frame while the others still copy */
                scatter_blocks(fb->argb, g_slots[r].h_argb, sizeof(uint32_t), W, H, r, n);
                if (fb->radiance)
                    scatter_blocks(fb->radiance, g_slots[r].h_rad, 3 * sizeof(float), W, H, r, n);
            }
        }
    return rc;
}

int rt_hip_render_devices(const rt_scene *scene, const rt_camera *cam, rt_framebuffer *fb, const int *devices,
                          uint32_t ndevices)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (;;) { /* (as rt_hip_render: a frame that does not fit the contexts' queues is rendered again with four times the room) */
        const int rc = render_devices_locked(scene, cam, fb, devices, ndevices);
        uint64_t cap = 0;
        for (const DevSlot &s : g_slots)
            if (s.ctx && s.ctx->cap > cap)
                cap = s.ctx->cap;
        if (rc != RT_E_OVERFLOW || cap == 0 || cap >= kDefaultCap)
            return rc;
        g_cap_floor = cap * 4u;
    }
}

/* RT_HIP_DEVICES = "all" | "0,1,2,3": the device list raytrace() spreads a frame over (unset: the
 * current device).  Returns the count written to out[]. */
static uint32_t devices_from_env(int *out, uint32_t room)
{
    const char *env = getenv("RT_HIP_DEVICES");
    if (!env || !*env)
        return 0;
    uint32_t n = 0;
    if (!strcmp(env, "all")) {
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess)
            return 0;
        for (int d = 0; d < have && n < room; ++d)
            out[n++] = d;
        return n;
    }
    for (const char *p = env; *p && n < room;) {
        char *end = nullptr;
        const long d = strtol(p, &end, 10);
        if (end == p)
            break;
        out[n++] = (int)d;
        p = (*end == ',') ? end + 1 : end;
        if (*end && *end != ',')
            break;
    }
    return n;
}

void raytrace(unsigned int *buffer, rt_scene_t *scene, float fovY, float aspectRatio, float nearZ,
              float farZ, unsigned int width, unsigned int height, unsigned int samplesPerPixelSq,
              unsigned int depth)
{
    (void)farZ; /* never read by the reference either */
    rt_camera cam;
    rt_camera_prepare(&cam, scene, width, height, samplesPerPixelSq, fovY, aspectRatio, nearZ, depth);
    /* prepare_scene writes these into the caller's scene (raytrace.c:77-84, 894) */
    scene->viewDistance = cam.viewDistance;
    scene->viewPlaneHalfHeight = cam.viewPlaneHalfHeight;
    scene->viewPlaneHalfWidth = cam.viewPlaneHalfWidth;
    scene->sppWidth = cam.sppWidth;
    scene->frameBufferWidth = width;
    scene->frameBufferHeight = height;
    scene->sqrtSpp = samplesPerPixelSq;

    rt_scene view;
    view.bgColor = scene->bgColor;
    view.ambientLightColor = scene->ambientLightColor;
    view.nLights = scene->nLights;
    view.nObjects = scene->nObjects;
    view.lights = scene->lights;
    view.objects = scene->objects;
    rt_framebuffer fb;
    fb.argb = buffer;
    fb.radiance = nullptr;
    fb.width = width;
    fb.height = height;
    fb.row0 = 0;
    fb.nrows = height;
    int devs[64];
    const uint32_t ndev = devices_from_env(devs, 64);
    const int rc = ndev > 1 ? rt_hip_render_devices(&view, &cam, &fb, devs, ndev) : rt_hip_render(&view, &cam, &fb);
    if (rc) {
        fprintf(stderr, "raytrace (HIP): error %d: %s\n", rc, rt_hip_last_error());
        exit(1);
    }
}

} /* extern "C" */
