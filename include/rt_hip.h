/*
 * rt_hip.h - C-ABI of the MI355X (gfx950) trace-and-shade library `librt_hip.so`.
 *
 * This is the drop-in boundary for the reference's raytrace.c (davidoguns/raytracer).
 * Every declaration cites the reference interface it replaces as file:line relative to
 * the reference tree.  Plain C: pointers and sizes only, no torch / C++ types.
 *
 * Layout contract: the POD types below are written from scratch but are byte-for-byte
 * layout-compatible with the reference's x86-64 (SysV, non-Cell) structs, so that a host
 * program compiled against the reference headers can hand its scene_t straight to
 * raytrace() in this library.  Sizes/offsets are pinned by _Static_assert at the bottom.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------ */
/* Reference-layout data types                                                          */
/* ------------------------------------------------------------------------------------ */

/* vector4_t / point_t  -- vector4.h:14-29 (16 B; w is carried but ignored by the math) */
typedef struct { float x, y, z, w; } rt_vec4;

/* color_t -- color.h:17-22 (12 B) */
typedef struct { float r, g, b; } rt_color;

/* plane_t -- plane.h:14-29: A*x + B*y + C*z + F = 0 (16 B) */
typedef struct { float A, B, C, F; } rt_plane;

/* sphere_t -- geometry.h:22-26 (20 B) */
typedef struct { rt_vec4 center; float radius; } rt_sphere;

/* polygon_t -- geometry.h:34-48 (32 B): plane + vertex list (host pointer) */
typedef struct {
    rt_plane  plane;
    uint32_t  nVerticies;
    rt_vec4  *vertex;
} rt_polygon;

/* material_t -- materials.h:24-34 (52 B); colors[0] = diffuse, colors[1] = specular */
#define RT_MATERIAL_DIFFUSECOLOR  0
#define RT_MATERIAL_SPECULARCOLOR 1
typedef struct {
    rt_color colors[2];
    float    phong_ka, phong_kd, phong_ks, phong_ke;
    float    kr, kt, n;
} rt_material;

/* geometry tags -- geometry.h:17-18 */
#define RT_GEOMETRY_SPHERE  0x01
#define RT_GEOMETRY_POLYGON 0x02

/* object3d_t -- object3d.h:17-35, x86 variant with debugName (104 B) */
typedef struct {
    uint16_t    geometryType;
    char       *debugName;
    rt_material material;
    union {
        rt_sphere  sphr_obj;
        rt_polygon poly_obj;
    };
} rt_object3d;

/* pointlight_t -- light.h:17-26 (32 B); range is never read by the reference */
typedef struct { rt_vec4 position; rt_color color; float range; } rt_pointlight;

/* scene_t -- scene.h:22-64, non-SPU variant (184 B).  The seven view-plane fields are
 * written by raytrace() exactly like the reference's prepare_scene (raytrace.c:72-85). */
typedef struct {
    rt_vec4        eyePos, lookAt, upVec;
    rt_vec4        U, V, N;
    uint32_t       frameBufferWidth, frameBufferHeight, sqrtSpp;
    float          viewDistance, viewPlaneHalfWidth, viewPlaneHalfHeight, sppWidth;
    rt_color       bgColor, ambientLightColor;
    float          ldMax, lMax;
    uint32_t       nLights, nObjects;
    rt_pointlight *lights;
    rt_object3d   *objects;
} rt_scene_t;

/* ------------------------------------------------------------------------------------ */
/* north_star entry point types                                                         */
/* ------------------------------------------------------------------------------------ */

/* The part of scene_t the trace-and-shade loop reads (scene.h:39-40,45-46,58-59). */
typedef struct {
    rt_color             bgColor;
    rt_color             ambientLightColor;
    uint32_t             nLights;
    uint32_t             nObjects;
    const rt_pointlight *lights;    /* host pointer, nLights entries   */
    const rt_object3d   *objects;   /* host pointer, nObjects entries  */
} rt_scene;

/* The reference has no camera type: camera = scene_t.{eyePos,U,V,N} (scene.h:24-29) + the
 * view-plane fields prepare_scene fills (scene.h:31-37, raytrace.c:77-84) + MAX_DEPTH
 * (raytrace.c:26,891). */
typedef struct {
    rt_vec4  eyePos, U, V, N;
    uint32_t frameBufferWidth, frameBufferHeight, sqrtSpp;
    float    viewDistance, viewPlaneHalfWidth, viewPlaneHalfHeight, sppWidth;
    uint32_t depth;
} rt_camera;

/* Caller-owned frame buffer.  argb[x + y*width] = 0xFFRRGGBB, row 0 = top
 * (raytrace.c:783-790, pam_output/output.c:25-27).  radiance (optional, may be NULL) receives
 * the un-quantised per-pixel colour (the reference's internal colorbuffer, raytrace.c:887)
 * as 3 floats per pixel.  Only rows [row0, row0+nrows) are rendered and written; both
 * arrays are indexed with full-frame coordinates. */
typedef struct {
    uint32_t *argb;
    float    *radiance;
    uint32_t  width, height;
    uint32_t  row0, nrows;
} rt_framebuffer;

/* Ray / work counters of the last render (deterministic functions of scene + camera). */
typedef struct {
    uint64_t primary_rays;      /* W*nrows*sqrtSpp^2                  raytrace.c:746-772 */
    uint64_t shadow_rays;       /* one per shade node per light       raytrace.c:327-335 */
    uint64_t reflection_rays;   /*                                    raytrace.c:405-431 */
    uint64_t transmission_rays; /* refracted + total-internal-refl.   raytrace.c:452-539 */
    uint64_t shade_nodes;       /* calls of get_shade_color_phong     raytrace.c:282     */
    uint64_t tree_nodes;        /* shade nodes that spawned children AND were stored in HBM: the sparse deep
                                   levels of a tree are finished in registers once a context knows a frame  */
    uint64_t queued_rays;       /* secondary-ray hits that went through the HBM queues (likewise)            */
    uint32_t batches;           /* image batches the render was split into               */
    uint32_t overflow_retries;  /* batches redone because a queue overflowed             */
    /* algorithmic work, filled only by rt_hip_render_stats (instrumented kernels):       */
    uint64_t sphere_tests, sphere_hits;
    uint64_t polygon_tests, polygon_rej_parallel, polygon_rej_behind, polygon_full, polygon_hits;
    uint64_t lights_lit;        /* (node, light) pairs that passed shadow + facing tests */
    uint64_t nodes_sphere, nodes_polygon;
    /* polygon inside tests the cheap screening pass could not decide (they took the exact
     * path), and screening verdicts that contradict the exact test - must be 0 */
    uint64_t polygon_screen_unsure, polygon_screen_mismatch;
    /* the same, restricted to the fused primary kernel (the dominant kernel):            */
    uint64_t k0_sphere_tests, k0_sphere_hits;
    uint64_t k0_polygon_tests, k0_polygon_rej_parallel, k0_polygon_rej_behind, k0_polygon_full;
    uint64_t k0_shade_nodes, k0_nodes_sphere, k0_nodes_polygon, k0_lights_lit;
    uint64_t k0_shadow_rays, k0_reflection_rays, k0_transmission_rays;
    /* device time of the fused primary kernel, HIP events on the render stream (ms),
     * summed over batches; 0 unless timing was enabled with rt_hip_ctx_set_timing.        */
    float    k0_ms;
    uint32_t k0_launches;
    /* primary rays of pixel tiles that were finished WITHOUT tracing: every sample of such a tile was
     * shown to miss every object (the reference traces them and gets the background, raytrace.c:700-703).
     * They are part of primary_rays (the reference's count); primary_rays - primary_rays_culled were
     * generated and traced.  0 for instrumented renders, which trace everything. */
    uint64_t primary_rays_culled;
    /* k_primary launches of this render that followed a launch plan, and plans built at its end
     * (rt_hip_ctx_set_launch_plan) */
    uint32_t planned_launches, plan_builds;
    uint32_t work_mib;          /* MiB of device memory the context holds for queues, nodes and per-sample arrays */
    uint32_t reserved_;
    /* frames of this render launched as a HIP graph, and graphs captured for them (rt_hip_ctx_set_graph) */
    uint32_t graph_launches, graph_captures;
} rt_hip_stats;

/* ------------------------------------------------------------------------------------ */
/* Entry points                                                                         */
/* ------------------------------------------------------------------------------------ */

/* Drop-in for the reference's render entry -- raytrace.h:17-20 / raytrace.c:874-931, sole
 * call site main.c:159-160.  Same argument meaning; mutates *scene like prepare_scene
 * (raytrace.c:894); blocking; on a HIP failure prints to stderr and exit(1) (the reference
 * has no error channel: void return, main.c:153-157 style).  farZ is ignored (as in the
 * reference). */
void raytrace(unsigned int *buffer, rt_scene_t *scene,
              float fovY, float aspectRatio, float nearZ, float farZ,
              unsigned int width, unsigned int height,
              unsigned int samplesPerPixelSq, unsigned int depth);

/* north_star entry: render `fb` rows of `scene` seen through `camera` on the current HIP
 * device.  Host pointers in, host pixels out (H2D of the flattened scene, kernels, D2H).
 * Replaces the body of raytrace() after prepare_scene (raytrace.c:903-927).
 * Returns 0 or a negative hipError_t / RT_E_* code.  The queues are the call's own business: a
 * frame whose ray trees do not fit them even one tile row at a time is rendered again with four
 * times the room, up to the default capacity (rt_hip_ctx_create), before RT_E_OVERFLOW is returned. */
int rt_hip_render(const rt_scene *scene, const rt_camera *camera, rt_framebuffer *fb);

/* The same blocking, host-pointer render spread over several GPUs of THIS process: device
 * devices[r] renders the 8-row blocks b with b % ndevices == r (rt_hip_render_interleaved) and its
 * blocks are copied straight into the caller's frame - the host frame is the gather point, so no
 * collective is involved (the reference's Cell host collects its workers' pixels the same way,
 * ppu/ppu_raytrace.c:312-338).  Whole frames only (row0 = 0, nrows = height).  A device may be
 * listed more than once.  raytrace() takes this path when the environment names more than one
 * device: RT_HIP_DEVICES=all or RT_HIP_DEVICES=0,1,2,3.  Pixels are identical to rt_hip_render's. */
int rt_hip_render_devices(const rt_scene *scene, const rt_camera *camera, rt_framebuffer *fb,
                          const int *devices, uint32_t ndevices);

/* prepare_scene -- raytrace.c:72-85: fills the view-plane fields of *cam from the
 * projection parameters (double-precision tan, stored to float), copies eye/U/V/N. */
void rt_camera_prepare(rt_camera *cam, const rt_scene_t *scene,
                       unsigned int width, unsigned int height, unsigned int sqrtSpp,
                       float fovY, float aspectRatio, float nearZ, unsigned int depth);

/* ---- device-resident interface (what bench.py and the multi-GPU host use) ----------- */

typedef struct rt_hip_ctx rt_hip_ctx;

#define RT_E_INVALID   (-10001)   /* bad argument                                   */
#define RT_E_OVERFLOW  (-10002)   /* ray/node queue overflow even at 1-row batches  */
#define RT_E_NOSCENE   (-10003)   /* render before rt_hip_scene_upload              */

/* Create a render context on HIP device `device` (-1 = current).  max_queue_rays = capacity of each per-level
 * hit queue and of the node pool, 152 bytes per entry.  0 = sized by the renders, from what they USE: the first
 * frame of a shape gets an eighth of an entry per primary sample (at least 1 Mi entries) and is rendered in row
 * batches; from the second frame on the context holds what the last frame of that shape needed per sample plus a
 * quarter - 0.08 entries per sample for the reference's scene (a 1920x1080 x 16 spp context: 0.5 GiB; 3840x2160:
 * 2 GiB), up to 2 for sphere clouds - so that the frame is ONE batch (k_primary's launch plan needs that, and
 * every batch pays the latency of the tree's sparse deep levels once).  At most RT_HIP_QUEUE_CAP entries from
 * the environment, else 256 Mi = 38 GiB of the GPU's 288; when the device cannot give that much the context
 * settles for half, a quarter, ... and renders in batches.  stats.work_mib reports what is held. */
int  rt_hip_ctx_create(rt_hip_ctx **out, int device, uint64_t max_queue_rays);
void rt_hip_ctx_destroy(rt_hip_ctx *ctx);

/* Flatten the AoS host scene (object3d_t[], pointlight_t[], polygon vertex lists --
 * the arrays the Cell build DMAs in spu/spu_main.c:54-148) into SoA device arrays.  Refused with RT_E_INVALID and
 * a message (rt_hip_last_error), the context keeping the scene it had: polygon data beyond a block's LDS (some
 * hundred quads), and - in scenes of more spheres than a block's LDS holds, which are searched through grids - a
 * sphere or light whose coordinates are not finite or lie beyond 1e15. */
int  rt_hip_scene_upload(rt_hip_ctx *ctx, const rt_scene *scene);

/* Enqueue the render of rows [row0,row0+nrows) on `stream` (a hipStream_t, NULL = default
 * stream).  d_argb / d_radiance are DEVICE pointers indexed with full-frame coordinates
 * (d_radiance may be NULL).  Asynchronous with respect to the host. */
int  rt_hip_render_rows(rt_hip_ctx *ctx, const rt_camera *camera,
                        uint32_t *d_argb, float *d_radiance,
                        uint32_t row0, uint32_t nrows, void *stream);

/* Multi-GPU partition (the Cell build's pixel-cyclic split, ppu/ppu_raytrace.c:237-263 and
 * spu/spu_raytrace.c:559-565, at 8-row granularity): render the 8-row blocks b of the frame with
 * b % nphases == phase into a COMPACT local buffer - block b lands in rows [8*(b/nphases), +8) of
 * d_argb_local / d_radiance_local, which hold 8*ceil(ceil(H/8)/nphases) rows of `width` pixels.
 * Cyclic blocks balance the very uneven cost of sky and sphere rows; an all-gather of the local
 * buffers followed by one de-interleave copy rebuilds the frame.  Asynchronous like
 * rt_hip_render_rows. */
int  rt_hip_render_interleaved(rt_hip_ctx *ctx, const rt_camera *camera,
                               uint32_t *d_argb_local, float *d_radiance_local,
                               uint32_t phase, uint32_t nphases, void *stream);

/* After the all-gather of the ranks' compact buffers (rank r's buffer at rows [r*L, (r+1)*L) of
 * d_gathered, L = 8*ceil(ceil(H/8)/nphases) rows of `width` pixels): one streaming copy that puts
 * block k of rank r at frame block k*nphases + r.  d_frame holds height rows.  No counterpart in the
 * reference (its Cell host copies pixel by pixel, ppu/ppu_raytrace.c:312-338).  Asynchronous. */
int  rt_hip_deinterleave(const uint32_t *d_gathered, uint32_t *d_frame, uint32_t width, uint32_t height,
                         uint32_t nphases, void *stream);

/* Tone reproduction -- apply_tone / ward_tone / reinhard_tone, raytrace.c:792-871: the step the
 * reference keeps between its pixel loop and the 8-bit pack (its call is commented out at
 * raytrace.c:917).  d_radiance: DEVICE frame of 3 floats per pixel as the render calls write it,
 * changed in place; d_argb (may be NULL) is re-packed from it (raytrace.c:783-790).
 * op: RT_TONE_WARD (what apply_tone does, :861) or RT_TONE_REINHARD (reinhard_tone in its place;
 * key_pixel < 0: log-average key, else that pixel's luminance, :816-819).  ldMax / lMax: scene_t's
 * display and scene luminance maxima (scene.h:42-43).  Asynchronous.  The two frame-wide sums are
 * double tree sums, not the reference's float running sums (which lose ~3e-4 at 2 M pixels).
 * The calls share one per-device scratch block: calls that overlap on one GPU race. */
#define RT_TONE_WARD      1
#define RT_TONE_REINHARD  2
int  rt_hip_tone(float *d_radiance, uint32_t *d_argb, uint64_t npixels, int op, float ldMax, float lMax,
                 int64_t key_pixel, void *stream);
/* The same in two halves for frames that are spread over several GPUs: each rank sums its own pixels
 * into d_sums (3 doubles on the device: sum L, sum log(1e-5 + L), L of the key pixel if this rank
 * holds it, else 0), the ranks all-reduce (SUM) those 3 doubles - the path's only data exchange
 * besides the frame gather - and each applies the operator to its pixels with the frame's pixel count. */
int  rt_hip_tone_sums(const float *d_radiance, uint64_t npixels, float lMax, int64_t key_pixel,
                      double *d_sums, void *stream);
int  rt_hip_tone_apply(float *d_radiance, uint32_t *d_argb, uint64_t npixels, int op, float ldMax,
                       float lMax, const double *d_sums, uint64_t total_pixels, int use_key, void *stream);

/* rt_hip_tone that allocates nothing.  No counterpart in the reference: it is for callers that cannot
 * call hipMalloc - code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same arguments, same kernels and bit-identical results to rt_hip_tone; the partial sums live in a
 * static device array of the library (16 KiB + 32 B per GPU, held until the process ends) instead of a
 * block allocated on first use.  Limit: npixels <= RT_TONE_STATIC_MAX_PIXELS, larger frames are rejected
 * with RT_E_INVALID.  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two
 * streams, two threads) race, like rt_hip_tone's.  The array's address is resolved once per device, by
 * rt_hip_ctx_create or by the first call on that device (hipGetSymbolAddress; no allocation). */
#define RT_TONE_STATIC_MAX_PIXELS (1ull << 28)
int  rt_hip_tone_static(float *d_radiance, uint32_t *d_argb, uint64_t npixels, int op, float ldMax, float lMax,
                        int64_t key_pixel, void *stream);

/* Wait for the last rt_hip_render_rows, verify no queue overflowed (re-rendering finer
 * batches if one did) and return its counters.  stats may be NULL.
 * The pixels of an asynchronous render are FINAL only after this call has returned 0: a batch whose
 * queues overflowed is abandoned on the device (its rows keep whatever the buffers held) and is rendered
 * again here, in halves, so work ordered behind the render on a stream may read unfinished rows when
 * stats->overflow_retries comes back non-zero.  Contexts created with capacity 0 size their queues from
 * the frame and do not overflow on scenes whose ray trees stay within one queue entry per primary
 * sample and level (every BASELINE configuration). */
int  rt_hip_finish(rt_hip_ctx *ctx, rt_hip_stats *stats);

/* Same render with instrumented kernels that also count ray-object tests by outcome
 * (for the algorithmic-FLOP numerator of the roofline).  Blocking. */
int  rt_hip_render_stats(rt_hip_ctx *ctx, const rt_camera *camera,
                         uint32_t *d_argb, float *d_radiance,
                         uint32_t row0, uint32_t nrows, void *stream, rt_hip_stats *stats);

/* Which of the reference's two shading routines the context's renders use: RT_SHADE_PHONG =
 * get_shade_color_phong (raytrace.c:282-564, the one get_ray_color calls at :706; the default) or
 * RT_SHADE_PHONGBLINN = get_shade_color_phongblinn (raytrace.c:574-685, kept in the reference but called
 * by nothing): local illumination only - material colour without the checkerboard, shadow rays that are
 * not pushed off the surface, halfway-vector highlight, no reflection or transmission (depth is ignored).
 * rt_hip_render / raytrace() follow RT_HIP_SHADING=phong|phongblinn in the environment.
 * Returns 0 or RT_E_INVALID. */
#define RT_SHADE_PHONG       0
#define RT_SHADE_PHONGBLINN  1
int  rt_hip_ctx_set_shading(rt_hip_ctx *ctx, int model);

/* Numerics of the context's renders.  RT_NUMERICS_EXACT (the default): every float operation rounded like
 * the reference's x86-64 build, in its order - divides and square roots correctly rounded, the reference's
 * double-precision islands in double, no fused multiply-adds - so that pixels equal the reference's bit
 * for bit (but for glibc's powf rounding).  RT_NUMERICS_FAST: the tolerance the north star grants (8-bit
 * pixels within 1 LSB and radiance within 1e-4 relative on >= 99.9 % of the pixels): fused multiply-adds,
 * the hardware's 1-ulp reciprocal / reciprocal square root / square root, float sphere roots and
 * refraction term, v_exp/v_log for the highlight power.  Same pipeline, ~2x fewer vector instructions.
 * (Which products the compiler fuses differs between the kernels that shade a hit, so a context's first frame of
 * a shape - every level a launch - and its later ones - the sparse deep levels in one kernel - can differ in the
 * last bit of a few hundred pixels; the exact build's frames are bit-identical whatever kernels render them.)
 * Instrumented renders (rt_hip_render_stats) and RT_SHADE_PHONGBLINN always run the exact kernels.
 * rt_hip_render / raytrace() follow RT_HIP_NUMERICS=exact|fast in the environment.  No counterpart in
 * the reference (one build, one set of compiler flags: Makefile:5-20).  Returns 0 or RT_E_INVALID. */
#define RT_NUMERICS_EXACT 0
#define RT_NUMERICS_FAST  1
int  rt_hip_ctx_set_numerics(rt_hip_ctx *ctx, int mode);

/* Optional second stream (a hipStream_t; NULL = none, the default) for the tail of every batch this
 * context renders.  A batch of the reference's scene spends 0.29 ms in kernels that fill the GPU
 * (k_primary, the first two bounce levels) and 0.14 ms in the sparse deep levels, the folds and the
 * resolve, which are latency- and HBM-bound.  With a tail stream the second part is enqueued there
 * behind an event, so a host that renders frame after frame (each frame in flight on a context of its
 * own) gets the next frame's k_primary beside it.  The frame is then complete on the TAIL stream: order
 * work that reads the pixels behind it, or call rt_hip_finish.  No counterpart in the reference (its
 * frames are strictly sequential, main.c:159-165).  Pixels and counters are unchanged. */
int  rt_hip_ctx_set_tail_stream(rt_hip_ctx *ctx, void *stream);

/* Which kernels render big scenes (more than 128 spheres; the reference has one loop over all objects for every
 * ray, raytrace.c:188-272).  RT_PIPELINE_AUTO (the default): the job pipeline - rays as jobs of persistent lanes,
 * shadow rays through per-light direction grids, child rays over a sphere grid (csrc/rt_jobs.inc).
 * RT_PIPELINE_FUSED: the fused primary / bounce kernels that render small scenes, walking the sphere grid per lane
 * (2-3x slower there; what instrumented and phong-blinn renders always use).  Same pixels, bit for bit, in the
 * exact build.  No counterpart in the reference.  Returns 0 or RT_E_INVALID. */
#define RT_PIPELINE_AUTO  0
#define RT_PIPELINE_FUSED 1
int  rt_hip_ctx_set_pipeline(rt_hip_ctx *ctx, int mode);

/* Kernels compiled for the scene shape the reference's parse_scene always returns (2 spheres, 1 quad, 1
 * light: every count a compile-time constant) are used when the uploaded scene has that shape; enable = 0
 * makes the context use the generic kernels for every scene (same pixels; what any other scene gets).
 * No counterpart in the reference. */
void rt_hip_ctx_set_specialization(rt_hip_ctx *ctx, int enable);

/* Trace kernels compiled at run time for the uploaded scene's SHAPE.  The library holds the fused trace kernels for any
 * scene (object and light counts are run-time values) and for the one shape the reference's parse_scene always
 * returns (scene.c:86-258: every count a compile-time constant, a quarter faster).  For every other scene small
 * enough for the fused kernels' LDS tables (up to RT_TILE_SPHERES spheres, tables within 20 KB) the context can have
 * the same kernels compiled for that scene's counts (hiprtc, from the sources the library was built from, about a
 * second per numerics build) - same expressions, same pixels; the exact build stays bit-identical.
 *   RT_JIT_OFF      never
 *   RT_JIT_AUTO     (default) once the context has submitted RT_JIT_AFTER frames of the scene, on a thread of its
 *                   own: frames go on with the generic kernels and switch when the code is there; a failed compile
 *                   changes nothing
 *   RT_JIT_NOW      at the next render, which waits for the compile
 *   RT_JIT_NOW_ANY  like RT_JIT_NOW, and also for the reference's shape instead of the library's kernels for it
 *                   (measurement: what the run-time compile gives a scene the library has kernels for)
 * The environment variable RT_HIP_JIT=<mode number> overrides every context.  Instrumented renders, the phong-blinn
 * routine and big scenes take the library's kernels.  rt_hip_ctx_jit_status: 0 nothing asked for, 1 compiling,
 * 2 ready, -1 failed, for the context's scene and numerics; the compiler's messages and the compile's duration on
 * request.  No counterpart in the reference (its loops run over run-time counts, raytrace.c:188-272). */
#define RT_JIT_OFF 0
#define RT_JIT_AUTO 1
#define RT_JIT_NOW 2
#define RT_JIT_NOW_ANY 3
#define RT_JIT_AFTER 8
int rt_hip_ctx_set_jit(rt_hip_ctx *ctx, int mode);
int rt_hip_ctx_jit_status(rt_hip_ctx *ctx, char *log, size_t log_bytes, double *compile_seconds);

/* Launch plan of the fused primary kernel.  mode 0: the raster top-down, always.  mode 1 (the default): a render that
 * fits one image batch records how long every 8x8-pixel tile took; the next renders of the same frame shape on
 * this context launch the 32x8-pixel blocks slowest first, so that the kernel does not end on tiles that start
 * late and run long (reference scene, 1080p 16 spp: 0.124 -> 0.099 ms).  The plan is a permutation of the blocks -
 * pixels and ray counters do not depend on it - built once per view (camera, rows, scene) at the end of the
 * recording render (rt_hip_finish: one 4-byte-per-wave copy to the host, a counting sort, one copy back) and
 * refreshed every 8th render while the view keeps changing.  Big scenes (RT_PIPELINE_AUTO below) have no such
 * kernel and no plan.  No counterpart in the reference (its pixels are one serial loop, raytrace.c:746-792). */
void rt_hip_ctx_set_launch_plan(rt_hip_ctx *ctx, int mode);

/* HIP graph of a frame.  enable = 1: once a render's launches repeat - one image batch of the fused pipeline, on one
 * stream of the caller's (not the null stream; no tail stream), the view, buffers and level counts of the last frame - they are captured from
 * the stream once (hipStreamBeginCapture on the render's stream, thread-local mode) and every later frame is ONE
 * hipGraphLaunch on it instead of a dozen kernel launches; a changed camera, scene, buffer or level count captures
 * anew.  What a rank of a frame split over eight GPUs needs: there the host's 50 us of launches per frame, not the
 * GPU, set the frame rate.  Same kernels, same pixels; stats.graph_launches / graph_captures count.  With
 * rt_hip_ctx_set_timing on, every 8th frame is launched the plain way, between its events (k0_ms, k0_launches).
 * Off by default.  No counterpart in the reference. */
void rt_hip_ctx_set_graph(rt_hip_ctx *ctx, int enable);

/* The host half of the launch plan, exported for tests (no GPU involved): the order rt_hip_finish builds from
 * wave_ticks[(tile row * block_columns + block column) * 4 + wave], the time each of a block's four waves took -
 * blocks by their slowest wave, slowest first, equal times in raster order.  plan[i] = block column | tile row << 15,
 * block_columns * tile_rows entries (both below 32768).  No counterpart in the reference. */
void rt_launch_plan_order(const uint32_t *wave_ticks, uint32_t block_columns, uint32_t tile_rows, uint32_t *plan);

/* Time the fused primary kernel (k0_ms, k0_launches).  enable = 1: two HIP events recorded around its launch on the
 * render stream (they bracket the dispatch packet: the kernel plus its queue latency); 2: events that take the kernel's
 * own start and end (hipExtLaunchKernel) - what a profiler reports, but ~25 us of host time a launch; 0: off. */
void rt_hip_ctx_set_timing(rt_hip_ctx *ctx, int enable);
/* Primary samples per image batch, as given (0 = automatic: what the queues hold).  A batch that outgrows the
 * queues is abandoned, halved and rendered again by rt_hip_finish. */
void rt_hip_ctx_set_batch_samples(rt_hip_ctx *ctx, uint64_t max_samples);

const char *rt_hip_last_error(void);
const char *rt_hip_version(void);

/* Diagnostic (no counterpart in the reference): runs the kernels' correctly rounded divide /
 * square-root helpers (csrc/rt_arith.h) over n operand triples so a caller can compare them with
 * IEEE arithmetic on the host - the reference's `/`, sqrt() and vec4_normalize (vector4.c:219-262).
 * All pointers are DEVICE pointers.  d_out holds 7 floats per element i:
 *   sqrt(a), a/b with special-operand fix-up, a/b without, normalize(a,b,c).xyz, 1.0f/b
 * d_out64[i] = sqrt((double)a) by the positive-normal-only helper.  Blocking. */
int  rt_hip_selftest_arith(const float *d_a, const float *d_b, const float *d_c, uint32_t n,
                           float *d_out, double *d_out64, void *stream);

/* Diagnostic (no counterpart in the reference): for n rays (DEVICE array, 7 floats each: origin,
 * direction, magnitude) the sphere search of the uploaded big scene twice - through the grid the
 * kernels walk, and as the reference's loop over every sphere (raytrace.c:188-272; filtered != 0:
 * the _excl variant, 0 < d < magnitude).  d_out (DEVICE, 4 uint32 per ray) = object and distance bits
 * from the grid, object and distance bits from the loop; they must be equal.  Blocking. */
int  rt_hip_selftest_grid(rt_hip_ctx *ctx, const float *d_rays, uint32_t n, int filtered, uint32_t *d_out);

/* ---- host-side scene helpers (reference scene.c / geometry.c equivalents) ---------- */

/* poly_plane -- geometry.c:14-34 (normal from the first three vertices, F forced >= 0). */
void rt_poly_plane(rt_plane *planeout, const rt_polygon *poly);

/* The hard-coded Whitted scene that parse_scene always produces -- scene.c:86-258.
 * Allocates lights/objects/vertices with malloc; release with rt_scene_free (scene.c:284). */
int  rt_scene_builtin(rt_scene_t *scene);
/* parse_scene -- scene.c:24-281: opens `filename` (error if unreadable, like scene.c:31-35),
 * ignores its contents exactly as the reference does, and fills in the builtin scene.
 * Returns 1 on success, 0 on failure (reference convention). */
int  rt_parse_scene(const char *filename, rt_scene_t *scene);
void rt_scene_free(rt_scene_t *scene);
/* A loader that really reads the grammar sketched in the reference's `scene` file (scene:1-47;
 * the reference's parse_scene, scene.c:36-84, tokenises it and throws the tokens away):
 * desc / camera / world / pointlight / sphere / polygon blocks, '#' comments, material keys
 * diffuse specular ka kd ks ke kr kt n (grammar in csrc/rt_host.c).  Camera basis per
 * scene.c:236-258.  Returns 1 on success, 0 on failure with a message on stdout (the reference's
 * convention, scene.c:31-35); release with rt_scene_free. */
int  rt_load_scene(const char *filename, rt_scene_t *scene);

/* Seeded synthetic scenes of BASELINE configs 4/5: nSpheres random
 * spheres + nPolys (0..4) axis-aligned quads + nLights lights; SplitMix64 generator.
 * kind 0: config-4 distribution, kind 1: config-5 distribution. */
int  rt_scene_synthetic(rt_scene_t *scene, uint64_t seed, uint32_t nSpheres,
                        uint32_t nPolys, uint32_t nLights, int kind);

/* write_image -- output.h:15-16 / pam_output/output.c:30-71: raw P6 "P6\n<W> <H>\n255\n"
 * + RGB bytes, row 0 first.  Returns 1 on success, 0 on failure. */
int  rt_write_image(const char *filename, const unsigned int *buffer,
                    unsigned int width, unsigned int height);
/* The same writer under the reference's own name and prototype (output.h:15-16), so that the
 * reference's main.c links against this library alone when libnetpbm / GLUT are not installed.
 * A host that links its own output.o keeps its own definition (executable symbols win). */
int  write_image(const char *filename, unsigned int *buffer,
                 unsigned int width, unsigned int height);

#ifdef __cplusplus
} /* extern "C" */
#endif

/* ---- layout pins (measured on the compiled reference) ------- */
#if defined(__cplusplus)
#define RT_SA(c, m) static_assert(c, m)
#else
#define RT_SA(c, m) _Static_assert(c, m)
#endif
RT_SA(sizeof(rt_vec4) == 16, "vector4_t");
RT_SA(sizeof(rt_color) == 12, "color_t");
RT_SA(sizeof(rt_plane) == 16, "plane_t");
RT_SA(sizeof(rt_sphere) == 20, "sphere_t");
RT_SA(sizeof(rt_polygon) == 32 && offsetof(rt_polygon, nVerticies) == 16 &&
      offsetof(rt_polygon, vertex) == 24, "polygon_t");
RT_SA(sizeof(rt_material) == 52, "material_t");
RT_SA(sizeof(rt_pointlight) == 32 && offsetof(rt_pointlight, color) == 16 &&
      offsetof(rt_pointlight, range) == 28, "pointlight_t");
RT_SA(sizeof(rt_object3d) == 104 && offsetof(rt_object3d, debugName) == 8 &&
      offsetof(rt_object3d, material) == 16 && offsetof(rt_object3d, sphr_obj) == 72 &&
      offsetof(rt_object3d, poly_obj) == 72, "object3d_t");
RT_SA(sizeof(rt_scene_t) == 184 && offsetof(rt_scene_t, U) == 48 &&
      offsetof(rt_scene_t, frameBufferWidth) == 96 && offsetof(rt_scene_t, sppWidth) == 120 &&
      offsetof(rt_scene_t, bgColor) == 124 && offsetof(rt_scene_t, ambientLightColor) == 136 &&
      offsetof(rt_scene_t, ldMax) == 148 && offsetof(rt_scene_t, nLights) == 156 &&
      offsetof(rt_scene_t, nObjects) == 160 && offsetof(rt_scene_t, lights) == 168 &&
      offsetof(rt_scene_t, objects) == 176, "scene_t");
#undef RT_SA

#endif /* RT_HIP_H */
