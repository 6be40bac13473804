/*
 * rt_device.h - device-side data layout shared by rt_kernels.hip and rt_api.hip.
 *
 * Everything the kernels read lives in HBM as SoA / float4 streams (no pointers to host
 * structs).  See DESIGN.md "Data layout in HBM".
 */
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else /* hiprtc (rt_jit.cpp): its built-in declarations, and the fixed-width names */
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned long uintptr_t;
#endif

#define RT_MAX_LEVELS   64          /* supported recursion depth: 0 .. RT_MAX_LEVELS-2 */
#define RT_NONE         0xFFFFFFFFu
#define RT_LINK_NONE    0x1FFFFFFFu /* 29-bit "no parent / end of chain" */
#define RT_TILE_SPHERES 128         /* most spheres searched by a loop over an LDS copy; scenes with more walk a grid in HBM */
#define RT_MAX_POLY_LDS 1024        /* float4 slots for planes + info + vertices in LDS */
#define RT_BLOCK        256
/* per-thread parking area at the start of every trace kernel's LDS (row-major [word][thread],
 * conflict-free; rows listed in shade_and_spawn).  Values parked here do not occupy VGPRs across
 * the intersection searches. */
#define RT_PARK_WORDS   20u
#define RT_PARK_FLOATS  (RT_PARK_WORDS * RT_BLOCK)
#define RT_DEEP_SUB     4u          /* k_fold_deep: RT_BLOCK * RT_DEEP_SUB threads per block */

/* Camera constants, pre-folded on the host with the reference's float rounding
 * (raytrace.c:737,751-768): everything that does not depend on the pixel. */
struct DevCamera {
    float eye[3];
    float U[3], V[3];
    float shiftN[3];     /* N * viewDistance                 raytrace.c:751 */
    float halfW, halfH;  /* viewPlaneHalfWidth/Height                       */
    float hfbW, hfbH;    /* frameBufferWidth/2.0f, frameBufferHeight/2.0f   */
    float sppW, sppHalf; /* sppWidth, sppWidth/2.0f                         */
    float scale;         /* 1.0f/(float)(sqrtSpp*sqrtSpp)    raytrace.c:737 */
    uint32_t W, H, sqrtSpp, depth;
};

/* Flattened scene (all device pointers). */
struct DevScene {
    const float4 *sph;        /* [nSph]  (cx, cy, cz, r*r)                           */
    const uint32_t *sph_obj;  /* [nSph]  object index of sphere k (ascending)        */
    const float4 *poly;       /* [nPolySlots] planes, then info, then vertex pool:
                                 poly[p]            = (A,B,C,F)
                                 poly[nPoly+p]      = bits(vtx offset, nv, obj, 0)
                                 poly[2*nPoly+v]    = vertex v (x,y,z,w)              */
    const float4 *tab;        /* [tabSlots] shading tables, records of whole float4s:
                                 tab[2*l + 0]              = light l position
                                 tab[2*l + 1]              = light l colour
                                 tab[2*nLights + 5*o + k]  = object o, RT_OBJ_* field k
                                 (one 80-byte record per object: a lane that shades object o
                                 touches one or two cache lines, not five arrays)            */
    const uint32_t *grid;     /* scenes with more than RT_TILE_SPHERES spheres: GridHeader, then
                                 cellStart[cells + 1], then the cells' sphere indices (ascending
                                 within a cell); nullptr otherwise                              */
    const uint32_t *lgrid;    /* with `grid`: the lights' direction grids (LightGridHeader ...), or nullptr */
    uint32_t nSph, nPoly, nPolySlots, nObj, nLights;
    uint32_t tabSlots;        /* 2*nLights + 5*nObj                                          */
    uint32_t tabInLds;        /* the tables are small enough to be staged in LDS per block   */
    uint32_t noShape;         /* use the generic kernels whatever the scene's shape (rt_hip_ctx_set_specialization) */
    float bg[3];
    float ambient[3];
};

/* Uniform grid over the spheres of a big scene (built by rt_hip_scene_upload).  A sphere is listed in
 * every cell its GROWN bounding box overlaps.  The growth covers what the reference's sphere test
 * (ray.c:185-209: B*B - 4*C without the quadratic's leading coefficient) accepts beyond geometry: with
 * a direction of squared length 1 +- kappa and float rounding, a sphere at distance <= reach whose
 * centre is up to sqrt(r*r + 2*kappa*reach*reach) from the ray's line can still come out as a hit, and
 * the line's point at the reported distance lies that close to the centre (rt_api.hip, build_grid).
 * Rays outside those assumptions (direction not of unit length: the reference's refraction formula
 * produces such rays; origin farther than `reach` allows) loop over all spheres instead. */
#define RT_GRID_KAPPA 3.0e-6f /* bound of | |dir|^2 - 1 | + float rounding of the discriminant, per unit reach^2 */
#define RT_GRID_UNIT  1.0e-6f /* a ray walks the grid only if | |dir|^2 - 1 | <= this */
struct GridHeader {
    float gmin[3];   /* lower corner                     */
    float gmax[3];   /* upper corner                     */
    float h[3];      /* cell size per axis               */
    float inv_h[3];
    uint32_t dim[3]; /* cells per axis                   */
    uint32_t words;  /* size of this header in uint32_t  */
    float centre[3]; /* of the box                       */
    float origin_reach2; /* a ray walks the grid only if its origin is within sqrt(this) of the centre */
    uint32_t rec_words;  /* where the cell chunks start: words from the grid's base, a multiple of 16; 0: none (too many spheres) */
    uint32_t pad_[3];    /* [0], [1]: where the sphere GROUPS of k_loop start (words, a multiple of 4) and their number (build_grid); [2]: 0 */
};
/* The same cells once more as 64-byte CHUNKS of up to three spheres, for the walks of the job pipeline (rt_jobs.inc):
 * a lane gets a chunk with ONE round trip instead of three dependent ones (cellStart -> items -> sph), every walking
 * lane of a wave does the same thing per iteration - one chunk, three tests - and the address of the next chunk is
 * known before the tests begin, so it is fetched beside them.
 * History: three inline spheres in a 64-byte record and the fuller cells' lists one sphere per iteration - four
 * fifths of the walk went into those lists (the cells list the spheres GROWN by what the reference's test accepts
 * beyond geometry at the grid's reach: config 5's 0.15-radius spheres count as 0.35, four to a cell); six spheres
 * in a 128-byte record - every iteration ran six test blocks for the one lane in a fuller cell (config 4: 2.2
 * spheres per occupied cell, 13 % of them hold more than three), and a walking lane's eight 16-byte loads each cost
 * the CU's vector memory pipe a cache-line look-up per lane.
 *   chunk c < cells: the first chunk of cell c: count (low 24 bits) = the cell's spheres, the first three inline; if
 *   count > 3 the others follow from chunk `next` on, three to a chunk, one behind the other (count and next 0 there).
 *   Per sphere: its OBJECT index, 21 bits (RT_CHUNK_MAX_SPHERES; object indices ascend with sphere indices, so "the
 *   lower index wins a tie" reads the same on either), i0 | i1 << 21 in idx_lo, i1 >> 11 | i2 << 10 in idx_hi, and in
 *   bits 24 + 2u, 25 + 2u of `count` whether a hit on sphere u spawns a reflection / a transmission ray (material
 *   kr != 0, kt != 0: raytrace.c:405,452) - a hit is queued as object | flags << 30 (RT_HIT_*), so that the next
 *   level knows which rays a hit asks for without looking its material up. */
#define RT_CHUNK_SPHERES 3u
#define RT_CHUNK_MAX_SPHERES (1u << 21)
#define RT_HIT_OBJ 0x3FFFFFFFu  /* a queued hit of the job pipeline: the object ... */
#define RT_HIT_REFL 0x40000000u /* ... spawns a reflection ray */
#define RT_HIT_TRANS 0x80000000u /* ... spawns a transmission ray */
struct CellChunk {
    uint32_t count;
    uint32_t idx_lo, idx_hi;
    uint32_t next;
    float4 sph[RT_CHUNK_SPHERES]; /* (centre, r*r): DevScene::sph's words */
};
static_assert(sizeof(CellChunk) == 64, "a chunk is half a cache line");

/*
 * Shadow rays end at a light: every sphere such a ray can meet lies in a narrow cone around the ray as seen FROM
 * the light.  Per light, the directions from it are cut into the cells of a cube map (6 faces of res x res
 * cells), and a cell lists every sphere whose grown ball (the growth of GridHeader, plus the distance by which a
 * shadow ray's line can miss its light after float rounding) is touched by some line through the light with a
 * direction in the cell; spheres that hold the light (nearly) are in the light's "global" list.  A shadow ray's
 * any-hit search is then ONE cell lookup and a scan of that cell's list and the global list - no walk.  Built by
 * rt_hip_scene_upload (rt_api.hip, build_light_grids); the same exclusions as the grid: direction of unit length,
 * origin within the grid's reach.
 *   words: LightGridHeader | cellStart[nLights * 6 * res * res + 1] | gstart[nLights + 1] | chunk[nitems], 64 bytes
 *          each (LightChunk): up to three spheres of a list - (centre, r*r), the sphere's index (21 bits, packed as
 *          in CellChunk) and its key as the upper half of a float, cut off (+inf: no sphere); key = the distance from
 *          the light at which the grown ball begins, a cell's list sorted by it (a ray that starts nearer the light
 *          than that cannot meet the sphere: the scan of the cell's list ends there; cutting a key off only makes it
 *          smaller).  cellStart and gstart count chunks.  Cell lists first, the global lists (key 0) behind them.
 * cell of direction u (from the light): m = axis of the largest |u|, face = 2 m + (u[m] < 0),
 *   (a, b) = the other two components in axis order / |u[m]|, (ia, ib) = floor((a + 1) / 2 * res) clamped,
 *   cell = (face * res + ib) * res + ia.
 */
struct LightGridHeader {
    uint32_t res, ncell;     /* ncell = 6 * res * res, cells per light */
    uint32_t off_cells, off_gstart, off_idx, off_sph; /* word offsets from the blob's base (off_sph a multiple of 4) */
    uint32_t nitems, pad_; /* nitems: chunks */
};
struct LightChunk {
    uint32_t idx_lo, idx_hi; /* i0 | i1 << 21, i1 >> 11 | i2 << 10 */
    uint32_t keys01, key2;   /* key0 | key1 << 16, key2: upper halves of floats */
    float4 sph[RT_CHUNK_SPHERES];
};
static_assert(sizeof(LightChunk) == 64, "a chunk is half a cache line");

/* fields of an object record in DevScene::tab */
enum {
    RT_OBJ_DIFFUSE = 0, /* (diffuse rgb, ka)                                   */
    RT_OBJ_SPECULAR,    /* (specular rgb, ks)                                  */
    RT_OBJ_PHONG,       /* (kd, ke, kr, kt)                                    */
    RT_OBJ_GEOM,        /* (n, bits type, bits geometry-list index, 0)         */
    RT_OBJ_SHAPE,       /* sphere centre xyz | polygon plane normal ABC        */
    RT_OBJ_FIELDS
};
#define RT_LDS_TABLE_BUDGET 20480u /* bytes of LDS per block up to which the tables are staged too
                                      (8 blocks per CU still fit the 160 KiB) */

/* Rows of one image batch, as 8-row tile rows (a wave owns an 8x8 pixel tile).
 * Tile row j of the batch covers frame rows y_first + j*y_step + [0,8), clipped to y_limit, and
 * is written to output rows out_first + 8*j + [0,8).
 *   contiguous slab [row0,row0+n): y_first = out_first = row0 + 8*t0, y_step = 8, y_limit = row0+n
 *   cyclic 8-row blocks (phase r of N ranks, multi-GPU): y_first = (t0*N + r)*8, y_step = 8*N,
 *   y_limit = H, out_first = 8*t0 (rows of the rank's compact local buffer)               */
struct DevRows {
    uint32_t y_first, y_step, y_limit, out_first, ntiles;
    /* k_primary's launch order (rt_api.hip, "launch plan"): block b of a 1-D grid of plan_len blocks renders block
     * column plan[b] & 0x7fff of tile row (plan[b] >> 15) & 0x7fff.  Every block of the raster is listed once - so
     * any plan renders the same frame.  nullptr: the raster itself, top-down. */
    const uint32_t *plan = nullptr;
    uint32_t plan_len = 0;
    /* where every wave of k_primary leaves the time it took, in 10 ns ticks of the wall clock:
     * wave_ticks[(tile row * block columns + block column) * 4 + wave]; nullptr: not recorded */
    uint32_t *wave_ticks = nullptr;
    /* Pixels of tiles none of whose samples became a root (sky, plain floor: 85 % of the tiles of the reference's
     * scene) are finished by k_primary itself - scaled, packed and stored to the frame (argb, radiance or nullptr),
     * tile_done[(tile row * block columns + block column) * 4 + wave] = 1 - and k_resolve skips them: their 16
     * bytes of pixel sums are neither written nor read back.  Every wave of k_primary writes its flag. */
    uint32_t *argb = nullptr;
    float *radiance = nullptr;
    uint8_t *tile_done = nullptr;
};

/* Per-batch work state in HBM. */
struct DevWork {
    /* Ten arrays of `cap` entries in one allocation, `stride` bytes apart, addressed from ONE base
     * pointer: thirteen pointers as kernel arguments kept ~26 scalar registers live through the
     * trace kernels, which spilled them into VGPR lanes (300 v_readlane / v_writelane in k_primary).
     *   slot 3p+0  qA[p]  hit queue of level parity p: (origin of the ray that found the hit, distance along it)
     *   slot 3p+1  qB[p]  (direction of that ray, bits parent<<1|slot)
     *   slot 3p+2  qObj[p] object hit, 4 bytes an entry (the rest of the slot is unused): 36 bytes a hit - the
     *                     consumer forms the hit point as the producer would (hit_point, rt_kernels.hip); with
     *                     the point stored too an entry was 48 bytes, 28 MB more written and read per frame
     *   slot 6     nodeA  (local rgb, kr)                 written when the node is made
     *   slot 7     childR (reflection child rgb, -)       written by the child, as a whole float4:
     *   slot 8     childT (transmission child rgb, -)     consecutive rays have consecutive parents,
     *                     so a wave's results are full-line stores (12-byte pieces of a shared float4
     *                     made every store a partial-sector read-modify-write)
     *   slot 9     nodeK  (kt, bits link), float2         written when the node is made           */
    char *base;
    uint64_t stride;
    float4 *pixAcc;       /* per pixel of the batch: (immediate sum rgb, bits #roots)  */
    uint32_t *roots;      /* [spp^2][pixels of the batch] k-th root node of a pixel: the roots of
                             a pixel are independent loads for k_resolve (a linked chain through
                             the nodes cost one dependent HBM round trip per sample)   */
    uint32_t *counters;   /* counter block of this batch (layout below)                */
    float4 *loopq;        /* the job pipeline's loop queues (rt_jobs.inc, k_loop): RT_SHARDS x loop_capS entries of 4 float4 */
    uint32_t loop_capS;
    uint32_t cap;         /* entries per queue / node pool                             */
    uint32_t capS;        /* entries per shard = cap / RT_SHARDS                       */

    __host__ __device__ __forceinline__ float4 *slot(uint32_t k) const { return (float4 *)(base + (uint64_t)k * stride); }
    __host__ __device__ __forceinline__ float4 *qA(uint32_t parity) const { return slot(3u * parity + 0u); }
    __host__ __device__ __forceinline__ float4 *qB(uint32_t parity) const { return slot(3u * parity + 1u); }
    __host__ __device__ __forceinline__ uint32_t *qObj(uint32_t parity) const { return (uint32_t *)slot(3u * parity + 2u); }
    __host__ __device__ __forceinline__ float4 *nodeA() const { return slot(6u); }
    __host__ __device__ __forceinline__ float4 *childR() const { return slot(7u); }
    __host__ __device__ __forceinline__ float4 *childT() const { return slot(8u); }
    __host__ __device__ __forceinline__ float2 *nodeK() const { return (float2 *)slot(9u); }
};

/* Per-sample arrays of a batch of a BIG scene (the job pipeline, rt_jobs.inc): sample = k * npix + pixel of the
 * batch, k = j * sqrtSpp + i the sub-sample in the reference's order (raytrace.c:746-776). */
struct DevLevel0 {
    uint2 *hit;        /* [nsamples] (object hit by the primary ray or RT_NONE, bits of the distance along it) */
    uint8_t *vis;      /* [nLights][nsamples] 1: the shadow ray to that light met nothing                        */
    float4 *childR;    /* [nsamples] colour the reflection ray brought back (the deeper levels use DevWork's)    */
    float4 *childT;    /* [nsamples] ... the transmission ray                                                    */
    uint32_t nsamples; /* sqrtSpp^2 * npix                                                                       */
    uint32_t npix;     /* pixels of the batch: its tile rows x 8 x W                                             */
};
#define RT_JOBS_MAX_LIGHTS 32u /* visibility bytes of the deeper levels: two slots of 16 bytes an entry */

/*
 * Counter block of a batch (uint32 words).  Zero when a batch starts: k_publish, the last kernel of every
 * batch, copies the block to the host's mapped mirror and zeroes it for its next use (the host clears
 * the blocks once at context creation, and again only after a render died midway).
 *
 * Queues and the node pool are split into RT_SHARDS shards, each with its own counters on its
 * own 128-byte line: one L2 word sustains only ~90 returning atomics per microsecond on MI355X,
 * and a 1080p frame issues ~10^5 wave-level allocations, so a single tail counter serialises
 * the whole frame (measured: 1.6 ms -> see DESIGN.md).  A block produces into / consumes from
 * shard (block index % RT_SHARDS).
 *
 *   word CNT_OVERFLOW (0)       : ONE overflow flag for the whole batch, raised by any kernel whose
 *                                 reservation did not fit; every later kernel of the batch returns at
 *                                 once when it is set (slots reserved by an overflowing wave stay
 *                                 unwritten, so nothing downstream may consume the queues) and the host
 *                                 renders the batch again in two halves.  Words 1 .. RT_MAX_LEVELS-1 are spare.
 *   CNT_STATS + s*64            : uint64 statistics of shard s (ST_* slots)
 *   CNT_LEVELS + (L*RT_SHARDS+s)*RT_LINE_WORDS : one uint64 for (level L, shard s):
 *                                 low 32 bits  = tree nodes made at level L
 *                                 high 32 bits = rays queued for level L+1
 *                                 (both are produced by the kernel that processes level L, so a
 *                                 wave reserves nodes and ray slots with ONE 64-bit atomic)
 */
#define RT_SHARDS      64
#define RT_LINE_WORDS  32
enum {
    CNT_OVERFLOW = 0,              /* word 0: the batch overflowed somewhere (the other RT_MAX_LEVELS - 1 words are spare) */
    CNT_STATS = RT_MAX_LEVELS,
    CNT_STATS_STRIDE = 64,
    CNT_LEVELS = CNT_STATS + RT_SHARDS * CNT_STATS_STRIDE,
};
#define RT_COUNTER_WORDS_FOR(levels) (CNT_LEVELS + (levels) * RT_SHARDS * RT_LINE_WORDS)
#define RT_COUNTER_WORDS RT_COUNTER_WORDS_FOR(RT_MAX_LEVELS)
static_assert(CNT_LEVELS % 4 == 0 && (RT_SHARDS * RT_LINE_WORDS) % 4 == 0, "counter blocks are copied 16 bytes at a time");

/* uint64 statistic slots of a shard */
enum {
    ST_K0_SHADE_NODES = 0, ST_K0_TRANS, ST_K0_REFL,
    ST_KB_SHADE_NODES, ST_KB_TRANS, ST_KB_REFL,
    /* instrumented build only */
    ST_SPH_TESTS, ST_SPH_HITS, ST_POLY_TESTS, ST_POLY_PAR, ST_POLY_BEHIND, ST_POLY_FULL, ST_POLY_HITS,
    ST_LIT, ST_NODES_SPH, ST_NODES_POLY,
    ST_K0_SPH_TESTS, ST_K0_SPH_HITS, ST_K0_POLY_TESTS, ST_K0_POLY_PAR, ST_K0_POLY_BEHIND,
    ST_K0_POLY_FULL, ST_K0_LIT, ST_K0_NODES_SPH, ST_K0_NODES_POLY,
    ST_POLY_SCREEN_MISMATCH, ST_POLY_SCREEN_UNSURE,
    ST_K0_CULLED, /* primary rays of tiles finished without tracing */
    ST_COUNT
};
static_assert(2 * ST_COUNT <= CNT_STATS_STRIDE, "statistics do not fit their line");

#define RT_SUBTREE_LEVELS 8 /* k_subtree: at most this many levels below the one it takes its hits from */
#ifndef __HIPCC_RTC__
/* launch wrappers implemented in rt_kernels.hip (all asynchronous on `stream`).  `jit`: kernels compiled at run time for
 * the scene's shape (rt_jit.cpp), or nullptr: the library's own */
struct RtJitKernels;
void rt_launch_fold_deep(const DevWork &Wk, uint32_t top_level, uint32_t bottom_level, hipStream_t stream);
/* ev0 / ev1 (both or neither): events that take the kernel's own start and end (hipExtLaunchKernel) */
void rt_launch_primary(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                       bool stats, bool blinn, hipStream_t stream, const RtJitKernels *jit = nullptr,
                       hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
/* the same two from the tolerance build of the kernels (rt_kernels.hip compiled with -DRT_FAST) */
void rt_launch_primary_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                            hipStream_t stream, const RtJitKernels *jit = nullptr, hipEvent_t ev0 = nullptr,
                            hipEvent_t ev1 = nullptr);
void rt_launch_bounce_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, uint32_t level,
                           uint32_t expect, hipStream_t stream, const RtJitKernels *jit = nullptr);
/* `expect` = entries per shard the launch is expected to find (0 = unknown: a full resident set).
 * It only sizes the grid; the kernels stride over whatever the counters say. */
void rt_launch_bounce(const DevScene &S, const DevCamera &C, const DevWork &Wk, uint32_t level,
                      bool stats, uint32_t expect, hipStream_t stream, const RtJitKernels *jit = nullptr);
/* the hits queued for `level` and everything below them, depth first, in one launch (k_subtree): at most
 * RT_SUBTREE_LEVELS levels, i.e. depth - level < RT_SUBTREE_LEVELS */
void rt_launch_subtree(const DevScene &S, const DevCamera &C, const DevWork &Wk, uint32_t level,
                       uint32_t expect, hipStream_t stream, const RtJitKernels *jit = nullptr);
void rt_launch_subtree_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, uint32_t level,
                            uint32_t expect, hipStream_t stream, const RtJitKernels *jit = nullptr);
void rt_launch_fold(const DevWork &Wk, uint32_t level, uint32_t expect, hipStream_t stream);
void rt_launch_resolve(const DevCamera &C, const DevWork &Wk, const DevRows &R, uint32_t *d_argb,
                       float *d_radiance, uint32_t *d_argb_host, hipStream_t stream);
size_t rt_scene_lds_bytes(const DevScene &S);
/* the job pipeline of big scenes (rt_jobs.inc), exact build and tolerance build (`_fast`) */
void rt_launch_jobs_hits(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                         const DevLevel0 &Z, hipStream_t stream);
void rt_launch_jobs_trace(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                          const DevLevel0 &Z, uint32_t level, uint32_t expect, bool spawns, hipStream_t stream);
void rt_launch_jobs_shadefold(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevLevel0 &Z,
                              uint32_t level, uint32_t expect, hipStream_t stream);
void rt_launch_jobs_resolve(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                            const DevLevel0 &Z, hipStream_t stream);
void rt_launch_jobs_hits_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                              const DevLevel0 &Z, hipStream_t stream);
void rt_launch_jobs_trace_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                               const DevLevel0 &Z, uint32_t level, uint32_t expect, bool spawns, hipStream_t stream);
void rt_launch_jobs_shadefold_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevLevel0 &Z,
                                   uint32_t level, uint32_t expect, hipStream_t stream);
void rt_launch_jobs_resolve_fast(const DevScene &S, const DevCamera &C, const DevWork &Wk, const DevRows &R,
                                 const DevLevel0 &Z, hipStream_t stream);
void rt_launch_selftest_grid(const DevScene &S, const float *rays, uint32_t n, bool filtered, uint32_t *out,
                             hipStream_t stream);
int rt_scene_shape(const DevScene &S); /* 0 = generic kernels, else the specialisation (rt_kernels.hip) */
void rt_launch_publish(uint32_t *counters, uint32_t *mirror, uint32_t words, hipStream_t stream);
void rt_launch_deinterleave(const uint32_t *gathered, uint32_t *frame, uint32_t width, uint32_t height,
                            uint32_t nphases, hipStream_t stream);
void rt_launch_selftest_arith(const float *a, const float *b, const float *c, uint32_t n, float *out,
                              double *out64, hipStream_t stream);
#endif /* !__HIPCC_RTC__ */
