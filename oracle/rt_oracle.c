/*
 * rt_oracle.c - CPU restatement of the reference's trace-and-shade loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: a scalar, recursive, plain-C
 * restatement of davidoguns/raytracer's x86 algorithm (raytrace.c, ray.c, vector4.c,
 * color.c, geometry.c, object3d.c), written from the behavioural spec in SURVEY.md
 * appendix A.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product path (librt_hip.so) never links, imports or calls anything here.
 *
 * Parity status: PINNED.  oracle/Makefile builds the untouched reference sources into
 * oracle/_ref/libref_raytracer.so; tests/golden/make_golden.py runs both and the committed
 * fixtures under tests/golden/ hold the reference's outputs (image hashes, float radiance,
 * ray/test counts, intersection KATs).  tests/test_oracle_golden.py requires this file to
 * reproduce them bit-for-bit.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (see oracle/Makefile).  -ffp-contract=off keeps
 * every float operation individually rounded, as in the reference's x86-64 build.
 * Each function cites the reference file:line it follows.
 */
#include "../include/rt_hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

/* counters (per render; merged over threads) */
typedef struct {
    uint64_t primary_rays, shadow_rays, reflection_rays, transmission_rays;
    uint64_t shade_nodes, nodes_sphere, nodes_polygon, lights_lit;
    uint64_t sphere_tests, sphere_hits;
    uint64_t polygon_tests, polygon_rej_parallel, polygon_rej_behind, polygon_full, polygon_hits;
    uint64_t nodes_at_depth[64];
} rt_oracle_counts;

typedef struct {
    v3 origin, direction;
    float magnitude;
} ray_t;

typedef struct {
    const rt_scene *scene;
    const rt_camera *cam;
    rt_oracle_counts *cnt;
} ctx_t;

/* DIAGNOSTIC switch (off by default: the oracle then calls libm's powf, like the reference,
 * raytrace.c:371).  glibc 2.35's powf is faithfully but not correctly rounded: for ~0.07 % of
 * arguments it returns the float on the other side of the exact power.  With the switch on, the
 * specular power is the correctly rounded one (exact product chain in double for an integral
 * exponent), which is what the HIP kernels compute; tests use it to show that libm's rounding is
 * the ONLY difference left between the kernels and the reference. */
static int g_rounded_pow;
void rt_oracle_set_rounded_pow(int on) { g_rounded_pow = on; }
static float rounded_pow(float s, float ke)
{
    const int e = (int)ke;
    if ((float)e != ke || e < 0 || e > 4096)
        return powf(s, ke);
    double r = 1.0, b = (double)s;
    for (int k = e; k; k >>= 1) {
        if (k & 1)
            r *= b;
        b *= b;
    }
    return (float)r;
}

/* ---- vector4.c ------------------------------------------------------------------------ */

static v3 v3_from(const rt_vec4 *p) { v3 r = {p->x, p->y, p->z}; return r; }
/* vec4_add  vector4.c:60-75 */
static v3 add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
/* vec4_sub  vector4.c:78-93 */
static v3 sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
/* vec4_scale  vector4.c:96-118 */
static v3 scale(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
/* vec4_dot  vector4.c:121-144: x,y,z only, left to right */
static float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vec4_magnitude  vector4.c:219-233: float sum, double sqrt, float result */
static float magnitude(v3 a) { return (float)sqrt((double)(a.x * a.x + a.y * a.y + a.z * a.z)); }
/* vec4_normalize  vector4.c:268-278: divides by the magnitude */
static v3 normalize(v3 a)
{
    float m = magnitude(a);
    v3 r = {a.x / m, a.y / m, a.z / m};
    return r;
}
/* vec4_costheta  vector4.c:281-285 */
static float costheta(v3 a, v3 b) { return dot(a, b) / (magnitude(a) * magnitude(b)); }

/* ---- ray.c ---------------------------------------------------------------------------- */

/* ray_create  ray.c:29-41: direction scaled by the float reciprocal of the magnitude */
static ray_t ray_create(v3 p1, v3 p2)
{
    ray_t r;
    r.origin = p1;
    r.direction = sub(p2, p1);
    r.magnitude = magnitude(r.direction);
    r.direction = scale(r.direction, 1.0f / r.magnitude);
    return r;
}

/* ray_tinypush  ray.c:72-86: origin += normalize(direction) * 0.001; rest unchanged */
static void ray_tinypush(ray_t *r)
{
    v3 v = normalize(r->direction);
    v = scale(v, 0.001f);
    r->origin = add(r->origin, v);
}

/* ray_intersect_sphere  ray.c:181-249 */
static int intersect_sphere(const ctx_t *c, const ray_t *ray, const rt_sphere *s, v3 *pt, float *distance)
{
    float A = 1;
    float dx = ray->origin.x - s->center.x;
    float dy = ray->origin.y - s->center.y;
    float dz = ray->origin.z - s->center.z;
    float B = 2 * (ray->direction.x * dx + ray->direction.y * dy + ray->direction.z * dz);
    float C = (dx * dx + dy * dy + dz * dz) - (s->radius * s->radius);
    float det = (B * B) - (4 * A * C);
    float w = 0.0f;
    c->cnt->sphere_tests++;
    if (det < 0.0f)
        return 0;
    /* roots are formed in double and rounded to float (ray.c:217-218) */
    float wOne = (float)(((double)(-B) - sqrt((double)det)) / (double)(2.0f * A));
    float wTwo = (float)(((double)(-B) + sqrt((double)det)) / (double)(2.0f * A));
    c->cnt->sphere_hits++;
    if (det == 0.0f) {
        *pt = add(ray->origin, scale(ray->direction, wOne));
        *distance = wOne;
        return 1;
    }
    if (wOne > 0.0f)
        w = wOne;
    if (wTwo > 0.0f && wTwo < w)
        w = wTwo;
    *pt = add(ray->origin, scale(ray->direction, w));
    *distance = w;
    return 1;
}

/* ray_intersect_polygon  ray.c:89-177 */
static int intersect_polygon(const ctx_t *c, const ray_t *ray, const rt_polygon *poly, v3 *pt,
                             float *distance)
{
    v3 n = {poly->plane.A, poly->plane.B, poly->plane.C};
    float num = -1.0f * (dot(n, ray->origin) + poly->plane.F);
    float den = dot(n, ray->direction);
    double angleTotal = 0.0;
    c->cnt->polygon_tests++;
    if (den == 0.0f) {
        c->cnt->polygon_rej_parallel++;
        return 0;
    }
    float w = num / den;
    if (w < 0.0f) {
        c->cnt->polygon_rej_behind++;
        return 0;
    }
    c->cnt->polygon_full++;
    *pt = add(ray->origin, scale(ray->direction, w));
    *distance = w;
    for (uint32_t i = 0; i < poly->nVerticies; ++i) {
        uint32_t j = (i == poly->nVerticies - 1) ? 0 : i + 1;
        v3 a = sub(v3_from(&poly->vertex[i]), *pt);
        v3 b = sub(v3_from(&poly->vertex[j]), *pt);
        double ang = (double)costheta(a, b);
        angleTotal += acos(ang);
    }
    if ((angleTotal + .001) > (2.0 * M_PI)) {
        c->cnt->polygon_hits++;
        return 1;
    }
    return 0;
}

/* ray_intersect_object  raytrace.c:170-182 */
static int intersect_object(const ctx_t *c, const ray_t *ray, const rt_object3d *obj, v3 *pt, float *d)
{
    switch (obj->geometryType) {
    case RT_GEOMETRY_POLYGON: return intersect_polygon(c, ray, &obj->poly_obj, pt, d);
    case RT_GEOMETRY_SPHERE:  return intersect_sphere(c, ray, &obj->sphr_obj, pt, d);
    default: return 0;
    }
}

/* get_object3d_intersect  raytrace.c:188-222: closest hit, no distance filter */
static const rt_object3d *closest(const ctx_t *c, const ray_t *ray, v3 *pt, float *d)
{
    const rt_object3d *best = 0;
    for (uint32_t i = 0; i < c->scene->nObjects; ++i) {
        v3 tp;
        float td;
        if (intersect_object(c, ray, &c->scene->objects[i], &tp, &td)) {
            if (best == 0 || td < *d) {
                *d = td;
                *pt = tp;
                best = &c->scene->objects[i];
            }
        }
    }
    return best;
}

/* get_object3d_intersect_excl  raytrace.c:231-272: skips `exc`, accepts 0 < d < magnitude */
static const rt_object3d *closest_excl(const ctx_t *c, const ray_t *ray, const rt_object3d *exc,
                                       v3 *pt, float *d)
{
    const rt_object3d *best = 0;
    for (uint32_t i = 0; i < c->scene->nObjects; ++i) {
        v3 tp;
        float td;
        if (&c->scene->objects[i] == exc)
            continue;
        if (intersect_object(c, ray, &c->scene->objects[i], &tp, &td)) {
            if (td < ray->magnitude && td > 0.0f) {
                if (best == 0 || td < *d) {
                    *d = td;
                    *pt = tp;
                    best = &c->scene->objects[i];
                }
            }
        }
    }
    return best;
}

/* ---- object3d.c / geometry.c ------------------------------------------------------------ */

/* get_object_normal object3d.c:15-27; get_sphere_normal geometry.c:37-45;
 * get_polygon_normal geometry.c:48-57 */
static v3 object_normal(const rt_object3d *obj, v3 pt)
{
    if (obj->geometryType == RT_GEOMETRY_SPHERE)
        return normalize(sub(pt, v3_from(&obj->sphr_obj.center)));
    v3 n = {obj->poly_obj.plane.A, obj->poly_obj.plane.B, obj->poly_obj.plane.C};
    return n;
}

/* get_object_color object3d.c:64-117 with get_2dproject_position object3d.c:30-39 */
static rt_color object_color(const rt_object3d *obj, v3 pt)
{
    if (obj->geometryType == RT_GEOMETRY_SPHERE)
        return obj->material.colors[RT_MATERIAL_DIFFUSECOLOR];
    float u = (pt.x + 15.0f) / 22.0f;
    float v = (pt.z + 100.0f) / 100.0f;
    int xtile = (int)((float)11 * u);
    int ytile = (int)((float)50 * v);
    rt_color col = {1.0f, 0.0f, 0.0f};
    if ((xtile % 2) != (ytile % 2))
        col.g = 1.0f;
    return col;
}

/* ---- raytrace.c ------------------------------------------------------------------------- */

/* get_shade_color_phong  raytrace.c:282-564 */
static rt_color shade(const ctx_t *c, const rt_object3d *obj, const ray_t *eye, v3 pt, unsigned depth)
{
    const rt_scene *sc = c->scene;
    const rt_material *m = &obj->material;
    const unsigned max_depth = c->cam->depth;
    rt_color out, diff = {0, 0, 0}, spec = {0, 0, 0};

    c->cnt->shade_nodes++;
    if (depth < 64)
        c->cnt->nodes_at_depth[depth]++;
    if (obj->geometryType == RT_GEOMETRY_SPHERE)
        c->cnt->nodes_sphere++;
    else
        c->cnt->nodes_polygon++;

    rt_color oc = object_color(obj, pt);
    /* ambient: colour * ambient light, then * ka  (:303-307) */
    out.r = oc.r * sc->ambientLightColor.r;
    out.g = oc.g * sc->ambientLightColor.g;
    out.b = oc.b * sc->ambientLightColor.b;
    out.r = out.r * m->phong_ka;
    out.g = out.g * m->phong_ka;
    out.b = out.b * m->phong_ka;

    v3 N = object_normal(obj, pt);
    v3 V = normalize(sub(eye->origin, pt)); /* :318-320 */

    for (uint32_t i = 0; i < sc->nLights; ++i) { /* :327-384 */
        const rt_pointlight *L = &sc->lights[i];
        ray_t shadow = ray_create(pt, v3_from(&L->position));
        ray_tinypush(&shadow);
        c->cnt->shadow_rays++;
        v3 hp;
        float hd;
        if (closest_excl(c, &shadow, obj, &hp, &hd))
            continue;
        v3 S = shadow.direction;
        float t = dot(S, N);
        v3 R = sub(scale(N, 2.0f * t), S);
        if (t <= 0.0f)
            continue;
        c->cnt->lights_lit++;
        diff.r += L->color.r * t;
        diff.g += L->color.g * t;
        diff.b += L->color.b * t;
        float s = dot(R, V);
        if (s < 0.0f)
            s = 0.0f;
        s = g_rounded_pow ? rounded_pow(s, m->phong_ke) : powf(s, m->phong_ke);
        spec.r += L->color.r * s;
        spec.g += L->color.g * s;
        spec.b += L->color.b * s;
    }
    const rt_color *sp = &m->colors[RT_MATERIAL_SPECULARCOLOR];
    out.r += m->phong_kd * diff.r * oc.r + m->phong_ks * spec.r * sp->r; /* :390-401 */
    out.g += m->phong_kd * diff.g * oc.g + m->phong_ks * spec.g * sp->g;
    out.b += m->phong_kd * diff.b * oc.b + m->phong_ks * spec.b * sp->b;

    if (m->kr != 0.0f && depth != max_depth) { /* reflection :405-451 */
        ray_t rr;
        rr.origin = pt;
        rr.magnitude = 9999999.9f;
        float t = dot(V, N);
        rr.direction = sub(scale(N, 2.0f * t), V);
        ray_tinypush(&rr);
        c->cnt->reflection_rays++;
        v3 hp;
        float hd;
        const rt_object3d *ho = closest_excl(c, &rr, 0, &hp, &hd);
        rt_color rc = ho ? shade(c, ho, &rr, hp, depth + 1) : sc->bgColor;
        out.r += m->kr * rc.r;
        out.g += m->kr * rc.g;
        out.b += m->kr * rc.b;
    }
    if (m->kt != 0.0f && depth != max_depth) { /* transmission :452-559 */
        ray_t tr;
        float nit = 1.0f / m->n;
        v3 T = scale(eye->direction, nit);
        float cs = dot(V, N);
        if (cs < 0.0f) { /* leaving the object: invert the ratio, flip N (:459-464) */
            nit = 1.0f / nit;
            N = scale(N, -1.0f);
        }
        float det = 1 + nit * nit * (cs * cs - 1);
        if (det < 0) { /* total internal reflection :470-513 */
            tr.origin = pt;
            tr.magnitude = 999999.0f;
            float t = dot(V, N);
            tr.direction = sub(scale(N, 2.0f * t), V);
        } else { /* refraction :514-539; note 1/n (not nit) and the double sqrt */
            float k = (float)((double)(1.0f / m->n * cs) - sqrt((double)det));
            tr.direction = add(T, scale(N, k));
            tr.origin = pt;
            tr.magnitude = 99999.9f;
        }
        ray_tinypush(&tr);
        c->cnt->transmission_rays++;
        v3 hp;
        float hd;
        const rt_object3d *ho = closest_excl(c, &tr, 0, &hp, &hd);
        rt_color tc = ho ? shade(c, ho, &tr, hp, depth + 1) : sc->bgColor;
        out.r += m->kt * tc.r;
        out.g += m->kt * tc.g;
        out.b += m->kt * tc.b;
    }
    /* color_clamp color.c:44-54: upper bound only (:562) */
    if (out.r > 1.0f) out.r = 1.0f;
    if (out.g > 1.0f) out.g = 1.0f;
    if (out.b > 1.0f) out.b = 1.0f;
    return out;
}

/* get_shade_color_phongblinn  raytrace.c:574-685: the reference's older shading routine, which nothing
 * calls (get_ray_color calls get_shade_color_phong, :706).  Local illumination only: the material's
 * diffuse colour (no checkerboard, :589), shadow rays that are NOT pushed off the surface (:613-616),
 * the halfway vector H = (V+S) * (1.0 / |V+S|) in place of the mirror direction (:632-634, :647),
 * no reflection, no transmission. */
static rt_color shade_blinn(const ctx_t *c, const rt_object3d *obj, const ray_t *eye, v3 pt)
{
    const rt_scene *sc = c->scene;
    const rt_material *m = &obj->material;
    const rt_color *dc = &m->colors[RT_MATERIAL_DIFFUSECOLOR];
    rt_color out, diff = {0, 0, 0}, spec = {0, 0, 0};

    c->cnt->shade_nodes++;
    c->cnt->nodes_at_depth[0]++;
    if (obj->geometryType == RT_GEOMETRY_SPHERE)
        c->cnt->nodes_sphere++;
    else
        c->cnt->nodes_polygon++;

    out.r = dc->r * sc->ambientLightColor.r; /* :589-593 */
    out.g = dc->g * sc->ambientLightColor.g;
    out.b = dc->b * sc->ambientLightColor.b;
    out.r = out.r * m->phong_ka;
    out.g = out.g * m->phong_ka;
    out.b = out.b * m->phong_ka;

    v3 N = object_normal(obj, pt);
    v3 V = normalize(sub(eye->origin, pt)); /* :604-606 */

    for (uint32_t i = 0; i < sc->nLights; ++i) { /* :609-663 */
        const rt_pointlight *L = &sc->lights[i];
        ray_t shadow = ray_create(pt, v3_from(&L->position));
        c->cnt->shadow_rays++;
        v3 hp;
        float hd;
        if (closest_excl(c, &shadow, obj, &hp, &hd))
            continue;
        v3 S = shadow.direction;
        v3 tmp = add(V, S);
        float inv = (float)(1.0 / (double)magnitude(tmp)); /* :633 double quotient stored to float */
        v3 H = scale(tmp, inv);
        float t = dot(S, N);
        if (t <= 0.0f)
            continue;
        c->cnt->lights_lit++;
        diff.r += L->color.r * t;
        diff.g += L->color.g * t;
        diff.b += L->color.b * t;
        float s = dot(H, N);
        if (s < 0.0f)
            s = 0.0f;
        s = g_rounded_pow ? rounded_pow(s, m->phong_ke) : powf(s, m->phong_ke);
        spec.r += L->color.r * s;
        spec.g += L->color.g * s;
        spec.b += L->color.b * s;
    }
    const rt_color *sp = &m->colors[RT_MATERIAL_SPECULARCOLOR];
    out.r += m->phong_kd * diff.r * dc->r + m->phong_ks * spec.r * sp->r; /* :669-680 */
    out.g += m->phong_kd * diff.g * dc->g + m->phong_ks * spec.g * sp->g;
    out.b += m->phong_kd * diff.b * dc->b + m->phong_ks * spec.b * sp->b;
    if (out.r > 1.0f) out.r = 1.0f; /* :683 */
    if (out.g > 1.0f) out.g = 1.0f;
    if (out.b > 1.0f) out.b = 1.0f;
    return out;
}

/* which of the reference's two shading routines get_ray_color calls: 0 = get_shade_color_phong (what
 * the reference does, raytrace.c:706), 1 = get_shade_color_phongblinn */
static int g_shading;
void rt_oracle_set_shading(int model) { g_shading = model; }

/* get_ray_color  raytrace.c:688-708 */
static rt_color ray_color(const ctx_t *c, const ray_t *ray)
{
    v3 pt;
    float d;
    const rt_object3d *obj = closest(c, ray, &pt, &d);
    if (!obj)
        return c->scene->bgColor;
    if (g_shading == 1)
        return shade_blinn(c, obj, ray, pt);
    return shade(c, obj, ray, pt, 0);
}

/* get_pixel_color  raytrace.c:732-780 */
static rt_color pixel_color(const ctx_t *c, unsigned x, unsigned y)
{
    const rt_camera *cam = c->cam;
    rt_color acc = {0.0f, 0.0f, 0.0f}; /* color_init, raytrace.c:908 */
    float sc = 1.0f / (float)(cam->sqrtSpp * cam->sqrtSpp);
    v3 eye = v3_from(&cam->eyePos), U = v3_from(&cam->U), V = v3_from(&cam->V), N = v3_from(&cam->N);
    for (unsigned j = 0; j < cam->sqrtSpp; ++j) {
        for (unsigned i = 0; i < cam->sqrtSpp; ++i) {
            v3 shift = scale(N, cam->viewDistance);
            float fu = ((((float)x - ((float)cam->frameBufferWidth / 2.0f)) /
                         ((float)cam->frameBufferWidth / 2.0f)) * cam->viewPlaneHalfWidth) +
                       (cam->sppWidth / 2.0f + (float)i * cam->sppWidth);
            shift = add(shift, scale(U, fu));
            float fv = (((((float)cam->frameBufferHeight / 2.0f) - (float)y) /
                         ((float)cam->frameBufferHeight / 2.0f)) * cam->viewPlaneHalfHeight) +
                       (cam->sppWidth / 2.0f + (float)j * cam->sppWidth);
            shift = add(shift, scale(V, fv));
            v3 target = add(eye, shift);
            ray_t ray = ray_create(eye, target);
            c->cnt->primary_rays++;
            rt_color col = ray_color(c, &ray);
            acc.r = acc.r + col.r;
            acc.g = acc.g + col.g;
            acc.b = acc.b + col.b;
        }
    }
    acc.r = acc.r * sc;
    acc.g = acc.g * sc;
    acc.b = acc.b * sc;
    return acc;
}

/* write_color_32  raytrace.c:783-790: truncating float -> unsigned char conversion as the
 * x86-64 build performs it (cvttss2si, low byte) */
static uint32_t pack_argb(rt_color col)
{
    uint32_t p = 0xFF000000u;
    p |= (uint32_t)((unsigned char)(int)(255 * col.r)) << 16;
    p |= (uint32_t)((unsigned char)(int)(255 * col.g)) << 8;
    p |= (uint32_t)((unsigned char)(int)(255 * col.b));
    return p;
}

static void merge_counts(rt_oracle_counts *dst, const rt_oracle_counts *src)
{
    const uint64_t *s = (const uint64_t *)src;
    uint64_t *d = (uint64_t *)dst;
    for (size_t i = 0; i < sizeof(*dst) / sizeof(uint64_t); ++i)
        d[i] += s[i];
}

/* ---- exported test API ------------------------------------------------------------------ */

/* prepare_scene  raytrace.c:72-85 */
void rt_oracle_prepare_camera(rt_camera *cam, const rt_scene_t *scene, unsigned width,
                              unsigned height, unsigned sqrtSpp, float fovY, float aspectRatio,
                              float nearZ, unsigned depth)
{
    cam->eyePos = scene->eyePos;
    cam->U = scene->U;
    cam->V = scene->V;
    cam->N = scene->N;
    cam->viewDistance = nearZ;
    cam->viewPlaneHalfHeight = cam->viewDistance * tan((fovY / 2.0f) * (M_PI / 180.0f));
    cam->viewPlaneHalfWidth = cam->viewPlaneHalfHeight * aspectRatio;
    cam->sppWidth = (cam->viewPlaneHalfWidth / (width / 2.0f)) / (float)sqrtSpp;
    cam->frameBufferWidth = width;
    cam->frameBufferHeight = height;
    cam->sqrtSpp = sqrtSpp;
    cam->depth = depth;
}

/* The pixel loop + pack of raytrace()  raytrace.c:903-927, restricted to rows
 * [row0,row0+nrows).  argb / radiance (either may be NULL) use full-frame indexing.
 * nthreads > 1 runs rows in parallel (OpenMP); results do not depend on nthreads. */
void rt_oracle_render(const rt_scene *scene, const rt_camera *cam, uint32_t *argb, float *radiance,
                      uint32_t row0, uint32_t nrows, rt_oracle_counts *counts, int nthreads)
{
    rt_oracle_counts total;
    memset(&total, 0, sizeof total);
    const uint32_t W = cam->frameBufferWidth;
    if (nthreads < 1)
        nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        rt_oracle_counts local;
        memset(&local, 0, sizeof local);
        ctx_t c = {scene, cam, &local};
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t y = row0; y < (int64_t)row0 + nrows; ++y) {
            for (uint32_t x = 0; x < W; ++x) {
                rt_color col = pixel_color(&c, x, (unsigned)y);
                size_t p = (size_t)x + (size_t)y * W;
                if (radiance) {
                    radiance[3 * p + 0] = col.r;
                    radiance[3 * p + 1] = col.g;
                    radiance[3 * p + 2] = col.b;
                }
                if (argb)
                    argb[p] = pack_argb(col);
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        merge_counts(&total, &local);
    }
    if (counts)
        *counts = total;
}

/* one pixel (get_pixel_color) - for sparse pins of configurations too big to render on a CPU */
void rt_oracle_pixel(const rt_scene *scene, const rt_camera *cam, uint32_t x, uint32_t y, float rgb[3])
{
    rt_oracle_counts local;
    memset(&local, 0, sizeof local);
    ctx_t c = {scene, cam, &local};
    rt_color col = pixel_color(&c, x, y);
    rgb[0] = col.r;
    rgb[1] = col.g;
    rgb[2] = col.b;
}

/* known-answer access to the two intersection routines and the procedural colour */
int rt_oracle_intersect_sphere(const float origin[3], const float dir[3], const rt_sphere *s,
                               float pt[3], float *distance)
{
    rt_oracle_counts local;
    memset(&local, 0, sizeof local);
    ctx_t c = {0, 0, &local};
    ray_t r = {{origin[0], origin[1], origin[2]}, {dir[0], dir[1], dir[2]}, 0.0f};
    v3 p = {0, 0, 0};
    *distance = 0.0f;
    int hit = intersect_sphere(&c, &r, s, &p, distance);
    pt[0] = p.x; pt[1] = p.y; pt[2] = p.z;
    return hit;
}

int rt_oracle_intersect_polygon(const float origin[3], const float dir[3], const rt_polygon *poly,
                                float pt[3], float *distance)
{
    rt_oracle_counts local;
    memset(&local, 0, sizeof local);
    ctx_t c = {0, 0, &local};
    ray_t r = {{origin[0], origin[1], origin[2]}, {dir[0], dir[1], dir[2]}, 0.0f};
    v3 p = {0, 0, 0};
    *distance = 0.0f;
    int hit = intersect_polygon(&c, &r, poly, &p, distance);
    pt[0] = p.x; pt[1] = p.y; pt[2] = p.z;
    return hit;
}

void rt_oracle_object_color(const rt_object3d *obj, const float pt[3], float rgb[3])
{
    v3 p = {pt[0], pt[1], pt[2]};
    rt_color c = object_color(obj, p);
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
}

/* ray_create (ray.c:29-41) from p1 to p2: out = origin[3], direction[3], magnitude */
void rt_oracle_ray_create(const float p1[3], const float p2[3], float out[7])
{
    v3 a = {p1[0], p1[1], p1[2]}, b = {p2[0], p2[1], p2[2]};
    ray_t r = ray_create(a, b);
    out[0] = r.origin.x; out[1] = r.origin.y; out[2] = r.origin.z;
    out[3] = r.direction.x; out[4] = r.direction.y; out[5] = r.direction.z;
    out[6] = r.magnitude;
}

size_t rt_oracle_counts_size(void) { return sizeof(rt_oracle_counts); }

/* ---- tone reproduction: apply_tone / ward_tone / reinhard_tone -- raytrace.c:792-871 ------------
 * The step the reference keeps between its pixel loop and the 8-bit pack (the call is commented
 * out at raytrace.c:917).  rgb: n pixels of 3 floats (the reference's colorbuffer), changed in
 * place.  op 1 = Ward (what apply_tone calls, :861), op 2 = Reinhard (reinhard_tone, :808-831, put in
 * ward_tone's place; key_pixel < 0: log-average key, else the luminance of that pixel, :816-819).
 * All sums are float and sequential, in pixel order, as in the reference.  sums_out (may be NULL)
 * receives totalLum, totalLogLum. */
static float luminance(const float *c) { return 0.27f * c[0] + 0.67f * c[1] + 0.06f * c[2]; } /* color.c:168-173 */
/* DIAGNOSTIC switch (off by default): the two frame-wide sums in double instead of the reference's
 * float running sums.  At 2 M pixels the float sum of log-luminances has an ulp of 1 against terms
 * of ~6, so the reference's own total is off by percents (2.4 % in the Ward factor at 1920x1080);
 * the HIP kernels sum in double, and the full-size test compares them with this variant. */
static int g_tone_wide_sums;
void rt_oracle_set_tone_wide_sums(int on) { g_tone_wide_sums = on; }

void rt_oracle_tone(float *rgb, uint64_t n, int op, float ldMax, float lMax, int64_t key_pixel,
                    float sums_out[2])
{
    const float ldMaxInv = 1.0f / ldMax; /* :838 */
    float totalLum = 0.0f, totalLogLum = 0.0f;
    double wideLum = 0.0, wideLogLum = 0.0;
    const float delta = .00001f;
    for (uint64_t i = 0; i < n; ++i) { /* :851-859 */
        float *c = rgb + 3 * i;
        c[0] = c[0] * lMax; c[1] = c[1] * lMax; c[2] = c[2] * lMax;
        const float l = luminance(c);
        totalLum += l;
        totalLogLum += logf(delta + l);
        wideLum += (double)l;
        wideLogLum += (double)logf(delta + l);
    }
    if (g_tone_wide_sums) {
        totalLum = (float)wideLum;
        totalLogLum = (float)wideLogLum;
    }
    if (sums_out) {
        sums_out[0] = totalLum;
        sums_out[1] = totalLogLum;
    }
    const float logAvgLum = expf((1.0f / (float)n) * totalLogLum); /* :795, :811 */
    if (op == 2) {
        float scale = 0.18f / logAvgLum; /* :814 */
        if (key_pixel > -1)
            scale = 0.18 / luminance(rgb + 3 * key_pixel); /* :818: a double quotient stored to float */
        for (uint64_t i = 0; i < n; ++i) { /* :821-830 */
            float *c = rgb + 3 * i;
            for (int k = 0; k < 3; ++k) {
                c[k] = c[k] * scale;
                c[k] = (c[k] * ldMax) / (1.0f + c[k]);
            }
        }
    } else {
        const float sf = powf((1.219f + powf(ldMax / 2.0f, 0.4f)) / (1.219f + powf(logAvgLum, 0.4f)), 2.5f); /* :797-798 */
        for (uint64_t i = 0; i < 3 * n; ++i)
            rgb[i] = rgb[i] * sf; /* :801-805 */
    }
    for (uint64_t i = 0; i < 3 * n; ++i) { /* back to the display model, clamped: :865-868 */
        rgb[i] = rgb[i] * ldMaxInv;
        if (rgb[i] > 1.0f)
            rgb[i] = 1.0f;
    }
}
