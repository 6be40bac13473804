/*
 * ref_wrap.c - link-time counters around the UNTOUCHED reference objects.
 *
 * TEST INFRASTRUCTURE ONLY (used by oracle/Makefile when it builds oracle/_ref/).
 * The reference's ray.c functions are called from raytrace.c across translation units, so
 * `ld --wrap=<sym>` can count them without editing any reference source.  Pointers are
 * passed through opaquely; nothing here depends on the reference's headers.
 */
#include <stdint.h>

uint64_t ref_n_ray_create, ref_n_ray_tinypush;
uint64_t ref_n_sphere_tests, ref_n_sphere_hits, ref_n_polygon_tests, ref_n_polygon_hits;

void *__real_ray_create(void *out, const void *p1, const void *p2);
void *__real_ray_tinypush(void *out, const void *ray);
int __real_ray_intersect_sphere(const void *ray, const void *sphere, void *pt, float *d);
int __real_ray_intersect_polygon(const void *ray, const void *poly, void *pt, float *d);

void *__wrap_ray_create(void *out, const void *p1, const void *p2)
{
    ++ref_n_ray_create;
    return __real_ray_create(out, p1, p2);
}

void *__wrap_ray_tinypush(void *out, const void *ray)
{
    ++ref_n_ray_tinypush;
    return __real_ray_tinypush(out, ray);
}

int __wrap_ray_intersect_sphere(const void *ray, const void *sphere, void *pt, float *d)
{
    int hit = __real_ray_intersect_sphere(ray, sphere, pt, d);
    ++ref_n_sphere_tests;
    ref_n_sphere_hits += (uint64_t)(hit != 0);
    return hit;
}

int __wrap_ray_intersect_polygon(const void *ray, const void *poly, void *pt, float *d)
{
    int hit = __real_ray_intersect_polygon(ray, poly, pt, d);
    ++ref_n_polygon_tests;
    ref_n_polygon_hits += (uint64_t)(hit != 0);
    return hit;
}

void ref_counts_reset(void)
{
    ref_n_ray_create = ref_n_ray_tinypush = 0;
    ref_n_sphere_tests = ref_n_sphere_hits = ref_n_polygon_tests = ref_n_polygon_hits = 0;
}

void ref_counts_get(uint64_t out[6])
{
    out[0] = ref_n_ray_create;
    out[1] = ref_n_ray_tinypush;
    out[2] = ref_n_sphere_tests;
    out[3] = ref_n_sphere_hits;
    out[4] = ref_n_polygon_tests;
    out[5] = ref_n_polygon_hits;
}

/* unwrapped access to the two intersection routines for known-answer tests */
int ref_intersect_sphere(const void *ray, const void *sphere, void *pt, float *d)
{
    return __real_ray_intersect_sphere(ray, sphere, pt, d);
}
int ref_intersect_polygon(const void *ray, const void *poly, void *pt, float *d)
{
    return __real_ray_intersect_polygon(ray, poly, pt, d);
}
