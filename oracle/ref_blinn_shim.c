/*
 * oracle/ref_blinn_shim.c - test infrastructure, never shipped.
 *
 * The reference keeps a second shading routine, get_shade_color_phongblinn (raytrace.c:574-685), that
 * nothing calls: get_ray_color calls get_shade_color_phong (raytrace.c:706).  To get the REFERENCE's own
 * pixels with that routine without touching its sources, this shim defines get_shade_color_phong and
 * forwards to the routine of the untouched build; loaded into the global symbol scope ahead of
 * oracle/_ref/libref_raytracer.so it takes the place of the library's own definition (the library is
 * position-independent code, so get_ray_color calls through the PLT).  Used by
 * tests/golden/make_golden_blinn.py only, in a process of its own.
 */
void *get_shade_color_phongblinn(void *colorout, const void *obj, const void *eye, const void *pt, const void *scene);

static unsigned long g_calls;
unsigned long ref_blinn_shim_calls(void) { return g_calls; }

void *get_shade_color_phong(void *colorout, const void *obj, const void *eye, const void *pt, const void *scene,
                            unsigned int depth)
{
    (void)depth;
    ++g_calls;
    return get_shade_color_phongblinn(colorout, obj, eye, pt, scene);
}
