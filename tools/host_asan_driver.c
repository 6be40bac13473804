/* development: exercises the HOST side of librt_hip.so (scene flattening, grid build, edge equations, batching,
 * device-list path, tail stream, tone, shading switch) in a build whose host code carries AddressSanitizer and
 * UBSan (tools/host_asan.sh; device code is untouched).  Runs on the GPU box. */
#include "rt_hip.h"
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { int rc_ = (x); if (rc_) { printf("%s -> %d: %s\n", #x, rc_, rt_hip_last_error()); return 1; } } while (0)

static int render_host(rt_scene_t *sc, unsigned W, unsigned H, unsigned s, unsigned d, int ndev)
{
    rt_camera cam;
    rt_camera_prepare(&cam, sc, W, H, s, 45.0f, (float)W / (float)H, 1.0f, d);
    rt_scene view = {sc->bgColor, sc->ambientLightColor, sc->nLights, sc->nObjects, sc->lights, sc->objects};
    unsigned *argb = malloc(sizeof(unsigned) * W * H);
    float *rad = malloc(sizeof(float) * 3 * W * H);
    rt_framebuffer fb = {argb, rad, W, H, 0, H};
    int devs[3] = {0, 0, 0};
    int rc = ndev ? rt_hip_render_devices(&view, &cam, &fb, devs, (uint32_t)ndev) : rt_hip_render(&view, &cam, &fb);
    if (rc)
        printf("render -> %d: %s\n", rc, rt_hip_last_error());
    free(argb);
    free(rad);
    return rc;
}

int main(void)
{
    rt_scene_t sc;
    memset(&sc, 0, sizeof sc);
    rt_scene_builtin(&sc);
    CHECK(render_host(&sc, 333, 217, 3, 5, 0));
    CHECK(render_host(&sc, 64, 48, 1, 0, 3));
    unsigned *buf = calloc(40 * 30, sizeof(unsigned));
    raytrace(buf, &sc, 45.0f, 40.0f / 30.0f, 1.0f, 200.0f, 40, 30, 2, 4);
    free(buf);
    rt_scene_free(&sc);
    const unsigned shapes[][3] = {{0, 1, 1}, {5, 0, 1}, {64, 4, 1}, {129, 4, 8}, {300, 2, 40}, {3000, 4, 2}, {20000, 0, 3}};
    for (unsigned k = 0; k < sizeof shapes / sizeof shapes[0]; ++k) {
        memset(&sc, 0, sizeof sc);
        if (!rt_scene_synthetic(&sc, 77 + k, shapes[k][0], shapes[k][1], shapes[k][2], (int)(k & 1))) {
            printf("synthetic %u failed\n", k);
            return 1;
        }
        CHECK(render_host(&sc, 96, 64, 2, 6, (int)(k % 3)));
        rt_scene_free(&sc);
    }
    memset(&sc, 0, sizeof sc);
    if (!rt_load_scene("tests/golden/triangle_two_lights.scene", &sc)) return 1;
    CHECK(render_host(&sc, 80, 60, 2, 8, 0));

    /* device-resident interface: small queues (overflow retries), many batches, tail stream, shading, tone */
    rt_hip_ctx *ctx = NULL;
    CHECK(rt_hip_ctx_create(&ctx, 0, 1u << 18));
    rt_scene view = {sc.bgColor, sc.ambientLightColor, sc.nLights, sc.nObjects, sc.lights, sc.objects};
    CHECK(rt_hip_scene_upload(ctx, &view));
    rt_camera cam;
    rt_camera_prepare(&cam, &sc, 160, 120, 2, 45.0f, 160.0f / 120.0f, 1.0f, 8);
    uint32_t *d_argb = NULL;
    float *d_rad = NULL;
    hipStream_t main_s, tail_s;
    if (hipMalloc((void **)&d_argb, 160 * 120 * 4) || hipMalloc((void **)&d_rad, 160 * 120 * 12) ||
        hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking) || hipStreamCreateWithFlags(&tail_s, hipStreamNonBlocking))
        return 1;
    rt_hip_stats st;
    for (int pass = 0; pass < 5; ++pass) {
        CHECK(rt_hip_ctx_set_jit(ctx, pass == 4 ? RT_JIT_NOW : RT_JIT_OFF)); /* (pass 4: kernels compiled for this scene's counts) */
        CHECK(rt_hip_ctx_set_tail_stream(ctx, pass >= 2 ? tail_s : NULL));
        CHECK(rt_hip_ctx_set_shading(ctx, pass == 3 ? RT_SHADE_PHONGBLINN : RT_SHADE_PHONG));
        if (pass == 4) {
            char log[256];
            double sec = 0.0;
            CHECK(rt_hip_render_rows(ctx, &cam, d_argb, d_rad, 0, 120, main_s));
            CHECK(rt_hip_finish(ctx, &st));
            printf("jit status %d (%.2f s)\n", rt_hip_ctx_jit_status(ctx, log, sizeof log, &sec), sec);
        }
        CHECK(rt_hip_render_rows(ctx, &cam, d_argb, d_rad, 0, 120, main_s));
        CHECK(rt_hip_finish(ctx, &st));
        printf("pass %d: batches %u retries %u nodes %llu\n", pass, st.batches, st.overflow_retries,
               (unsigned long long)st.tree_nodes);
    }
    CHECK(rt_hip_render_stats(ctx, &cam, d_argb, d_rad, 0, 120, main_s, &st));
    CHECK(rt_hip_render_interleaved(ctx, &cam, d_argb, d_rad, 1, 3, main_s));
    CHECK(rt_hip_finish(ctx, NULL));
    CHECK(rt_hip_tone(d_rad, d_argb, 160 * 40, RT_TONE_WARD, 100.0f, 1000.0f, -1, main_s));
    CHECK(rt_hip_tone(d_rad, d_argb, 160 * 40, RT_TONE_REINHARD, 100.0f, 1000.0f, 17, main_s));
    if (hipStreamSynchronize(main_s)) return 1;
    if (rt_hip_ctx_set_shading(ctx, 9) != RT_E_INVALID) return 1;
    rt_hip_ctx_destroy(ctx);
    rt_scene_free(&sc);
    (void)hipFree(d_argb);
    (void)hipFree(d_rad);
    puts("host side clean");
    return 0;
}
