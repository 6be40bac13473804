/*
 * rt_main.c - the command-line host of the ray tracer, on top of librt_hip.so.
 *
 * Same contract as the reference's program (main.c:96-173):
 *     rt outputFile sceneFile imgWidth imgHeight samplesPerPixel^2 depth
 * parse the scene, render, write the image (raw P6, output.h:15-16), print the run time; the
 * same messages for too few arguments (main.c:127-131), for a scene that cannot be parsed
 * (main.c:153-157) and at the start / end of a run (main.c:144,170).  Options after the six
 * arguments are additions:
 *     --load-scene          read sceneFile for real (rt_load_scene) instead of the reference's
 *                           behaviour of checking that it opens and using the built-in scene
 *     --gpus N              spread the frame over HIP devices 0..N-1 (rt_hip_render_devices)
 *     --devices a,b,...     the same with an explicit device list
 *     --dump-radiance FILE  also write the un-quantised radiance: W*H*3 float32, row 0 on top
 *     --shading MODEL       phong (get_shade_color_phong, what the reference calls, raytrace.c:706) or
 *                           phongblinn (its unused get_shade_color_phongblinn, raytrace.c:574-685)
 * The run time is printed with sub-second resolution (the reference's time() has whole seconds).
 */
#include "rt_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int parse_list(const char *text, int *out, int room)
{
    int n = 0;
    while (*text && n < room) {
        char *end;
        const long d = strtol(text, &end, 10);
        if (end == text || d < 0)
            return -1;
        out[n++] = (int)d;
        if (*end == ',')
            ++end;
        else if (*end)
            return -1;
        text = end;
    }
    return n;
}

int main(int argc, char **argv)
{
    const double start = now_s();
    if (argc < 7) {
        printf("raytrace outputFile sceneFile imgWidth imgHeight samplesPerPixel^2 depth\n");
        return 1;
    }
    const char *img_out = argv[1], *scene_file = argv[2], *radiance_out = NULL;
    const unsigned width = (unsigned)atoi(argv[3]), height = (unsigned)atoi(argv[4]);
    const unsigned sqrt_spp = (unsigned)atoi(argv[5]), depth = (unsigned)atoi(argv[6]);
    int load_scene = 0, devices[64], ndevices = 0;
    for (int a = 7; a < argc; ++a) {
        if (!strcmp(argv[a], "--load-scene")) {
            load_scene = 1;
        } else if (!strcmp(argv[a], "--gpus") && a + 1 < argc) {
            ndevices = atoi(argv[++a]);
            if (ndevices < 1 || ndevices > 64) {
                printf("--gpus takes a count from 1 to 64\n");
                return 1;
            }
            for (int d = 0; d < ndevices; ++d)
                devices[d] = d;
        } else if (!strcmp(argv[a], "--devices") && a + 1 < argc) {
            ndevices = parse_list(argv[++a], devices, 64);
            if (ndevices < 1) {
                printf("--devices takes a comma-separated list of device numbers\n");
                return 1;
            }
        } else if (!strcmp(argv[a], "--dump-radiance") && a + 1 < argc) {
            radiance_out = argv[++a];
        } else if (!strcmp(argv[a], "--shading") && a + 1 < argc) {
            ++a;
            if (strcmp(argv[a], "phong") && strcmp(argv[a], "phongblinn")) {
                printf("--shading takes phong or phongblinn\n");
                return 1;
            }
            setenv("RT_HIP_SHADING", argv[a], 1); /* what raytrace() / rt_hip_render_devices follow */
        } else {
            printf("unknown option %s\n", argv[a]);
            return 1;
        }
    }
    if (width == 0 || height == 0 || sqrt_spp == 0) {
        printf("image width, height and samplesPerPixel^2 must be positive\n");
        return 1;
    }

    unsigned int *frame = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)width * height);
    float *radiance = radiance_out ? (float *)malloc(sizeof(float) * 3 * (size_t)width * height) : NULL;
    if (!frame || (radiance_out && !radiance)) {
        printf("out of memory for a %ux%u frame\n", width, height);
        return 1;
    }

    printf("Running Ray Tracer...\n");

    rt_scene_t scene;
    if (!(load_scene ? rt_load_scene(scene_file, &scene) : rt_parse_scene(scene_file, &scene))) {
        printf("Error parsing scene file.  Exiting...\n");
        return 1;
    }

    if (!radiance && ndevices <= 1 && (ndevices == 0 || devices[0] == 0)) {
        /* the reference's own call (main.c:159-160) */
        raytrace(frame, &scene, 45.0f, width / (float)height, 1.0f, 200.0f, width, height, sqrt_spp, depth);
    } else {
        rt_camera cam;
        rt_camera_prepare(&cam, &scene, width, height, sqrt_spp, 45.0f, width / (float)height, 1.0f, depth);
        rt_scene view;
        view.bgColor = scene.bgColor;
        view.ambientLightColor = scene.ambientLightColor;
        view.nLights = scene.nLights;
        view.nObjects = scene.nObjects;
        view.lights = scene.lights;
        view.objects = scene.objects;
        rt_framebuffer fb = {frame, radiance, width, height, 0, height};
        if (ndevices == 0) {
            devices[0] = 0;
            ndevices = 1;
        }
        const int rc = rt_hip_render_devices(&view, &cam, &fb, devices, (uint32_t)ndevices);
        if (rc) {
            fprintf(stderr, "raytrace (HIP): error %d: %s\n", rc, rt_hip_last_error());
            return 1;
        }
    }

    if (!write_image(img_out, frame, width, height)) {
        printf("Error writing %s\n", img_out);
        return 1;
    }
    if (radiance) {
        FILE *f = fopen(radiance_out, "wb");
        const size_t n = 3 * (size_t)width * height;
        if (!f || fwrite(radiance, sizeof(float), n, f) != n) {
            printf("Error writing %s\n", radiance_out);
            return 1;
        }
        fclose(f);
    }

    rt_scene_free(&scene);
    free(frame);
    free(radiance);
    printf("Run time:\t%f\n", now_s() - start);
    return 0;
}
