/*
 * rt_host.c - host-side (CPU, plain C) half of librt_hip.so: camera preparation, the
 * builtin Whitted scene, seeded synthetic scenes, plane construction and the P6 writer.
 *
 * None of this is on the GPU hot path; it is the glue the reference keeps in scene.c,
 * geometry.c (poly_plane), raytrace.c (prepare_scene) and pam_output/output.c, re-stated
 * here so that a host program can build exactly the inputs the reference would.
 * Compile with -ffp-contract=off: the float expressions below must round exactly like the
 * reference's x86-64 build.
 */
#include "rt_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- small vector helpers with the reference's rounding behaviour ------------------- */

/* vec4_magnitude (vector4.c:219-233): float sum of squares, double sqrt, float result. */
static float mag3(float x, float y, float z)
{
    float s = x * x + y * y + z * z;
    return (float)sqrt((double)s);
}

/* vec4_normalize (vector4.c:268-278): divides every component by the magnitude. */
static void normalize3(float *x, float *y, float *z)
{
    float m = mag3(*x, *y, *z);
    *x /= m;
    *y /= m;
    *z /= m;
}

/* ---- prepare_scene (raytrace.c:72-85) ----------------------------------------------- */

void rt_camera_prepare(rt_camera *cam, const rt_scene_t *scene,
                       unsigned int width, unsigned int height, unsigned int sqrtSpp,
                       float fovY, float aspectRatio, float nearZ, unsigned int depth)
{
    cam->eyePos = scene->eyePos;
    cam->U = scene->U;
    cam->V = scene->V;
    cam->N = scene->N;
    cam->viewDistance = nearZ;
    /* float * double tan(double): the product is formed in double, then stored to float */
    cam->viewPlaneHalfHeight =
        (float)((double)cam->viewDistance * tan((double)(fovY / 2.0f) * (M_PI / (double)180.0f)));
    cam->viewPlaneHalfWidth = cam->viewPlaneHalfHeight * aspectRatio;
    cam->sppWidth = (cam->viewPlaneHalfWidth / ((float)width / 2.0f)) / (float)sqrtSpp;
    cam->frameBufferWidth = width;
    cam->frameBufferHeight = height;
    cam->sqrtSpp = sqrtSpp;
    cam->depth = depth;
}

/* ---- poly_plane (geometry.c:14-34) --------------------------------------------------- */

void rt_poly_plane(rt_plane *planeout, const rt_polygon *poly)
{
    const rt_vec4 *v = poly->vertex;
    /* a = v0 - v1, b = v2 - v1, n = b x a */
    float ax = v[0].x - v[1].x, ay = v[0].y - v[1].y, az = v[0].z - v[1].z;
    float bx = v[2].x - v[1].x, by = v[2].y - v[1].y, bz = v[2].z - v[1].z;
    float nx = by * az - bz * ay;
    float ny = bz * ax - bx * az;
    float nz = bx * ay - by * ax;
    normalize3(&nx, &ny, &nz);
    float f = nx * v[0].x + ny * v[0].y + nz * v[0].z;
    if (f < 0.0f)
        f = -f; /* the reference forces F >= 0 (geometry.c:31-32) */
    planeout->A = nx;
    planeout->B = ny;
    planeout->C = nz;
    planeout->F = f;
}

/* ---- scene construction -------------------------------------------------------------- */

static void set_color(rt_color *c, float r, float g, float b)
{
    c->r = r;
    c->g = g;
    c->b = b;
}

static void set_vec(rt_vec4 *v, float x, float y, float z, float w)
{
    v->x = x;
    v->y = y;
    v->z = z;
    v->w = w;
}

static void set_material(rt_material *m, float dr, float dg, float db, float ka, float kd,
                         float ks, float ke, float kr, float kt, float n)
{
    set_color(&m->colors[RT_MATERIAL_DIFFUSECOLOR], dr, dg, db);
    set_color(&m->colors[RT_MATERIAL_SPECULARCOLOR], 1.0f, 1.0f, 1.0f);
    m->phong_ka = ka;
    m->phong_kd = kd;
    m->phong_ks = ks;
    m->phong_ke = ke;
    m->kr = kr;
    m->kt = kt;
    m->n = n;
}

static void make_sphere(rt_object3d *o, const char *name, float cx, float cy, float cz, float r)
{
    memset(o, 0, sizeof *o);
    o->geometryType = RT_GEOMETRY_SPHERE;
    o->debugName = (char *)name;
    set_vec(&o->sphr_obj.center, cx, cy, cz, 1.0f);
    o->sphr_obj.radius = r;
}

static int make_quad(rt_object3d *o, const char *name, const float v[4][3])
{
    memset(o, 0, sizeof *o);
    o->geometryType = RT_GEOMETRY_POLYGON;
    o->debugName = (char *)name;
    o->poly_obj.nVerticies = 4;
    o->poly_obj.vertex = (rt_vec4 *)malloc(4 * sizeof(rt_vec4));
    if (!o->poly_obj.vertex)
        return 0;
    for (int i = 0; i < 4; ++i)
        set_vec(&o->poly_obj.vertex[i], v[i][0], v[i][1], v[i][2], 1.0f);
    rt_poly_plane(&o->poly_obj.plane, &o->poly_obj);
    return 1;
}

/* camera basis the way scene.c:236-258 builds it: N = normalize(lookAt - eye), V = up
 * (not re-orthogonalised), U = N x V */
static void set_camera(rt_scene_t *s, float ex, float ey, float ez, float lx, float ly,
                       float lz)
{
    set_vec(&s->eyePos, ex, ey, ez, 1.0f);
    /* lookAt is given relative to the eye and added with a w-clearing add (scene.c:241-245) */
    set_vec(&s->lookAt, ex + lx, ey + ly, ez + lz, 0.0f);
    set_vec(&s->upVec, 0.0f, 1.0f, 0.0f, 1.0f);
    float nx = s->lookAt.x - s->eyePos.x, ny = s->lookAt.y - s->eyePos.y,
          nz = s->lookAt.z - s->eyePos.z;
    normalize3(&nx, &ny, &nz);
    set_vec(&s->N, nx, ny, nz, 0.0f);
    s->V = s->upVec;
    set_vec(&s->U, s->N.y * s->V.z - s->N.z * s->V.y, s->N.z * s->V.x - s->N.x * s->V.z,
            s->N.x * s->V.y - s->N.y * s->V.x, 0.0f);
}

static const float k_floor[4][3] = {
    {7.0f, 0.0f, 0.0f}, {7.0f, 0.0f, -100.0f}, {-15.0f, 0.0f, -100.0f}, {-15.0f, 0.0f, 0.0f}};

static void scene_common(rt_scene_t *s)
{
    memset(s, 0, sizeof *s);
    s->ldMax = 100.0f;
    s->lMax = 1000.0f;
    set_color(&s->bgColor, .4470588f, .6274501f, .8705882f);
    set_color(&s->ambientLightColor, 1.0f, 1.0f, 1.0f);
    set_camera(s, 0.0f, 4.5f, 0.0f, 0.0f, 0.0f, -1.0f);
}

/* The values of scene.c:86-258 (two spheres, the checkerboard floor, one light). */
int rt_scene_builtin(rt_scene_t *s)
{
    scene_common(s);
    s->nLights = 1;
    s->nObjects = 3;
    s->lights = (rt_pointlight *)calloc(s->nLights, sizeof(rt_pointlight));
    s->objects = (rt_object3d *)calloc(s->nObjects, sizeof(rt_object3d));
    if (!s->lights || !s->objects)
        return 0;

    make_sphere(&s->objects[0], "Sphere 1", 0.0f, 5.0f, -6.0f, 1.15f);
    set_material(&s->objects[0].material, 1.0f, 1.0f, 1.0f, 0.075f, 0.075f, 0.2f, 20.0f, 0.01f,
                 0.85f, 0.95f);

    make_sphere(&s->objects[1], "Sphere 2", -1.25f, 3.75f, -7.25f, 1.0f);
    set_material(&s->objects[1].material, 0.7f, 0.7f, 0.7f, 0.15f, 0.25f, 1.0f, 20.0f, 0.75f,
                 0.0f, 1.0f);

    if (!make_quad(&s->objects[2], "Floor", k_floor))
        return 0;
    set_material(&s->objects[2].material, 0.0f, 1.0f, 0.0f, 0.1f, 0.7f, 0.2f, 2.0f, 0.0f, 0.0f,
                 1.0f);

    set_vec(&s->lights[0].position, 1.0f, 8.0f, 1.0f, 1.0f);
    set_color(&s->lights[0].color, 1.0f, 1.0f, 1.0f);
    s->lights[0].range = 150.0f;
    return 1;
}

/* parse_scene (scene.c:24-281): the reference tokenises the file and discards every token,
 * then installs the hard-coded scene; only an unreadable file is an error. */
int rt_parse_scene(const char *filename, rt_scene_t *scene)
{
    FILE *fp = fopen(filename, "r");
    if (!fp) {
        printf("Error opening file {%s} for reading.\n", filename);
        return 0;
    }
    char tok[1024];
    while (fscanf(fp, "%1023s", tok) == 1) {
        if (tok[0] == '#') { /* comment: drop the rest of the line */
            int ch;
            while ((ch = fgetc(fp)) != EOF && ch != '\n')
                ;
        }
    }
    fclose(fp);
    return rt_scene_builtin(scene);
}

void rt_scene_free(rt_scene_t *s)
{
    if (s->objects) {
        for (uint32_t i = 0; i < s->nObjects; ++i)
            if (s->objects[i].geometryType == RT_GEOMETRY_POLYGON)
                free(s->objects[i].poly_obj.vertex);
    }
    free(s->objects);
    free(s->lights);
    s->objects = NULL;
    s->lights = NULL;
    s->nObjects = s->nLights = 0;
}

/* ---- scene-file loader ------------------------------------- */
/*
 * The reference ships a `scene` file whose grammar its parse_scene never interprets
 * (scene.c:36-84 reads tokens and drops them).  rt_load_scene reads that grammar for real:
 *
 *   desc       { <objects> <lights> }                     advisory counts, checked when present
 *   camera     { position x y z   lookat x y z   upvector x y z   projection fov near far }
 *   world      { ambient r g b   background r g b }       (extension; defaults = builtin scene)
 *   pointlight { position x y z   diffuse r g b   range r }
 *   sphere     { name s   position x y z   radius r   <material keys> }
 *   polygon    { name s   point x y z  (3 or more)    <material keys> }
 *   material keys: diffuse r g b | specular r g b | ka v | kd v | ks v | ke v | kr v | kt v | n v
 *   '#' starts a comment; numbers may carry a C 'f' suffix (the reference's file has "1.0f").
 *
 * Camera basis as scene.c:236-258: N = normalize(lookat - position), V = upvector as given (not
 * re-orthogonalised), U = N x V.  `projection` is parsed and validated but not stored: fovY / near
 * are arguments of raytrace() (main.c:159-160), not fields of scene_t.
 * Names live in the same allocation as the object array, so rt_scene_free releases them.
 */
typedef struct {
    FILE *fp;
    const char *file;
    int line;
    char tok[256];
} rt_lexer;

static int lex_next(rt_lexer *lx)
{
    int ch;
    for (;;) {
        ch = fgetc(lx->fp);
        if (ch == EOF)
            return 0;
        if (ch == '\n')
            lx->line++;
        if (ch == '#') {
            while ((ch = fgetc(lx->fp)) != EOF && ch != '\n')
                ;
            lx->line++;
            continue;
        }
        if (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n')
            continue;
        break;
    }
    size_t n = 0;
    if (ch == '{' || ch == '}') {
        lx->tok[n++] = (char)ch;
    } else {
        while (ch != EOF && ch != ' ' && ch != '\t' && ch != '\r' && ch != '\n' && ch != '{' && ch != '}' &&
               ch != '#') {
            if (n + 1 < sizeof lx->tok)
                lx->tok[n++] = (char)ch;
            ch = fgetc(lx->fp);
        }
        if (ch != EOF)
            ungetc(ch, lx->fp);
    }
    lx->tok[n] = 0;
    return 1;
}

static int lex_fail(const rt_lexer *lx, const char *what)
{
    printf("Error parsing scene file {%s} line %d: %s (at \"%s\").\n", lx->file, lx->line, what, lx->tok);
    fflush(stdout);
    return 0;
}

static int lex_floats(rt_lexer *lx, float *out, int n)
{
    for (int i = 0; i < n; ++i) {
        if (!lex_next(lx))
            return lex_fail(lx, "number expected, file ends");
        char *end = NULL;
        out[i] = strtof(lx->tok, &end);
        if (end == lx->tok || !(end[0] == 0 || ((end[0] == 'f' || end[0] == 'F') && end[1] == 0)))
            return lex_fail(lx, "number expected");
    }
    return 1;
}

typedef struct {
    rt_object3d obj;
    char name[64];
    float (*pts)[3];
    uint32_t npts, cap;
} rt_parsed_obj;

static int parse_material_key(rt_lexer *lx, rt_material *m, int *known)
{
    float v[3];
    *known = 1;
    if (!strcmp(lx->tok, "diffuse")) {
        if (!lex_floats(lx, v, 3)) return 0;
        set_color(&m->colors[RT_MATERIAL_DIFFUSECOLOR], v[0], v[1], v[2]);
    } else if (!strcmp(lx->tok, "specular")) {
        if (!lex_floats(lx, v, 3)) return 0;
        set_color(&m->colors[RT_MATERIAL_SPECULARCOLOR], v[0], v[1], v[2]);
    } else if (!strcmp(lx->tok, "ka")) { if (!lex_floats(lx, &m->phong_ka, 1)) return 0;
    } else if (!strcmp(lx->tok, "kd")) { if (!lex_floats(lx, &m->phong_kd, 1)) return 0;
    } else if (!strcmp(lx->tok, "ks")) { if (!lex_floats(lx, &m->phong_ks, 1)) return 0;
    } else if (!strcmp(lx->tok, "ke")) { if (!lex_floats(lx, &m->phong_ke, 1)) return 0;
    } else if (!strcmp(lx->tok, "kr")) { if (!lex_floats(lx, &m->kr, 1)) return 0;
    } else if (!strcmp(lx->tok, "kt")) { if (!lex_floats(lx, &m->kt, 1)) return 0;
    } else if (!strcmp(lx->tok, "n")) {  if (!lex_floats(lx, &m->n, 1)) return 0;
    } else {
        *known = 0;
    }
    return 1;
}

int rt_load_scene(const char *filename, rt_scene_t *scene)
{
    rt_lexer lx;
    memset(&lx, 0, sizeof lx);
    lx.fp = fopen(filename, "r");
    lx.file = filename;
    lx.line = 1;
    if (!lx.fp) {
        printf("Error opening file {%s} for reading.\n", filename);
        return 0;
    }
    scene_common(scene);
    float eye[3] = {scene->eyePos.x, scene->eyePos.y, scene->eyePos.z};
    float look[3] = {scene->lookAt.x, scene->lookAt.y, scene->lookAt.z};
    float up[3] = {0.0f, 1.0f, 0.0f};
    rt_parsed_obj *objs = NULL;
    rt_pointlight *lights = NULL;
    uint32_t nobj = 0, cobj = 0, nlight = 0, clight = 0;
    long want_obj = -1, want_light = -1;
    int ok = 1;

    while (ok && lex_next(&lx)) {
        char block[32];
        snprintf(block, sizeof block, "%.31s", lx.tok);
        if (!lex_next(&lx) || strcmp(lx.tok, "{")) { ok = lex_fail(&lx, "'{' expected after block name"); break; }
        if (!strcmp(block, "desc")) {
            float v[2];
            if (!(ok = lex_floats(&lx, v, 2))) break;
            want_obj = (long)v[0];
            want_light = (long)v[1];
            if (!lex_next(&lx) || strcmp(lx.tok, "}")) { ok = lex_fail(&lx, "'}' expected"); break; }
            continue;
        }
        rt_parsed_obj *po = NULL;
        rt_pointlight *pl = NULL;
        const int is_sphere = !strcmp(block, "sphere"), is_poly = !strcmp(block, "polygon");
        const int is_light = !strcmp(block, "pointlight"), is_cam = !strcmp(block, "camera");
        const int is_world = !strcmp(block, "world");
        if (is_sphere || is_poly) {
            if (nobj == cobj) {
                cobj = cobj ? 2 * cobj : 16;
                rt_parsed_obj *g = (rt_parsed_obj *)realloc(objs, cobj * sizeof *objs);
                if (!g) { ok = lex_fail(&lx, "out of memory"); break; }
                objs = g;
            }
            po = &objs[nobj++];
            memset(po, 0, sizeof *po);
            po->obj.geometryType = is_sphere ? RT_GEOMETRY_SPHERE : RT_GEOMETRY_POLYGON;
            po->obj.sphr_obj.center.w = 1.0f;
            /* the floor's material of the builtin scene is the default */
            set_material(&po->obj.material, 0.5f, 0.5f, 0.5f, 0.1f, 0.7f, 0.2f, 2.0f, 0.0f, 0.0f, 1.0f);
        } else if (is_light) {
            if (nlight == clight) {
                clight = clight ? 2 * clight : 8;
                rt_pointlight *g = (rt_pointlight *)realloc(lights, clight * sizeof *lights);
                if (!g) { ok = lex_fail(&lx, "out of memory"); break; }
                lights = g;
            }
            pl = &lights[nlight++];
            memset(pl, 0, sizeof *pl);
            set_vec(&pl->position, 0.0f, 0.0f, 0.0f, 1.0f);
            set_color(&pl->color, 1.0f, 1.0f, 1.0f);
            pl->range = 150.0f;
        } else if (!is_cam && !is_world) {
            ok = lex_fail(&lx, "unknown block");
            break;
        }
        for (;;) { /* keys of the block */
            if (!lex_next(&lx)) { ok = lex_fail(&lx, "'}' expected, file ends"); break; }
            if (!strcmp(lx.tok, "}"))
                break;
            float v[3];
            int known = 0;
            if (po) {
                if (!(ok = parse_material_key(&lx, &po->obj.material, &known))) break;
                if (known) continue;
                if (!strcmp(lx.tok, "name")) {
                    if (!lex_next(&lx)) { ok = lex_fail(&lx, "name expected"); break; }
                    snprintf(po->name, sizeof po->name, "%.63s", lx.tok);
                } else if (is_sphere && !strcmp(lx.tok, "position")) {
                    if (!(ok = lex_floats(&lx, v, 3))) break;
                    set_vec(&po->obj.sphr_obj.center, v[0], v[1], v[2], 1.0f);
                } else if (is_sphere && !strcmp(lx.tok, "radius")) {
                    if (!(ok = lex_floats(&lx, &po->obj.sphr_obj.radius, 1))) break;
                } else if (is_poly && !strcmp(lx.tok, "point")) {
                    if (!(ok = lex_floats(&lx, v, 3))) break;
                    if (po->npts == po->cap) {
                        po->cap = po->cap ? 2 * po->cap : 8;
                        float(*g)[3] = (float(*)[3])realloc(po->pts, po->cap * sizeof *po->pts);
                        if (!g) { ok = lex_fail(&lx, "out of memory"); break; }
                        po->pts = g;
                    }
                    memcpy(po->pts[po->npts++], v, sizeof v);
                } else { ok = lex_fail(&lx, "unknown key"); break; }
            } else if (pl) {
                if (!strcmp(lx.tok, "position")) {
                    if (!(ok = lex_floats(&lx, v, 3))) break;
                    set_vec(&pl->position, v[0], v[1], v[2], 1.0f);
                } else if (!strcmp(lx.tok, "diffuse")) {
                    if (!(ok = lex_floats(&lx, v, 3))) break;
                    set_color(&pl->color, v[0], v[1], v[2]);
                } else if (!strcmp(lx.tok, "range")) {
                    if (!(ok = lex_floats(&lx, &pl->range, 1))) break;
                } else { ok = lex_fail(&lx, "unknown key"); break; }
            } else if (is_cam) {
                if (!strcmp(lx.tok, "position")) { if (!(ok = lex_floats(&lx, eye, 3))) break;
                } else if (!strcmp(lx.tok, "lookat")) { if (!(ok = lex_floats(&lx, look, 3))) break;
                } else if (!strcmp(lx.tok, "upvector")) { if (!(ok = lex_floats(&lx, up, 3))) break;
                } else if (!strcmp(lx.tok, "projection")) {
                    if (!(ok = lex_floats(&lx, v, 3))) break;
                    if (!(v[0] > 0.0f && v[0] < 180.0f && v[1] > 0.0f)) { ok = lex_fail(&lx, "bad projection"); break; }
                } else { ok = lex_fail(&lx, "unknown key"); break; }
            } else { /* world */
                if (!strcmp(lx.tok, "ambient")) {
                    if (!(ok = lex_floats(&lx, v, 3))) break;
                    set_color(&scene->ambientLightColor, v[0], v[1], v[2]);
                } else if (!strcmp(lx.tok, "background")) {
                    if (!(ok = lex_floats(&lx, v, 3))) break;
                    set_color(&scene->bgColor, v[0], v[1], v[2]);
                } else { ok = lex_fail(&lx, "unknown key"); break; }
            }
        }
        if (ok && is_poly && po->npts < 3)
            ok = lex_fail(&lx, "a polygon needs at least three points");
    }
    fclose(lx.fp);
    if (ok && want_obj >= 0 && ((long)nobj != want_obj || (long)nlight != want_light)) {
        printf("Error parsing scene file {%s}: desc announces %ld objects and %ld lights, found %u and %u.\n",
               filename, want_obj, want_light, nobj, nlight);
        fflush(stdout);
        ok = 0;
    }

    if (ok) { /* camera: scene.c:236-258 */
        set_vec(&scene->eyePos, eye[0], eye[1], eye[2], 1.0f);
        set_vec(&scene->lookAt, look[0], look[1], look[2], 0.0f);
        set_vec(&scene->upVec, up[0], up[1], up[2], 1.0f);
        float nx = look[0] - eye[0], ny = look[1] - eye[1], nz = look[2] - eye[2];
        normalize3(&nx, &ny, &nz);
        set_vec(&scene->N, nx, ny, nz, 0.0f);
        scene->V = scene->upVec;
        set_vec(&scene->U, scene->N.y * scene->V.z - scene->N.z * scene->V.y,
                scene->N.z * scene->V.x - scene->N.x * scene->V.z,
                scene->N.x * scene->V.y - scene->N.y * scene->V.x, 0.0f);
        /* objects + their names in one allocation; vertices per polygon (rt_scene_free) */
        const size_t names = (size_t)nobj * sizeof objs[0].name;
        scene->objects = (rt_object3d *)calloc(1, (nobj ? nobj : 1) * sizeof(rt_object3d) + names);
        scene->lights = (rt_pointlight *)calloc(nlight ? nlight : 1, sizeof(rt_pointlight));
        if (!scene->objects || !scene->lights)
            ok = 0;
        for (uint32_t i = 0; ok && i < nobj; ++i) {
            rt_object3d *o = &scene->objects[i];
            *o = objs[i].obj;
            char *nm = (char *)(scene->objects + nobj) + (size_t)i * sizeof objs[0].name;
            memcpy(nm, objs[i].name, sizeof objs[0].name);
            o->debugName = nm;
            if (o->geometryType == RT_GEOMETRY_POLYGON) {
                o->poly_obj.nVerticies = objs[i].npts;
                o->poly_obj.vertex = (rt_vec4 *)malloc(objs[i].npts * sizeof(rt_vec4));
                if (!o->poly_obj.vertex) { ok = 0; break; }
                for (uint32_t k = 0; k < objs[i].npts; ++k)
                    set_vec(&o->poly_obj.vertex[k], objs[i].pts[k][0], objs[i].pts[k][1], objs[i].pts[k][2], 1.0f);
                rt_poly_plane(&o->poly_obj.plane, &o->poly_obj);
            }
            scene->nObjects = i + 1; /* what rt_scene_free must walk if a later allocation fails */
        }
        if (ok) {
            if (nlight) /* (a scene without lights has no array to copy from) */
                memcpy(scene->lights, lights, nlight * sizeof *lights);
            scene->nLights = nlight;
        }
    }
    for (uint32_t i = 0; i < nobj; ++i)
        free(objs[i].pts);
    free(objs);
    free(lights);
    if (!ok)
        rt_scene_free(scene);
    return ok;
}

/* ---- seeded synthetic scenes (BASELINE configs 4 and 5) --------- */

static uint64_t splitmix64(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* uniform float in [lo, hi): 24 random bits, all arithmetic in float */
static float uniformf(uint64_t *state, float lo, float hi)
{
    float u = (float)(splitmix64(state) >> 40) * (1.0f / 16777216.0f);
    return lo + (hi - lo) * u;
}

int rt_scene_synthetic(rt_scene_t *s, uint64_t seed, uint32_t nSpheres, uint32_t nPolys,
                       uint32_t nLights, int kind)
{
    /* quads whose vertex order makes poly_plane produce the inward normal with n.v0 <= 0,
     * as the reference's F = |n.v0| requires (geometry.c:29-32) */
    static const float back[4][3] = {
        {-14.0f, 12.0f, -45.0f}, {-14.0f, 0.0f, -45.0f}, {14.0f, 0.0f, -45.0f}, {14.0f, 12.0f, -45.0f}};
    static const float left[4][3] = {
        {-14.0f, 12.0f, 0.0f}, {-14.0f, 0.0f, 0.0f}, {-14.0f, 0.0f, -45.0f}, {-14.0f, 12.0f, -45.0f}};
    static const float right[4][3] = {
        {14.0f, 12.0f, -45.0f}, {14.0f, 0.0f, -45.0f}, {14.0f, 0.0f, 0.0f}, {14.0f, 12.0f, 0.0f}};
    static const char *qname[4] = {"Floor", "Back wall", "Left wall", "Right wall"};
    const float(*quads[4])[3] = {k_floor, back, left, right};

    if (nPolys > 4 || nLights == 0 || nLights > 64 || nSpheres + nPolys == 0)
        return 0;
    scene_common(s);
    s->nLights = nLights;
    s->nObjects = nSpheres + nPolys;
    s->lights = (rt_pointlight *)calloc(s->nLights, sizeof(rt_pointlight));
    s->objects = (rt_object3d *)calloc(s->nObjects, sizeof(rt_object3d));
    if (!s->lights || !s->objects)
        return 0;

    const float zfar = kind ? -60.0f : -40.0f;
    const float rlo = kind ? 0.05f : 0.15f, rhi = kind ? 0.25f : 0.6f;
    uint64_t st = seed;
    for (uint32_t i = 0; i < nSpheres; ++i) {
        rt_object3d *o = &s->objects[i];
        float cx = uniformf(&st, -12.0f, 12.0f);
        float cy = uniformf(&st, 0.3f, 8.0f);
        float cz = uniformf(&st, zfar, -4.0f);
        float r = uniformf(&st, rlo, rhi);
        float dr = uniformf(&st, 0.2f, 1.0f);
        float dg = uniformf(&st, 0.2f, 1.0f);
        float db = uniformf(&st, 0.2f, 1.0f);
        make_sphere(o, "sphere", cx, cy, cz, r);
        float kr = 0.0f, kt = 0.0f, n = 1.0f;
        switch (i & 3u) {
        case 1: kr = 0.75f; break;                         /* mirror */
        case 2: kr = 0.01f; kt = 0.85f; n = 0.95f; break;  /* glass  */
        case 3: kr = 0.3f;  kt = 0.4f;  n = 1.1f;  break;  /* mixed  */
        default: break;                                    /* diffuse */
        }
        set_material(&o->material, dr, dg, db, 0.1f, 0.5f, 0.5f, 20.0f, kr, kt, n);
    }
    for (uint32_t q = 0; q < nPolys; ++q) {
        rt_object3d *o = &s->objects[nSpheres + q];
        if (!make_quad(o, qname[q], quads[q]))
            return 0;
        set_material(&o->material, 0.0f, 1.0f, 0.0f, 0.1f, 0.7f, 0.2f, 2.0f, 0.0f, 0.0f, 1.0f);
    }
    if (nLights == 1) {
        set_vec(&s->lights[0].position, 1.0f, 8.0f, 1.0f, 1.0f);
        set_color(&s->lights[0].color, 1.0f, 1.0f, 1.0f);
        s->lights[0].range = 150.0f;
    } else {
        /* ring of radius 6 at y = 10 around (0,10,-20); intensity 1/nLights each */
        static const float cs[8][2] = {{1.0f, 0.0f},  {0.70710678f, 0.70710678f},
                                       {0.0f, 1.0f},  {-0.70710678f, 0.70710678f},
                                       {-1.0f, 0.0f}, {-0.70710678f, -0.70710678f},
                                       {0.0f, -1.0f}, {0.70710678f, -0.70710678f}};
        float inten = 1.0f / (float)nLights;
        for (uint32_t l = 0; l < nLights; ++l) {
            const float *c = cs[l & 7u];
            float rad = 6.0f + 0.5f * (float)(l >> 3);
            set_vec(&s->lights[l].position, rad * c[0], 10.0f, -20.0f + rad * c[1], 1.0f);
            set_color(&s->lights[l].color, inten, inten, inten);
            s->lights[l].range = 150.0f;
        }
    }
    return 1;
}

/* ---- write_image (pam_output/output.c:30-71): raw P6, maxval 255 --------------------- */

int rt_write_image(const char *filename, const unsigned int *buffer, unsigned int width,
                   unsigned int height)
{
    FILE *fp = fopen(filename, "wb");
    if (!fp) {
        printf("Could not open file {%s} for writing.\n", filename);
        return 0;
    }
    fprintf(fp, "P6\n%u %u\n255\n", width, height);
    size_t n = (size_t)width * height;
    unsigned char *rgb = (unsigned char *)malloc(n ? n * 3 : 1);
    if (!rgb) {
        fclose(fp);
        return 0;
    }
    for (size_t i = 0; i < n; ++i) {
        unsigned int p = buffer[i];
        rgb[3 * i + 0] = (unsigned char)((p >> 16) & 0xFFu);
        rgb[3 * i + 1] = (unsigned char)((p >> 8) & 0xFFu);
        rgb[3 * i + 2] = (unsigned char)(p & 0xFFu);
    }
    int ok = fwrite(rgb, 3, n, fp) == n;
    free(rgb);
    fclose(fp);
    return ok;
}

/* output.h:15-16 */
int write_image(const char *filename, unsigned int *buffer, unsigned int width, unsigned int height)
{
    return rt_write_image(filename, buffer, width, height);
}
