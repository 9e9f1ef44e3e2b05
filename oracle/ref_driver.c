/*
 * Headless driver for the UNMODIFIED reference renderer (test infrastructure).
 *
 * This file is the only thing in the reference build that is ours: it is
 * compiled together with the reference's own sources where they lie under
 * /root/reference (see oracle/Makefile) and calls the reference's public API
 * -- scene factories (renderer/scenes/{blinn,pbr}_scenes.h), camera_create
 * (renderer/core/camera.h), test_build_perframe / test_draw_scene
 * (renderer/tests/test_helper.h:27-28) -- to render one frame and dump the
 * colour and depth buffers.  The dumps are the golden vectors that pin the
 * CPU restatement in oracle/ and, through it, the HIP path.
 *
 * Never runs on the GPU box (the reference does not travel).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "core/api.h"
#include "scenes/blinn_scenes.h"
#include "scenes/pbr_scenes.h"
#include "scenes/scene_helper.h"
#include "shaders/cache_helper.h"
#include "tests/test_helper.h"

#ifdef RZ_DROP_IN
/* Built on top of the C ABI instead of the reference's core/graphics.c and
 * core/texture.c (oracle/Makefile, target refhost): draws are recorded, so a
 * host that reads color_buffer / depth_buffer itself synchronises first. */
void framebuffer_sync(framebuffer_t *framebuffer);
#endif

typedef scene_t *factory_t(void);
typedef struct {const char *name; factory_t *make;} entry_t;

static entry_t g_table[] = {
    {"blinn:azura", blinn_azura_scene}, {"blinn:centaur", blinn_centaur_scene},
    {"blinn:craftsman", blinn_craftsman_scene}, {"blinn:elfgirl", blinn_elfgirl_scene},
    {"blinn:kgirl", blinn_kgirl_scene}, {"blinn:lighthouse", blinn_lighthouse_scene},
    {"blinn:mccree", blinn_mccree_scene}, {"blinn:nier2b", blinn_nier2b_scene},
    {"blinn:phoenix", blinn_phoenix_scene}, {"blinn:vivi", blinn_vivi_scene},
    {"blinn:whip", blinn_whip_scene}, {"blinn:witch", blinn_witch_scene},
    {"pbr:assassin", pbr_assassin_scene}, {"pbr:buster", pbr_buster_scene},
    {"pbr:crab", pbr_crab_scene}, {"pbr:dieselpunk", pbr_dieselpunk_scene},
    {"pbr:drone", pbr_drone_scene}, {"pbr:helmet", pbr_helmet_scene},
    {"pbr:horse", pbr_horse_scene}, {"pbr:junkrat", pbr_junkrat_scene},
    {"pbr:ornitier", pbr_ornitier_scene}, {"pbr:ponycar", pbr_ponycar_scene},
    {"pbr:robot", pbr_robot_scene}, {"pbr:sphere", pbr_sphere_scene},
    {"pbr:spheres", pbr_spheres_scene}, {NULL, NULL},
};

/* root transforms of the scenes for which this repo ships derived .scn files
 * (same constants and product order as the reference's factories) */
static int derived_root(const char *like, mat4_t *root) {
    if (strcmp(like, "blinn:azura") == 0) {
        mat4_t t = mat4_translate(-6.073f, -1.278f, 0.280f);
        mat4_t s = mat4_scale(0.378f, 0.378f, 0.378f);
        *root = mat4_mul_mat4(s, t);
    } else if (strcmp(like, "blinn:kgirl") == 0) {
        mat4_t t = mat4_translate(0, -4.937f, -96.547f);
        mat4_t rx = mat4_rotate_x(TO_RADIANS(-90));
        mat4_t ry = mat4_rotate_y(TO_RADIANS(90));
        mat4_t r = mat4_mul_mat4(ry, rx);
        mat4_t s = mat4_scale(0.005f, 0.005f, 0.005f);
        *root = mat4_mul_mat4(s, mat4_mul_mat4(r, t));
    } else if (strcmp(like, "pbr:dieselpunk") == 0) {
        mat4_t t = mat4_translate(1.036f, -114.817f, 27.682f);
        mat4_t r = mat4_rotate_y(TO_RADIANS(-90));
        mat4_t s = mat4_scale(0.0012f, 0.0012f, 0.0012f);
        *root = mat4_mul_mat4(s, mat4_mul_mat4(r, t));
    } else if (strcmp(like, "pbr:helmet") == 0) {
        mat4_t t = mat4_translate(0.002f, 0.187f, 0);
        mat4_t r = mat4_rotate_x(TO_RADIANS(90));
        mat4_t s = mat4_scale(0.5f, 0.5f, 0.5f);
        *root = mat4_mul_mat4(s, mat4_mul_mat4(r, t));
    } else if (strcmp(like, "identity") == 0) {
        *root = mat4_identity();
    } else {
        return 0;
    }
    return 1;
}

/* 2-D overlays drawn over the finished frame through the reference's draw2d API
 * (core/draw2d.h:8-15), the way renderer/tests/test_pbr.c:71-106 draws the layer
 * inspector's edges and labels: --point r g b a x y | --line r g b a x0 y0 x1 y1 |
 * --outline r g b a x0 y0 x1 y1 x2 y2 | --label file.tga x y, in command-line order */
typedef struct {int kind; float v[10]; const char *file;} overlay_t;
#define MAX_OVERLAYS 64
static overlay_t g_overlays[MAX_OVERLAYS];
static int g_num_overlays = 0;

static int parse_overlay(int argc, char **argv, int *i) {
    static const struct {const char *flag; int kind, floats;} forms[] = {
        {"--point", 0, 6}, {"--line", 1, 8}, {"--outline", 2, 10}, {"--label", 3, 2},
    };
    int f, k;
    for (f = 0; f < 4; f++) {
        int wanted = forms[f].floats + (forms[f].kind == 3 ? 1 : 0);
        overlay_t *o;
        if (strcmp(argv[*i], forms[f].flag) != 0) continue;
        if (*i + wanted >= argc || g_num_overlays == MAX_OVERLAYS) return -1;
        o = &g_overlays[g_num_overlays++];
        o->kind = forms[f].kind;
        o->file = forms[f].kind == 3 ? argv[++*i] : NULL;
        for (k = 0; k < forms[f].floats; k++) o->v[k] = (float)atof(argv[++*i]);
        return 1;
    }
    return 0;
}

static void draw_overlays(framebuffer_t *fb) {
    int i;
    for (i = 0; i < g_num_overlays; i++) {
        const overlay_t *o = &g_overlays[i];
        const float *v = o->v;
        vec4_t color = vec4_new(v[0], v[1], v[2], v[3]);
        if (o->kind == 0) {
            draw2d_draw_point(fb, color, vec2_new(v[4], v[5]));
        } else if (o->kind == 1) {
            draw2d_draw_line(fb, color, vec2_new(v[4], v[5]), vec2_new(v[6], v[7]));
        } else if (o->kind == 2) {
            draw2d_draw_triangle(fb, color, vec2_new(v[4], v[5]), vec2_new(v[6], v[7]), vec2_new(v[8], v[9]));
        } else {
            texture_t *label = cache_acquire_texture(o->file, USAGE_LDR_COLOR);      /* test_pbr.c:115-117 */
            draw2d_draw_texture(fb, label, vec2_new(v[0], v[1]));
            cache_release_texture(label);
        }
    }
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* --poses file.ani joints count out: the reference's own pose evaluation (core/skeleton.h:13-15) at the times
 * i * 0.137f, i < count; per time `joints` 4x4 joint matrices, then `joints` 3x3 normal matrices, as floats */
static int dump_poses(const char *assets, const char *ani, int joints, int count, const char *out) {
    skeleton_t *skeleton;
    FILE *f;
    int i;
    if (chdir(assets) != 0) { perror("chdir assets"); return 2; }
    skeleton = skeleton_load(ani);
    f = fopen(out, "wb");
    if (!f) { perror(out); return 2; }
    for (i = 0; i < count; i++) {
        skeleton_update_joints(skeleton, (float)i * 0.137f);
        fwrite(skeleton_get_joint_matrices(skeleton), sizeof(mat4_t), (size_t)joints, f);
        fwrite(skeleton_get_normal_matrices(skeleton), sizeof(mat3_t), (size_t)joints, f);
    }
    fclose(f);
    skeleton_release(skeleton);
    return 0;
}

/* --texture file usage out: the reference's own image loading and usage conversion (core/texture.c:32-128,
 * core/image.c) -- width, height, then the float RGBA texels of texture->buffer */
static int dump_texture(const char *assets, const char *file, int usage, const char *out) {
    texture_t *texture;
    FILE *f;
    int size[2];
    if (chdir(assets) != 0) { perror("chdir assets"); return 2; }
    texture = texture_from_file(file, (usage_t)usage);
    f = fopen(out, "wb");
    if (!f) { perror(out); return 2; }
    size[0] = texture->width; size[1] = texture->height;
    fwrite(size, sizeof(int), 2, f);
    fwrite(texture->buffer, sizeof(vec4_t), (size_t)texture->width * texture->height, f);
    fclose(f);
    texture_release(texture);
    return 0;
}

/* --mesh file out: the reference's own OBJ loading (core/mesh.c: de-indexed vertex stream with tangents, joints and
 * weights) -- the number of faces, then 3 x faces vertex_t of 80 bytes */
static int dump_mesh(const char *assets, const char *file, const char *out) {
    mesh_t *mesh;
    FILE *f;
    int faces;
    if (chdir(assets) != 0) { perror("chdir assets"); return 2; }
    mesh = mesh_load(file);
    f = fopen(out, "wb");
    if (!f) { perror(out); return 2; }
    faces = mesh_get_num_faces(mesh);
    fwrite(&faces, sizeof(int), 1, f);
    fwrite(mesh_get_vertices(mesh), sizeof(vertex_t), (size_t)faces * 3, f);
    fclose(f);
    mesh_release(mesh);
    return 0;
}

int main(int argc, char **argv) {
    const char *assets = "/root/reference/assets";
    const char *scene_name = NULL, *scn_file = NULL, *out = NULL;
    int width = 320, height = 180, timed = 0, layer = -1, i;
    vec3_t eye = {0, 0, 1.5f}, target = {0, 0, 0};
    float frame_time = 0;
    float theta = TO_RADIANS(45), phi = TO_RADIANS(45);
    scene_t *scene = NULL;
    framebuffer_t *fb;
    camera_t *camera;
    context_t context;
    perframe_t perframe;

    for (i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], "--assets")) assets = argv[i + 1];
    for (i = 1; i + 2 < argc; i++)
        if (!strcmp(argv[i], "--mesh")) return dump_mesh(assets, argv[i + 1], argv[i + 2]);
    for (i = 1; i + 3 < argc; i++)
        if (!strcmp(argv[i], "--texture")) return dump_texture(assets, argv[i + 1], atoi(argv[i + 2]), argv[i + 3]);
    for (i = 1; i + 4 < argc; i++)
        if (!strcmp(argv[i], "--poses")) return dump_poses(assets, argv[i + 1], atoi(argv[i + 2]), atoi(argv[i + 3]), argv[i + 4]);
    for (i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--scene") && i + 1 < argc) scene_name = argv[++i];
        else if (!strcmp(argv[i], "--scn") && i + 1 < argc) scn_file = argv[++i];
        else if (!strcmp(argv[i], "--assets") && i + 1 < argc) assets = argv[++i];
        else if (!strcmp(argv[i], "--size") && i + 2 < argc) { width = atoi(argv[++i]); height = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--eye") && i + 3 < argc) { eye.x = (float)atof(argv[++i]); eye.y = (float)atof(argv[++i]); eye.z = (float)atof(argv[++i]); }
        else if (!strcmp(argv[i], "--target") && i + 3 < argc) { target.x = (float)atof(argv[++i]); target.y = (float)atof(argv[++i]); target.z = (float)atof(argv[++i]); }
        else if (!strcmp(argv[i], "--time") && i + 1 < argc) frame_time = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--layer") && i + 1 < argc) layer = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--bench") && i + 1 < argc) timed = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else if (parse_overlay(argc, argv, &i) > 0) continue;
        else { fprintf(stderr, "bad argument: %s\n", argv[i]); return 2; }
    }
    if (scene_name == NULL) { fprintf(stderr, "--scene required\n"); return 2; }
    if (chdir(assets) != 0) { perror("chdir assets"); return 2; }

    if (scn_file) {                     /* derived .scn from this repo */
        mat4_t root;
        if (!derived_root(scene_name, &root)) { fprintf(stderr, "no root for %s\n", scene_name); return 2; }
        scene = scene_from_file(scn_file, root);
    } else {
        for (i = 0; g_table[i].name; i++) {
            if (!strcmp(g_table[i].name, scene_name)) scene = g_table[i].make();
        }
    }
    if (scene == NULL) { fprintf(stderr, "unknown scene %s\n", scene_name); return 2; }

    fb = framebuffer_create(width, height);
    camera = camera_create(eye, target, (float)width / (float)height);
    memset(&context, 0, sizeof(context));
    context.framebuffer = fb;
    context.camera = camera;
    /* same expression as get_light_dir, renderer/tests/test_helper.c:162-169 */
    context.light_dir = vec3_new(-((float)sin(phi) * (float)sin(theta)),
                                 -((float)cos(phi)),
                                 -((float)sin(phi) * (float)cos(theta)));
    context.frame_time = frame_time;
    context.delta_time = 0;

    perframe = test_build_perframe(scene, &context);
    perframe.layer_view = layer;
    test_draw_scene(scene, fb, &perframe);

    if (timed > 0) {
        double *samples = (double*)malloc(sizeof(double) * timed);
        for (i = 0; i < timed; i++) {
            double t0 = now_ms();
            perframe = test_build_perframe(scene, &context);
            perframe.layer_view = layer;
            test_draw_scene(scene, fb, &perframe);
            samples[i] = now_ms() - t0;
        }
        qsort(samples, timed, sizeof(double), cmp_double);
        printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"frames\": %d, \"median_ms\": %.4f, \"min_ms\": %.4f}\n",
               scene_name, width, height, timed, samples[timed / 2], samples[0]);
        free(samples);
    }

    /* in the drop-in build the overlays are the library's too (oracle/Makefile leaves core/draw2d.c
     * out): they follow the recorded draws without a synchronisation in between */
    draw_overlays(fb);
#ifdef RZ_DROP_IN
    framebuffer_sync(fb);
#endif
    if (out) {
        FILE *f = fopen(out, "wb");
        int hdr[4];
        if (!f) { perror(out); return 2; }
        hdr[0] = 0x31475a52; hdr[1] = width; hdr[2] = height; hdr[3] = 0;   /* "RZG1" */
        fwrite(hdr, sizeof(int), 4, f);
        fwrite(fb->color_buffer, 1, (size_t)width * height * 4, f);
        fwrite(fb->depth_buffer, sizeof(float), (size_t)width * height, f);
        fclose(f);
    }
    return 0;
}
