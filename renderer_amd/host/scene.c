/*
 * Scene layer: models of the blinn / pbr / skybox families, .scn files, the
 * scene catalogue and the per-frame driver.  This is the CALLER of the hot
 * path.  It follows the behaviour of the reference's
 *   renderer/shaders/{blinn,pbr,skybox}_shader.c  "high-level api",
 *   renderer/scenes/scene_helper.c + {blinn,pbr}_scenes.c,
 *   renderer/core/scene.c, renderer/core/camera.c,
 *   renderer/tests/test_helper.c:382-507 (frame driver)
 * and talks to the pipeline only through the functions declared in
 * include/mi355r.h -- the same functions the reference's scene code calls,
 * plus graphics_draw_mesh for whole meshes.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "scene.h"
#include "hmath.h"

/* shader pairs: identity tags for program_create (the GPU library never calls
 * them; the CPU oracle build supplies real ones) */
rz_vec4 blinn_vertex_shader(void *attribs, void *varyings, void *uniforms);
rz_vec4 blinn_fragment_shader(void *varyings, void *uniforms, int *discard, int backface);
rz_vec4 pbr_vertex_shader(void *attribs, void *varyings, void *uniforms);
rz_vec4 pbr_fragment_shader(void *varyings, void *uniforms, int *discard, int backface);
rz_vec4 skybox_vertex_shader(void *attribs, void *varyings, void *uniforms);
rz_vec4 skybox_fragment_shader(void *varyings, void *uniforms, int *discard, int backface);

static void fail(const char *what, const char *detail) {
    fprintf(stderr, "mi355r scene: %s %s\n", what, detail ? detail : "");
    abort();
}

/* ======================= per-frame state ================================ */

typedef struct {                /* renderer/core/scene.h:10-23 */
    float frame_time;
    rz_vec3 light_dir, camera_pos;
    rz_mat4 light_view, light_proj, camera_view, camera_proj;
    float ambient, punctual;
    rz_texture *shadow_map;
    int layer_view;
} frame_t;

enum { MODEL_BLINN, MODEL_PBR, MODEL_SKYBOX };

typedef struct {                /* renderer/core/scene.h:25-40 */
    int kind;
    rzh_mesh *mesh;
    rz_program *program;
    rz_mat4 transform;
    rzh_skeleton *skeleton;
    int attached;
    int opaque;
    float distance;
    /* this frame's pose is evaluated by the pipeline (graphics_draw_mesh_posed) at this time */
    int posed;
    float pose_time;
} model_t;

#define RZS_SHADOW_RING 128          /* at most; a scene takes as many as fit a budget, at least 32 (scene_create) */

struct rzs_scene {              /* renderer/core/scene.h:42-52 */
    rz_vec4 background;
    model_t *skybox;
    model_t **models;
    int num_models;
    float ambient, punctual;
    /* Shadow buffer and map, RZS_SHADOW_RING of each, taken in turn by successive frames: a frame
     * then never overwrites what the frames before it still sample, so the GPU library can keep
     * several frames (shadow pass, copy, main pass each) in one flush (scene.c:10-27 has one of
     * each, and with one the library has to finish every frame before the next starts). */
    rz_framebuffer *shadow_buffer, *shadow_buffers[RZS_SHADOW_RING];
    rz_texture *shadow_map, *shadow_maps[RZS_SHADOW_RING];
    int shadow_turn, shadow_ring;
};

/* ======================= models ========================================= */

/* blinn_shader.c:264-304 and pbr_shader.c:506-547 share this prologue: the
 * skeleton is advanced to the frame time; a node-attached model folds its
 * node matrix into the model matrix and is NOT skinned. */
static int g_per_triangle = 0;
static int g_device_poses = 1;

void rzs_set_device_poses(int on) { g_device_poses = on; }

static void model_frame_matrices(model_t *m, const frame_t *f, rz_mat4 *model_matrix,
                                 rz_mat3 *normal_matrix, rz_mat4 **joints, rz_mat3 **joint_normals) {
    *model_matrix = m->transform;
    *joints = NULL;
    *joint_normals = NULL;
    m->posed = 0;
    if (m->skeleton && g_device_poses && !g_per_triangle && rzh_skeleton_pose_source(m->skeleton)) {
        /* the pipeline evaluates the pose and, for a node-attached model, folds the node's matrix into
         * the model matrix and derives the normal matrix (graphics_draw_mesh_posed) */
        m->posed = 1;
        m->pose_time = f->frame_time;
    } else if (m->skeleton) {
        rzh_skeleton_update(m->skeleton, f->frame_time);
        *joints = rzh_skeleton_joint_matrices(m->skeleton);
        *joint_normals = rzh_skeleton_normal_matrices(m->skeleton);
        if (m->attached >= 0) {
            *model_matrix = hm_m4_mul(*model_matrix, (*joints)[m->attached]);
            *joints = NULL;
            *joint_normals = NULL;
        }
    }
    *normal_matrix = hm_m3_inverse_transpose(hm_m3_from_m4(*model_matrix));
}

static void model_update(model_t *m, const frame_t *f) {
    if (m->kind == MODEL_SKYBOX) {                      /* skybox_shader.c:33-45 */
        rz_skybox_uniforms *u = (rz_skybox_uniforms*)program_get_uniforms(m->program);
        rz_mat4 view = f->camera_view;
        view.m[0][3] = 0; view.m[1][3] = 0; view.m[2][3] = 0;
        u->vp_matrix = hm_m4_mul(f->camera_proj, view);
    } else if (m->kind == MODEL_BLINN) {
        rz_blinn_uniforms *u = (rz_blinn_uniforms*)program_get_uniforms(m->program);
        model_frame_matrices(m, f, &u->model_matrix, &u->normal_matrix,
                             &u->joint_matrices, &u->joint_n_matrices);
        u->light_dir = f->light_dir;
        u->camera_pos = f->camera_pos;
        u->light_vp_matrix = hm_m4_mul(f->light_proj, f->light_view);
        u->camera_vp_matrix = hm_m4_mul(f->camera_proj, f->camera_view);
        u->ambient_intensity = hm_clamp(f->ambient, 0, 5);
        u->punctual_intensity = hm_clamp(f->punctual, 0, 5);
        u->shadow_map = f->shadow_map;
    } else {
        rz_pbr_uniforms *u = (rz_pbr_uniforms*)program_get_uniforms(m->program);
        model_frame_matrices(m, f, &u->model_matrix, &u->normal_matrix,
                             &u->joint_matrices, &u->joint_n_matrices);
        u->light_dir = f->light_dir;
        u->camera_pos = f->camera_pos;
        u->light_vp_matrix = hm_m4_mul(f->light_proj, f->light_view);
        u->camera_vp_matrix = hm_m4_mul(f->camera_proj, f->camera_view);
        u->ambient_intensity = hm_clamp(f->ambient, 0, 5);
        u->punctual_intensity = hm_clamp(f->punctual, 0, 5);
        u->shadow_map = f->shadow_map;
        u->layer_view = f->layer_view;
    }
}

/* blinn_shader.c:306-342, pbr_shader.c:549-578, skybox_shader.c:47-66: one
 * triangle per face in mesh order; the skybox takes no part in shadow passes */
void rzs_set_per_triangle(int on) { g_per_triangle = on; }

static void model_draw(model_t *m, rz_framebuffer *fb, int shadow_pass) {
    if (m->kind == MODEL_SKYBOX) {
        if (shadow_pass) return;
    } else if (m->kind == MODEL_BLINN) {
        ((rz_blinn_uniforms*)program_get_uniforms(m->program))->shadow_pass = shadow_pass;
    } else {
        ((rz_pbr_uniforms*)program_get_uniforms(m->program))->shadow_pass = shadow_pass;
    }
    if (m->posed) {
        graphics_draw_mesh_posed(fb, m->program, m->mesh->vertices, m->mesh->num_faces,
                                 rzh_skeleton_pose_source(m->skeleton), m->pose_time, m->attached);
    } else if (!g_per_triangle) {
        graphics_draw_mesh(fb, m->program, m->mesh->vertices, m->mesh->num_faces);
    } else {
        /* the reference's own loop: three attribute blocks filled per face,
         * then one graphics_draw_triangle call */
        const rz_vertex *verts = m->mesh->vertices;
        int f, j;
        for (f = 0; f < m->mesh->num_faces; f++) {
            for (j = 0; j < 3; j++) {
                const rz_vertex *v = &verts[f * 3 + j];
                void *attribs = program_get_attribs(m->program, j);
                if (m->kind == MODEL_SKYBOX) {
                    ((rz_skybox_attribs*)attribs)->position = v->position;
                } else if (m->kind == MODEL_BLINN) {
                    rz_blinn_attribs *a = (rz_blinn_attribs*)attribs;
                    a->position = v->position; a->texcoord = v->texcoord; a->normal = v->normal;
                    a->joint = v->joint; a->weight = v->weight;
                } else {
                    rz_pbr_attribs *a = (rz_pbr_attribs*)attribs;
                    a->position = v->position; a->texcoord = v->texcoord; a->normal = v->normal;
                    a->tangent = v->tangent; a->joint = v->joint; a->weight = v->weight;
                }
            }
            graphics_draw_triangle(fb, m->program);
        }
    }
}

static void model_free(model_t *m) {
    if (!m) return;
    if (m->kind == MODEL_SKYBOX) {
        rz_skybox_uniforms *u = (rz_skybox_uniforms*)program_get_uniforms(m->program);
        rzh_release_skybox(u->skybox);
    } else if (m->kind == MODEL_BLINN) {
        rz_blinn_uniforms *u = (rz_blinn_uniforms*)program_get_uniforms(m->program);
        rzh_release_texture(u->diffuse_map);
        rzh_release_texture(u->specular_map);
        rzh_release_texture(u->emission_map);
    } else {
        rz_pbr_uniforms *u = (rz_pbr_uniforms*)program_get_uniforms(m->program);
        rzh_release_texture(u->basecolor_map); rzh_release_texture(u->metalness_map);
        rzh_release_texture(u->roughness_map); rzh_release_texture(u->diffuse_map);
        rzh_release_texture(u->specular_map); rzh_release_texture(u->glossiness_map);
        rzh_release_texture(u->normal_map); rzh_release_texture(u->occlusion_map);
        rzh_release_texture(u->emission_map);
        rzh_release_ibldata(u->ibldata);
    }
    program_release(m->program);
    rzh_release_skeleton(m->skeleton);
    rzh_release_mesh(m->mesh);
    free(m);
}

static model_t *model_new(int kind, const char *mesh, rz_mat4 transform, const char *skeleton,
                          int attached, int double_sided, int enable_blend) {
    model_t *m = (model_t*)calloc(1, sizeof *m);
    m->kind = kind;
    if (kind == MODEL_BLINN) {
        m->program = program_create(blinn_vertex_shader, blinn_fragment_shader,
                                    (int)sizeof(rz_blinn_attribs), 44, (int)sizeof(rz_blinn_uniforms),
                                    double_sided, enable_blend);
    } else if (kind == MODEL_PBR) {
        m->program = program_create(pbr_vertex_shader, pbr_fragment_shader,
                                    (int)sizeof(rz_pbr_attribs), 84, (int)sizeof(rz_pbr_uniforms),
                                    double_sided, enable_blend);
    } else {
        m->program = program_create(skybox_vertex_shader, skybox_fragment_shader,
                                    (int)sizeof(rz_skybox_attribs), 12, (int)sizeof(rz_skybox_uniforms),
                                    1, 0);
    }
    m->mesh = rzh_acquire_mesh(mesh);
    m->transform = transform;
    m->skeleton = rzh_acquire_skeleton(skeleton);
    m->attached = attached;
    m->opaque = !enable_blend;
    m->distance = 0;
    return m;
}

/* ======================= .scn files ===================================== */

/* A .scn file is a flat sequence of whitespace-separated tokens
 * ("key:" value...), always in the same order (scene_helper.c:98-361). */
typedef struct { FILE *f; const char *name; } reader_t;

static void rd_word(reader_t *r, char *out) {
    if (fscanf(r->f, " %255s", out) != 1) fail("unexpected end of", r->name);
}
static void rd_expect(reader_t *r, const char *label) {
    char w[RZ_LINE];
    rd_word(r, w);
    if (strcmp(w, label) != 0) {
        fprintf(stderr, "mi355r scene: %s: expected '%s', found '%s'\n", r->name, label, w);
        abort();
    }
}
static float rd_float(reader_t *r) {
    char w[RZ_LINE];
    rd_word(r, w);
    return strtof(w, NULL);
}
static int rd_int(reader_t *r) {            /* also accepts "3:" */
    char w[RZ_LINE];
    rd_word(r, w);
    return atoi(w);
}
static void rd_key_word(reader_t *r, const char *key, char *out) { rd_expect(r, key); rd_word(r, out); }
static float rd_key_float(reader_t *r, const char *key) { rd_expect(r, key); return rd_float(r); }
static int rd_key_int(reader_t *r, const char *key) { rd_expect(r, key); return rd_int(r); }
static const char *path_or_null(const char *s) { return strcmp(s, "null") == 0 ? NULL : s; }
static int on_off(const char *s, const char *file) {
    if (strcmp(s, "on") == 0) return 1;
    if (strcmp(s, "off") != 0) fail("expected on/off in", file);
    return 0;
}

typedef struct {
    /* blinn */
    rz_vec4 basecolor; float shininess;
    /* pbrm */
    float metalness, roughness;
    /* pbrs */
    rz_vec3 specular; float glossiness;
    char map[6][RZ_LINE];       /* family-specific order, see read_material */
    int double_sided, enable_blend;
    float alpha_cutoff;
} material_t;

typedef struct {
    char mesh[RZ_LINE], skeleton[RZ_LINE];
    int attached, material, transform;
} modeldesc_t;

enum { SCN_BLINN, SCN_PBRM, SCN_PBRS };

static void read_material(reader_t *r, int type, int index, material_t *m) {
    char w[RZ_LINE];
    int i, nmaps;
    memset(m, 0, sizeof *m);
    rd_expect(r, "material");
    if (rd_int(r) != index) fail("material index out of order in", r->name);
    if (type == SCN_BLINN) {
        static const char *keys[3] = {"diffuse_map:", "specular_map:", "emission_map:"};
        rd_expect(r, "basecolor:");
        m->basecolor.x = rd_float(r); m->basecolor.y = rd_float(r);
        m->basecolor.z = rd_float(r); m->basecolor.w = rd_float(r);
        m->shininess = rd_key_float(r, "shininess:");
        nmaps = 3;
        for (i = 0; i < nmaps; i++) rd_key_word(r, keys[i], m->map[i]);
    } else if (type == SCN_PBRM) {
        static const char *keys[6] = {"basecolor_map:", "metalness_map:", "roughness_map:",
                                      "normal_map:", "occlusion_map:", "emission_map:"};
        rd_expect(r, "basecolor_factor:");
        m->basecolor.x = rd_float(r); m->basecolor.y = rd_float(r);
        m->basecolor.z = rd_float(r); m->basecolor.w = rd_float(r);
        m->metalness = rd_key_float(r, "metalness_factor:");
        m->roughness = rd_key_float(r, "roughness_factor:");
        nmaps = 6;
        for (i = 0; i < nmaps; i++) rd_key_word(r, keys[i], m->map[i]);
    } else {
        static const char *keys[6] = {"diffuse_map:", "specular_map:", "glossiness_map:",
                                      "normal_map:", "occlusion_map:", "emission_map:"};
        rd_expect(r, "diffuse_factor:");
        m->basecolor.x = rd_float(r); m->basecolor.y = rd_float(r);
        m->basecolor.z = rd_float(r); m->basecolor.w = rd_float(r);
        rd_expect(r, "specular_factor:");
        m->specular.x = rd_float(r); m->specular.y = rd_float(r); m->specular.z = rd_float(r);
        m->glossiness = rd_key_float(r, "glossiness_factor:");
        nmaps = 6;
        for (i = 0; i < nmaps; i++) rd_key_word(r, keys[i], m->map[i]);
    }
    rd_key_word(r, "double_sided:", w); m->double_sided = on_off(w, r->name);
    rd_key_word(r, "enable_blend:", w); m->enable_blend = on_off(w, r->name);
    m->alpha_cutoff = rd_key_float(r, "alpha_cutoff:");
}

/* blinn_shader.c:361-407 */
static model_t *make_blinn(const modeldesc_t *d, rz_mat4 transform, const material_t *mat) {
    model_t *m = model_new(MODEL_BLINN, path_or_null(d->mesh), transform, path_or_null(d->skeleton),
                           d->attached, mat->double_sided, mat->enable_blend);
    rz_blinn_uniforms *u = (rz_blinn_uniforms*)program_get_uniforms(m->program);
    u->basecolor = mat->basecolor;
    u->shininess = mat->shininess;
    u->diffuse_map = rzh_acquire_texture(path_or_null(mat->map[0]), RZ_USAGE_LDR_COLOR);
    u->specular_map = rzh_acquire_texture(path_or_null(mat->map[1]), RZ_USAGE_LDR_COLOR);
    u->emission_map = rzh_acquire_texture(path_or_null(mat->map[2]), RZ_USAGE_LDR_COLOR);
    u->alpha_cutoff = mat->alpha_cutoff;
    return m;
}

/* pbr_shader.c:636-688: colour maps are linearised at load, data maps are not */
static model_t *make_pbr(const modeldesc_t *d, rz_mat4 transform, const material_t *mat,
                         int specular_workflow, const char *env) {
    model_t *m = model_new(MODEL_PBR, path_or_null(d->mesh), transform, path_or_null(d->skeleton),
                           d->attached, mat->double_sided, mat->enable_blend);
    rz_pbr_uniforms *u = (rz_pbr_uniforms*)program_get_uniforms(m->program);
    if (!specular_workflow) {
        u->basecolor_factor = mat->basecolor;
        u->metalness_factor = hm_saturate(mat->metalness);
        u->roughness_factor = hm_saturate(mat->roughness);
        u->basecolor_map = rzh_acquire_texture(path_or_null(mat->map[0]), RZ_USAGE_HDR_COLOR);
        u->metalness_map = rzh_acquire_texture(path_or_null(mat->map[1]), RZ_USAGE_LDR_DATA);
        u->roughness_map = rzh_acquire_texture(path_or_null(mat->map[2]), RZ_USAGE_LDR_DATA);
    } else {
        u->diffuse_factor = mat->basecolor;
        u->specular_factor = mat->specular;
        u->glossiness_factor = hm_saturate(mat->glossiness);
        u->diffuse_map = rzh_acquire_texture(path_or_null(mat->map[0]), RZ_USAGE_HDR_COLOR);
        u->specular_map = rzh_acquire_texture(path_or_null(mat->map[1]), RZ_USAGE_HDR_COLOR);
        u->glossiness_map = rzh_acquire_texture(path_or_null(mat->map[2]), RZ_USAGE_LDR_DATA);
    }
    u->normal_map = rzh_acquire_texture(path_or_null(mat->map[3]), RZ_USAGE_LDR_DATA);
    u->occlusion_map = rzh_acquire_texture(path_or_null(mat->map[4]), RZ_USAGE_LDR_DATA);
    u->emission_map = rzh_acquire_texture(path_or_null(mat->map[5]), RZ_USAGE_HDR_COLOR);
    u->ibldata = rzh_acquire_ibldata(env);
    u->alpha_cutoff = mat->alpha_cutoff;
    u->workflow = specular_workflow;
    u->layer_view = -1;
    return m;
}

/* skybox_shader.c:77-105 */
static model_t *make_skybox(const char *env, int blur_level) {
    model_t *m = model_new(MODEL_SKYBOX, "common/box.obj", hm_m4_identity(), NULL, -1, 1, 0);
    rz_skybox_uniforms *u = (rz_skybox_uniforms*)program_get_uniforms(m->program);
    u->skybox = rzh_acquire_skybox(env, blur_level);
    m->opaque = 1;
    return m;
}

/* scene_helper.c:584-640 (parse) + 363-404 (skybox / shadow set-up) */
static rzs_scene *scene_from_scn(const char *filename, rz_mat4 root) {
    reader_t r;
    char type_word[RZ_LINE], env[RZ_LINE], skybox[RZ_LINE], shadow[RZ_LINE];
    rz_vec3 background;
    float ambient, punctual;
    int type, num_materials, num_transforms, num_models, i, k;
    material_t *materials;
    rz_mat4 *transforms;
    rzs_scene *scene;
    int shadow_w = -1, shadow_h = -1;

    r.f = rzh_open_asset(filename);
    r.name = filename;
    rd_key_word(&r, "type:", type_word);
    if (strcmp(type_word, "blinn") == 0) type = SCN_BLINN;
    else if (strcmp(type_word, "pbrm") == 0) type = SCN_PBRM;
    else if (strcmp(type_word, "pbrs") == 0) type = SCN_PBRS;
    else { fail("unknown scene type in", filename); return NULL; }

    rd_expect(&r, "lighting:");
    rd_expect(&r, "background:");
    background.x = rd_float(&r); background.y = rd_float(&r); background.z = rd_float(&r);
    rd_key_word(&r, "environment:", env);
    rd_key_word(&r, "skybox:", skybox);
    rd_key_word(&r, "shadow:", shadow);
    ambient = rd_key_float(&r, "ambient:");
    punctual = rd_key_float(&r, "punctual:");

    num_materials = rd_key_int(&r, "materials");
    materials = (material_t*)calloc(num_materials > 0 ? num_materials : 1, sizeof *materials);
    for (i = 0; i < num_materials; i++) read_material(&r, type, i, &materials[i]);

    num_transforms = rd_key_int(&r, "transforms");
    transforms = (rz_mat4*)calloc(num_transforms > 0 ? num_transforms : 1, sizeof *transforms);
    for (i = 0; i < num_transforms; i++) {
        rd_expect(&r, "transform");
        if (rd_int(&r) != i) fail("transform index out of order in", filename);
        for (k = 0; k < 16; k++) transforms[i].m[k / 4][k % 4] = rd_float(&r);
    }

    num_models = rd_key_int(&r, "models");
    scene = (rzs_scene*)calloc(1, sizeof *scene);
    scene->models = (model_t**)calloc(num_models > 0 ? num_models : 1, sizeof(model_t*));
    scene->num_models = num_models;
    for (i = 0; i < num_models; i++) {
        modeldesc_t d;
        rz_mat4 transform;
        rd_expect(&r, "model");
        if (rd_int(&r) != i) fail("model index out of order in", filename);
        rd_key_word(&r, "mesh:", d.mesh);
        rd_key_word(&r, "skeleton:", d.skeleton);
        d.attached = rd_key_int(&r, "attached:");
        d.material = rd_key_int(&r, "material:");
        d.transform = rd_key_int(&r, "transform:");
        if (d.material < 0 || d.material >= num_materials || d.transform < 0 || d.transform >= num_transforms)
            fail("model refers to a missing material/transform in", filename);
        transform = hm_m4_mul(root, transforms[d.transform]);
        if (type == SCN_BLINN) scene->models[i] = make_blinn(&d, transform, &materials[d.material]);
        else scene->models[i] = make_pbr(&d, transform, &materials[d.material], type == SCN_PBRS,
                                         path_or_null(env));
    }
    fclose(r.f);
    free(materials);
    free(transforms);

    if (strcmp(skybox, "off") != 0) {
        int blur = strcmp(skybox, "ambient") == 0 ? -1 : (strcmp(skybox, "blurred") == 0 ? 1 : 0);
        if (blur == 0 && strcmp(skybox, "on") != 0) fail("bad skybox mode in", filename);
        if (!path_or_null(env)) fail("skybox without environment in", filename);
        scene->skybox = make_skybox(env, blur);
    }
    if (strcmp(shadow, "off") != 0) {
        if (strcmp(shadow, "on") == 0) { shadow_w = 512; shadow_h = 512; }
        else if (sscanf(shadow, "%dx%d", &shadow_w, &shadow_h) != 2 || shadow_w <= 0 || shadow_h <= 0)
            fail("bad shadow size in", filename);
    }
    /* scene.c:10-27 */
    scene->background = hm_v4(background.x, background.y, background.z, 1);
    scene->ambient = ambient;
    scene->punctual = punctual;
    if (shadow_w > 0 && shadow_h > 0) {
        int k;
        /* The ring is what bounds a flush: the frame that takes a slot again has to wait for the frame that used it
         * before, and the library finishes everything recorded so far.  The longer a flush, the more of its batches run
         * side by side on the library's lanes (a frame's shadow pass, main pass and blended pass depend on each other;
         * those of different frames do not).  A slot holds a framebuffer and a texture with their host images, 36 bytes
         * per texel: as many slots as fit in 1 GiB, 32 at least (a 2048 x 2048 map: 151 MB per slot). */
        const double slot_bytes = 36.0 * (double)shadow_w * (double)shadow_h;
        scene->shadow_ring = (int)(1073741824.0 / slot_bytes);
        if (scene->shadow_ring < 32) scene->shadow_ring = 32;
        if (scene->shadow_ring > RZS_SHADOW_RING) scene->shadow_ring = RZS_SHADOW_RING;
        for (k = 0; k < scene->shadow_ring; k++) {
            scene->shadow_buffers[k] = framebuffer_create(shadow_w, shadow_h);
            scene->shadow_maps[k] = texture_create(shadow_w, shadow_h);
            /* on the GPU library a texture's first copy from a framebuffer moves its host image into page-locked memory
             * and allocates its device image: paid here, not by the first frames that take the slot */
            if (strncmp(mi355r_backend(), "hip", 3) == 0) texture_from_depthbuffer(scene->shadow_maps[k], scene->shadow_buffers[k]);
        }
        scene->shadow_buffer = scene->shadow_buffers[0];
        scene->shadow_map = scene->shadow_maps[0];
    }
    return scene;
}

/* ======================= catalogue ====================================== */

/* Root transforms of the shipped scenes (the constants of
 * renderer/scenes/blinn_scenes.c and pbr_scenes.c):
 * root = S * T, or S * (R * T) with R = Ry * Rx, Rx or Ry. */
typedef struct {
    const char *name, *file;
    float tx, ty, tz;
    int use_t, use_rx, use_ry;
    float rx_deg, ry_deg;
    float scale;
} entry_t;

static const entry_t g_catalogue[] = {
    {"azura", "azura/azura.scn", -6.073f, -1.278f, 0.280f, 1, 0, 0, 0, 0, 0.378f},
    {"centaur", "centaur/centaur.scn", 0.154f, -7.579f, -30.749f, 1, 1, 1, -90, -90, 0.016f},
    {"craftsman", "craftsman/craftsman.scn", -1.668f, -27.061f, -10.834f, 1, 0, 0, 0, 0, 0.016f},
    {"elfgirl", "elfgirl/elfgirl.scn", 2.449f, -2.472f, -20.907f, 1, 1, 0, -90, 0, 0.023f},
    {"kgirl", "kgirl/kgirl.scn", 0, -4.937f, -96.547f, 1, 1, 1, -90, 90, 0.005f},
    {"lighthouse", "lighthouse/lighthouse.scn", -78.203f, -222.929f, 16.181f, 1, 0, 1, 0, -135, 0.0016f},
    {"mccree", "mccree/mccree.scn", 0.108f, -1.479f, 0.034f, 1, 0, 0, 0, 0, 0.337f},
    {"nier2b", "nier2b/nier2b.scn", 4.785f, -105.275f, -23.067f, 1, 0, 1, 0, 90, 0.004f},
    {"phoenix", "phoenix/phoenix.scn", 376.905f, -169.495f, 0, 1, 0, 1, 0, 180, 0.001f},
    {"vivi", "vivi/vivi.scn", 0, 0, -1.369f, 1, 1, 0, -90, 0, 0.331f},
    {"whip", "whip/whip.scn", -3732.619f, -93.643f, -1561.663f, 1, 1, 0, 90, 0, 0.0004f},
    {"witch", "witch/witch.scn", -17.924f, -16.974f, -32.691f, 1, 1, 0, -90, 0, 0.02f},
    {"assassin", "assassin/assassin.scn", 0, -125.815f, 18.898f, 1, 0, 0, 0, 0, 0.0038f},
    {"buster", "buster/buster.scn", 0, 15.918f, -5.720f, 1, 1, 0, 90, 0, 0.0045f},
    {"crab", "crab/crab.scn", -0.285f, 0.780f, 0.572f, 1, 0, 1, 0, 180, 0.167f},
    {"dieselpunk", "dieselpunk/dieselpunk.scn", 1.036f, -114.817f, 27.682f, 1, 0, 1, 0, -90, 0.0012f},
    {"drone", "drone/drone.scn", 0, -138.288f, -4.447f, 1, 0, 1, 0, 180, 0.004f},
    {"helmet", "helmet/helmet.scn", 0.002f, 0.187f, 0, 1, 1, 0, 90, 0, 0.5f},
    {"horse", "horse/horse.scn", 0.069f, 0, -0.845f, 1, 1, 0, -90, 0, 0.475f},
    {"junkrat", "junkrat/junkrat.scn", 3.735f, -382.993f, 57.980f, 1, 0, 0, 0, 0, 0.0013f},
    {"ornitier", "ornitier/ornitier.scn", -111.550f, -67.795f, 178.647f, 1, 0, 0, 0, 0, 0.00095f},
    {"ponycar", "ponycar/ponycar.scn", -10.343f, -13.252f, -186.343f, 1, 1, 0, -90, 0, 0.0015f},
    {"robot", "robot/robot.scn", -0.478f, 2.465f, -7.816f, 1, 1, 1, -90, -45, 0.0575f},
    {"sphere", "common/sphere.scn", 0, 0, 0, 0, 0, 0, 0, 0, 1.1f},
    {"spheres", "common/spheres.scn", 0, 0, 0, 0, 0, 0, 0, 0, 0.125f},
    {NULL, NULL, 0, 0, 0, 0, 0, 0, 0, 0, 0},
};

static rz_mat4 entry_root(const entry_t *e) {
    rz_mat4 s = hm_m4_scale(e->scale, e->scale, e->scale);
    rz_mat4 t, r;
    if (!e->use_t) return s;
    t = hm_m4_translate(e->tx, e->ty, e->tz);
    if (e->use_rx && e->use_ry) {
        rz_mat4 rx = hm_m4_rotate_x(RZ_RADIANS(e->rx_deg));
        rz_mat4 ry = hm_m4_rotate_y(RZ_RADIANS(e->ry_deg));
        r = hm_m4_mul(ry, rx);
    } else if (e->use_rx) {
        r = hm_m4_rotate_x(RZ_RADIANS(e->rx_deg));
    } else if (e->use_ry) {
        r = hm_m4_rotate_y(RZ_RADIANS(e->ry_deg));
    } else {
        return hm_m4_mul(s, t);
    }
    return hm_m4_mul(s, hm_m4_mul(r, t));
}

static const entry_t *find_entry(const char *name) {
    int i;
    for (i = 0; g_catalogue[i].name; i++) if (strcmp(g_catalogue[i].name, name) == 0) return &g_catalogue[i];
    return NULL;
}

const char *rzs_catalogue(int index) {
    int n = (int)(sizeof g_catalogue / sizeof g_catalogue[0]) - 1;
    return index >= 0 && index < n ? g_catalogue[index].name : NULL;
}

int rzs_set_assets_dir(const char *dir) {
    rzh_set_assets_dir(dir);
    return 0;
}

rzs_scene *rzs_scene_open(const char *name, const char *like) {
    size_t n = strlen(name);
    if (n > 4 && strcmp(name + n - 4, ".scn") == 0) {
        const entry_t *e = like ? find_entry(like) : NULL;
        if (like && !e) return NULL;
        return scene_from_scn(name, e ? entry_root(e) : hm_m4_identity());
    } else {
        const entry_t *e = find_entry(name);
        if (!e) return NULL;
        return scene_from_scn(e->file, entry_root(e));
    }
}

void rzs_scene_close(rzs_scene *scene) {            /* scene.c:29-48 */
    int i;
    if (!scene) return;
    graphics_flush();
    model_free(scene->skybox);
    for (i = 0; i < scene->num_models; i++) model_free(scene->models[i]);
    free(scene->models);
    if (scene->shadow_buffer) {
        int k;
        for (k = 0; k < scene->shadow_ring; k++) {
            framebuffer_release(scene->shadow_buffers[k]);
            texture_release(scene->shadow_maps[k]);
        }
    }
    free(scene);
}

void rzs_scene_info(rzs_scene *scene, int out[8]) {
    int i;
    memset(out, 0, sizeof(int) * 8);
    out[0] = scene->num_models;
    for (i = 0; i < scene->num_models; i++) {
        model_t *m = scene->models[i];
        out[1] += m->mesh->num_faces;
        if (!m->opaque) out[5]++;
        if (m->skeleton && m->attached < 0) out[6]++;
        if (m->skeleton && m->attached >= 0) out[7]++;
    }
    out[2] = scene->skybox != NULL;
    if (scene->shadow_buffer) { out[3] = scene->shadow_buffer->width; out[4] = scene->shadow_buffer->height; }
}

rz_framebuffer *rzs_scene_shadow_buffer(rzs_scene *scene) { return scene->shadow_buffer; }

/* how many frames' shadow buffers and maps the scene alternates between (0: it casts no shadows) */
int rzs_scene_shadow_ring(rzs_scene *scene) { return scene->shadow_buffer ? scene->shadow_ring : 0; }

/* Joint and normal matrices of `count` poses of a .ani skeleton, for tests: through the pipeline's pose
 * source (on_host == 0) or this layer's own per-frame evaluation (rzh_skeleton_update).  Returns the
 * number of joints (also with count == 0), -1 when the pipeline does not take the skeleton. */
int rzs_evaluate_poses(const char *ani, const float *times, int count, float *joints, float *normals, int on_host) {
    rzh_skeleton *sk = rzh_acquire_skeleton(ani);
    int J = rzh_skeleton_num_joints(sk), i;
    if (!on_host) {
        rz_pose_source *src = rzh_skeleton_pose_source(sk);
        if (!src) J = -1;
        else if (count > 0) pose_source_evaluate(src, times, count, (rz_mat4*)joints, (rz_mat3*)normals);
    } else {
        for (i = 0; i < count; i++) {
            rzh_skeleton_update(sk, times[i]);
            memcpy(joints + (size_t)i * J * 16, rzh_skeleton_joint_matrices(sk), sizeof(rz_mat4) * J);
            memcpy(normals + (size_t)i * J * 9, rzh_skeleton_normal_matrices(sk), sizeof(rz_mat3) * J);
        }
    }
    rzh_release_skeleton(sk);
    return J;
}

/* ======================= views ========================================== */

/* test_helper.c:14-18 and 162-169: eye (0,0,1.5) looking at the origin;
 * light direction from theta = phi = 45 degrees */
void rzs_default_view(rzs_view *v) {
    float theta = RZ_RADIANS(45), phi = RZ_RADIANS(45);
    v->eye[0] = 0; v->eye[1] = 0; v->eye[2] = 1.5f;
    v->target[0] = v->target[1] = v->target[2] = 0;
    v->light_dir[0] = -((float)sin(phi) * (float)sin(theta));
    v->light_dir[1] = -((float)cos(phi));
    v->light_dir[2] = -((float)sin(phi) * (float)cos(theta));
    v->frame_time = 0;
    v->layer_view = -1;
}

/* Orbit used by the batch benchmarks: eye = R (sin a, 0.3, cos a), R = 1.5,
 * a = 2 pi index / count, animation time index / 30 s. */
void rzs_orbit_view(rzs_view *v, int index, int count) {
    double a = 2.0 * 3.14159265358979323846 * (double)index / (double)(count > 0 ? count : 1);
    rzs_default_view(v);
    v->eye[0] = (float)(1.5 * sin(a));
    v->eye[1] = (float)(1.5 * 0.3);
    v->eye[2] = (float)(1.5 * cos(a));
    v->frame_time = (float)index / 30.0f;
}

/* ======================= frame driver =================================== */

/* test_helper.c:417-430: opaque models first, front to back; blended models
 * after them, back to front.  The comparator never returns 0. */
static int model_order(const void *ap, const void *bp) {
    const model_t *a = *(model_t* const*)ap, *b = *(model_t* const*)bp;
    if (a->opaque && b->opaque) return a->distance < b->distance ? -1 : 1;
    if (a->opaque && !b->opaque) return -1;
    if (!a->opaque && b->opaque) return 1;
    return a->distance < b->distance ? 1 : -1;
}

/* test_helper.c:432-446: distance = -(view * transform * mesh centre).z */
static void sort_models(rzs_scene *scene, rz_mat4 view) {
    int i;
    if (scene->num_models <= 1) return;
    for (i = 0; i < scene->num_models; i++) {
        model_t *m = scene->models[i];
        rz_vec3 c = m->mesh->center;
        rz_vec4 world = hm_m4_mul_v4(m->transform, hm_v4(c.x, c.y, c.z, 1));
        rz_vec4 viewp = hm_m4_mul_v4(view, world);
        m->distance = -viewp.z;
    }
    qsort(scene->models, scene->num_models, sizeof(model_t*), model_order);
}

static void scene_draw(rzs_scene *scene, rz_framebuffer *fb, const rzs_view *view, int clear);

void rzs_scene_draw(rzs_scene *scene, rz_framebuffer *fb, const rzs_view *view) { scene_draw(scene, fb, view, 1); }
void rzs_scene_draw_no_clear(rzs_scene *scene, rz_framebuffer *fb, const rzs_view *view) { scene_draw(scene, fb, view, 0); }

static void scene_draw(rzs_scene *scene, rz_framebuffer *fb, const rzs_view *view, int clear) {
    frame_t f;
    rz_vec3 eye = hm_v3(view->eye[0], view->eye[1], view->eye[2]);
    rz_vec3 target = hm_v3(view->target[0], view->target[1], view->target[2]);
    rz_vec3 up = hm_v3(0, 1, 0);
    float aspect = (float)fb->width / (float)fb->height;
    int i, num_opaque;

    /* test_helper.c:395-415 with camera.c:13-16, 101-107 */
    f.frame_time = view->frame_time;
    f.light_dir = hm_v3_unit(hm_v3(view->light_dir[0], view->light_dir[1], view->light_dir[2]));
    f.camera_pos = eye;
    f.light_view = hm_m4_lookat(hm_v3_neg(f.light_dir), hm_v3(0, 0, 0), up);
    f.light_proj = hm_m4_orthographic(1, 1, 0, 2);
    f.camera_view = hm_m4_lookat(eye, target, up);
    f.camera_proj = hm_m4_perspective(RZ_RADIANS(60), aspect, 0.1f, 10000);
    if (scene->shadow_buffer) {
        const int slot = scene->shadow_turn++ % scene->shadow_ring;
        scene->shadow_buffer = scene->shadow_buffers[slot];
        scene->shadow_map = scene->shadow_maps[slot];
    }
    f.ambient = scene->ambient;
    f.punctual = scene->punctual;
    f.shadow_map = scene->shadow_map;
    f.layer_view = view->layer_view;

    /* test_helper.c:448-507 */
    for (i = 0; i < scene->num_models; i++) model_update(scene->models[i], &f);
    if (scene->skybox) model_update(scene->skybox, &f);

    if (scene->shadow_buffer && scene->shadow_map) {
        sort_models(scene, f.light_view);
        framebuffer_clear_depth(scene->shadow_buffer, 1);
        for (i = 0; i < scene->num_models; i++) {
            if (scene->models[i]->opaque) model_draw(scene->models[i], scene->shadow_buffer, 1);
        }
        texture_from_depthbuffer(scene->shadow_map, scene->shadow_buffer);
    }

    sort_models(scene, f.camera_view);
    if (clear) {
        framebuffer_clear_color(fb, scene->background);
        framebuffer_clear_depth(fb, 1);
    }
    if (scene->skybox == NULL || f.layer_view >= 0) {
        for (i = 0; i < scene->num_models; i++) model_draw(scene->models[i], fb, 0);
    } else {
        for (num_opaque = 0; num_opaque < scene->num_models; num_opaque++) {
            if (!scene->models[num_opaque]->opaque) break;
        }
        for (i = 0; i < num_opaque; i++) model_draw(scene->models[i], fb, 0);
        model_draw(scene->skybox, fb, 0);
        for (i = num_opaque; i < scene->num_models; i++) model_draw(scene->models[i], fb, 0);
    }
}
