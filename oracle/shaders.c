/*
 * TEST INFRASTRUCTURE -- CPU oracle of the three shader families.
 *
 * Restates the "low-level api" halves of renderer/shaders/blinn_shader.c
 * (9-260), pbr_shader.c (10-502) and skybox_shader.c (8-29): what each vertex
 * and fragment shader computes, in the reference's evaluation order.
 */
#include <math.h>
#include <string.h>
#include "omath.h"
#include "oracle.h"

/* ---------------------------------------------------------------- shared -- */

/* blinn_shader.c:9-43 == pbr_shader.c:10-44: skinned vertices use
 * model * sum(w_i joint_i) and normal * sum(w_i jointnormal_i) */
static rz_mat4 skinned_model(const rz_mat4 *model, const rz_mat4 *joints, rz_vec4 joint, rz_vec4 weight) {
    if (joints) {
        rz_mat4 skin = o_blend4(joints, joint, weight);
        return o_m4m4(model, &skin);
    }
    return *model;
}

static rz_mat3 skinned_normal(const rz_mat3 *normal, const rz_mat3 *joints, rz_vec4 joint, rz_vec4 weight) {
    if (joints) {
        rz_mat3 skin = o_blend3(joints, joint, weight);
        return o_m3m3(normal, &skin);
    }
    return *normal;
}

/* blinn_shader.c:178-195 == pbr_shader.c:311-328: light-space position to
 * [0,1]^3, slope-scaled bias, compare with the nearest shadow-map texel */
static int shadowed(rz_texture *shadow_map, rz_vec3 depth_position, float n_dot_l) {
    if (shadow_map) {
        float u = (depth_position.x + 1) * 0.5f;
        float v = (depth_position.y + 1) * 0.5f;
        float d = (depth_position.z + 1) * 0.5f;
        float bias = o_max(0.05f * (1 - n_dot_l), 0.005f);
        float current = d - bias;
        float closest = texture_sample(shadow_map, o_v2(u, v)).x;
        return current > closest;
    }
    return 0;
}

/* ----------------------------------------------------------------- blinn -- */

rz_vec4 blinn_vertex_shader(void *attribs_, void *varyings_, void *uniforms_) {
    rz_blinn_attribs *a = (rz_blinn_attribs*)attribs_;
    o_blinn_varyings *v = (o_blinn_varyings*)varyings_;
    rz_blinn_uniforms *u = (rz_blinn_uniforms*)uniforms_;
    rz_mat4 model = skinned_model(&u->model_matrix, u->joint_matrices, a->joint, a->weight);
    rz_vec4 world = o_m4v4(&model, o_v4(a->position.x, a->position.y, a->position.z, 1));
    rz_vec4 light_space = o_m4v4(&u->light_vp_matrix, world);
    if (u->shadow_pass) {                               /* :45-57 */
        v->texcoord = a->texcoord;
        return light_space;
    } else {                                            /* :59-87 */
        rz_mat3 nmat = skinned_normal(&u->normal_matrix, u->joint_n_matrices, a->joint, a->weight);
        rz_vec4 clip = o_m4v4(&u->camera_vp_matrix, world);
        rz_vec3 n = o_m3v3(&nmat, a->normal);
        v->world_position = o_xyz(world);
        v->depth_position = o_xyz(light_space);
        v->texcoord = a->texcoord;
        v->normal = o_unit3(n);
        return clip;
    }
}

rz_vec4 blinn_fragment_shader(void *varyings_, void *uniforms_, int *discard, int backface) {
    o_blinn_varyings *v = (o_blinn_varyings*)varyings_;
    rz_blinn_uniforms *u = (rz_blinn_uniforms*)uniforms_;
    rz_vec3 diffuse, specular, normal, emission, color;
    float alpha;

    if (u->shadow_pass) {                               /* :101-115 */
        if (u->alpha_cutoff > 0) {
            float a = u->basecolor.w;
            if (u->diffuse_map) a *= texture_sample(u->diffuse_map, v->texcoord).w;
            if (a < u->alpha_cutoff) *discard = 1;
        }
        return o_v4(0, 0, 0, 0);
    }

    /* material (:126-169) */
    diffuse = o_xyz(u->basecolor);
    alpha = u->basecolor.w;
    if (u->diffuse_map) {
        rz_vec4 s = texture_sample(u->diffuse_map, v->texcoord);
        diffuse = o_mul3(diffuse, o_xyz(s));
        alpha *= s.w;
    }
    specular = o_v3(0, 0, 0);
    if (u->specular_map) specular = o_xyz(texture_sample(u->specular_map, v->texcoord));
    normal = o_unit3(v->normal);
    if (backface) normal = o_neg3(normal);
    emission = o_v3(0, 0, 0);
    if (u->emission_map) emission = o_xyz(texture_sample(u->emission_map, v->texcoord));

    if (u->alpha_cutoff > 0 && alpha < u->alpha_cutoff) {       /* :220-222 */
        *discard = 1;
        return o_v4(0, 0, 0, 0);
    }

    color = emission;                                   /* :224-245 */
    if (u->ambient_intensity > 0) color = o_add3(color, o_scale3(diffuse, u->ambient_intensity));
    if (u->punctual_intensity > 0) {
        rz_vec3 light = o_neg3(u->light_dir);
        float n_dot_l = o_dot3(normal, light);
        if (n_dot_l > 0 && !shadowed(u->shadow_map, v->depth_position, n_dot_l)) {
            rz_vec3 view = o_unit3(o_sub3(u->camera_pos, v->world_position));
            rz_vec3 spec = o_v3(0, 0, 0);
            rz_vec3 lit;
            if (!(specular.x == 0 && specular.y == 0 && specular.z == 0)) {       /* :201-212 */
                rz_vec3 half = o_unit3(o_add3(light, view));
                float n_dot_h = o_dot3(normal, half);
                if (n_dot_h > 0) spec = o_scale3(specular, (float)pow(n_dot_h, u->shininess));
            }
            lit = o_add3(o_scale3(diffuse, n_dot_l), spec);
            color = o_add3(color, o_scale3(lit, u->punctual_intensity));
        }
    }
    return o_v4(color.x, color.y, color.z, alpha);
}

/* ------------------------------------------------------------------- pbr -- */

rz_vec4 pbr_vertex_shader(void *attribs_, void *varyings_, void *uniforms_) {
    rz_pbr_attribs *a = (rz_pbr_attribs*)attribs_;
    o_pbr_varyings *v = (o_pbr_varyings*)varyings_;
    rz_pbr_uniforms *u = (rz_pbr_uniforms*)uniforms_;
    rz_mat4 model = skinned_model(&u->model_matrix, u->joint_matrices, a->joint, a->weight);
    rz_vec4 world = o_m4v4(&model, o_v4(a->position.x, a->position.y, a->position.z, 1));
    if (u->shadow_pass) {                               /* :46-58 */
        v->texcoord = a->texcoord;
        return o_m4v4(&u->light_vp_matrix, world);
    } else {                                            /* :60-99 */
        rz_mat3 nmat = skinned_normal(&u->normal_matrix, u->joint_n_matrices, a->joint, a->weight);
        rz_vec4 clip = o_m4v4(&u->camera_vp_matrix, world);
        rz_vec4 light_space = o_m4v4(&u->light_vp_matrix, world);
        rz_vec3 n = o_m3v3(&nmat, a->normal);
        if (u->normal_map) {
            rz_mat3 tmat = o_upper3(&model);
            rz_vec3 t = o_m3v3(&tmat, o_xyz(a->tangent));
            rz_vec3 b;
            n = o_unit3(n);
            t = o_unit3(t);
            b = o_scale3(o_cross3(n, t), a->tangent.w);
            v->world_normal = n;
            v->world_tangent = t;
            v->world_bitangent = b;
        } else {
            v->world_normal = o_unit3(n);
        }
        v->world_position = o_xyz(world);
        v->depth_position = o_xyz(light_space);
        v->clip_position = clip;
        v->texcoord = a->texcoord;
        return clip;
    }
}

typedef struct {
    rz_vec3 diffuse, specular;
    float alpha, roughness;
    rz_vec3 normal;
    float occlusion;
    rz_vec3 emission;
} surface_t;

static float largest(rz_vec3 v) {                       /* :184-186 */
    return v.x > v.y && v.x > v.z ? v.x : (v.y > v.z ? v.y : v.z);
}

static surface_t surface_of(o_pbr_varyings *v, rz_pbr_uniforms *u, int backface) {
    surface_t s;
    rz_vec2 uv = v->texcoord;
    memset(&s, 0, sizeof s);
    if (u->workflow == 0) {                             /* metal-rough, :148-182 */
        rz_vec3 base = o_xyz(u->basecolor_factor);
        float metal = u->metalness_factor, rough = u->roughness_factor;
        s.alpha = u->basecolor_factor.w;
        if (u->basecolor_map) {
            rz_vec4 t = texture_sample(u->basecolor_map, uv);
            base = o_mul3(base, o_xyz(t));
            s.alpha *= t.w;
        }
        if (u->metalness_map) metal *= texture_sample(u->metalness_map, uv).x;
        if (u->roughness_map) rough *= texture_sample(u->roughness_map, uv).x;
        s.diffuse = o_scale3(base, (1 - 0.04f) * (1 - metal));
        s.specular = o_mix3(o_v3(0.04f, 0.04f, 0.04f), base, metal);
        s.roughness = rough;
    } else {                                            /* spec-gloss, :188-222 */
        rz_vec3 diff = o_xyz(u->diffuse_factor), spec = u->specular_factor;
        float gloss = u->glossiness_factor;
        s.alpha = u->diffuse_factor.w;
        if (u->diffuse_map) {
            rz_vec4 t = texture_sample(u->diffuse_map, uv);
            diff = o_mul3(diff, o_xyz(t));
            s.alpha *= t.w;
        }
        if (u->specular_map) spec = o_mul3(spec, o_xyz(texture_sample(u->specular_map, uv)));
        if (u->glossiness_map) gloss *= texture_sample(u->glossiness_map, uv).x;
        s.diffuse = o_scale3(diff, 1 - largest(spec));
        s.specular = spec;
        s.roughness = 1 - gloss;
    }
    if (u->normal_map) {                                /* :224-243 */
        rz_vec4 t = texture_sample(u->normal_map, uv);
        rz_vec3 tn = o_v3(t.x * 2 - 1, t.y * 2 - 1, t.z * 2 - 1);
        rz_mat3 tbn;
        /* columns: tangent, bitangent, normal */
        tbn.m[0][0] = v->world_tangent.x; tbn.m[1][0] = v->world_tangent.y; tbn.m[2][0] = v->world_tangent.z;
        tbn.m[0][1] = v->world_bitangent.x; tbn.m[1][1] = v->world_bitangent.y; tbn.m[2][1] = v->world_bitangent.z;
        tbn.m[0][2] = v->world_normal.x; tbn.m[1][2] = v->world_normal.y; tbn.m[2][2] = v->world_normal.z;
        s.normal = o_unit3(o_m3v3(&tbn, tn));
    } else {
        s.normal = o_unit3(v->world_normal);
    }
    if (backface) s.normal = o_neg3(s.normal);
    s.occlusion = u->occlusion_map ? texture_sample(u->occlusion_map, uv).x : 1;       /* :259-264 */
    s.emission = u->emission_map ? o_xyz(texture_sample(u->emission_map, uv)) : o_v3(0, 0, 0);
    return s;
}

/* image-based lighting, :281-309 */
static rz_vec3 ambient_shade(const surface_t *s, rz_ibldata *ibl, rz_vec3 n, rz_vec3 view) {
    rz_vec3 diffuse_color = o_scale3(s->diffuse, s->occlusion);
    rz_vec3 irradiance = o_xyz(cubemap_clamp_sample(ibl->diffuse_map, n));
    rz_vec3 diffuse = o_mul3(irradiance, diffuse_color);
    float n_dot_v = o_dot3(n, view);
    rz_vec4 lut = texture_clamp_sample(ibl->brdf_lut, o_v2(n_dot_v, s->roughness));
    rz_vec3 spec_color = o_v3(s->specular.x * lut.x + lut.y,
                              s->specular.y * lut.x + lut.y,
                              s->specular.z * lut.x + lut.y);
    rz_vec3 reflected = o_sub3(o_scale3(n, 2 * n_dot_v), view);               /* :276-279 */
    float top = (float)(ibl->mip_levels - 1);
    int level = (int)(s->roughness * top + 0.5f);
    rz_vec3 radiance = o_xyz(cubemap_clamp_sample(ibl->specular_maps[level], reflected));
    return o_add3(diffuse, o_mul3(radiance, spec_color));
}

/* directional light: GGX D, height-correlated Smith V, Schlick F; :334-393 */
static rz_vec3 light_shade(const surface_t *s, rz_vec3 light, rz_vec3 n, rz_vec3 view) {
    float n_dot_l = o_dot3(n, light);
    float n_dot_v = o_dot3(n, view);
    if (n_dot_l > 0 && n_dot_v > 0) {
        rz_vec3 half = o_unit3(o_add3(light, view));
        float n_dot_h = o_max(o_dot3(n, half), 0);
        float v_dot_h = o_max(o_dot3(view, half), 0);
        float a1 = s->roughness * s->roughness;
        float a2 = a1 * a1;
        /* D */
        float ndh2 = n_dot_h * n_dot_h;
        float fd = ndh2 * (a2 - 1) + 1;
        float d_term = a2 / (O_PI * fd * fd);
        /* V */
        float ndv2 = n_dot_v * n_dot_v, ndl2 = n_dot_l * n_dot_l;
        float gv = n_dot_l * (float)sqrt(ndv2 * (1 - a2) + a2);
        float gl = n_dot_v * (float)sqrt(ndl2 * (1 - a2) + a2);
        float v_term = 0.5f / (gv + gl);
        /* F */
        float ff = (float)pow(1 - v_dot_h, 5);
        float f90 = o_sat(largest(s->specular) * 50);
        rz_vec3 f = o_v3(s->specular.x + (f90 - s->specular.x) * ff,
                         s->specular.y + (f90 - s->specular.y) * ff,
                         s->specular.z + (f90 - s->specular.z) * ff);
        rz_vec3 dl = o_div3(s->diffuse, O_PI);
        rz_vec3 sl = o_scale3(f, v_term * d_term);
        rz_vec3 lobe = o_v3((1 - f.x) * dl.x + sl.x, (1 - f.y) * dl.y + sl.y, (1 - f.z) * dl.z + sl.z);
        return o_scale3(lobe, n_dot_l);
    }
    return o_v3(0, 0, 0);
}

static rz_vec4 display_encode(rz_vec3 c, float alpha) {                 /* :401-406 */
    return o_v4(o_encode_srgb(o_aces(c.x)), o_encode_srgb(o_aces(c.y)), o_encode_srgb(o_aces(c.z)), alpha);
}

/* layer inspector (:408-438): five slanted bands across the screen */
static int right_of_band_edge(int edge, rz_vec2 p) {
    float space = 0.15f;
    float offset = space * (float)edge;
    float sx = (1 - space * 0.5f) - offset, ex = (space * (5 - 0.5f)) - offset;
    /* vec2_edge(start, end, p), maths.c:108 with start.y = 0, end.y = 1 */
    return (p.x - sx) * (1.0f - 0.0f) - (p.y - 0.0f) * (ex - sx) > 0;
}

static rz_vec4 layer_color(int layer, const surface_t *s) {
    if (layer == 1) return display_encode(s->diffuse, s->alpha);
    if (layer == 2) return display_encode(s->specular, s->alpha);
    if (layer == 3) return o_v4(s->roughness, s->roughness, s->roughness, s->alpha);
    if (layer == 4) return o_v4(s->occlusion, s->occlusion, s->occlusion, s->alpha);
    return o_v4(s->normal.x * 0.5f + 0.5f, s->normal.y * 0.5f + 0.5f, s->normal.z * 0.5f + 0.5f, s->alpha);
}

rz_vec4 pbr_fragment_shader(void *varyings_, void *uniforms_, int *discard, int backface) {
    o_pbr_varyings *v = (o_pbr_varyings*)varyings_;
    rz_pbr_uniforms *u = (rz_pbr_uniforms*)uniforms_;
    surface_t s;
    rz_vec2 screen;

    if (u->shadow_pass) {                               /* :113-136 */
        if (u->alpha_cutoff > 0) {
            float a;
            if (u->workflow == 0) {
                a = u->basecolor_factor.w;
                if (u->basecolor_map) a *= texture_sample(u->basecolor_map, v->texcoord).w;
            } else {
                a = u->diffuse_factor.w;
                if (u->diffuse_map) a *= texture_sample(u->diffuse_map, v->texcoord).w;
            }
            if (a < u->alpha_cutoff) *discard = 1;
        }
        return o_v4(0, 0, 0, 0);
    }

    s = surface_of(v, u, backface);                     /* :446-490 */
    screen = o_v2(v->clip_position.x / v->clip_position.w * 0.5f + 0.5f,
                  v->clip_position.y / v->clip_position.w * 0.5f + 0.5f);
    if (u->alpha_cutoff > 0 && s.alpha < u->alpha_cutoff) {
        *discard = 1;
        return o_v4(0, 0, 0, 0);
    } else if (u->layer_view > 0) {
        return layer_color(u->layer_view, &s);
    } else if (u->layer_view == 0 && !right_of_band_edge(0, screen)) {
        int edge;
        for (edge = 1; edge < 5; edge++) if (right_of_band_edge(edge, screen)) break;
        return layer_color(edge, &s);
    } else {
        rz_vec3 view = o_unit3(o_sub3(u->camera_pos, v->world_position));
        rz_vec3 light = o_neg3(u->light_dir);
        float n_dot_l = o_dot3(s.normal, light);
        rz_vec3 color = s.emission;
        if (u->ambient_intensity > 0 && u->ibldata) {
            color = o_add3(color, o_scale3(ambient_shade(&s, u->ibldata, s.normal, view), u->ambient_intensity));
        }
        if (u->punctual_intensity > 0 && n_dot_l > 0) {
            if (!shadowed(u->shadow_map, v->depth_position, n_dot_l)) {
                color = o_add3(color, o_scale3(light_shade(&s, light, s.normal, view), u->punctual_intensity));
            }
        }
        return display_encode(color, s.alpha);
    }
}

/* ---------------------------------------------------------------- skybox -- */

rz_vec4 skybox_vertex_shader(void *attribs_, void *varyings_, void *uniforms_) {      /* :8-19 */
    rz_skybox_attribs *a = (rz_skybox_attribs*)attribs_;
    o_skybox_varyings *v = (o_skybox_varyings*)varyings_;
    rz_skybox_uniforms *u = (rz_skybox_uniforms*)uniforms_;
    rz_vec4 clip = o_m4v4(&u->vp_matrix, o_v4(a->position.x, a->position.y, a->position.z, 1));
    clip.z = clip.w * (1 - O_EPS);
    v->direction = a->position;
    return clip;
}

rz_vec4 skybox_fragment_shader(void *varyings_, void *uniforms_, int *discard, int backface) {   /* :21-29 */
    o_skybox_varyings *v = (o_skybox_varyings*)varyings_;
    rz_skybox_uniforms *u = (rz_skybox_uniforms*)uniforms_;
    (void)discard; (void)backface;
    return cubemap_sample(u->skybox, v->direction);
}
