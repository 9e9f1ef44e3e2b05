/*
 * Asset ingest for the scene layer: Wavefront OBJ meshes with the
 * "# ext.*" comment extensions, TGA and Radiance HDR images, .ani skeletons,
 * and the name-keyed caches the model layer shares them through.
 * Load-time only; nothing here runs per frame except skeleton evaluation.
 */
#ifndef RZ_ASSETS_H
#define RZ_ASSETS_H

#include <stdio.h>
#include "../../include/mi355r.h"

#define RZ_LINE 256

void rzh_set_assets_dir(const char *dir);
FILE *rzh_open_asset(const char *filename);

/* ---- images (renderer/core/image.h:4-13) ---- */
typedef struct {
    int is_hdr;                 /* 0: bytes in ldr, 1: floats in hdr */
    int width, height, channels;
    unsigned char *ldr;
    float *hdr;
    unsigned char *rgbe;        /* .hdr files: the file's own quads (mantissas + shared exponent), width * height * 4, same row order as hdr */
} rzh_image;

rzh_image *rzh_image_load(const char *filename);    /* .tga or .hdr */
void rzh_image_free(rzh_image *image);

/* ---- meshes (renderer/core/mesh.h) ---- */
typedef struct {
    int num_faces;
    rz_vertex *vertices;        /* 3 * num_faces */
    rz_vec3 center;             /* bounding-box centre */
} rzh_mesh;

rzh_mesh *rzh_mesh_load(const char *filename);
void rzh_mesh_free(rzh_mesh *mesh);

/* ---- skeletons (renderer/core/skeleton.h) ---- */
typedef struct rzh_skeleton rzh_skeleton;
rzh_skeleton *rzh_skeleton_load(const char *filename);
void rzh_skeleton_free(rzh_skeleton *skeleton);
void rzh_skeleton_update(rzh_skeleton *skeleton, float frame_time);
rz_mat4 *rzh_skeleton_joint_matrices(rzh_skeleton *skeleton);
rz_mat3 *rzh_skeleton_normal_matrices(rzh_skeleton *skeleton);
int rzh_skeleton_num_joints(rzh_skeleton *skeleton);
/* the skeleton's key tracks as a pose source of the pipeline (include/mi355r.h); NULL if it does not take them */
rz_pose_source *rzh_skeleton_pose_source(rzh_skeleton *skeleton);
/* off: find key segments by the reference's linear scan (skeleton.c:231-313) even where bisection applies; for tests */
void rzh_skeleton_use_bisection(rzh_skeleton *skeleton, int on);

/* ---- caches (renderer/shaders/cache_helper.h) ---- */
rzh_mesh *rzh_acquire_mesh(const char *filename);               /* NULL in, NULL out */
void rzh_release_mesh(rzh_mesh *mesh);
rzh_skeleton *rzh_acquire_skeleton(const char *filename);
void rzh_release_skeleton(rzh_skeleton *skeleton);
rz_texture *rzh_acquire_texture(const char *filename, rz_usage usage);
void rzh_release_texture(rz_texture *texture);
rz_cubemap *rzh_acquire_skybox(const char *env_name, int blur_level);
void rzh_release_skybox(rz_cubemap *skybox);
rz_ibldata *rzh_acquire_ibldata(const char *env_name);
void rzh_release_ibldata(rz_ibldata *ibldata);

#endif
