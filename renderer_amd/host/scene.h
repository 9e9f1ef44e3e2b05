/*
 * Scene layer on top of the pipeline API in include/mi355r.h: models of the
 * three shader families, .scn scene files, and the per-frame driver
 * (update -> shadow pass -> depth-to-texture -> opaque -> skybox -> blended).
 * It is the caller of the hot path, written against the same functions the
 * reference's scene code calls, and exposes a small C ABI (rzs_*) for the
 * Python tests and benchmarks.
 */
#ifndef RZ_SCENE_H
#define RZ_SCENE_H

#include "../../include/mi355r.h"
#include "assets.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rzs_scene rzs_scene;

/* Directory that holds the scene data (the reference's assets/ layout).  All
 * relative paths inside .scn files resolve against it. */
int rzs_set_assets_dir(const char *dir);

/* Opens a scene by catalogue name ("helmet", "azura", ... see rzs_catalogue)
 * or, when name ends in ".scn", from that file with the root transform of the
 * catalogue entry `like` (NULL = identity). Returns NULL on unknown names. */
rzs_scene *rzs_scene_open(const char *name, const char *like);
void rzs_scene_close(rzs_scene *scene);
const char *rzs_catalogue(int index);           /* NULL past the end */

/* Per-frame inputs (renderer/tests/test_helper.h:6-15 context_t, minus UI). */
typedef struct {
    float eye[3];
    float target[3];
    float light_dir[3];         /* need not be unit length */
    float frame_time;
    int layer_view;             /* -1 = normal rendering */
} rzs_view;

void rzs_default_view(rzs_view *view);          /* eye (0,0,1.5), light 45/45 deg */
void rzs_orbit_view(rzs_view *view, int index, int count);

/* Renders one frame into the framebuffer (asynchronously on the GPU build;
 * read the result with framebuffer_sync / framebuffer_read). */
/* Submission mode of rzs_scene_draw: 0 (default) whole meshes through
 * graphics_draw_mesh; 1 the reference's per-face loop (fill three attribute
 * blocks, graphics_draw_triangle), e.g. pbr_shader.c:561-577. */
void rzs_set_per_triangle(int on);
/* Skeletal poses: 1 (default) evaluated by the pipeline (graphics_draw_mesh_posed; not in per-triangle
 * mode, whose graphics_draw_triangle takes joint matrices from the uniform block), 0 on the host per
 * frame (rzh_skeleton_update, the reference's skeleton_update_joints). */
void rzs_set_device_poses(int on);
void rzs_scene_draw(rzs_scene *scene, rz_framebuffer *framebuffer, const rzs_view *view);
/* the same without clearing the framebuffer first: the frame is drawn over what is there */
void rzs_scene_draw_no_clear(rzs_scene *scene, rz_framebuffer *framebuffer, const rzs_view *view);

int rzs_evaluate_poses(const char *ani, const float *times, int count, float *joints, float *normals, int on_host);

/* info: [0] models, [1] faces, [2] has skybox, [3] shadow width, [4] shadow
 * height, [5] blended models, [6] skinned models, [7] node-attached models */
void rzs_scene_info(rzs_scene *scene, int out[8]);
rz_framebuffer *rzs_scene_shadow_buffer(rzs_scene *scene);
int rzs_scene_shadow_ring(rzs_scene *scene);

#ifdef __cplusplus
}
#endif
#endif
