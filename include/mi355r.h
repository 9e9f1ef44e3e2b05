/*
 * mi355r -- C ABI of the MI355X (gfx950) triangle pipeline.
 *
 * The library is a drop-in for the two translation units of the reference
 * software renderer that make up its hot path,
 *     renderer/core/graphics.c   (framebuffer, program, graphics_draw_triangle)
 *     renderer/core/texture.c    (texture / cubemap objects, samplers, depth->texture)
 * and for the one that draws 2-D overlays straight into a colour buffer,
 *     renderer/core/draw2d.c     (points, lines, outlines, label images)
 * and exports their public functions under their original names with
 * identical signatures and struct layouts (x86-64 SysV), so the reference's
 * scene files, shader "high-level api" and frame driver link against it
 * unchanged.  Every declaration below cites the reference interface it
 * replaces.  All work runs on the GPU through hand-written HIP kernels; there
 * is no CPU rendering path in this library.
 *
 * Plain C: pointers and sizes only.
 */
#ifndef MI355R_H
#define MI355R_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- value types: layout-identical to renderer/core/maths.h:4-9 ---------- */
typedef struct { float x, y; } rz_vec2;
typedef struct { float x, y, z; } rz_vec3;
typedef struct { float x, y, z, w; } rz_vec4;
typedef struct { float m[3][3]; } rz_mat3;     /* row-major, column vectors */
typedef struct { float m[4][4]; } rz_mat4;

/* ---- public objects ------------------------------------------------------ */

/* renderer/core/graphics.h:6-10.  Row 0 is the BOTTOM row of the image.
 * color_buffer (RGBA8) and depth_buffer (f32) are host mirrors of the device
 * buffers: they are only valid after framebuffer_sync(). */
typedef struct {
    int width, height;
    unsigned char *color_buffer;
    float *depth_buffer;
} rz_framebuffer;

/* renderer/core/texture.h:14-17.  buffer is the host image (float RGBA,
 * row 0 = bottom); the device copy is (re)uploaded lazily, see
 * texture_invalidate(). */
typedef struct {
    int width, height;
    rz_vec4 *buffer;
} rz_texture;

/* renderer/core/texture.h:19-21; faces in the order +x -x +y -y +z -z */
typedef struct {
    rz_texture *faces[6];
} rz_cubemap;

/* renderer/core/texture.h:7-12 */
typedef enum {
    RZ_USAGE_LDR_COLOR,
    RZ_USAGE_LDR_DATA,
    RZ_USAGE_HDR_COLOR,
    RZ_USAGE_HDR_DATA
} rz_usage;

/* renderer/core/mesh.h:8-15: the de-indexed vertex stream a mesh holds,
 * 3 consecutive entries per face, 80 bytes each */
typedef struct {
    rz_vec3 position;
    rz_vec2 texcoord;
    rz_vec3 normal;
    rz_vec4 tangent;
    rz_vec4 joint;
    rz_vec4 weight;
} rz_vertex;

/* renderer/core/graphics.h:12-15 */
typedef struct rz_program rz_program;
typedef rz_vec4 rz_vertex_shader(void *attribs, void *varyings, void *uniforms);
typedef rz_vec4 rz_fragment_shader(void *varyings, void *uniforms,
                                   int *discard, int backface);

/* ---- shader families ------------------------------------------------------
 * The device pipeline implements the reference's three shader pairs natively.
 * program_create() recognises them by function-pointer identity (the symbols
 * blinn_/pbr_/skybox_{vertex,fragment}_shader of the running program, or
 * pointers registered with program_register_family) and, failing that, by the
 * attribs/varyings/uniforms sizes.  An unknown pair aborts: there is no CPU
 * fallback. */
typedef enum {
    RZ_FAMILY_BLINN = 0,    /* renderer/shaders/blinn_shader.h:8-44  */
    RZ_FAMILY_PBR = 1,      /* renderer/shaders/pbr_shader.h:20-76   */
    RZ_FAMILY_SKYBOX = 2    /* renderer/shaders/skybox_shader.h:8-19 */
} rz_family;

/* uniform blocks: layout-identical to the reference structs cited above */
typedef struct rz_ibldata {                 /* pbr_shader.h:13-18 */
    int mip_levels;
    rz_cubemap *diffuse_map;
    rz_cubemap *specular_maps[15];
    rz_texture *brdf_lut;
} rz_ibldata;

typedef struct {                            /* blinn_shader.h:23-44, 344 bytes */
    rz_vec3 light_dir;
    rz_vec3 camera_pos;
    rz_mat4 model_matrix;
    rz_mat3 normal_matrix;
    rz_mat4 light_vp_matrix;
    rz_mat4 camera_vp_matrix;
    rz_mat4 *joint_matrices;
    rz_mat3 *joint_n_matrices;
    float ambient_intensity;
    float punctual_intensity;
    rz_texture *shadow_map;
    rz_vec4 basecolor;
    float shininess;
    rz_texture *diffuse_map;
    rz_texture *specular_map;
    rz_texture *emission_map;
    float alpha_cutoff;
    int shadow_pass;
} rz_blinn_uniforms;

typedef struct {                            /* pbr_shader.h:39-76, 440 bytes */
    rz_vec3 light_dir;
    rz_vec3 camera_pos;
    rz_mat4 model_matrix;
    rz_mat3 normal_matrix;
    rz_mat4 light_vp_matrix;
    rz_mat4 camera_vp_matrix;
    rz_mat4 *joint_matrices;
    rz_mat3 *joint_n_matrices;
    float ambient_intensity;
    float punctual_intensity;
    rz_texture *shadow_map;
    rz_vec4 basecolor_factor;
    float metalness_factor;
    float roughness_factor;
    rz_texture *basecolor_map;
    rz_texture *metalness_map;
    rz_texture *roughness_map;
    rz_vec4 diffuse_factor;
    rz_vec3 specular_factor;
    float glossiness_factor;
    rz_texture *diffuse_map;
    rz_texture *specular_map;
    rz_texture *glossiness_map;
    rz_texture *normal_map;
    rz_texture *occlusion_map;
    rz_texture *emission_map;
    rz_ibldata *ibldata;
    int workflow;                           /* 0 metalness, 1 specular */
    float alpha_cutoff;
    int shadow_pass;
    int layer_view;
} rz_pbr_uniforms;

typedef struct {                            /* skybox_shader.h:16-19, 72 bytes */
    rz_mat4 vp_matrix;
    rz_cubemap *skybox;
} rz_skybox_uniforms;

/* attribute blocks (what program_get_attribs hands out) */
typedef struct {                            /* blinn_shader.h:8-14, 64 bytes */
    rz_vec3 position; rz_vec2 texcoord; rz_vec3 normal;
    rz_vec4 joint; rz_vec4 weight;
} rz_blinn_attribs;
typedef rz_vertex rz_pbr_attribs;           /* pbr_shader.h:20-27, 80 bytes */
typedef struct { rz_vec3 position; } rz_skybox_attribs;  /* skybox_shader.h:8-10 */

/* ---- framebuffer management: renderer/core/graphics.h:17-21 -------------- */
rz_framebuffer *framebuffer_create(int width, int height);          /* graphics.c:11 */
void framebuffer_release(rz_framebuffer *framebuffer);              /* graphics.c:32 */
void framebuffer_clear_color(rz_framebuffer *framebuffer, rz_vec4 color);   /* graphics.c:38 */
void framebuffer_clear_depth(rz_framebuffer *framebuffer, float depth);     /* graphics.c:49 */

/* ---- program management: renderer/core/graphics.h:23-30 ------------------ */
rz_program *program_create(rz_vertex_shader *vertex_shader,
                           rz_fragment_shader *fragment_shader,
                           int sizeof_attribs, int sizeof_varyings,
                           int sizeof_uniforms,
                           int double_sided, int enable_blend);     /* graphics.c:82 */
void program_release(rz_program *program);                          /* graphics.c:120 */
void *program_get_attribs(rz_program *program, int nth_vertex);     /* graphics.c:134 */
void *program_get_uniforms(rz_program *program);                    /* graphics.c:139 */

/* ---- the pipeline: renderer/core/graphics.h:33 --------------------------- */
/* graphics.c:537.  Records one triangle from the three attribute blocks; the
 * triangle is rendered, in submission order, at the next flush point. */
void graphics_draw_triangle(rz_framebuffer *framebuffer, rz_program *program);

/* ---- textures: renderer/core/texture.h:23-41 ----------------------------- */
rz_texture *texture_create(int width, int height);                  /* texture.c:12 */
void texture_release(rz_texture *texture);                          /* texture.c:27 */
rz_texture *texture_from_file(const char *filename, rz_usage usage);        /* texture.c:108 */
/* The two copies below run on the device.  While the framebuffer has recorded draws the copy is
 * recorded behind them and runs at the next flush point, between the passes on either side of it;
 * texture->buffer on the host is valid only after texture_sync. */
void texture_from_colorbuffer(rz_texture *texture, rz_framebuffer *framebuffer);  /* texture.c:130 */
void texture_from_depthbuffer(rz_texture *texture, rz_framebuffer *framebuffer);  /* texture.c:147 */
rz_vec4 texture_repeat_sample(rz_texture *texture, rz_vec2 texcoord);       /* texture.c:160 */
rz_vec4 texture_clamp_sample(rz_texture *texture, rz_vec2 texcoord);        /* texture.c:169 */
rz_vec4 texture_sample(rz_texture *texture, rz_vec2 texcoord);              /* texture.c:178 */
rz_cubemap *cubemap_from_files(const char *positive_x, const char *negative_x,
                               const char *positive_y, const char *negative_y,
                               const char *positive_z, const char *negative_z,
                               rz_usage usage);                     /* texture.c:184 */
void cubemap_release(rz_cubemap *cubemap);                          /* texture.c:198 */
rz_vec4 cubemap_repeat_sample(rz_cubemap *cubemap, rz_vec3 direction);      /* texture.c:257 */
rz_vec4 cubemap_clamp_sample(rz_cubemap *cubemap, rz_vec3 direction);       /* texture.c:264 */
rz_vec4 cubemap_sample(rz_cubemap *cubemap, rz_vec3 direction);             /* texture.c:271 */

/* ---- additions (no counterpart in the reference) -------------------------
 * The reference is immediate-mode and synchronous; a GPU back end records
 * draws and needs explicit points at which host-visible memory is valid. */

/* Batched form of the per-face loop in the shaders' draw_model
 * (blinn_shader.c:323-341, pbr_shader.c:561-577, skybox_shader.c:57-64):
 * draws num_faces triangles from a de-indexed vertex stream with the
 * program's current uniforms.  The stream is uploaded once and cached by
 * address; call mesh_invalidate() if its contents change. */
void graphics_draw_mesh(rz_framebuffer *framebuffer, rz_program *program,
                        const rz_vertex *vertices, int num_faces);
void mesh_invalidate(const rz_vertex *vertices);

/* Skeletal poses evaluated on the device: renderer/core/skeleton.c:231-341 (key search, lerp / slerp,
 * joint hierarchy, inverse bind, normal matrices) for every pose a flush needs, in one kernel launch,
 * straight into the memory the vertex kernel skins from -- instead of one serial evaluation on the host
 * per frame plus a copy of all joint matrices per draw.  A pose source is an immutable copy of a
 * skeleton's key tracks; joints come parents first.  pose_source_create returns NULL for a skeleton the
 * device path does not take (more than 256 joints, a parent behind its child): the caller then poses on
 * the host as before.
 *   graphics_draw_mesh_posed = graphics_draw_mesh with the pose (source, frame_time) in place of the
 *   uniform block's joint_matrices / joint_n_matrices (which are ignored).  attached < 0: the mesh is
 *   skinned with the pose.  attached >= 0: rigid node animation as in blinn_shader.c:270-283 /
 *   pbr_shader.c:512-525 -- the uniform block's model_matrix is the model's own transform, and the draw
 *   uses model_matrix * joint[attached] and the normal matrix derived from that product.
 *   pose_source_evaluate hands the matrices back (count poses, num_joints matrices each): for tests. */
typedef struct {
    int num_joints;
    float max_time;                     /* time wraps: fmod(frame_time, max_time), skeleton.c:316 */
    const int *parent;                  /* [num_joints], -1 = root */
    const rz_mat4 *inverse_bind;        /* [num_joints] */
    const int *key_count;               /* [num_joints][3]: translation, rotation, scale keys */
    const float *const *key_times;      /* [num_joints][3] -> key_count floats */
    const float *const *key_values;     /* [num_joints][3] -> key_count x (3, 4 (x y z w), 3) floats */
} rz_skeleton_desc;
typedef struct rz_pose_source rz_pose_source;
rz_pose_source *pose_source_create(const rz_skeleton_desc *skeleton);
void pose_source_release(rz_pose_source *source);
void graphics_draw_mesh_posed(rz_framebuffer *framebuffer, rz_program *program, const rz_vertex *vertices,
                              int num_faces, rz_pose_source *source, float frame_time, int attached);
void pose_source_evaluate(rz_pose_source *source, const float *frame_times, int count,
                          rz_mat4 *joint_matrices, rz_mat3 *joint_n_matrices);

/* Executes everything recorded so far (asynchronously, in order). */
void graphics_flush(void);
/* Flushes, waits, and makes color_buffer/depth_buffer of this framebuffer
 * valid on the host.  Any code that reads or writes those arrays directly
 * (blit, 2D overlay, dumps) calls this first. */
void framebuffer_sync(rz_framebuffer *framebuffer);
/* The same for color_buffer alone: what a blit needs (renderer/core/private.c:8-28 reads nothing
 * else), half the bytes over the bus. */
void framebuffer_sync_color(rz_framebuffer *framebuffer);
/* Presentation: renderer/core/private.c:8-50 (private_blit_bgr / private_blit_rgb) with the flip and
 * the channel swap done on the device.  destination is the 4-channel image of the window surface
 * (width x height x 4 bytes, row 0 at the TOP): pixel (r, c) of the framebuffer goes to row
 * height - 1 - r; bytes 0..2 are written (blue, green, red when swap_red_blue, else red, green, blue),
 * byte 3 of every destination pixel is left as it is, as in the reference.  Flushes and waits for this
 * framebuffer only.  A maintainer replaces the body of private_blit_bgr by
 * framebuffer_blit(src, dst->ldr_buffer, 1). */
void framebuffer_blit(rz_framebuffer *framebuffer, unsigned char *destination, int swap_red_blue);

/* ---- 2-D overlays: renderer/core/draw2d.h:8-15 ---------------------------
 * The reference's own names and signatures (the PBR layer inspector draws its edges and labels with
 * them, renderer/tests/test_pbr.c:71-106), executed on the device's colour image behind everything
 * recorded for the framebuffer so far, so that a host which links these instead of core/draw2d.c
 * needs no framebuffer_sync in front of its overlays.  Points are in [0,1]^2 with (0,0) the bottom
 * left pixel; a point or line replaces all four bytes of the pixels it touches (draw2d.c:30-37), a
 * texture is added byte-wise, wrapping, to red, green and blue (draw2d.c:75-93). */
void draw2d_draw_point(rz_framebuffer *framebuffer, rz_vec4 color, rz_vec2 point);         /* draw2d.c:95 */
void draw2d_draw_line(rz_framebuffer *framebuffer, rz_vec4 color,
                      rz_vec2 point0, rz_vec2 point1);                                     /* draw2d.c:104 */
void draw2d_draw_triangle(rz_framebuffer *framebuffer, rz_vec4 color,
                          rz_vec2 point0, rz_vec2 point1, rz_vec2 point2);                 /* draw2d.c:114 */
void draw2d_draw_texture(rz_framebuffer *framebuffer, rz_texture *texture, rz_vec2 origin); /* draw2d.c:128 */
/* Same for a texture written by texture_from_{color,depth}buffer. */
void texture_sync(rz_texture *texture);
/* texture_sync for callers that cannot allocate device memory.  No counterpart in the reference (whose
 * textures live in host memory only); it is for callers that must not call hipMalloc: tight per-frame
 * loops (a viewer that shows the shadow map every frame), harnesses that forbid allocation.  Same result
 * as texture_sync -- texture->buffer valid on the host -- but the row-major staging image lives in a
 * fixed 16 MiB array inside the library's device image instead of a buffer allocated and freed per
 * call; the call allocates and frees nothing and creates no event.  It does SYNCHRONISE: it waits for
 * the stream that wrote the texture, before and after its own kernel and copy, so it is not a call to
 * make while a stream is being captured into a graph.
 * Limits: at most 1024 x 1024 = 1 048 576 texels; the copy into the texture must have been submitted
 * (graphics_flush) -- a recorded copy is not submitted here, because submitting sizes scratch; the
 * host image must be page-locked, which texture_from_colorbuffer / texture_from_depthbuffer see to
 * (a pageable destination would make the runtime stage the copy in memory of its own).  One staging
 * array per GPU, shared by all calls on it: calls that overlap on one GPU race.
 * Returns RZ_NOALLOC_OK, or a negative code after a message on stderr (nothing is written then). */
enum { RZ_NOALLOC_OK = 0, RZ_NOALLOC_TOO_LARGE = -1, RZ_NOALLOC_NOT_SUBMITTED = -2, RZ_NOALLOC_NO_HOST_IMAGE = -3,
       RZ_NOALLOC_NOT_PINNED = -4 };
int texture_sync_noalloc(rz_texture *texture);
/* Tells the library that texture->buffer was modified by the host. */
void texture_invalidate(rz_texture *texture);
/* Copies the finished device images into caller memory (host pointers) or
 * into device memory (device pointers, e.g. a torch tensor), without touching
 * the framebuffer's own host mirrors.  Either pointer may be NULL. */
void framebuffer_read(rz_framebuffer *framebuffer, void *color_rgba8, void *depth_f32);
void framebuffer_read_device(rz_framebuffer *framebuffer, void *color_rgba8_dev,
                             void *depth_f32_dev);

/* The finished colour image without its alpha bytes, into device memory: width * height * 3 bytes, RGB,
 * row 0 = bottom.  draw_fragment never writes alpha (graphics.c:466-468), so the fourth byte of every
 * pixel is whatever the last clear put there: a receiver that knows the clear colour loses nothing,
 * and a gather of frames over GPU links moves a quarter less. */
void framebuffer_read_device_rgb(rz_framebuffer *framebuffer, void *color_rgb8_dev);
/* The same for `count` framebuffers at once, frame i at color_rgb8_dev + i * stride_bytes (stride and
 * pointer multiples of 4, stride at least width * height * 3): the copies follow the rendering on each
 * framebuffer's own stream and the host waits once for all of them -- what a step's frames cost on
 * their way into the buffer a gather sends. */
void framebuffers_read_device_rgb(rz_framebuffer *const *framebuffers, int count, void *color_rgb8_dev,
                                  size_t stride_bytes);
/* The same in two halves.  _begin submits what is recorded, launches the copies behind the rendering and
 * returns without waiting; _end waits for those copies only, not for what was submitted after _begin.  In
 * between the caller may record and submit more -- also into the same framebuffers: later passes come after
 * the copies -- but must leave the destination alone.  One set of frames at a time. */
void framebuffers_read_device_rgb_begin(rz_framebuffer *const *framebuffers, int count, void *color_rgb8_dev,
                                        size_t stride_bytes);
void framebuffers_read_device_rgb_end(void);

/* Explicit family registration for hosts whose shader functions are not
 * visible as dynamic symbols. */
void program_register_family(rz_vertex_shader *vs, rz_fragment_shader *fs,
                             rz_family family);

/* Library control / introspection.  mi355r_init selects the device
 * (-1 = current) and creates the stream; it is called implicitly by the
 * first object creation.  mi355r_backend returns "hip:gfx950". */
int mi355r_init(int device);
void mi355r_shutdown(void);
const char *mi355r_backend(void);
void *mi355r_stream(void);              /* hipStream_t of the pipeline */

/* Workload counters since the last reset (measurement only; the device-side
 * ones cost atomics and are off unless enabled): index 0 triangles submitted,
 * 1 triangles rasterised after clip/cull, 2 fragments shaded (the reference's
 * draw_fragment calls, graphics.c:437), 3 bin-list entries, 4 passes,
 * 5 draws, 6 skybox fragments, 7 blended fragments, 8 fragments the device
 * actually shaded (deferred shading skips fragments that get overdrawn),
 * 9 skybox fragments among those. */
void mi355r_get_counters(unsigned long long out[12]);
void mi355r_reset_counters(void);
void mi355r_enable_counters(int on);

/* Device time per kernel, from HIP events recorded on the pipeline's stream
 * around each launch while enabled (passes then run on one stream, so that a
 * kernel has the device to itself): out[0..6] = summed milliseconds of the
 * vertex, setup+clip, bin, raster, skybox-shade, blinn-shade and pbr-shade
 * kernels, out[7] = launch sequences timed (one sequence covers a batch of
 * passes); returns the number of passes. */
void mi355r_enable_timing(int on);
unsigned long long mi355r_get_timing(double out[8], int reset);

/* Inspection of the geometry stage of the most recently submitted pass
 * (a pass = a maximal run of draws into one framebuffer, cut once more in
 * front of its first blended draw).  No counterpart in
 * the reference, whose rasterize_triangle walks each bounding box at once
 * (graphics.c:505-532); this is how tests compare the integer binning stage
 * with its CPU statement.  Triangles are numbered in submission order.
 *   boxes[t] = {x0 | x1 << 16, y0 | y1 << 16}: the pixel box of find_bounding_box
 *              (graphics.c:355-364), for a clipped triangle the union over its
 *              fan; x0 > x1 when nothing is rasterised
 *   fans[t]  = triangles rasterised for input triangle t (0 when culled or
 *              without pixels, 1 for an unclipped one, else the fan length up
 *              to the first culled fan triangle, graphics.c:584-607)
 *   bins     = square screen bins of bin_edge pixels, row-major from the
 *              bottom-left; counts[b] entries each, concatenated in `entries`
 *              as {triangle, fan index} pairs in rasterisation order
 * mi355r_inspect_pass waits for the device and returns 0 if no pass ran. */
typedef struct {
    int num_tris, width, height, bin_edge, bins_x, bins_y;
    unsigned int bin_entries;
} rz_pass_info;
int mi355r_inspect_pass(rz_pass_info *info);
void mi355r_inspect_boxes(unsigned int *boxes, unsigned int *fans);
void mi355r_inspect_bins(unsigned int *counts, unsigned int *entries);
/* The last pass's first triangle that belongs to a sky box kept out of the bins (its triangles sit on the pass's table
 * of sky triangles: mi355r_inspect_boxes reports their boxes, mi355r_inspect_bins has no entries of theirs);
 * = the pass's triangle count when there is none. */
int mi355r_inspect_sky_first(void);

/* Scratch estimates (tests).  The clipper's block table and the bin lists of a pass are sized by an
 * estimate before the pass runs; a pass that needs more does not touch its framebuffer, reports what
 * it needs, and is run again -- together with the passes of its flush that may depend on it -- before
 * the next flush starts or its framebuffer is read, so no input makes the library give up
 * (graphics.c:537-608 renders any input).  mi355r_set_scratch_estimate forces the first-time sizes
 * (clipper blocks and list entries per pass; 0, 0 = the library's own estimate), which lets a test
 * send ordinary scenes down that road; mi355r_passes_run_again counts the passes that took it. */
void mi355r_set_scratch_estimate(unsigned long long fan_blocks, unsigned long long list_entries);
unsigned long long mi355r_passes_run_again(void);
/* Measurement: the name of the rasterising kernel the last batch of passes was launched with (the strip
 * height is chosen per batch), as a profiler spells it, e.g. "rz_strip_kernel<4, false>". */
const char *mi355r_last_raster_kernel(void);
/* graphics_draw_triangle remembers, per program, the vertex stream its last draw was made of (on the host to
 * compare against, on the device to draw from): a later draw of the program whose triangles repeat that
 * stream or a prefix of it -- the reference's draw_model loops submit the same faces every frame -- stores
 * and uploads nothing per triangle.  The first triangle that differs makes the draw an ordinary recorded
 * one; a program whose stream changed four times in a row is no longer remembered.  MI355R_REPLAY=0
 * switches it off.  Counts the draws that were drawn from a remembered stream. */
unsigned long long mi355r_replayed_draws(void);

/* Device image formats.  texture->buffer is float RGBA as in the reference
 * (texture.h:14-17); on the device an image is kept in 4 bytes per texel when
 * that loses nothing, which is decided by looking at every texel.  This entry
 * point exposes that decision (host code only, usable without a device):
 * returns 0 (stays float4), 4 (channel = byte / 255.0f, texture.c:40-61),
 * 8 (rgb = (float)pow(byte / 255.0f, 2.2), alpha = byte / 255.0f,
 * texture.c:63-75) or 16 (rgb = (float)mantissa * 2^(e - 136), alpha 1,
 * image.c:329-347) and, for a non-zero result, the packed words
 * r | g << 8 | b << 16 | a-or-e << 24 in packed[count]. */
int mi355r_pack_texels(const rz_vec4 *texels, size_t count, unsigned int *packed);

/* Ingest without the float detour (texture.c:32-128 with image.c:315-347).  A loader that has the
 * file's bytes in hand -- 8-bit channels of a TGA, RGBE quads of a Radiance file -- already holds one of
 * the formats above and hands the words over as they are: format 4, 8 or 16 as mi355r_pack_texels
 * names them, words[width * height] = r | g << 8 | b << 16 | a-or-e << 24, row 0 = bottom.  The image
 * goes to the device at once; texture->buffer stays NULL until texture_sync(texture) asks for it (the
 * device then decodes to the very floats the reference's conversion produces).  Returns NULL when the
 * library keeps float images only (MI355R_PACK=0): the loader then fills texture_create's buffer. */
rz_texture *texture_create_packed(int width, int height, int format, const unsigned int *words);

#ifdef __cplusplus
}
#endif
#endif /* MI355R_H */
