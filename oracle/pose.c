/*
 * TEST INFRASTRUCTURE -- CPU restatement of the reference's skeleton evaluation
 * (renderer/core/skeleton.c:231-341 with maths.c:291-318 quat_slerp, :492-523
 * mat4_from_quat / mat4_from_trs, :432-467 mat3_inverse_transpose) behind the
 * pose entry points of include/mi355r.h.  It is the checker of the device's
 * pose kernel: key segments are found by the reference's own linear scan, one
 * pose at a time, joints in index order.  Pinned by the golden frames of the
 * animated scenes (kgirl, buster, craftsman, phoenix, drone), which the
 * unmodified reference rendered through its own skeleton.c.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "omath.h"
#include "oracle.h"

typedef struct { int count; float *times; float *values; } o_track;

struct rz_pose_source {
    int num_joints;
    float max_time;
    int *parent;
    rz_mat4 *inverse_bind;
    o_track *tracks;                /* [num_joints][3] */
    rz_mat4 *world, *joints;        /* one pose: joint-to-model, joint matrices */
    rz_mat3 *normals;
};

static void *copy_of(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(p, src, bytes);
    return p;
}

rz_pose_source *pose_source_create(const rz_skeleton_desc *d) {
    int j, k, J = d->num_joints;
    rz_pose_source *s = (rz_pose_source*)calloc(1, sizeof *s);
    s->num_joints = J;
    s->max_time = d->max_time;
    s->parent = (int*)copy_of(d->parent, sizeof(int) * J);
    s->inverse_bind = (rz_mat4*)copy_of(d->inverse_bind, sizeof(rz_mat4) * J);
    s->tracks = (o_track*)calloc((size_t)3 * J, sizeof(o_track));
    for (j = 0; j < J; j++)
        for (k = 0; k < 3; k++) {
            o_track *t = &s->tracks[3 * j + k];
            t->count = d->key_count[3 * j + k];
            t->times = (float*)copy_of(d->key_times[3 * j + k], sizeof(float) * t->count);
            t->values = (float*)copy_of(d->key_values[3 * j + k], sizeof(float) * t->count * (k == 1 ? 4 : 3));
        }
    s->world = (rz_mat4*)calloc(J, sizeof(rz_mat4));
    s->joints = (rz_mat4*)calloc(J, sizeof(rz_mat4));
    s->normals = (rz_mat3*)calloc(J, sizeof(rz_mat3));
    return s;
}

void pose_source_release(rz_pose_source *s) {
    int i;
    for (i = 0; i < 3 * s->num_joints; i++) { free(s->tracks[i].times); free(s->tracks[i].values); }
    free(s->parent); free(s->inverse_bind); free(s->tracks); free(s->world); free(s->joints); free(s->normals);
    free(s);
}

/* get_translation / get_rotation / get_scale (skeleton.c:231-313) share this search */
static int segment(const o_track *t, float time, float *u) {
    int i;
    if (time <= t->times[0]) return -1;
    if (time >= t->times[t->count - 1]) return -2;
    for (i = 0; i < t->count - 1; i++) {
        float curr = t->times[i], next = t->times[i + 1];
        if (time >= curr && time < next) { *u = (time - curr) / (next - curr); return i; }
    }
    abort();
}

static rz_vec3 vec3_key(const o_track *t, float time, rz_vec3 none) {
    float u = 0;
    const float *v = t->values;
    int s;
    if (t->count == 0) return none;
    s = segment(t, time, &u);
    if (s == -1) return o_v3(v[0], v[1], v[2]);
    if (s == -2) { v += (size_t)(t->count - 1) * 3; return o_v3(v[0], v[1], v[2]); }
    v += (size_t)s * 3;
    return o_mix3(o_v3(v[0], v[1], v[2]), o_v3(v[3], v[4], v[5]), u);
}

static rz_vec4 slerp(rz_vec4 a, rz_vec4 b, float t) {               /* maths.c:291-318 */
    float c = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (c < 0) { b = o_v4(-b.x, -b.y, -b.z, -b.w); c = -c; }
    if (c > 1 - O_EPS) {
        return o_v4(o_mix(a.x, b.x, t), o_mix(a.y, b.y, t), o_mix(a.z, b.z, t), o_mix(a.w, b.w, t));
    } else {
        float angle = (float)acos(c);
        float s = (float)sin(angle);
        float fa = (float)sin((1 - t) * angle) / s;
        float fb = (float)sin(t * angle) / s;
        return o_v4(fa * a.x + fb * b.x, fa * a.y + fb * b.y, fa * a.z + fb * b.z, fa * a.w + fb * b.w);
    }
}

static rz_vec4 quat_key(const o_track *t, float time) {
    float u = 0;
    const float *v = t->values;
    int s;
    if (t->count == 0) return o_v4(0, 0, 0, 1);
    s = segment(t, time, &u);
    if (s == -1) return o_v4(v[0], v[1], v[2], v[3]);
    if (s == -2) { v += (size_t)(t->count - 1) * 4; return o_v4(v[0], v[1], v[2], v[3]); }
    v += (size_t)s * 4;
    return slerp(o_v4(v[0], v[1], v[2], v[3]), o_v4(v[4], v[5], v[6], v[7]), u);
}

static rz_mat4 identity4(void) {
    rz_mat4 m;
    memset(&m, 0, sizeof m);
    m.m[0][0] = m.m[1][1] = m.m[2][2] = m.m[3][3] = 1;
    return m;
}

static rz_mat4 from_trs(rz_vec3 t, rz_vec4 q, rz_vec3 s) {          /* maths.c:492-523 */
    rz_mat4 tm = identity4(), rm = identity4(), sm = identity4(), rs;
    float xx = q.x * q.x, xy = q.x * q.y, xz = q.x * q.z, xw = q.x * q.w;
    float yy = q.y * q.y, yz = q.y * q.z, yw = q.y * q.w, zz = q.z * q.z, zw = q.z * q.w;
    rm.m[0][0] = 1 - 2 * (yy + zz); rm.m[0][1] = 2 * (xy - zw); rm.m[0][2] = 2 * (xz + yw);
    rm.m[1][0] = 2 * (xy + zw); rm.m[1][1] = 1 - 2 * (xx + zz); rm.m[1][2] = 2 * (yz - xw);
    rm.m[2][0] = 2 * (xz - yw); rm.m[2][1] = 2 * (yz + xw); rm.m[2][2] = 1 - 2 * (xx + yy);
    tm.m[0][3] = t.x; tm.m[1][3] = t.y; tm.m[2][3] = t.z;
    sm.m[0][0] = s.x; sm.m[1][1] = s.y; sm.m[2][2] = s.z;
    rs = o_m4m4(&rm, &sm);
    return o_m4m4(&tm, &rs);
}

static rz_mat3 inverse_transpose(rz_mat3 m) {                       /* maths.c:432-467 */
    rz_mat3 adj, out;
    float a, b, c, det, inv;
    int r, k;
    adj.m[0][0] = +(m.m[1][1] * m.m[2][2] - m.m[2][1] * m.m[1][2]);
    adj.m[0][1] = -(m.m[1][0] * m.m[2][2] - m.m[2][0] * m.m[1][2]);
    adj.m[0][2] = +(m.m[1][0] * m.m[2][1] - m.m[2][0] * m.m[1][1]);
    adj.m[1][0] = -(m.m[0][1] * m.m[2][2] - m.m[2][1] * m.m[0][2]);
    adj.m[1][1] = +(m.m[0][0] * m.m[2][2] - m.m[2][0] * m.m[0][2]);
    adj.m[1][2] = -(m.m[0][0] * m.m[2][1] - m.m[2][0] * m.m[0][1]);
    adj.m[2][0] = +(m.m[0][1] * m.m[1][2] - m.m[1][1] * m.m[0][2]);
    adj.m[2][1] = -(m.m[0][0] * m.m[1][2] - m.m[1][0] * m.m[0][2]);
    adj.m[2][2] = +(m.m[0][0] * m.m[1][1] - m.m[1][0] * m.m[0][1]);
    a = +m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]);
    b = -m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]);
    c = +m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
    det = a + b + c;
    inv = 1 / det;
    for (r = 0; r < 3; r++) for (k = 0; k < 3; k++) out.m[r][k] = adj.m[r][k] * inv;
    return out;
}

/* skeleton_update_joints (skeleton.c:315-341) */
static void evaluate(rz_pose_source *s, float frame_time) {
    int i;
    frame_time = (float)fmod(frame_time, s->max_time);
    for (i = 0; i < s->num_joints; i++) {
        rz_vec3 t = vec3_key(&s->tracks[3 * i], frame_time, o_v3(0, 0, 0));
        rz_vec4 q = quat_key(&s->tracks[3 * i + 1], frame_time);
        rz_vec3 sc = vec3_key(&s->tracks[3 * i + 2], frame_time, o_v3(1, 1, 1));
        s->world[i] = from_trs(t, q, sc);
        if (s->parent[i] >= 0) s->world[i] = o_m4m4(&s->world[s->parent[i]], &s->world[i]);
        s->joints[i] = o_m4m4(&s->world[i], &s->inverse_bind[i]);
        s->normals[i] = inverse_transpose(o_upper3(&s->joints[i]));
    }
}

void pose_source_evaluate(rz_pose_source *s, const float *frame_times, int count, rz_mat4 *joints, rz_mat3 *normals) {
    int i;
    for (i = 0; i < count; i++) {
        evaluate(s, frame_times[i]);
        if (joints) memcpy(joints + (size_t)i * s->num_joints, s->joints, sizeof(rz_mat4) * s->num_joints);
        if (normals) memcpy(normals + (size_t)i * s->num_joints, s->normals, sizeof(rz_mat3) * s->num_joints);
    }
}

/* update_model's use of the pose (blinn_shader.c:264-304, pbr_shader.c:506-547) followed by the draw */
void graphics_draw_mesh_posed(rz_framebuffer *fb, rz_program *p, const rz_vertex *verts, int num_faces,
                              rz_pose_source *s, float frame_time, int attached) {
    /* the head of both uniform blocks is the same (blinn_shader.h:23-34, pbr_shader.h:39-50) */
    rz_blinn_uniforms *u = (rz_blinn_uniforms*)program_get_uniforms(p);
    rz_mat4 model = u->model_matrix;
    rz_mat3 normal = u->normal_matrix;
    rz_mat4 *joints = u->joint_matrices;
    rz_mat3 *joint_normals = u->joint_n_matrices;
    evaluate(s, frame_time);
    if (attached >= 0) {
        u->model_matrix = o_m4m4(&model, &s->joints[attached]);
        u->normal_matrix = inverse_transpose(o_upper3(&u->model_matrix));
        u->joint_matrices = NULL;
        u->joint_n_matrices = NULL;
    } else {
        u->joint_matrices = s->joints;
        u->joint_n_matrices = s->normals;
    }
    graphics_draw_mesh(fb, p, verts, num_faces);
    u->model_matrix = model;
    u->normal_matrix = normal;
    u->joint_matrices = joints;
    u->joint_n_matrices = joint_normals;
}
