/* Host side, part 3 of 6: the recorder -- a draw call snapshots its program's uniform block, translates the host
 * pointers in it to device handles and joins the list of pending draws (graphics.c's immediate-mode API, deferred). */

/* --------------------------------------------------------------- recording -- */

static void copy_floats(float *dst, const void *src, int n) { memcpy(dst, src, sizeof(float) * n); }

/* uniform block (host layout, host pointers) -> DrawRec (device form) */
static void snapshot_uniforms_once(rz_program *p, Pending &pd, int num_joints);
static bool g_ignore_host_joints = false;   /* while graphics_draw_mesh_posed takes its snapshot */

/* Resolving a texture can upload it, and an upload flushes what is recorded -- recorded
 * framebuffer copies included, which changes how a texture resolved a moment ago has to be waited
 * for.  A snapshot during which a flush happened is therefore taken again (nothing is left to
 * upload the second time). */
static void snapshot_uniforms(rz_program *p, Pending &pd, int num_joints) {
    for (int attempt = 0; attempt < 3; attempt++) {
        const unsigned long long epoch = g.flush_epoch;
        snapshot_uniforms_once(p, pd, num_joints);
        if (g.flush_epoch == epoch) return;
    }
    fatal("textures keep being uploaded while a draw is recorded");
}

static void snapshot_uniforms_once(rz_program *p, Pending &pd, int num_joints) {
    DrawRec &r = pd.rec;
    memset(&r, 0, sizeof r);
    g.current_gpu_textures.clear();
    g.current_pending_textures.clear();
    r.family = p->family;
    r.flags = (p->double_sided ? RZ_DRAW_DOUBLE_SIDED : 0) | (p->blend ? RZ_DRAW_BLEND : 0);
    const rz_mat4 *joints = nullptr;
    const rz_mat3 *joint_normals = nullptr;
    if (p->family == RZ_FAMILY_SKYBOX) {
        const rz_skybox_uniforms *u = (const rz_skybox_uniforms*)p->uniforms;
        r.num_varyings = 3;
        copy_floats(r.camera_vp, &u->vp_matrix, 16);
        r.skybox = dev_cube(u->skybox);
        if (!r.skybox) fatal("skybox draw without a cube map");
        bool uniform = true;
        for (int i = 0; i < 6; i++) {
            r.maps[i] = dev_tex(u->skybox->faces[i]);
            uniform = uniform && r.maps[i].w == r.maps[0].w && r.maps[i].h == r.maps[0].h && r.maps[i].flags == r.maps[0].flags;
        }
        if (uniform) r.flags |= RZ_DRAW_UNIFORM_CUBE;
        /* a program that blends needs the colour itself, not its bytes */
        if (!p->blend && cube_display(u->skybox, &r.maps[0])) r.flags |= RZ_DRAW_SKY_BYTES;
    } else if (p->family == RZ_FAMILY_BLINN) {
        const rz_blinn_uniforms *u = (const rz_blinn_uniforms*)p->uniforms;
        r.num_varyings = 11;
        copy_floats(r.light_dir, &u->light_dir, 3);
        copy_floats(r.camera_pos, &u->camera_pos, 3);
        copy_floats(r.model, &u->model_matrix, 16);
        copy_floats(r.normal, &u->normal_matrix, 9);
        copy_floats(r.light_vp, &u->light_vp_matrix, 16);
        copy_floats(r.camera_vp, &u->camera_vp_matrix, 16);
        joints = u->joint_matrices;
        joint_normals = u->joint_n_matrices;
        r.ambient = u->ambient_intensity;
        r.punctual = u->punctual_intensity;
        if (!u->shadow_pass) r.shadow_map = dev_tex(u->shadow_map);     /* the shadow pass renders it, it does not sample it */
        copy_floats(r.base, &u->basecolor, 4);
        r.shininess = u->shininess;
        r.maps[0] = dev_tex(u->diffuse_map);
        r.maps[1] = dev_tex(u->specular_map);
        r.maps[2] = dev_tex(u->emission_map);
        r.alpha_cutoff = u->alpha_cutoff;
        if (u->shadow_pass) r.flags |= RZ_DRAW_SHADOW_PASS;
        else {
            rz_texture *const maps[3] = {u->diffuse_map, u->specular_map, u->emission_map};
            r.bundle = material_bundle(maps, 3, r.maps, 4);
        }
    } else {
        const rz_pbr_uniforms *u = (const rz_pbr_uniforms*)p->uniforms;
        r.num_varyings = 21;
        copy_floats(r.light_dir, &u->light_dir, 3);
        copy_floats(r.camera_pos, &u->camera_pos, 3);
        copy_floats(r.model, &u->model_matrix, 16);
        copy_floats(r.normal, &u->normal_matrix, 9);
        copy_floats(r.light_vp, &u->light_vp_matrix, 16);
        copy_floats(r.camera_vp, &u->camera_vp_matrix, 16);
        joints = u->joint_matrices;
        joint_normals = u->joint_n_matrices;
        r.ambient = u->ambient_intensity;
        r.punctual = u->punctual_intensity;
        if (!u->shadow_pass) r.shadow_map = dev_tex(u->shadow_map);     /* the shadow pass renders it, it does not sample it */
        r.workflow = u->workflow;
        if (u->workflow == 0) {
            copy_floats(r.base, &u->basecolor_factor, 4);
            r.metalness = u->metalness_factor;
            r.roughness = u->roughness_factor;
            r.maps[0] = dev_tex(u->basecolor_map);
            r.maps[1] = dev_tex(u->metalness_map);
            r.maps[2] = dev_tex(u->roughness_map);
        } else {
            copy_floats(r.base, &u->diffuse_factor, 4);
            copy_floats(r.specular, &u->specular_factor, 3);
            r.glossiness = u->glossiness_factor;
            r.maps[0] = dev_tex(u->diffuse_map);
            r.maps[1] = dev_tex(u->specular_map);
            r.maps[2] = dev_tex(u->glossiness_map);
        }
        r.maps[3] = dev_tex(u->normal_map);
        r.maps[4] = dev_tex(u->occlusion_map);
        r.maps[5] = dev_tex(u->emission_map);
        r.ibl = dev_ibl(u->ibldata);
        r.alpha_cutoff = u->alpha_cutoff;
        r.layer_view = u->layer_view;
        if (u->shadow_pass) r.flags |= RZ_DRAW_SHADOW_PASS;
        else {
            rz_texture *const maps[6] = {u->workflow == 0 ? u->basecolor_map : u->diffuse_map, u->workflow == 0 ? u->metalness_map : u->specular_map,
                                         u->workflow == 0 ? u->roughness_map : u->glossiness_map, u->normal_map, u->occlusion_map, u->emission_map};
            r.bundle = material_bundle(maps, 6, r.maps, 8);
        }
    }
    if (p->family != RZ_FAMILY_SKYBOX) {
        bool narrow = !r.shadow_map.data || (r.shadow_map.flags & RZ_TEX_NARROW_FORMATS);
        for (int i = 0; i < 6; i++) narrow = narrow && (!r.maps[i].data || (r.maps[i].flags & RZ_TEX_NARROW_FORMATS));
        if (r.ibl) narrow = narrow && g.ibl_narrow[((const rz_pbr_uniforms*)p->uniforms)->ibldata];
        if (narrow) r.flags |= RZ_DRAW_NARROW_TEXELS;
    }
    pd.gpu_textures = g.current_gpu_textures;
    pd.pending_textures = g.current_pending_textures;
    /* joint matrices are rewritten in place by the skeleton every frame
     * (skeleton.c:336-337): copy them now */
    pd.num_joints = 0;
    if (joints && !g_ignore_host_joints) {
        if (!joint_normals && !(r.flags & RZ_DRAW_SHADOW_PASS)) fatal("joint matrices without joint normal matrices");
        pd.num_joints = num_joints;
        r.flags |= RZ_DRAW_SKINNED;
        pd.joints_at = g.side.size();
        g.side.insert(g.side.end(), (const char*)joints, (const char*)joints + sizeof(rz_mat4) * num_joints);
        pd.joint_normals_at = g.side.size();
        if (joint_normals)
            g.side.insert(g.side.end(), (const char*)joint_normals, (const char*)joint_normals + sizeof(rz_mat3) * num_joints);
        else
            g.side.insert(g.side.end(), sizeof(rz_mat3) * num_joints, 0);    /* a shadow pass reads none; the layout stays the same */
        while (g.side.size() % 16) g.side.push_back(0);
    }
}

extern "C" void graphics_draw_mesh(rz_framebuffer *fp, rz_program *program, const rz_vertex *vertices, int num_faces) {
    Fb *fb = as_fb(fp);
    if (num_faces <= 0) return;
    Pending pd;                     /* built aside: texture uploads below may flush the list */
    pd.fb = fb;
    pd.program = program;
    pd.transient = false;
    snapshot_uniforms(program, pd, device_mesh(vertices, num_faces).num_joints);
    Mesh &m = device_mesh(vertices, num_faces);
    pd.rec.vertices = m.d_vertices;
    pd.rec.num_faces = num_faces;
    g.pending.push_back(std::move(pd));
}

/* the joint matrices a program's uniform block points at right now (nullptr: not skinned) */
static void pose_of(const rz_program *p, const rz_mat4 **joints, const rz_mat3 **joint_normals) {
    *joints = nullptr;
    *joint_normals = nullptr;
    if (p->family == RZ_FAMILY_BLINN) {
        *joints = ((const rz_blinn_uniforms*)p->uniforms)->joint_matrices;
        *joint_normals = ((const rz_blinn_uniforms*)p->uniforms)->joint_n_matrices;
    } else if (p->family == RZ_FAMILY_PBR) {
        *joints = ((const rz_pbr_uniforms*)p->uniforms)->joint_matrices;
        *joint_normals = ((const rz_pbr_uniforms*)p->uniforms)->joint_n_matrices;
    }
}

/* Whether one more triangle that refers to joints [0, need) can join the draw `last` (the most
 * recent record; its uniform block already compared equal, so the matrices sit at the same
 * addresses): the matrices that draw snapshotted must still be what the host holds -- the skeleton
 * rewrites them in place (skeleton.c:336-337) without telling anybody.  A triangle that refers to
 * more joints than the draw has seen so far extends the snapshot, which is the last thing in the
 * side buffer. */
static bool same_pose(const rz_program *p, Pending &last, int need) {
    if (last.num_joints == 0) return true;          /* the draw is not skinned: its uniform block says all */
    const rz_mat4 *joints;
    const rz_mat3 *joint_normals;
    pose_of(p, &joints, &joint_normals);
    if (!joints) return false;
    const size_t n = (size_t)last.num_joints;
    if (memcmp(g.side.data() + last.joints_at, joints, sizeof(rz_mat4) * n) != 0) return false;
    if (joint_normals && memcmp(g.side.data() + last.joint_normals_at, joint_normals, sizeof(rz_mat3) * n) != 0) return false;
    if ((size_t)need > n) {
        g.side.resize(last.joints_at);
        g.side.insert(g.side.end(), (const char*)joints, (const char*)joints + sizeof(rz_mat4) * need);
        last.joint_normals_at = g.side.size();
        if (joint_normals)
            g.side.insert(g.side.end(), (const char*)joint_normals, (const char*)joint_normals + sizeof(rz_mat3) * need);
        else
            g.side.insert(g.side.end(), sizeof(rz_mat3) * need, 0);
        while (g.side.size() % 16) g.side.push_back(0);
        last.num_joints = need;
    }
    return true;
}

extern "C" void graphics_draw_mesh_posed(rz_framebuffer *fp, rz_program *program, const rz_vertex *vertices, int num_faces,
                                         rz_pose_source *source, float frame_time, int attached) {
    Fb *fb = as_fb(fp);
    as_pose_source(source);
    if (program->family == RZ_FAMILY_SKYBOX) fatal("graphics_draw_mesh_posed: the skybox family has no skeleton");
    if (attached >= source->num_joints) fatal("graphics_draw_mesh_posed: attached to joint %d of %d", attached, source->num_joints);
    if (num_faces <= 0) return;
    Pending pd;
    pd.fb = fb;
    pd.program = program;
    pd.transient = false;
    g_ignore_host_joints = true;
    snapshot_uniforms(program, pd, 0);
    g_ignore_host_joints = false;
    Mesh &m = device_mesh(vertices, num_faces);
    if (attached < 0 && m.num_joints > source->num_joints)
        fatal("graphics_draw_mesh_posed: the mesh refers to joint %d, the skeleton has %d", m.num_joints - 1, source->num_joints);
    pd.rec.vertices = m.d_vertices;
    pd.rec.num_faces = num_faces;
    pd.pose = source;
    pd.pose_time = wrapped_time(source, frame_time);
    pd.pose_attached = attached;
    if (attached < 0) pd.rec.flags |= RZ_DRAW_SKINNED;
    g.pending.push_back(std::move(pd));
}

/* one more triangle of a draw recorded triangle by triangle (the last one of the list) */
static void append_transient(Pending &pd, const rz_vertex *tri) {
    if (pd.replay) {
        const Replay &r = *pd.replay;
        const size_t have = (size_t)pd.rec.num_faces;
        if (have < r.faces && memcmp(r.host.data() + 3 * have, tri, 3 * sizeof(rz_vertex)) == 0) return;
        /* the stream departs from the remembered one: what repeated it so far is recorded after all */
        g.transient_verts.insert(g.transient_verts.end(), r.host.begin(), r.host.begin() + 3 * have);
        pd.verts_count = 3 * have;
        pd.replay = nullptr;
    }
    g.transient_verts.insert(g.transient_verts.end(), tri, tri + 3);
    pd.verts_count += 3;
}

static const size_t REPLAY_MIN_FACES = 32;
static std::vector<Replay*> g_replays_to_upload;

/* at the end of a flush, while the recorded vertices are still there: remember the stream of a draw that
 * repeated none, for the next frame */
static void remember_stream(Pending &pd) {
    if (pd.replay) { pd.replay->misses = 0; return; }
    if (!g.replay_streams || pd.verts_count < 3 * REPLAY_MIN_FACES) return;
    Replay &r = g.replays[pd.program];
    if (r.made_epoch == g.flush_epoch) return;          /* an earlier draw of this flush made it (shadow pass and main pass; several frames) */
    const rz_vertex *verts = g.transient_verts.data() + pd.verts_at;
    if (r.d_vertices && r.faces * 3 >= pd.verts_count && memcmp(r.host.data(), verts, pd.verts_count * sizeof(rz_vertex)) == 0) return;
    if (r.misses >= 4) return;                          /* this program's vertices change all the time */
    r.misses++;
    r.made_epoch = g.flush_epoch;
    r.host.assign(verts, verts + pd.verts_count);
    r.faces = pd.verts_count / 3;
    g_replays_to_upload.push_back(&r);
}

/* ... and, once the flush has been handed over, their copies on the device (passes in flight may still
 * draw from the copy that is replaced) */
static void upload_remembered_streams() {
    for (Replay *r : g_replays_to_upload) {
        if (r->d_vertices) { sync_stream(); HIP_OK(hipFree(r->d_vertices)); r->d_vertices = nullptr; }
        HIP_OK(hipMalloc(&r->d_vertices, r->host.size() * sizeof(rz_vertex)));
        HIP_OK(hipMemcpy(r->d_vertices, r->host.data(), r->host.size() * sizeof(rz_vertex), hipMemcpyHostToDevice));
    }
    g_replays_to_upload.clear();
}

/* graphics.c:537: one triangle from the program's three attribute blocks.
 * Consecutive calls with an unchanged uniform block -- and, for a skinned
 * program, unchanged joint matrices -- extend one draw. */
extern "C" void graphics_draw_triangle(rz_framebuffer *fp, rz_program *program) {
    Fb *fb = (Fb*)fp;
    if (fb != g.checked_fb || !fb) g.checked_fb = fb = as_fb(fp);
    rz_vertex built[3];
    const rz_vertex *tri = built;
    if (program->family == RZ_FAMILY_PBR && program->attribs[1] == (char*)program->attribs[0] + sizeof(rz_vertex) &&
        program->attribs[2] == (char*)program->attribs[0] + 2 * sizeof(rz_vertex)) {
        tri = (const rz_vertex*)program->attribs[0];    /* the three blocks are one allocation and already have the stream's layout */
    } else {
        for (int j = 0; j < 3; j++) {
            rz_vertex &v = built[j];
            if (program->family == RZ_FAMILY_BLINN) {
                const rz_blinn_attribs *a = (const rz_blinn_attribs*)program->attribs[j];
                memset(&v, 0, sizeof v);
                v.position = a->position; v.texcoord = a->texcoord; v.normal = a->normal;
                v.joint = a->joint; v.weight = a->weight;
            } else if (program->family == RZ_FAMILY_PBR) {
                v = *(const rz_vertex*)program->attribs[j];
            } else {
                memset(&v, 0, sizeof v);
                v.position = ((const rz_skybox_attribs*)program->attribs[j])->position;
            }
        }
    }
    /* joints a skinned program's triangle refers to (1 + the largest index): only such a program's draw keeps a pose */
    int joints_needed = 1;
    {
        const rz_mat4 *joints;
        const rz_mat3 *joint_normals;
        pose_of(program, &joints, &joint_normals);
        if (joints) {
            int top = 0;
            for (int j = 0; j < 3; j++) {
                const rz_vec4 &q = tri[j].joint;
                top = std::max(top, std::max(std::max((int)q.x, (int)q.y), std::max((int)q.z, (int)q.w)));
            }
            joints_needed = top + 1;
        }
    }
    const char *ub = (const char*)program->uniforms;
    if (!g.pending.empty()) {
        Pending &last = g.pending.back();
        const bool copy_between = !g.tex_writes.empty() && g.tex_writes.back().after == g.pending.size();
        if (!copy_between && last.transient && last.fb == fb && last.program == program &&
            last.uniform_bytes.size() == (size_t)program->uniform_bytes &&
            memcmp(last.uniform_bytes.data(), ub, program->uniform_bytes) == 0 &&
            same_pose(program, last, joints_needed)) {
            append_transient(last, tri);
            last.rec.num_faces++;
            return;
        }
    }
    Pending pd;
    pd.fb = fb;
    pd.program = program;
    pd.transient = true;
    snapshot_uniforms(program, pd, joints_needed);      /* may flush what is recorded: the draw's vertices are appended after it */
    pd.uniform_bytes.assign(ub, ub + program->uniform_bytes);
    pd.verts_at = g.transient_verts.size();
    pd.rec.num_faces = 0;
    if (g.replay_streams) {
        auto remembered = g.replays.find(program);
        if (remembered != g.replays.end() && remembered->second.d_vertices) pd.replay = &remembered->second;
    }
    append_transient(pd, tri);
    pd.rec.num_faces = 1;
    g.pending.push_back(std::move(pd));
}
