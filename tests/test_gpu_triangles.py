"""The reference's own entry points driven directly (program_create, program_get_attribs /
program_get_uniforms, graphics_draw_triangle), with hand-made triangles instead of scenes:
random overlapping triangles at equal depth (last writer wins), ordered blending over several
draws, triangles that need the clipper, degenerate and non-finite input, empty passes.
GPU library against the CPU oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd.binding import Vec4

pytestmark = pytest.mark.gpu

IDENTITY4 = (C.c_float * 16)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1)
IDENTITY3 = (C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)


class SkyboxUniforms(C.Structure):
    _fields_ = [("vp_matrix", C.c_float * 16), ("skybox", C.c_void_p)]


class BlinnUniforms(C.Structure):
    _fields_ = [("light_dir", C.c_float * 3), ("camera_pos", C.c_float * 3), ("model_matrix", C.c_float * 16),
                ("normal_matrix", C.c_float * 9), ("light_vp_matrix", C.c_float * 16), ("camera_vp_matrix", C.c_float * 16),
                ("joint_matrices", C.c_void_p), ("joint_n_matrices", C.c_void_p),
                ("ambient_intensity", C.c_float), ("punctual_intensity", C.c_float), ("shadow_map", C.c_void_p),
                ("basecolor", C.c_float * 4), ("shininess", C.c_float),
                ("diffuse_map", C.c_void_p), ("specular_map", C.c_void_p), ("emission_map", C.c_void_p),
                ("alpha_cutoff", C.c_float), ("shadow_pass", C.c_int)]


class BlinnAttribs(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("texcoord", C.c_float * 2), ("normal", C.c_float * 3),
                ("joint", C.c_float * 4), ("weight", C.c_float * 4)]


assert C.sizeof(SkyboxUniforms) == 72 and C.sizeof(BlinnUniforms) == 344 and C.sizeof(BlinnAttribs) == 64


def declare(lib):
    P = C.c_void_p
    lib.program_create.restype = P
    lib.program_create.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.program_release.argtypes = [P]
    lib.program_get_attribs.restype = P
    lib.program_get_attribs.argtypes = [P, C.c_int]
    lib.program_get_uniforms.restype = P
    lib.program_get_uniforms.argtypes = [P]
    lib.graphics_draw_triangle.argtypes = [P, P]
    lib.cubemap_from_files.restype = P
    lib.cubemap_from_files.argtypes = [C.c_char_p] * 6 + [C.c_int]
    lib.cubemap_release.argtypes = [P]


def fn(lib, name):
    return C.cast(getattr(lib, name), C.c_void_p)


def skybox_frame(pipe, size, triangles):
    """triangles: (n, 3, 3) float32 clip-space positions (w = 1 through an identity matrix)."""
    lib = pipe.lib
    declare(lib)
    faces = [("workshop/m1_%s.hdr" % f).encode() for f in ("px", "nx", "py", "ny", "pz", "nz")]
    cube = lib.cubemap_from_files(*faces, 0)
    prog = lib.program_create(fn(lib, "skybox_vertex_shader"), fn(lib, "skybox_fragment_shader"), 12, 12, 72, 1, 0)
    u = SkyboxUniforms.from_address(lib.program_get_uniforms(prog))
    u.vp_matrix[:] = IDENTITY4[:]
    u.skybox = cube
    fb = pipe.framebuffer(*size)
    lib.framebuffer_clear_color(fb, Vec4(0.1, 0.2, 0.3, 1.0))
    lib.framebuffer_clear_depth(fb, 1.0)
    for tri in triangles:
        for j in range(3):
            (C.c_float * 3).from_address(lib.program_get_attribs(prog, j))[:] = [float(x) for x in tri[j]]
        lib.graphics_draw_triangle(C.cast(fb, C.c_void_p), prog)
    out = pipe.read(fb)
    pipe.release(fb)
    lib.program_release(prog)
    lib.cubemap_release(cube)
    return out


def blinn_frame(pipe, size, triangles, colours, blend, depth_step=0.0):
    """One draw (own uniform block) per triangle: flat colours, optional alpha blending."""
    lib = pipe.lib
    declare(lib)
    prog = lib.program_create(fn(lib, "blinn_vertex_shader"), fn(lib, "blinn_fragment_shader"), 64, 44, 344, 1, 1 if blend else 0)
    u = BlinnUniforms.from_address(lib.program_get_uniforms(prog))
    u.light_dir[:] = [0.0, 0.0, -1.0]
    u.camera_pos[:] = [0.0, 0.0, 2.0]
    u.model_matrix[:] = IDENTITY4[:]
    u.normal_matrix[:] = IDENTITY3[:]
    u.light_vp_matrix[:] = IDENTITY4[:]
    u.camera_vp_matrix[:] = IDENTITY4[:]
    u.ambient_intensity = 0.7
    u.punctual_intensity = 0.6
    u.shininess = 12.0
    fb = pipe.framebuffer(*size)
    lib.framebuffer_clear_color(fb, Vec4(0.25, 0.25, 0.25, 1.0))
    lib.framebuffer_clear_depth(fb, 1.0)
    for i, (tri, colour) in enumerate(zip(triangles, colours)):
        u.basecolor[:] = [float(c) for c in colour]
        for j in range(3):
            a = BlinnAttribs.from_address(lib.program_get_attribs(prog, j))
            a.position[:] = [float(tri[j][0]), float(tri[j][1]), float(tri[j][2]) - depth_step * i]
            a.texcoord[:] = [0.5, 0.5]
            a.normal[:] = [0.0, 0.0, 1.0]
            a.joint[:] = [0, 0, 0, 0]
            a.weight[:] = [0, 0, 0, 0]
        lib.graphics_draw_triangle(C.cast(fb, C.c_void_p), prog)
    out = pipe.read(fb)
    pipe.release(fb)
    lib.program_release(prog)
    return out


def same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def random_triangles(seed, n, spread):
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-spread, spread, (n, 1, 3))
    tris = centre + rng.uniform(-0.6, 0.6, (n, 3, 3))
    tris[..., 2] = rng.uniform(-0.5, 0.5, (n, 3))
    return tris.astype(np.float32)


@pytest.mark.parametrize("size,n,spread", [((160, 90), 200, 0.9), ((333, 77), 64, 1.6), ((64, 64), 1000, 1.0)])
def test_equal_depth_overlap_last_writer_wins(gpu, oracle, size, n, spread):
    """The skybox vertex shader pins every depth to the same value: overlapping triangles are
    resolved purely by submission order (depth test <=, graphics.c:523)."""
    tris = random_triangles(11 + n, n, spread)
    assert same(skybox_frame(gpu, size, tris), skybox_frame(oracle, size, tris))


@pytest.mark.parametrize("size,n,extent", [((128, 96), 4000, 0.05), ((64, 64), 6000, 0.12), ((200, 40), 3000, 0.02)])
def test_many_small_triangles_at_equal_depth(gpu, oracle, size, n, extent):
    """Thousands of triangles of a few pixels each in a handful of bins, all at the same depth: the
    rasteriser takes them many at a time, one per lane, and settles every pixel by an atomic
    minimum over (depth, submission order) -- the last one submitted has to win exactly as in the
    reference's sequential loop, and triangles of one pixel or none must not disturb that."""
    rng = np.random.default_rng(100 + n)
    centre = rng.uniform(-0.9, 0.9, (n, 1, 3))
    tris = (centre + rng.uniform(-extent, extent, (n, 3, 3))).astype(np.float32)
    assert same(skybox_frame(gpu, size, tris), skybox_frame(oracle, size, tris))


def test_triangles_through_the_clipper(gpu, oracle):
    """Corners far outside the view volume in x, y and z, and behind w = 0 is impossible here
    (w = 1), so the +-x, +-y, +-z planes do the work (graphics.c:279-313)."""
    tris = random_triangles(5, 120, 2.5)
    tris[::3, :, 2] *= 6.0
    assert same(blinn_frame(gpu, (200, 120), tris, np.tile([[0.8, 0.3, 0.2, 1.0]], (120, 1)), False),
                blinn_frame(oracle, (200, 120), tris, np.tile([[0.8, 0.3, 0.2, 1.0]], (120, 1)), False))


def test_ordered_blending_over_many_draws(gpu, oracle):
    """Every triangle is its own draw with its own colour and alpha; the blend reads the 8-bit
    colour below it (graphics.c:455-463), so any reordering changes bytes."""
    n = 150
    rng = np.random.default_rng(21)
    tris = random_triangles(8, n, 0.8)
    colours = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.2, 0.9, (n, 1))], axis=1)
    for depth_step in (0.0, 0.001):     # equal depth, and each triangle slightly in front of the last
        assert same(blinn_frame(gpu, (192, 108), tris * [1, 1, 0], colours, True, depth_step),
                    blinn_frame(oracle, (192, 108), tris * [1, 1, 0], colours, True, depth_step))


def test_degenerate_and_non_finite_triangles(gpu, oracle):
    tris = random_triangles(3, 40, 0.7)
    tris[0, 1] = tris[0, 0]                     # two corners coincide
    tris[1] = tris[1, 0]                        # a point
    tris[2, 2] = (tris[2, 0] + tris[2, 1]) / 2  # collinear
    tris[3, 0, 0] = np.nan
    tris[4, 1, 1] = np.inf
    tris[5, :, :2] *= 1e20                      # far outside: every corner is clipped away or huge
    tris[6, :, :2] *= 1e-20                     # smaller than a pixel
    colours = np.tile([[0.2, 0.7, 0.4, 1.0]], (40, 1))
    assert same(blinn_frame(gpu, (128, 96), tris, colours, False), blinn_frame(oracle, (128, 96), tris, colours, False))


def test_empty_pass_and_clear_only(gpu, oracle):
    none = np.zeros((0, 3, 3), np.float32)
    a, b = skybox_frame(gpu, (70, 40), none), skybox_frame(oracle, (70, 40), none)
    assert same(a, b)
    assert np.all(a[0] == [25, 51, 76, 255]) and np.all(a[1] == 1.0)      # float_to_uchar truncates (maths.c:33)
    outside = random_triangles(1, 10, 0.3) + np.float32(5.0)             # all off screen
    assert same(skybox_frame(gpu, (70, 40), outside), b)


class Vertex(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("texcoord", C.c_float * 2), ("normal", C.c_float * 3),
                ("tangent", C.c_float * 4), ("joint", C.c_float * 4), ("weight", C.c_float * 4)]


class HostTexture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("buffer", C.POINTER(C.c_float))]


assert C.sizeof(Vertex) == 80

SHADER = C.CFUNCTYPE(None)


def test_gpu_invalidation_and_registered_shader_pairs(gpu, oracle):
    """The three calls a host needs when it goes around the recorded state: texture_invalidate after
    writing texture->buffer by hand, mesh_invalidate after changing a vertex array that was drawn
    with graphics_draw_mesh, and program_register_family for shader functions of its own (the
    library tells families by function pointer; the functions themselves are never called)."""
    w, h = 96, 64
    frames = []
    keep = []
    for pipe in (gpu, oracle):
        lib = pipe.lib
        declare(lib)
        P = C.c_void_p
        lib.texture_create.restype = C.POINTER(HostTexture)
        lib.texture_create.argtypes = [C.c_int, C.c_int]
        lib.texture_release.argtypes = [C.POINTER(HostTexture)]
        lib.texture_invalidate.argtypes = [C.POINTER(HostTexture)]
        lib.graphics_draw_mesh.argtypes = [P, P, P, C.c_int]
        lib.mesh_invalidate.argtypes = [P]
        lib.program_register_family.argtypes = [P, P, C.c_int]
        if pipe is gpu:
            # a shader pair of the host's own, registered as blinn (the GPU library never calls it)
            vs, fs = SHADER(lambda: None), SHADER(lambda: None)
            keep += [vs, fs]
            lib.program_register_family(C.cast(vs, P), C.cast(fs, P), 0)
            prog = lib.program_create(C.cast(vs, P), C.cast(fs, P), 64, 44, 344, 1, 0)
        else:
            # the oracle runs the shader functions it is given, like the reference: its own blinn pair
            prog = lib.program_create(fn(lib, "blinn_vertex_shader"), fn(lib, "blinn_fragment_shader"), 64, 44, 344, 1, 0)
        u = BlinnUniforms.from_address(lib.program_get_uniforms(prog))
        u.light_dir[:] = [0.0, 0.0, -1.0]
        u.camera_pos[:] = [0.0, 0.0, 2.0]
        u.model_matrix[:] = IDENTITY4[:]
        u.normal_matrix[:] = IDENTITY3[:]
        u.light_vp_matrix[:] = IDENTITY4[:]
        u.camera_vp_matrix[:] = IDENTITY4[:]
        u.ambient_intensity = 1.0
        u.punctual_intensity = 0.0
        u.shininess = 4.0
        u.basecolor[:] = [1.0, 1.0, 1.0, 1.0]
        tex = lib.texture_create(8, 8)
        texels = np.ctypeslib.as_array(tex.contents.buffer, (8, 8, 4))
        texels[...] = [0.25, 0.5, 0.75, 1.0]
        lib.texture_invalidate(tex)                 # texture_create's image was uploaded as zeros at most; this one counts
        u.diffuse_map = C.cast(tex, P)
        verts = (Vertex * 3)()
        for j, (x, y) in enumerate([(-0.8, -0.8), (0.8, -0.8), (0.0, 0.8)]):
            verts[j].position[:] = [x, y, 0.0]
            verts[j].texcoord[:] = [0.5, 0.5]
            verts[j].normal[:] = [0.0, 0.0, 1.0]
        fb = pipe.framebuffer(w, h)
        out = []
        lib.graphics_draw_mesh(C.cast(fb, P), prog, C.cast(verts, P), 1)
        out.append(pipe.read(fb))
        # new texels, same texture object
        texels[...] = [1.0, 0.125, 0.0, 1.0]
        lib.texture_invalidate(tex)
        lib.framebuffer_clear_color(fb, Vec4(0.0, 0.0, 0.0, 1.0))
        lib.framebuffer_clear_depth(fb, 1.0)
        lib.graphics_draw_mesh(C.cast(fb, P), prog, C.cast(verts, P), 1)
        out.append(pipe.read(fb))
        # new positions, same vertex array
        for j in range(3):
            verts[j].position[0] *= 0.5
            verts[j].position[1] *= -0.5
        lib.mesh_invalidate(C.cast(verts, P))
        lib.framebuffer_clear_color(fb, Vec4(0.0, 0.0, 0.0, 1.0))
        lib.framebuffer_clear_depth(fb, 1.0)
        lib.graphics_draw_mesh(C.cast(fb, P), prog, C.cast(verts, P), 1)
        out.append(pipe.read(fb))
        frames.append(out)
        pipe.release(fb)
        lib.program_release(prog)
        lib.texture_release(tex)
    got, want = frames
    for a, b in zip(got, want):
        assert same(a, b)
    covered = [int((f[1] < 1).sum()) for f in got]
    assert covered[0] == covered[1] > 500 and 0 < covered[2] < covered[0] // 2      # the mesh did shrink
    inside = got[0][1] < 1
    assert np.all(got[0][0][inside][:, :3] == [63, 127, 191]) and np.all(got[1][0][inside][:, :3] == [255, 31, 0])


class Cubemap(C.Structure):
    _fields_ = [("faces", C.POINTER(HostTexture) * 6)]


CUBE_CORNERS = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
CUBE_FACES = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]


def look_from_inside(aspect, yaw, pitch):
    """projection * rotation, row-major, for a camera at the cube's centre"""
    f = 1.0 / np.tan(np.radians(35.0))
    near, far = 0.1, 10.0
    proj = np.array([[f / aspect, 0, 0, 0], [0, f, 0, 0],
                     [0, 0, -(far + near) / (far - near), -2 * far * near / (far - near)], [0, 0, -1, 0]], np.float64)
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    rot_y = np.array([[cy, 0, sy, 0], [0, 1, 0, 0], [-sy, 0, cy, 0], [0, 0, 0, 1]])
    rot_x = np.array([[1, 0, 0, 0], [0, cp, -sp, 0], [0, sp, cp, 0], [0, 0, 0, 1]])
    return (proj @ rot_x @ rot_y).astype(np.float32)


@pytest.mark.parametrize("eight_bit", [False, True], ids=["float_faces", "eight_bit_faces"])
def test_skybox_of_caller_made_faces_replaced_between_frames(gpu, oracle, eight_bit):
    """A sky box seen from inside, its cube map made of six textures the caller fills by hand (values
    below 0 and above 1 among them: draw_fragment saturates, graphics.c:452), one opaque triangle in
    front of part of it, several views into framebuffers that are used again -- then two faces get new
    texels (texture_invalidate) and the views are drawn once more.  Whatever the library derives from
    a face's image has to follow the image."""
    size, views = (320, 200), [(0.3, 0.2), (2.1, -0.4), (4.0, 1.2), (5.5, -1.3)]
    rng = np.random.default_rng(77)
    if eight_bit:
        images = [rng.integers(0, 256, (32, 32, 4)).astype(np.float32) / np.float32(255) for _ in range(8)]
    else:
        images = [rng.uniform(-0.25, 1.5, (32, 32, 4)).astype(np.float32) for _ in range(8)]
    results = []
    for pipe in (gpu, oracle):
        lib = pipe.lib
        declare(lib)
        P = C.c_void_p
        lib.texture_create.restype = C.POINTER(HostTexture)
        lib.texture_create.argtypes = [C.c_int, C.c_int]
        lib.texture_release.argtypes = [C.POINTER(HostTexture)]
        lib.texture_invalidate.argtypes = [C.POINTER(HostTexture)]
        cube = Cubemap()
        texels = []
        for i in range(6):
            cube.faces[i] = lib.texture_create(32, 32)
            texels.append(np.ctypeslib.as_array(cube.faces[i].contents.buffer, (32, 32, 4)))
            texels[i][...] = images[i]
            lib.texture_invalidate(cube.faces[i])
        sky = lib.program_create(fn(lib, "skybox_vertex_shader"), fn(lib, "skybox_fragment_shader"), 12, 12, 72, 1, 0)
        su = SkyboxUniforms.from_address(lib.program_get_uniforms(sky))
        su.skybox = C.cast(C.pointer(cube), P)
        front = lib.program_create(fn(lib, "blinn_vertex_shader"), fn(lib, "blinn_fragment_shader"), 64, 44, 344, 1, 0)
        bu = BlinnUniforms.from_address(lib.program_get_uniforms(front))
        bu.light_dir[:] = [0.0, 0.0, -1.0]
        bu.camera_pos[:] = [0.0, 0.0, 2.0]
        bu.model_matrix[:] = IDENTITY4[:]
        bu.normal_matrix[:] = IDENTITY3[:]
        bu.light_vp_matrix[:] = IDENTITY4[:]
        bu.camera_vp_matrix[:] = IDENTITY4[:]
        bu.ambient_intensity = 1.0
        bu.punctual_intensity = 0.0
        bu.shininess = 4.0
        bu.basecolor[:] = [0.9, 0.4, 0.1, 1.0]
        fbs = [pipe.framebuffer(*size) for _ in views]

        def draw_all():
            for fb, (yaw, pitch) in zip(fbs, views):
                lib.framebuffer_clear_color(fb, Vec4(0.0, 0.0, 0.0, 1.0))
                lib.framebuffer_clear_depth(fb, 1.0)
                for j, (x, y) in enumerate([(-0.3, -0.2), (0.35, -0.25), (0.05, 0.4)]):
                    a = BlinnAttribs.from_address(lib.program_get_attribs(front, j))
                    a.position[:] = [x, y, 0.0]
                    a.texcoord[:] = [0.5, 0.5]
                    a.normal[:] = [0.0, 0.0, 1.0]
                    a.joint[:] = [0, 0, 0, 0]
                    a.weight[:] = [0, 0, 0, 0]
                lib.graphics_draw_triangle(C.cast(fb, P), front)
                su.vp_matrix[:] = [float(x) for x in look_from_inside(size[0] / size[1], yaw, pitch).reshape(-1)]
                for quad in CUBE_FACES:
                    for tri in ((quad[0], quad[1], quad[2]), (quad[0], quad[2], quad[3])):
                        for j in range(3):
                            (C.c_float * 3).from_address(lib.program_get_attribs(sky, j))[:] = [float(x) for x in CUBE_CORNERS[tri[j]]]
                        lib.graphics_draw_triangle(C.cast(fb, P), sky)
            return [pipe.read(fb) for fb in fbs]

        frames = draw_all()
        texels[0][...] = images[6]
        texels[3][...] = images[7]
        lib.texture_invalidate(cube.faces[0])
        lib.texture_invalidate(cube.faces[3])
        frames += draw_all()
        for fb in fbs:
            pipe.release(fb)
        lib.program_release(sky)
        lib.program_release(front)
        for i in range(6):
            lib.texture_release(cube.faces[i])
        results.append(frames)
    for got, want in zip(*results):
        assert same(got, want)
    n = len(views)
    assert any(not same(results[1][i], results[1][n + i]) for i in range(n))       # the new faces are in view somewhere
    assert all(int((f[1] < 0.9).sum()) > 1000 for f in results[0])                  # the triangle in front is there


@pytest.mark.parametrize("frames", [2, 12], ids=["two_frames", "twelve_frames"])
def test_triangles_that_reach_behind_the_sky(gpu, oracle, frames):
    """Opaque triangles whose depth runs from well in front of the sky box to beyond it (the sky's vertex shader
    puts it at z = w * (1 - EPSILON), skybox_shader.c:15, short of the far plane), then the sky box: drawn last, the
    sky takes every pixel where its depth is not greater (graphics.c:523), so the far part of each triangle is
    hidden behind it.  Several framebuffers in one flush: the library shades the sky and the triangles'
    fragments in different kernels, side by side when the flush is large enough, and a pixel both of them
    have a fragment for must end up with the sky's."""
    size = (448, 320)
    rng = np.random.default_rng(5)
    images = [rng.integers(0, 256, (32, 32, 4)).astype(np.float32) / np.float32(255) for _ in range(6)]
    views = [(0.5 * i, 0.2 * np.sin(i)) for i in range(frames)]
    results = []
    for pipe in (gpu, oracle):
        lib = pipe.lib
        declare(lib)
        P = C.c_void_p
        lib.texture_create.restype = C.POINTER(HostTexture)
        lib.texture_create.argtypes = [C.c_int, C.c_int]
        lib.texture_release.argtypes = [C.POINTER(HostTexture)]
        lib.texture_invalidate.argtypes = [C.POINTER(HostTexture)]
        cube = Cubemap()
        for i in range(6):
            cube.faces[i] = lib.texture_create(32, 32)
            np.ctypeslib.as_array(cube.faces[i].contents.buffer, (32, 32, 4))[...] = images[i]
            lib.texture_invalidate(cube.faces[i])
        sky = lib.program_create(fn(lib, "skybox_vertex_shader"), fn(lib, "skybox_fragment_shader"), 12, 12, 72, 1, 0)
        su = SkyboxUniforms.from_address(lib.program_get_uniforms(sky))
        su.skybox = C.cast(C.pointer(cube), P)
        front = lib.program_create(fn(lib, "blinn_vertex_shader"), fn(lib, "blinn_fragment_shader"), 64, 44, 344, 1, 0)
        bu = BlinnUniforms.from_address(lib.program_get_uniforms(front))
        bu.light_dir[:] = [0.0, 0.0, -1.0]
        bu.camera_pos[:] = [0.0, 0.0, 2.0]
        bu.model_matrix[:] = IDENTITY4[:]
        bu.normal_matrix[:] = IDENTITY3[:]
        bu.light_vp_matrix[:] = IDENTITY4[:]
        bu.camera_vp_matrix[:] = IDENTITY4[:]
        bu.ambient_intensity = 1.0
        bu.punctual_intensity = 0.0
        bu.shininess = 4.0
        fbs = [pipe.framebuffer(*size) for _ in views]
        for k, (fb, (yaw, pitch)) in enumerate(zip(fbs, views)):
            lib.framebuffer_clear_color(fb, Vec4(0.0, 0.0, 0.0, 1.0))
            lib.framebuffer_clear_depth(fb, 1.0)
            # clip space = position (identity matrices): z = 0.99999 is where the sky lies
            for t, (z_near, z_far) in enumerate([(0.2, 1.0), (0.9999, 1.0), (0.99998, 0.999995)]):
                bu.basecolor[:] = [0.9 - 0.3 * t, 0.3 + 0.2 * t, 0.1 + 0.4 * t, 1.0]
                corners = [(-0.9 + 0.1 * k / frames, -0.8 + 0.5 * t, z_near), (0.9, -0.7 + 0.5 * t, z_far), (-0.2, 0.1 + 0.4 * t, z_far)]
                for j, (x, y, z) in enumerate(corners):
                    a = BlinnAttribs.from_address(lib.program_get_attribs(front, j))
                    a.position[:] = [x, y, z]
                    a.texcoord[:] = [0.5, 0.5]
                    a.normal[:] = [0.0, 0.0, 1.0]
                    a.joint[:] = [0, 0, 0, 0]
                    a.weight[:] = [0, 0, 0, 0]
                lib.graphics_draw_triangle(C.cast(fb, P), front)
            su.vp_matrix[:] = [float(x) for x in look_from_inside(size[0] / size[1], yaw, pitch).reshape(-1)]
            for quad in CUBE_FACES:
                for tri in ((quad[0], quad[1], quad[2]), (quad[0], quad[2], quad[3])):
                    for j in range(3):
                        (C.c_float * 3).from_address(lib.program_get_attribs(sky, j))[:] = [float(x) for x in CUBE_CORNERS[tri[j]]]
                    lib.graphics_draw_triangle(C.cast(fb, P), sky)
        out = [pipe.read(fb) for fb in fbs]
        for fb in fbs:
            pipe.release(fb)
        lib.program_release(sky)
        lib.program_release(front)
        for i in range(6):
            lib.texture_release(cube.faces[i])
        results.append(out)
    for got, want in zip(*results):
        assert same(got, want)
    sky_depth = np.float32((1.0 - 1e-5 + 1.0) * 0.5)
    for colour, depth in results[1]:
        assert int((depth < 0.99).sum()) > 1000                     # the near parts of the triangles are there
        assert float(depth.max()) <= float(sky_depth) + 1e-6        # nothing shows from behind the sky


@pytest.mark.parametrize("frames", [2, 12], ids=["two_frames", "twelve_frames"])
def test_sky_over_unusual_clears(gpu, oracle, frames):
    """A triangle and the sky box into framebuffers that start differently: colour and depth cleared, depth cleared
    to 0.5 (nothing behind it shows, the sky least of all), colour cleared only, nothing cleared, and a second
    pass on top of a first without any clear between them.  Who carries a pending clear out where -- the bins'
    kernel, the strips, the sky -- differs from bin to bin and with the number of frames in the flush; the
    pictures must not."""
    size = (416, 288)
    rng = np.random.default_rng(15)
    images = [rng.integers(0, 256, (16, 16, 4)).astype(np.float32) / np.float32(255) for _ in range(6)]
    results = []
    for pipe in (gpu, oracle):
        lib = pipe.lib
        declare(lib)
        P = C.c_void_p
        lib.texture_create.restype = C.POINTER(HostTexture)
        lib.texture_create.argtypes = [C.c_int, C.c_int]
        lib.texture_release.argtypes = [C.POINTER(HostTexture)]
        lib.texture_invalidate.argtypes = [C.POINTER(HostTexture)]
        cube = Cubemap()
        for i in range(6):
            cube.faces[i] = lib.texture_create(16, 16)
            np.ctypeslib.as_array(cube.faces[i].contents.buffer, (16, 16, 4))[...] = images[i]
            lib.texture_invalidate(cube.faces[i])
        sky = lib.program_create(fn(lib, "skybox_vertex_shader"), fn(lib, "skybox_fragment_shader"), 12, 12, 72, 1, 0)
        su = SkyboxUniforms.from_address(lib.program_get_uniforms(sky))
        su.skybox = C.cast(C.pointer(cube), P)
        front = lib.program_create(fn(lib, "blinn_vertex_shader"), fn(lib, "blinn_fragment_shader"), 64, 44, 344, 1, 0)
        bu = BlinnUniforms.from_address(lib.program_get_uniforms(front))
        bu.light_dir[:] = [0.0, 0.0, -1.0]
        bu.camera_pos[:] = [0.0, 0.0, 2.0]
        bu.model_matrix[:] = IDENTITY4[:]
        bu.normal_matrix[:] = IDENTITY3[:]
        bu.light_vp_matrix[:] = IDENTITY4[:]
        bu.camera_vp_matrix[:] = IDENTITY4[:]
        bu.ambient_intensity = 1.0
        bu.punctual_intensity = 0.0
        bu.shininess = 4.0
        fbs = [pipe.framebuffer(*size) for _ in range(frames)]

        def one_pass(fb, k, shift):
            bu.basecolor[:] = [0.2 + 0.05 * (k % 7), 0.8 - 0.1 * shift, 0.3, 1.0]
            for j, (x, y, z) in enumerate([(-0.8 + shift, -0.7, 0.1), (0.6 + shift, -0.5, 0.7), (-0.1 + shift, 0.8, 1.0)]):
                a = BlinnAttribs.from_address(lib.program_get_attribs(front, j))
                a.position[:] = [x, y, z]
                a.texcoord[:] = [0.5, 0.5]
                a.normal[:] = [0.0, 0.0, 1.0]
                a.joint[:] = [0, 0, 0, 0]
                a.weight[:] = [0, 0, 0, 0]
            lib.graphics_draw_triangle(C.cast(fb, P), front)
            su.vp_matrix[:] = [float(x) for x in look_from_inside(size[0] / size[1], 0.7 * k, 0.3 * np.cos(k)).reshape(-1)]
            for quad in CUBE_FACES:
                for tri in ((quad[0], quad[1], quad[2]), (quad[0], quad[2], quad[3])):
                    for j in range(3):
                        (C.c_float * 3).from_address(lib.program_get_attribs(sky, j))[:] = [float(x) for x in CUBE_CORNERS[tri[j]]]
                    lib.graphics_draw_triangle(C.cast(fb, P), sky)

        for k, fb in enumerate(fbs):
            kind = k % 5
            if kind in (0, 4):
                lib.framebuffer_clear_color(fb, Vec4(0.3, 0.1, 0.6, 1.0))
                lib.framebuffer_clear_depth(fb, 1.0)
            elif kind == 1:
                lib.framebuffer_clear_depth(fb, 0.5)
            elif kind == 2:
                lib.framebuffer_clear_color(fb, Vec4(0.9, 0.9, 0.1, 0.5))
            one_pass(fb, k, 0.0)
            if kind == 4:
                one_pass(fb, k + 3, 0.25)           # a second pass on top: no clear
        out = [pipe.read(fb) for fb in fbs]
        for fb in fbs:
            pipe.release(fb)
        lib.program_release(sky)
        lib.program_release(front)
        for i in range(6):
            lib.texture_release(cube.faces[i])
        results.append(out)
    for got, want in zip(*results):
        assert same(got, want)
    assert float(results[1][1][1].max()) <= 0.5                     # behind a depth of 0.5 nothing shows


@pytest.mark.parametrize("blend", [0, 1], ids=["opaque", "second_blended"])
def test_two_sky_boxes_in_one_pass(gpu, oracle, blend):
    """A sky box from inside, then triangles of a SECOND sky program (another cube map) over parts of it, all
    in one pass and at the sky's one depth, so the later draw wins wherever it lies (graphics.c:523).  Bins
    whose lists hold entries of both draws and bins that hold one draw's entries only sit side by side; the
    library takes them through different kernels, the frames must not show where.  With the second program
    blending (its texels' alpha over the first sky's bytes, graphics.c:455-463) the pass is cut in two and the
    second sky is shaded from float texels inside the ordered kernel."""
    size, views = (384, 256), [(0.4, 0.1), (3.3, -0.7)]
    rng = np.random.default_rng(91)
    first = [rng.integers(0, 256, (64, 64, 4)).astype(np.float32) / np.float32(255) for _ in range(6)]
    second = [rng.uniform(0.0, 1.2, (16, 16, 4)).astype(np.float32) for _ in range(6)]
    results = []
    for pipe in (gpu, oracle):
        lib = pipe.lib
        declare(lib)
        P = C.c_void_p
        lib.texture_create.restype = C.POINTER(HostTexture)
        lib.texture_create.argtypes = [C.c_int, C.c_int]
        lib.texture_release.argtypes = [C.POINTER(HostTexture)]
        lib.texture_invalidate.argtypes = [C.POINTER(HostTexture)]
        cubes, programs = [], []
        for images in (first, second):
            cube = Cubemap()
            for i, image in enumerate(images):
                cube.faces[i] = lib.texture_create(image.shape[1], image.shape[0])
                np.ctypeslib.as_array(cube.faces[i].contents.buffer, image.shape)[...] = image
                lib.texture_invalidate(cube.faces[i])
            prog = lib.program_create(fn(lib, "skybox_vertex_shader"), fn(lib, "skybox_fragment_shader"), 12, 12, 72, 1,
                                      blend if images is second else 0)
            SkyboxUniforms.from_address(lib.program_get_uniforms(prog)).skybox = C.cast(C.pointer(cube), P)
            cubes.append(cube)
            programs.append(prog)
        frames = []
        for yaw, pitch in views:
            fb = pipe.framebuffer(*size)
            lib.framebuffer_clear_color(fb, Vec4(0.2, 0.2, 0.2, 1.0))
            lib.framebuffer_clear_depth(fb, 1.0)
            vp = [float(x) for x in look_from_inside(size[0] / size[1], yaw, pitch).reshape(-1)]
            for prog, quads in ((programs[0], CUBE_FACES), (programs[1], CUBE_FACES[:2])):
                SkyboxUniforms.from_address(lib.program_get_uniforms(prog)).vp_matrix[:] = vp
                for quad in quads:
                    for tri in ((quad[0], quad[1], quad[2]), (quad[0], quad[2], quad[3])):
                        for j in range(3):
                            corner = CUBE_CORNERS[tri[j]] * (np.float32(1.0) if prog == programs[0] else np.float32(0.6))
                            (C.c_float * 3).from_address(lib.program_get_attribs(prog, j))[:] = [float(x) for x in corner]
                        lib.graphics_draw_triangle(C.cast(fb, P), prog)
            frames.append(pipe.read(fb))
            pipe.release(fb)
        for prog in programs:
            lib.program_release(prog)
        for cube in cubes:
            for i in range(6):
                lib.texture_release(cube.faces[i])
        results.append(frames)
    for got, want in zip(*results):
        assert same(got, want)


def test_streams_that_change_between_frames(gpu, oracle):
    """One program, frame after frame: the same triangles (drawn from the remembered stream), a prefix of
    them, the same with one triangle changed in the middle, more triangles than remembered, and entirely
    different ones several times in a row -- every frame against the oracle."""
    rng = np.random.default_rng(77)
    base = rng.uniform(-0.9, 0.9, (200, 3, 3)).astype(np.float32)
    changed = base.copy()
    changed[120] = rng.uniform(-0.9, 0.9, (3, 3))
    longer = np.concatenate([base, rng.uniform(-0.9, 0.9, (40, 3, 3)).astype(np.float32)])
    streams = [base, base, base[:150], changed, changed, longer, longer[:10], longer]
    streams += [rng.uniform(-0.9, 0.9, (64, 3, 3)).astype(np.float32) for _ in range(6)]
    streams += [base, base]
    frames = []
    for pipe in (gpu, oracle):
        lib = pipe.lib
        declare(lib)
        prog = lib.program_create(fn(lib, "blinn_vertex_shader"), fn(lib, "blinn_fragment_shader"), 64, 44, 344, 1, 0)
        u = BlinnUniforms.from_address(lib.program_get_uniforms(prog))
        u.light_dir[:] = [0.0, 0.0, -1.0]
        u.camera_pos[:] = [0.0, 0.0, 2.0]
        u.model_matrix[:] = IDENTITY4[:]
        u.normal_matrix[:] = IDENTITY3[:]
        u.light_vp_matrix[:] = IDENTITY4[:]
        u.camera_vp_matrix[:] = IDENTITY4[:]
        u.ambient_intensity = 0.7
        u.punctual_intensity = 0.6
        u.shininess = 12.0
        out = []
        for n, stream in enumerate(streams):
            fb = pipe.framebuffer(160, 96)
            u.basecolor[:] = [0.2 + 0.05 * (n % 10), 0.9 - 0.04 * (n % 7), 0.5, 1.0]
            for tri in stream:
                for j in range(3):
                    a = BlinnAttribs.from_address(lib.program_get_attribs(prog, j))
                    a.position[:] = [float(x) for x in tri[j]]
                    a.texcoord[:] = [0.5, 0.5]
                    a.normal[:] = [0.0, 0.0, 1.0]
                    a.joint[:] = [0, 0, 0, 0]
                    a.weight[:] = [0, 0, 0, 0]
                lib.graphics_draw_triangle(C.cast(fb, C.c_void_p), prog)
            out.append(pipe.read(fb))
            pipe.release(fb)
        lib.program_release(prog)
        frames.append(out)
    if gpu.is_gpu:
        gpu.lib.mi355r_replayed_draws.restype = C.c_ulonglong
        assert gpu.lib.mi355r_replayed_draws() > 0
    for n, (g, o) in enumerate(zip(*frames)):
        assert same(g, o), n
