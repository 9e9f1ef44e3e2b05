"""Inputs that outgrow the pipeline's scratch estimates, and resources that are replaced while frames
are in flight.  The reference renders any input (graphics.c:537-608 has no capacity anywhere); the
GPU library sizes the clipper's block table and the bin lists of a pass by an estimate, and a pass
that needs more is run again with what it reported -- so the frames must still equal the oracle's,
bit for bit, and nothing may abort."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import golden_cases, load_golden, render_case
from renderer_amd.binding import SCENES, Vec4
from test_gpu_parity import assert_same
from test_gpu_triangles import (IDENTITY3, IDENTITY4, BlinnUniforms, HostTexture, Vertex, blinn_frame, declare, fn,
                                random_triangles, same)

pytestmark = pytest.mark.gpu


def test_full_screen_blended_triangles_beyond_the_list_estimate(gpu, oracle):
    """300 blended triangles whose boxes cover a 2048x2048 framebuffer: 300 x 1024 bins = 307 200 list
    entries, more than the first estimate (max(16 per triangle, 262 144)).  Forty of them fill the
    screen, the rest are long slivers (a full-screen box, few pixels)."""
    n = 300
    rng = np.random.default_rng(4)
    tris = np.zeros((n, 3, 3), np.float32)
    for i in range(n):
        if i % 8 == 0:          # covers the whole view (clipped to it)
            tris[i] = [[-3.0, -1.0, 0.0], [1.0, -1.0, 0.0], [1.0, 3.0, 0.0]] if i % 16 else [[-1.0, -1.0, 0.0], [3.0, -1.0, 0.0], [-1.0, 3.0, 0.0]]
        else:                   # corner to corner, a few pixels wide
            t = rng.uniform(0.002, 0.02)
            s = 1 if i % 2 else -1
            tris[i] = [[-0.999, -0.999 * s, 0.0], [0.999, 0.999 * s, 0.0], [0.999 - t, (0.999 - 4 * t) * s, 0.0]]
    colours = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.05, 0.6, (n, 1))], axis=1)
    before = gpu.lib.mi355r_passes_run_again()
    got = blinn_frame(gpu, (2048, 2048), tris, colours, True)
    assert gpu.lib.mi355r_passes_run_again() > before       # the estimate was too small, the pass ran again
    want = blinn_frame(oracle, (2048, 2048), tris, colours, True)
    assert same(got, want)
    # the library remembers what that pass needed: the same frame again fits at once
    before = gpu.lib.mi355r_passes_run_again()
    assert same(blinn_frame(gpu, (2048, 2048), tris, colours, True), want)
    assert gpu.lib.mi355r_passes_run_again() == before


def _mesh_frame(pipe, size, verts, faces):
    lib = pipe.lib
    declare(lib)
    P = C.c_void_p
    lib.graphics_draw_mesh.argtypes = [P, P, P, C.c_int]
    lib.mesh_invalidate.argtypes = [P]
    prog = lib.program_create(fn(lib, "blinn_vertex_shader"), fn(lib, "blinn_fragment_shader"), 64, 44, 344, 1, 0)
    u = BlinnUniforms.from_address(lib.program_get_uniforms(prog))
    u.light_dir[:] = [0.0, 0.0, -1.0]
    u.camera_pos[:] = [0.0, 0.0, 2.0]
    u.model_matrix[:] = IDENTITY4[:]
    u.normal_matrix[:] = IDENTITY3[:]
    u.light_vp_matrix[:] = IDENTITY4[:]
    u.camera_vp_matrix[:] = IDENTITY4[:]
    u.ambient_intensity = 0.9
    u.punctual_intensity = 0.3
    u.shininess = 8.0
    u.basecolor[:] = [0.9, 0.6, 0.3, 1.0]
    fb = pipe.framebuffer(*size)
    lib.framebuffer_clear_color(fb, Vec4(0.0, 0.1, 0.2, 1.0))
    lib.framebuffer_clear_depth(fb, 1.0)
    lib.graphics_draw_mesh(C.cast(fb, P), prog, verts.ctypes.data_as(P), faces)
    out = pipe.read(fb)
    lib.mesh_invalidate(verts.ctypes.data_as(P))
    pipe.release(fb)
    lib.program_release(prog)
    return out


def test_more_clipped_triangles_than_clipper_blocks(gpu, oracle):
    """270 000 small triangles that all straddle the right or the top edge of the view volume: every
    one of them goes through the clipper and takes a block of its table, whose first estimate is
    2^18 = 262 144 blocks."""
    faces = 270_000
    rng = np.random.default_rng(9)
    verts = np.zeros((faces * 3, 20), np.float32)            # rz_vertex: position 0-2, texcoord 3-4, normal 5-7, ...
    along = rng.uniform(-0.95, 0.95, faces).astype(np.float32)
    size = rng.uniform(0.01, 0.05, faces).astype(np.float32)
    on_top = rng.integers(0, 2, faces).astype(bool)
    depth = rng.uniform(-0.5, 0.5, faces).astype(np.float32)
    corner = np.zeros((faces, 3, 3), np.float32)
    # inside, inside, outside (x or y beyond 1)
    corner[:, 0] = np.stack([np.where(on_top, along, 1 - size), np.where(on_top, 1 - size, along), depth], 1)
    corner[:, 1] = np.stack([np.where(on_top, along + size, 1 - size), np.where(on_top, 1 - size, along + size), depth], 1)
    corner[:, 2] = np.stack([np.where(on_top, along, 1 + size), np.where(on_top, 1 + size, along), depth], 1)
    verts[:, 0:3] = corner.reshape(-1, 3)
    verts[:, 3:5] = 0.5
    verts[:, 7] = 1.0
    assert verts.nbytes == faces * 3 * C.sizeof(Vertex)
    before = gpu.lib.mi355r_passes_run_again()
    got = _mesh_frame(gpu, (96, 96), verts, faces)
    assert gpu.lib.mi355r_passes_run_again() > before
    assert same(got, _mesh_frame(oracle, (96, 96), verts, faces))


@pytest.fixture
def tiny_scratch(gpu):
    gpu.lib.mi355r_set_scratch_estimate(2, 48)
    yield gpu
    gpu.lib.mi355r_set_scratch_estimate(0, 0)


@pytest.mark.parametrize("case", [c for c in golden_cases() if c["name"] in
                                  ("helmet_inside", "kgirl_blend", "azura_shadow", "dieselpunk_shadow", "synthetic_blinn_low")],
                         ids=lambda c: c["name"])
def test_goldens_with_scratch_estimates_that_are_always_too_small(tiny_scratch, case):
    """Two clipper blocks and 48 list entries per pass to begin with: every pass of every frame runs out,
    passes behind it in the flush (the blended pass of the frame, the main pass that samples the
    shadow map the shadow pass was copied into) are left out on the device, and all of them run
    again in order -- with the recorded framebuffer -> texture copies between them."""
    gpu = tiny_scratch
    before = gpu.lib.mi355r_passes_run_again()
    got = render_case(gpu, case)
    assert gpu.lib.mi355r_passes_run_again() > before
    assert_same(got, load_golden(case["name"]), got[1].size)


def test_many_frames_in_one_flush_with_scratch_that_is_too_small(tiny_scratch, oracle):
    """24 shadowed + blended frames recorded together (three levels of passes, copies recorded between
    them), all of them out of scratch on the first attempt; and a second flush on top of the first
    without a read in between."""
    gpu = tiny_scratch
    path = os.path.join(SCENES, "kgirl_blend.scn")
    size, n = (160, 90), 24
    scene = gpu.open_scene(path, "kgirl")
    views = [gpu.orbit_view(i, n) for i in range(n)]
    fbs = [gpu.framebuffer(*size) for _ in range(n)]
    for fb, v in zip(fbs, views):
        gpu.draw(scene, fb, v)
    gpu.lib.graphics_flush()
    for fb, v in zip(fbs[:4], views[4:8]):              # frames 4..7 once more, into framebuffers 0..3
        gpu.draw(scene, fb, v)
    got = [gpu.read(fb) for fb in fbs]
    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    scene = oracle.open_scene(path, "kgirl")
    for i, view in [(0, 4), (3, 7), (4, 4), (11, 11), (23, 23)]:
        assert_same(got[i], oracle.render(scene, size[0], size[1], oracle.orbit_view(view, n)), size[0] * size[1])
    oracle.close_scene(scene)


def _textured_triangle_setup(pipe, texels_rgba):
    lib = pipe.lib
    declare(lib)
    P = C.c_void_p
    lib.texture_create.restype = C.POINTER(HostTexture)
    lib.texture_create.argtypes = [C.c_int, C.c_int]
    lib.texture_release.argtypes = [C.POINTER(HostTexture)]
    lib.texture_invalidate.argtypes = [C.POINTER(HostTexture)]
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
    tex = lib.texture_create(256, 256)
    texels = np.ctypeslib.as_array(tex.contents.buffer, (256, 256, 4))
    texels[...] = texels_rgba
    lib.texture_invalidate(tex)
    u.diffuse_map = C.cast(tex, P)
    return prog, tex, texels


def _draw_textured(pipe, prog, fb, scale):
    from test_gpu_triangles import BlinnAttribs
    lib = pipe.lib
    for tri in ([(-1, -1), (1, -1), (1, 1)], [(-1, -1), (1, 1), (-1, 1)]):
        for j, (x, y) in enumerate(tri):
            a = BlinnAttribs.from_address(lib.program_get_attribs(prog, j))
            a.position[:] = [x * scale, y * scale, 0.0]
            a.texcoord[:] = [(x + 1) / 2, (y + 1) / 2]
            a.normal[:] = [0.0, 0.0, 1.0]
            a.joint[:] = [0, 0, 0, 0]
            a.weight[:] = [0, 0, 0, 0]
        lib.graphics_draw_triangle(C.cast(fb, C.c_void_p), prog)


def test_texture_replaced_while_frames_that_sample_it_are_in_flight(gpu, oracle):
    """Frames into several framebuffers (several lanes), then the texture's host image is rewritten and
    invalidated WITHOUT a read in between, then more frames: the first set must show the old image,
    the second set the new one -- the upload of the new image waits for the passes that still sample
    the old one."""
    size, n = (512, 512), 8
    rng = np.random.default_rng(2)
    first = rng.integers(0, 256, (256, 256, 4)).astype(np.float32) / np.float32(255)
    second = rng.integers(0, 256, (256, 256, 4)).astype(np.float32) / np.float32(255)
    results = []
    for pipe in (gpu, oracle):
        prog, tex, texels = _textured_triangle_setup(pipe, first)
        old = [pipe.framebuffer(*size) for _ in range(n)]
        new = [pipe.framebuffer(*size) for _ in range(n)]
        for rep in range(3):                            # the GPU is busy with earlier flushes when the image is replaced
            for i, fb in enumerate(old):
                _draw_textured(pipe, prog, fb, 0.5 + 0.05 * i)
            pipe.lib.graphics_flush()
        texels[...] = second
        pipe.lib.texture_invalidate(tex)
        for i, fb in enumerate(new):
            _draw_textured(pipe, prog, fb, 0.5 + 0.05 * i)
        frames = [pipe.read(fb) for fb in old + new]
        for fb in old + new:
            pipe.release(fb)
        pipe.lib.program_release(prog)
        pipe.lib.texture_release(tex)
        results.append(frames)
    for got, want in zip(*results):
        assert same(got, want)
    assert not same(results[1][0], results[1][n])       # the two images do differ


def test_thousands_of_small_flushes_without_a_full_synchronisation(gpu, oracle):
    """A viewer's loop over a shadowed scene: draw, read the colour buffer, again -- framebuffer_read
    drains one lane only, so the events the flushes create pile up until the library bounds them, which
    it may only do at a point where no event is still waited for."""
    size = (64, 48)
    scene = gpu.open_scene("craftsman")
    fbs = [gpu.framebuffer(*size) for _ in range(3)]
    frames = 2600
    last = {}
    for i in range(frames):
        fb = fbs[i % 3]
        gpu.draw(scene, fb, gpu.orbit_view(i % 16, 16))
        if i % 3 == 2 or i >= frames - 3:
            last[i % 3] = (i % 16, gpu.read(fb))
    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    scene = oracle.open_scene("craftsman")
    for view, got in last.values():
        assert_same(got, oracle.render(scene, size[0], size[1], oracle.orbit_view(view, 16)), size[0] * size[1])
    oracle.close_scene(scene)


@pytest.mark.parametrize("tiny", [False, True], ids=["ordinary", "scratch-too-small"])
def test_frames_copied_out_in_two_halves(gpu, tiny):
    """framebuffers_read_device_rgb_begin / _end: the copies of a step's frames are launched behind the rendering
    and waited for later; in between the next step is recorded and submitted into the SAME framebuffers.  What
    arrives is the first step's frames, the framebuffers end up with the second's.  With scratch estimates that
    are always too small the first step's passes do not run when submitted: they run again when the flush is
    settled (before the next one's first launch), and the copies are launched again behind them."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    w, h, count = 320, 180, 6
    stride = w * h * 3
    scene = gpu.open_scene("helmet")
    fbs = [gpu.framebuffer(w, h) for _ in range(count)]
    handles = (type(fbs[0]) * count)(*fbs)
    first = [gpu.orbit_view(i, 16) for i in range(count)]
    second = [gpu.orbit_view(i + 8, 16) for i in range(count)]
    want = []
    for views in (first, second):
        for fb, v in zip(fbs, views):
            gpu.draw(scene, fb, v)
        want.append([gpu.read(fb)[0][..., :3].copy() for fb in fbs])
    assert any(not np.array_equal(a, b) for a, b in zip(*want))
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), stride * count) == 0 and hip.hipMemset(dev, 5, stride * count) == 0
    if tiny:
        gpu.lib.mi355r_set_scratch_estimate(2, 48)
    try:
        before = gpu.lib.mi355r_passes_run_again()
        for rounds in range(2):                         # twice: the second set begins while nothing else is in flight
            for fb, v in zip(fbs, first):
                gpu.draw(scene, fb, v)
            gpu.lib.framebuffers_read_device_rgb_begin(handles, count, dev, stride)
            for fb, v in zip(fbs, second):
                gpu.draw(scene, fb, v)
            gpu.lib.graphics_flush()
            if rounds:
                import time
                time.sleep(0.05)        # the second step's passes have finished (or reported that they did not run) when _end looks
            gpu.lib.framebuffers_read_device_rgb_end()
            got = np.zeros(stride * count, np.uint8)
            assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dev, stride * count, 2) == 0
            got = got.reshape(count, h, w, 3)
            for i, fb in enumerate(fbs):
                assert np.array_equal(got[i], want[0][i]), (rounds, i)
                assert np.array_equal(gpu.read(fb)[0][..., :3], want[1][i]), (rounds, i)
            assert hip.hipMemset(dev, 5, stride * count) == 0
        # begin and end with nothing submitted in between
        for fb, v in zip(fbs, first):
            gpu.draw(scene, fb, v)
        gpu.lib.framebuffers_read_device_rgb_begin(handles, count, dev, stride)
        gpu.lib.framebuffers_read_device_rgb_end()
        got = np.zeros(stride * count, np.uint8)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dev, stride * count, 2) == 0
        assert all(np.array_equal(got.reshape(count, h, w, 3)[i], want[0][i]) for i in range(count))
        if tiny:
            assert gpu.lib.mi355r_passes_run_again() > before
    finally:
        gpu.lib.mi355r_set_scratch_estimate(0, 0)
        hip.hipFree(dev)
        for fb in fbs:
            gpu.release(fb)
        gpu.close_scene(scene)


def test_clears_asked_for_while_copies_are_outstanding(gpu):
    """A render loop clears its framebuffers for the next step while the copies of this step are outstanding
    (framebuffers_read_device_rgb_begin .. _end).  Clears are flags until a pass consumes them.  When the step's
    passes have to run again (scratch too small) and the flush is settled by a full synchronisation BEFORE the
    next step is submitted, the copies are launched again behind the second run: they must carry this step's
    frames, not the colour the framebuffers are about to be cleared to -- and the clears must still happen."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    w, h, count = 320, 180, 4
    stride = w * h * 3
    scene = gpu.open_scene("helmet")
    fbs = [gpu.framebuffer(w, h) for _ in range(count)]
    handles = (type(fbs[0]) * count)(*fbs)
    views = [gpu.orbit_view(i, 16) for i in range(count)]
    for fb, v in zip(fbs, views):
        gpu.draw(scene, fb, v)
    want = [gpu.read(fb)[0][..., :3].copy() for fb in fbs]
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), stride * count) == 0 and hip.hipMemset(dev, 5, stride * count) == 0
    gpu.lib.mi355r_set_scratch_estimate(2, 48)
    try:
        for fb, v in zip(fbs, views):
            gpu.draw(scene, fb, v)
        gpu.lib.framebuffers_read_device_rgb_begin(handles, count, dev, stride)
        for fb in fbs:                                  # the next step's clears: flags only
            gpu.lib.framebuffer_clear_color(fb, Vec4(1.0, 0.0, 0.0, 1.0))
        before = gpu.lib.mi355r_passes_run_again()      # a full synchronisation: settles the flush, runs its passes again
        assert before > 0
        gpu.lib.framebuffers_read_device_rgb_end()
        got = np.zeros(stride * count, np.uint8)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dev, stride * count, 2) == 0
        got = got.reshape(count, h, w, 3)
        for i in range(count):
            assert np.array_equal(got[i], want[i]), i
        # the clears are still pending and happen when the framebuffer is looked at
        colour, _ = gpu.read(fbs[0])
        assert np.all(colour[..., :3] == [255, 0, 0])
    finally:
        gpu.lib.mi355r_set_scratch_estimate(0, 0)
        hip.hipFree(dev)
        for fb in fbs:
            gpu.release(fb)
        gpu.close_scene(scene)


def _blended_frames(pipe, size, sets):
    """Several framebuffers, each with its own list of (triangle, colour) blended draws, all recorded
    before any is read: their passes share one flush and, being of one kind and level, one batch."""
    lib = pipe.lib
    declare(lib)
    prog = lib.program_create(fn(lib, "blinn_vertex_shader"), fn(lib, "blinn_fragment_shader"), 64, 44, 344, 1, 1)
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
    fbs = []
    for triangles, colours in sets:
        fb = pipe.framebuffer(*size)
        fbs.append(fb)
        lib.framebuffer_clear_color(fb, Vec4(0.25, 0.25, 0.25, 1.0))
        lib.framebuffer_clear_depth(fb, 1.0)
    # a first, opaque-free flush consumes the clears: the blended passes below carry none
    lib.graphics_flush()
    for fb in fbs:
        pipe.read(fb)
    for fb, (triangles, colours) in zip(fbs, sets):
        for tri, colour in zip(triangles, colours):
            u.basecolor[:] = [float(c) for c in colour]
            for j in range(3):
                a = BlinnAttribs_of(lib, prog, j)
                a.position[:] = [float(tri[j][0]), float(tri[j][1]), float(tri[j][2])]
                a.texcoord[:] = [0.5, 0.5]
                a.normal[:] = [0.0, 0.0, 1.0]
                a.joint[:] = [0, 0, 0, 0]
                a.weight[:] = [0, 0, 0, 0]
            lib.graphics_draw_triangle(C.cast(fb, C.c_void_p), prog)
    lib.graphics_flush()
    out = [pipe.read(fb) for fb in fbs]
    for fb in fbs:
        pipe.release(fb)
    lib.program_release(prog)
    return out


def BlinnAttribs_of(lib, prog, j):
    from test_gpu_triangles import BlinnAttribs
    return BlinnAttribs.from_address(lib.program_get_attribs(prog, j))


def test_two_blended_passes_of_one_batch_where_one_needs_a_third_run(gpu, oracle):
    """Forced estimates of 2 clipper blocks and 48 list entries.  Framebuffer A gets blended triangles that fit the
    clipper but not the lists: its pass fails once and succeeds when run again.  Framebuffer B gets sixty blended
    triangles that all need the clipper: its first run counts list entries only for the two triangles it had blocks
    for, so its second run fails on the lists and it needs a third.  Both passes sit in one batch; the third run must
    not render A's pass once more -- a blended pass carries no clear, it would be blended twice."""
    rng = np.random.default_rng(31)
    size = (512, 512)
    # A: two view-filling triangles (64 bins each) and a few small ones, all inside the view volume
    a_tris = [np.array([[-1.0, -1.0, 0.0], [1.0, -1.0, 0.0], [1.0, 1.0, 0.0]], np.float32),
              np.array([[-1.0, -1.0, 0.1], [1.0, 1.0, 0.1], [-1.0, 1.0, 0.1]], np.float32)]
    a_tris += list(random_triangles(6, 6, 0.3) * np.float32(0.5))
    a_cols = np.concatenate([rng.uniform(0, 1, (len(a_tris), 3)), rng.uniform(0.3, 0.7, (len(a_tris), 1))], axis=1)
    # B: every triangle has a corner outside the view volume
    b_tris = random_triangles(7, 60, 0.8)
    b_tris[:, 2, 0] = 1.5
    b_tris[..., 2] *= 0.5
    b_cols = np.concatenate([rng.uniform(0, 1, (60, 3)), rng.uniform(0.3, 0.7, (60, 1))], axis=1)
    sets = [(a_tris, a_cols), (b_tris, b_cols)]
    want = _blended_frames(oracle, size, sets)
    gpu.lib.mi355r_set_scratch_estimate(2, 48)
    try:
        before = gpu.lib.mi355r_passes_run_again()
        got = _blended_frames(gpu, size, sets)
        assert gpu.lib.mi355r_passes_run_again() >= before + 2
    finally:
        gpu.lib.mi355r_set_scratch_estimate(0, 0)
    for g, w_ in zip(got, want):
        assert same(g, w_)
