"""GPU parity (the tests proper): the HIP pipeline, called through the C ABI,
against (1) the golden frames rendered by the unmodified reference and (2) the
CPU oracle on the same inputs.

Bar (BASELINE.json north_star): depth within 1e-5 relative, 8-bit colour
within 1 LSB on >= 99.9 % of pixels, coverage differing on <= 0.01 % of pixels.
The pipeline keeps the reference's f32 expression order with FMA contraction
off, so the expected result is much tighter: identical coverage and depth, and
colour differing only where the device's double-precision pow() rounds
differently from glibc's."""
import ctypes as C
import os

import numpy as np
import pytest

from __graft_entry__ import compare_frames
from conftest import REPO, golden_cases, load_golden, render_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_gpu_matches_reference_golden(gpu, case):
    want = load_golden(case["name"])
    got = render_case(gpu, case)
    n = got[1].size
    any_diff, big_diff, coverage, bad_depth, any_depth = compare_frames(got, want)
    print(f"{case['name']}: colour!= {any_diff} (>1LSB {big_diff}) coverage {coverage} "
          f"depth>1e-5 {bad_depth} depth!= {any_depth} of {n}")
    # north-star tolerance
    assert coverage <= max(1, n * 1e-4)
    assert bad_depth <= coverage
    assert big_diff <= n * 1e-3
    # what this implementation actually achieves
    assert coverage == 0 and any_depth == 0
    assert any_diff <= max(2, n * 2e-4)


@pytest.mark.parametrize("name,like,size,index", [
    ("helmet", None, (640, 360), 5),
    ("azura", None, (400, 300), 11),
    ("craftsman", None, (400, 300), 2),
    ("kgirl_blend.scn", "kgirl", (512, 288), 7),
])
def test_gpu_matches_oracle_on_orbit_views(gpu, oracle, name, like, size, index):
    import os
    from renderer_amd.binding import SCENES
    frames = []
    for pipe in (gpu, oracle):
        scene = pipe.open_scene(os.path.join(SCENES, name) if name.endswith(".scn") else name, like)
        frames.append(pipe.render(scene, size[0], size[1], pipe.orbit_view(index, 16)))
        pipe.close_scene(scene)
    n = size[0] * size[1]
    any_diff, big_diff, coverage, bad_depth, any_depth = compare_frames(frames[0], frames[1])
    print(f"{name}@{index}: colour!= {any_diff} (>1LSB {big_diff}) coverage {coverage} depth!= {any_depth} of {n}")
    assert coverage == 0 and any_depth == 0
    assert any_diff <= max(2, n * 2e-4) and big_diff <= n * 1e-3


def assert_same(got, want, n, exact_colour=False):
    any_diff, big_diff, coverage, bad_depth, any_depth = compare_frames(got, want)
    assert coverage == 0 and any_depth == 0, (coverage, any_depth)
    if exact_colour:
        assert any_diff == 0, any_diff
    else:
        assert any_diff <= max(2, n * 2e-4) and big_diff <= n * 1e-3, (any_diff, big_diff)


@pytest.mark.parametrize("case", [c for c in golden_cases() if c["name"] in ("helmet_inside", "kgirl_blend", "azura_shadow", "phoenix_cutoff")],
                         ids=lambda c: c["name"])
def test_gpu_per_triangle_submission_matches_reference_golden(gpu, case):
    """The compatible slow path: one graphics_draw_triangle call per face, attributes
    written through program_get_attribs, exactly like the reference's draw_model loops."""
    gpu.lib.rzs_set_per_triangle(1)
    try:
        got = render_case(gpu, case)
    finally:
        gpu.lib.rzs_set_per_triangle(0)
    assert_same(got, load_golden(case["name"]), got[1].size)


@pytest.mark.parametrize("case", [c for c in golden_cases() if c["name"] in ("helmet_inside", "kgirl_blend", "azura_shadow", "buster_anim")],
                         ids=lambda c: c["name"])
def test_gpu_per_triangle_streams_that_repeat_are_drawn_from_the_device(gpu, case):
    """Frame after frame the reference's per-face loops submit the same vertices: from the second frame on
    a model's draw comes from the stream remembered on the device (mi355r_replayed_draws counts them), also
    under another camera and time, and the frames stay the reference's."""
    import os
    from renderer_amd.binding import SCENES
    gpu.lib.mi355r_replayed_draws.restype = C.c_ulonglong
    scene = gpu.open_scene(os.path.join(SCENES, case["scene"]) if case["scene"].endswith(".scn") else case["scene"], case.get("like"))
    view = gpu.make_view(case.get("eye"), case.get("target"), case.get("time", 0.0), case.get("layer", -1))
    other = gpu.make_view((0.3, 0.8, -1.1), None, case.get("time", 0.0) + 0.37)
    w, h = case["size"]
    want = load_golden(case["name"])
    gpu.lib.rzs_set_per_triangle(1)
    try:
        first = gpu.render(scene, w, h, view)
        before = gpu.lib.mi355r_replayed_draws()
        second = gpu.render(scene, w, h, view)
        assert gpu.lib.mi355r_replayed_draws() > before
        elsewhere = gpu.render(scene, w, h, other)
        third = gpu.render(scene, w, h, view)
    finally:
        gpu.lib.rzs_set_per_triangle(0)
    by_mesh = gpu.render(scene, w, h, other)
    gpu.close_scene(scene)
    for got in (first, second, third):
        assert_same(got, want, w * h)
    assert np.array_equal(elsewhere[0], by_mesh[0]) and np.array_equal(elsewhere[1], by_mesh[1])


@pytest.mark.parametrize("name,like", [("kgirl_blend.scn", "kgirl"), ("helmet", None), ("azura_shadow.scn", "azura")])
def test_gpu_per_triangle_frames_recorded_together(gpu, name, like):
    """Several frames submitted face by face into one flush, three flushes in a row (the first remembers the
    models' vertex streams, the others draw from them: many draws of one program per flush, shadow pass and
    main pass, different views and times) against the same frames submitted mesh by mesh."""
    import os
    from renderer_amd import orbit
    from renderer_amd.binding import SCENES
    scene = gpu.open_scene(os.path.join(SCENES, name) if name.endswith(".scn") else name, like)
    w, h, count = 256, 144, 6
    fbs = [gpu.framebuffer(w, h) for _ in range(count)]
    for flush in range(3):
        views = [gpu.orbit_view(7 * flush + i, 40) for i in range(count)]
        orbit.render_views(gpu, scene, fbs, views)
        want = [gpu.read(fb) for fb in fbs]
        gpu.lib.rzs_set_per_triangle(1)
        try:
            orbit.render_views(gpu, scene, fbs, views)
        finally:
            gpu.lib.rzs_set_per_triangle(0)
        for i, fb in enumerate(fbs):
            got = gpu.read(fb)
            assert np.array_equal(got[0], want[i][0]) and np.array_equal(got[1], want[i][1]), (flush, i)
    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)


@pytest.mark.parametrize("name,like", [("kgirl_blend.scn", "kgirl"), ("helmet", None)])
def test_gpu_per_triangle_submission_records_one_draw_per_model(gpu, name, like):
    """graphics_draw_triangle merges consecutive triangles of one model into one draw: the uniform
    block is compared per call and, for a skinned program, the joint matrices it points at -- so the
    reference's per-face loop over a skinned mesh (blinn_shader.c:323-341) costs as many draws, and as
    many copies of the pose, as graphics_draw_mesh."""
    import os
    from renderer_amd.binding import SCENES
    scene = gpu.open_scene(os.path.join(SCENES, name) if name.endswith(".scn") else name, like)
    view = gpu.make_view((0.5, 0.2, -1.3), None, 1.9)
    draws = []
    for per_triangle in (0, 1):
        gpu.lib.rzs_set_per_triangle(per_triangle)
        try:
            gpu.render(scene, 256, 144, view)           # uploads
            gpu.lib.mi355r_reset_counters()
            gpu.render(scene, 256, 144, view)
            c = gpu.counters()
        finally:
            gpu.lib.rzs_set_per_triangle(0)
        draws.append((c[5], c[0]))
    gpu.close_scene(scene)
    assert draws[0] == draws[1], draws                  # (draws, triangles) of the frame
    assert draws[0][0] < 64 and draws[0][1] > 3000


@pytest.mark.parametrize("size", [(1, 1), (31, 9), (65, 64), (333, 77), (64, 1)])
def test_gpu_ragged_framebuffer_sizes(gpu, oracle, size):
    """Sizes that are not multiples of the 32x8 tile or the 64x64 bin."""
    frames = []
    for pipe in (gpu, oracle):
        scene = pipe.open_scene("helmet")
        frames.append(pipe.render(scene, size[0], size[1], pipe.make_view((0.4, 0.2, 0.8), None, 0.0)))
        pipe.close_scene(scene)
    assert_same(frames[0], frames[1], size[0] * size[1])


@pytest.mark.parametrize("name,like,size,eye,time", [
    ("helmet", None, (1920, 1080), None, 0.6),                                   # the benchmark configuration (C2)
    ("dieselpunk_shadow.scn", "dieselpunk", (3840, 2160), (-0.9, 0.6, 1.0), 0.6),  # C4: 4K + 2048^2 shadow map
    ("kgirl_blend.scn", "kgirl", (1920, 1080), (0.5, 0.2, -1.3), 0.6),           # C3: skinning + ordered blend
    ("azura_shadow.scn", "azura", (800, 600), None, 0.0),                        # C1 at its quoted size: dense mesh, shadow pass
    ("buster", None, (1920, 1080), None, 0.4),                                   # C5 at its quoted size, two animation times
    ("buster", None, (1920, 1080), (1.1, 0.45, -1.0), 2.9),
])
def test_gpu_full_size_frames_match_oracle(gpu, oracle, name, like, size, eye, time):
    import os
    from renderer_amd.binding import SCENES
    frames = []
    for pipe in (gpu, oracle):
        scene = pipe.open_scene(os.path.join(SCENES, name) if name.endswith(".scn") else name, like)
        frames.append(pipe.render(scene, size[0], size[1], pipe.make_view(eye, None, time)))
        pipe.close_scene(scene)
    assert_same(frames[0], frames[1], size[0] * size[1])


def test_gpu_full_size_animated_frames_of_one_flush_match_oracle(gpu, oracle):
    """C3 as the orbit bench submits it, at its quoted size: eight 1080p frames of the skinned, shadowed, blended
    scene with eight different animation times recorded into ONE flush (eight poses evaluated by one launch, shadow
    passes, recorded copies, main and blended passes batched level by level), each against the oracle."""
    import os
    from renderer_amd.binding import SCENES
    path, size, n = os.path.join(SCENES, "kgirl_blend.scn"), (1920, 1080), 8
    scene = gpu.open_scene(path, "kgirl")
    views = [gpu.orbit_view(i * 29, 240) for i in range(n)]         # frame_time = index / 30 s
    assert len({v.frame_time for v in views}) == n
    fbs = [gpu.framebuffer(*size) for _ in range(n)]
    for fb, v in zip(fbs, views):
        gpu.draw(scene, fb, v)
    gpu.lib.graphics_flush()
    got = [gpu.read(fb) for fb in fbs]
    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    scene = oracle.open_scene(path, "kgirl")
    for i in range(n):
        assert_same(got[i], oracle.render(scene, size[0], size[1], oracle.orbit_view(i * 29, 240)), size[0] * size[1])
    oracle.close_scene(scene)


def test_gpu_batched_frames_equal_single_frames(gpu):
    """Size-independent property at the benchmark size: N views recorded and flushed together
    (one launch sequence over all of them) give byte-identical frames to rendering them one
    by one, and rendering twice gives the same bytes (no order dependence between waves)."""
    w, h, n = 1920, 1080, 6
    scene = gpu.open_scene("helmet")
    views = [gpu.orbit_view(i * 3, 32) for i in range(n)]
    single = [gpu.render(scene, w, h, v) for v in views]
    fbs = [gpu.framebuffer(w, h) for _ in range(n)]
    for rep in range(2):
        for fb, v in zip(fbs, views):
            gpu.draw(scene, fb, v)
        gpu.lib.graphics_flush()
        for fb, want in zip(fbs, single):
            color, depth = gpu.read(fb)
            assert np.array_equal(color, want[0]) and np.array_equal(depth, want[1])
    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    # every pixel was written: alpha stays at the cleared value, depth < 1 where the model or the sky is
    assert np.all(single[0][0][..., 3] == 255)


@pytest.mark.parametrize("name,like,size", [("azura_shadow.scn", "azura", (200, 150)), ("synthetic_blinn.scn", None, (224, 126)),
                                            ("kgirl_blend.scn", "kgirl", (192, 108))])
def test_gpu_shadowed_frames_recorded_together_match_oracle(gpu, oracle, name, like, size):
    """Frames with a shadow pass, a framebuffer -> texture copy and a main (+ blended) pass each,
    recorded one after the other and read only at the end: the copies are recorded between the
    draws and the frames share launch sequences.  Eight frames more than the scene layer has shadow
    buffers to alternate between, so the case where a shadow map comes round again while recorded
    draws still sample it (which has to flush) is part of it.  Frames from the start, the middle and
    either side of that point against the oracle's."""
    import os
    from renderer_amd.binding import SCENES
    path = os.path.join(SCENES, name)
    scene = gpu.open_scene(path, like)
    ring = gpu.lib.rzs_scene_shadow_ring(scene)
    assert 32 <= ring <= 128
    n = ring + 8
    views = [gpu.orbit_view(i, n) for i in range(n)]
    gpu.render(scene, size[0], size[1], views[0])       # uploads
    fbs = [gpu.framebuffer(*size) for _ in range(n)]
    gpu.lib.mi355r_enable_timing(1)
    gpu.kernel_times()                                  # start counting from here
    for fb, v in zip(fbs, views):
        gpu.draw(scene, fb, v)
    got = [gpu.read(fb) for fb in fbs]
    passes, sequences, _ = gpu.kernel_times()
    gpu.lib.mi355r_enable_timing(0)
    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    assert passes >= 2 * n and sequences <= passes // 4, (passes, sequences)     # frames did share launches
    scene = oracle.open_scene(path, like)
    for i in (0, 1, 15, ring - 1, ring, ring + 1, n - 1):
        want = oracle.render(scene, size[0], size[1], oracle.orbit_view(i, n))
        assert_same(got[i], want, size[0] * size[1])
    oracle.close_scene(scene)


def test_gpu_counters_match_oracle_workload(gpu, oracle):
    """Triangles in, triangles rasterised and fragments shaded (graphics.c:437 calls) of one frame."""
    size = (480, 270)
    scene = gpu.open_scene("helmet")
    view = gpu.orbit_view(3, 16)
    gpu.render(scene, *size, view)                      # uploads
    gpu.lib.mi355r_reset_counters()
    gpu.lib.mi355r_enable_counters(1)
    gpu.render(scene, *size, view)
    got = gpu.counters()
    gpu.lib.mi355r_enable_counters(0)
    gpu.close_scene(scene)
    scene = oracle.open_scene("helmet")
    oracle.lib.oracle_reset_stats()
    oracle.render(scene, *size, oracle.orbit_view(3, 16))
    want = oracle.counters()          # triangles, prims, tests, covered, shaded
    oracle.close_scene(scene)
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[4]
    assert got[8] <= got[2]           # deferred shading never shades more than the reference


def test_gpu_passes_into_one_framebuffer_stay_ordered(gpu, oracle):
    """Several frames recorded into the SAME framebuffer before anything is read: with a clear in
    between the last one must win, without a clear they pile up in call order (the second frame's
    depth test sees the first frame's depth).  Independent frames into other framebuffers are
    recorded in between, so the scheduler has something to batch them with."""
    w, h = 320, 180
    results = []
    for pipe in (gpu, oracle):
        scene = pipe.open_scene("helmet")
        other = pipe.open_scene("azura")
        fb = pipe.framebuffer(w, h)
        side = [pipe.framebuffer(w, h) for _ in range(3)]
        views = [pipe.orbit_view(i, 8) for i in range(4)]
        pipe.draw(scene, fb, views[0])
        pipe.draw(other, side[0], views[1])
        pipe.draw(scene, fb, views[2])                  # rzs_scene_draw clears first: replaces frame 0
        pipe.draw(other, side[1], views[3])
        cleared = pipe.read(fb)
        # now without a clear: the blinn scene's models are drawn over the finished pbr frame
        pipe.lib.rzs_scene_draw_no_clear(other, fb, C.byref(views[1]))
        pipe.draw(scene, side[2], views[0])
        piled = pipe.read(fb)
        sides = [pipe.read(f) for f in side]
        for f in side + [fb]:
            pipe.release(f)
        pipe.close_scene(scene)
        pipe.close_scene(other)
        results.append([cleared, piled] + sides)
    for got, want in zip(*results):
        assert_same(got, want, w * h)


class _Texture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("buffer", C.POINTER(C.c_float))]


def _texture_api(lib):
    lib.texture_create.restype = C.POINTER(_Texture)
    lib.texture_create.argtypes = [C.c_int, C.c_int]
    lib.texture_release.argtypes = [C.POINTER(_Texture)]
    lib.texture_sync.argtypes = [C.POINTER(_Texture)]
    lib.texture_from_colorbuffer.argtypes = [C.POINTER(_Texture), C.c_void_p]
    lib.texture_from_depthbuffer.argtypes = [C.POINTER(_Texture), C.c_void_p]


@pytest.mark.parametrize("recorded", [True, False], ids=["recorded-behind-the-draws", "after-a-read"])
def test_gpu_framebuffer_copies_into_textures(gpu, oracle, recorded):
    """texture_from_colorbuffer / texture_from_depthbuffer (texture.c:130-158): texels = byte / 255.0f
    per channel, and (d, d, d, 1).  Called right after the draws (the copy is then recorded behind them
    and runs at the next flush) and after the frame has been read back (nothing pending: it runs at
    once); either way the texture's host image, made valid by texture_sync, equals the oracle's."""
    w, h = 160, 90
    images = []
    for pipe in (gpu, oracle):
        _texture_api(pipe.lib)
        scene = pipe.open_scene("helmet")
        fb = pipe.framebuffer(w, h)
        colour, depth = pipe.lib.texture_create(w, h), pipe.lib.texture_create(w, h)
        pipe.draw(scene, fb, pipe.orbit_view(2, 16))
        if not recorded:
            pipe.read(fb)
        pipe.lib.texture_from_colorbuffer(colour, fb)
        pipe.lib.texture_from_depthbuffer(depth, fb)
        frame = pipe.read(fb)
        out = []
        for t in (colour, depth):
            pipe.lib.texture_sync(t)
            out.append(np.ctypeslib.as_array(t.contents.buffer, (h, w, 4)).copy())
            pipe.lib.texture_release(t)
        images.append((frame, out))
        pipe.release(fb)
        pipe.close_scene(scene)
    (gf, (gc, gd)), (of, (oc, od)) = images
    assert_same(gf, of, w * h)
    # the copies against the frame they were taken from, exactly
    assert np.array_equal(gc, gf[0].astype(np.float32) / np.float32(255.0))
    assert np.array_equal(gd[..., 0], gf[1]) and np.array_equal(gd[..., 1], gf[1]) and np.array_equal(gd[..., 2], gf[1])
    assert np.all(gd[..., 3] == 1)
    assert np.array_equal(od[..., 0], of[1]) and np.array_equal(oc, of[0].astype(np.float32) / np.float32(255.0))


@pytest.mark.parametrize("size", [(320, 180), (67, 33)])
def test_gpu_blit_equals_the_reference_blit(gpu, oracle, size):
    """framebuffer_blit (private.c:8-50): vertical flip, red/blue swap or not, and the destination's fourth
    byte left alone -- against the restated host loop, on a destination filled with noise."""
    w, h = size
    out = []
    for pipe in (gpu, oracle):
        pipe.lib.framebuffer_blit.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        scene = pipe.open_scene("helmet")
        fb = pipe.framebuffer(w, h)
        pipe.draw(scene, fb, pipe.orbit_view(4, 16))
        images = []
        for swap in (1, 0):
            dst = np.random.default_rng(5).integers(0, 256, (h, w, 4)).astype(np.uint8)
            pipe.lib.framebuffer_blit(C.cast(fb, C.c_void_p), dst.ctypes.data_as(C.c_void_p), swap)
            images.append(dst)
        frame = pipe.read(fb)
        out.append((images, frame))
        pipe.release(fb)
        pipe.close_scene(scene)
    (gi, gf), (oi, of) = out
    noise = np.random.default_rng(5).integers(0, 256, (h, w, 4)).astype(np.uint8)
    for g, o, swap in zip(gi, oi, (1, 0)):
        assert np.array_equal(g[..., 3], noise[..., 3])
        order = [2, 1, 0] if swap else [0, 1, 2]
        assert np.array_equal(g[..., :3], gf[0][::-1, :, order])       # the library's own frame, flipped
        assert np.array_equal(o[..., :3], of[0][::-1, :, order])
        assert (g[..., :3] != o[..., :3]).any(axis=2).sum() <= max(2, w * h * 2e-4)   # colour bar of the frames themselves


@pytest.mark.parametrize("size", [(1920, 1080), (333, 77), (64, 1), (1, 1)])
def test_gpu_overlays_equal_the_reference_overlays(gpu, oracle, size):
    """draw2d_draw_point / _line / _triangle / _texture (draw2d.c) on the device against the restated host
    loops: random points inside and outside [0,1]^2, steep, flat, diagonal and single-pixel lines, the label
    images, a float image a host filled by hand (values up to 1, where the byte sum wraps) and labels that
    stick out of the frame, over a non-black clear colour and over one another; every byte identical."""
    from renderer_amd.binding import Vec2, Vec4
    w, h = size
    rng = np.random.default_rng(w * 131 + h)
    unit = lambda n: [float(np.float32(x)) for x in rng.uniform(-0.2, 1.2, n)]
    ops = []
    for i in range(120):
        color = [float(np.float32(x)) for x in rng.uniform(0, 1, 4)]
        kind = ("point", "line", "outline", "label")[i % 4]
        if kind == "label":
            ops.append(["label", "common/%s.tga" % ("diffuse", "specular", "roughness", "occlusion", "normal")[i // 4 % 5], *unit(2)])
        else:
            ops.append([kind, *color, *unit({"point": 2, "line": 4, "outline": 6}[kind])])
    ops.append(["line", 1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 1.0])          # corner to corner
    ops.append(["line", 0.0, 1.0, 0.0, 1.0, 0.25, 0.5, 0.25, 0.5])        # both distances zero
    hand = np.random.default_rng(9).uniform(0, 1, (37, 53, 4)).astype(np.float32)
    hand[::5, ::3] = 1.0
    frames = []
    for pipe in (gpu, oracle):
        lib = pipe.lib
        P = C.c_void_p
        texture_create = C.CFUNCTYPE(P, C.c_int, C.c_int)(("texture_create", lib))
        texture_invalidate = C.CFUNCTYPE(None, P)(("texture_invalidate", lib))
        fb = pipe.framebuffer(w, h)
        lib.framebuffer_clear_color(fb, Vec4(0.3, 0.6, 0.9, 0.5))
        pipe.overlay(fb, ops)
        tex = texture_create(53, 37)
        buffer = C.cast(C.cast(tex + 8, C.POINTER(C.c_void_p))[0], C.POINTER(C.c_float))     # texture_t: width, height, buffer
        C.memmove(buffer, hand.ctypes.data, hand.nbytes)
        texture_invalidate(tex)
        for origin in ((0.1, 0.2), (0.97, 0.4), (0.5, 0.99), (0.11, 0.21)):
            pipe._draw_texture(fb, tex, Vec2(*origin))
        pipe.overlay(fb, ops[:8])
        frames.append(pipe.read(fb)[0])
        pipe._texture_release(tex)
        pipe.release(fb)
    assert np.array_equal(frames[0], frames[1])
    assert (frames[0] != np.array([76, 153, 229, 127], np.uint8)).any()


@pytest.mark.parametrize("size", [(320, 180), (256, 64), (67, 33), (5, 1)])
def test_gpu_frames_without_their_alpha_bytes(gpu, size):
    """framebuffer_read_device_rgb: the colour image as RGB8 in device memory (what the multi-GPU gather
    moves), for pixel counts that are and are not multiples of four."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    w, h = size
    scene = gpu.open_scene("helmet")
    fb = gpu.framebuffer(w, h)
    gpu.draw(scene, fb, gpu.orbit_view(6, 16))
    n = h * w * 3 + 8
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), n) == 0 and hip.hipMemset(dev, 7, n) == 0
    gpu.lib.framebuffer_read_device_rgb(fb, dev)
    got = np.zeros(n, np.uint8)
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dev, n, 2) == 0        # device to host
    hip.hipFree(dev)
    color, _ = gpu.read(fb)
    assert np.array_equal(got[:h * w * 3].reshape(h, w, 3), color[..., :3])
    assert np.all(got[h * w * 3:] == 7)                  # nothing behind the image is touched
    gpu.release(fb)
    gpu.close_scene(scene)


def test_gpu_many_frames_without_their_alpha_bytes_at_once(gpu):
    """framebuffers_read_device_rgb: the frames of a step copied out together (one wait for all lanes) equal
    the frames copied out one by one, each at its stride, nothing in between touched; also right after the
    draws were recorded, and with a framebuffer that was only cleared."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    from renderer_amd.binding import Vec4
    w, h, count = 322, 181, 9
    stride = (w * h * 3 + 3) // 4 * 4 + 64
    scene = gpu.open_scene("helmet")
    fbs = [gpu.framebuffer(w, h) for _ in range(count)]
    gpu.lib.framebuffer_clear_color(fbs[-1], Vec4(0.2, 0.4, 0.6, 1.0))
    for i, fb in enumerate(fbs[:-1]):
        gpu.draw(scene, fb, gpu.orbit_view(i, 16))
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), stride * count) == 0 and hip.hipMemset(dev, 9, stride * count) == 0
    handles = (type(fbs[0]) * count)(*fbs)
    gpu.lib.framebuffers_read_device_rgb(handles, count, dev, stride)     # the draws are still only recorded
    got = np.zeros(stride * count, np.uint8)
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dev, stride * count, 2) == 0
    hip.hipFree(dev)
    got = got.reshape(count, stride)
    for i, fb in enumerate(fbs):
        color, _ = gpu.read(fb)
        assert np.array_equal(got[i, :w * h * 3].reshape(h, w, 3), color[..., :3]), i
        assert np.all(got[i, w * h * 3:] == 9)
        gpu.release(fb)
    assert np.all(got[-1, :3] == np.array([51, 102, 153], np.uint8))
    gpu.close_scene(scene)


def test_gpu_frames_into_the_gather_buffer(gpu):
    """orbit.frames_to_tensor with a device tensor, as bench.py --gpus N fills the buffer its gather sends:
    RGB8 (all frames in one call) and RGBA8 (frame by frame) against the frames read back one by one."""
    import torch
    from renderer_amd import orbit
    w, h, count = 200, 120, 6
    scene = gpu.open_scene("helmet")
    fbs = [gpu.framebuffer(w, h) for _ in range(count)]
    orbit.render_views(gpu, scene, fbs, [gpu.orbit_view(i, 16) for i in range(count)])
    rgb = orbit.frames_to_tensor(gpu, fbs, count, torch.zeros((count + 1, h, w, 3), dtype=torch.uint8, device="cuda"))
    rgba = orbit.frames_to_tensor(gpu, fbs, count, torch.zeros((count, h, w, 4), dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    rgb, rgba = rgb.cpu().numpy(), rgba.cpu().numpy()
    for i, fb in enumerate(fbs):
        color, _ = gpu.read(fb)
        assert np.array_equal(rgb[i], color[..., :3]) and np.array_equal(rgba[i], color)
        gpu.release(fb)
    assert not rgb[count].any()
    gpu.close_scene(scene)


@pytest.mark.gpu
def test_gpu_random_views_match_oracle():
    """tests/fuzz_parity.py for a few dozen views: ragged sizes, cameras inside the models and near the sky
    box, random animation times, frames rendered twice into one framebuffer."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "fuzz_parity.py"), "40", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FUZZ 40 views 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
