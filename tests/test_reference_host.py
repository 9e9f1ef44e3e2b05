"""The reference's OWN host code on top of the C ABI (CPU container only: needs /root/reference).

oracle/Makefile (target refhost) compiles every source file of the reference except the ones the
library replaces (core/graphics.c, core/texture.c, and core/draw2d.c for the 2-D overlays) -- scenes/*.c with the .scn parser, the shaders'
update_model / draw_model loops (per-face program_get_attribs + graphics_draw_triangle), the caches,
loaders and tests/test_helper.c's frame driver, unchanged, where they lie -- and links them
  (1) against renderer_amd/libmi355r.so with --no-undefined: every symbol the reference's host
      needs from graphics.c / texture.c is exported by the library (it cannot run here: no GPU;
      tests/test_00_reference_host_gpu.py runs it on the GPU box);
  (2) against oracle/liboracle.so, and runs that through ref_driver.c: the frames must equal the
      goldens (rendered by the unmodified reference) bit for bit, which pins the C ABI's semantics
      -- recorded draws, texture_from_file / cubemap_from_files, the samplers the reference's CPU
      shaders call back into, texture_from_depthbuffer -- under the reference's own callers."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, golden_cases, load_golden

REFERENCE = "/root/reference/renderer"
REF_OUT = os.path.join(REPO, "oracle", "_ref")

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference sources are not present")


@pytest.fixture(scope="module")
def refhost():
    subprocess.run(["make", "-C", os.path.join(REPO, "renderer_amd")], check=True, capture_output=True)
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "oracle", "refhost"], check=True, capture_output=True)
    return {k: os.path.join(REF_OUT, "refhost_" + k) for k in ("oracle", "mi355r")}


def test_reference_host_links_against_the_gpu_library(refhost):
    """The reference's host (everything but core/graphics.c and core/texture.c) linked against the library with
    -Wl,--no-undefined: nothing it calls is missing."""
    exe = refhost["mi355r"]
    assert os.path.exists(exe)
    dynamic = subprocess.run(["readelf", "-d", exe], check=True, capture_output=True, text=True).stdout
    assert "libmi355r.so" in dynamic
    # the entry points it binds are the reference's own names
    undefined = subprocess.run(["nm", "-D", "--undefined-only", exe], check=True, capture_output=True, text=True).stdout
    for name in ("graphics_draw_triangle", "program_create", "program_get_attribs", "program_get_uniforms",
                 "framebuffer_create", "framebuffer_clear_color", "framebuffer_clear_depth", "texture_from_file",
                 "cubemap_from_files", "texture_from_depthbuffer", "texture_create", "framebuffer_sync",
                 "draw2d_draw_line", "draw2d_draw_texture"):
        assert f" U {name}" in undefined, name
    # and none of them is the batched addition: the reference's loops are per triangle
    assert "graphics_draw_mesh" not in undefined


@pytest.mark.parametrize("name", ["helmet_default", "kgirl_anim", "azura_shadow", "helmet_inside", "kgirl_blend",
                                  "craftsman_default", "buster_anim", "helmet_overlay"])
def test_reference_host_on_the_c_abi_reproduces_goldens(refhost, name):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_goldens import run_case
    case = next(c for c in golden_cases() if c["name"] == name)
    color, depth = run_case(case, os.path.join(REPO, "assets"), binary=refhost["oracle"])
    want_color, want_depth = load_golden(name)
    assert np.array_equal(depth, want_depth)
    assert np.array_equal(color, want_color)


@pytest.mark.parametrize("seed,size", [(1, (97, 61)), (2, (160, 90)), (3, (33, 200))])
def test_random_overlays_equal_the_unmodified_reference(oracle, seed, size):
    """draw2d.c of the UNMODIFIED reference (oracle/_ref/refdump) against the restatement in oracle/pipeline.c on
    random points, lines, outlines and labels over a rendered frame (60 per case, ragged frame sizes, points
    outside [0,1]^2): pins the overlay restatement beyond the one golden frame."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_goldens import run_case, REFDUMP
    if not os.path.exists(REFDUMP):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "ref"], check=True, capture_output=True)
    rng = np.random.default_rng(seed)
    f32 = lambda xs: [float(np.float32(x)) for x in xs]
    ops = []
    for i in range(60):
        kind = ("point", "line", "outline", "label")[int(rng.integers(4))]
        if kind == "label":
            ops.append(["label", "common/%s.tga" % ("diffuse", "specular", "roughness", "occlusion", "normal")[int(rng.integers(5))],
                        *f32(rng.uniform(-0.1, 1.1, 2))])
        else:
            ops.append([kind, *f32(rng.uniform(0, 1, 4)), *f32(rng.uniform(-0.2, 1.2, {"point": 2, "line": 4, "outline": 6}[kind]))])
    case = {"name": "overlays", "scene": "helmet", "ref": "pbr:helmet", "size": list(size), "overlay": ops}
    want_color, want_depth = run_case(case, os.path.join(REPO, "assets"))
    from conftest import render_case
    color, depth = render_case(oracle, case)
    assert np.array_equal(depth, want_depth)
    assert np.array_equal(color, want_color)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18])
def test_random_views_equal_the_unmodified_reference(oracle, seed):
    """The restatement against the unmodified reference (oracle/_ref/refdump) beyond the golden cameras: a random
    scene, frame size, eye (also inside the model and next to the sky box), target and animation time per
    case; colour and depth bit for bit."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_goldens import run_case
    from conftest import render_case
    rng = np.random.default_rng(seed)
    scenes = [("helmet", "pbr:helmet"), ("kgirl", "blinn:kgirl"), ("buster", "pbr:buster"), ("azura", "blinn:azura"),
              ("craftsman", "blinn:craftsman"), ("drone", "pbr:drone"), ("dieselpunk", "pbr:dieselpunk"), ("phoenix", "blinn:phoenix")]
    scene, ref = scenes[seed % len(scenes)]
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction)
    radius = float(rng.choice([0.08, 0.4, 1.2, 2.5]))
    case = {"name": "random", "scene": scene, "ref": ref,
            "size": [int(rng.integers(40, 150)), int(rng.integers(30, 100))],
            "eye": [float(np.float32(x)) for x in direction * radius],
            "target": [float(np.float32(x)) for x in rng.uniform(-0.2, 0.2, 3)],
            "time": float(np.float32(rng.uniform(0, 3)))}
    want_color, want_depth = run_case(case, os.path.join(REPO, "assets"))
    color, depth = render_case(oracle, case)
    assert np.array_equal(depth, want_depth), case
    assert np.array_equal(color, want_color), case


@pytest.mark.parametrize("ani", ["kgirl/kgirl.ani", "buster/buster.ani", "drone/drone.ani", "drone/fire.ani", "phoenix/phoenix.ani"])
def test_poses_equal_the_unmodified_reference(oracle, ani, tmp_path):
    """skeleton_update_joints of the UNMODIFIED reference (core/skeleton.c:231-341, through oracle/_ref/refdump
    --poses) against the restated evaluation (oracle/pose.c) and the scene layer's (host/assets.c): joint and
    normal matrices of every joint at 220 times (i * 0.137, past the animations' ends where time wraps), bit
    for bit -- the matrices themselves, not frames rendered with them."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_goldens import REFDUMP
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "ref"], check=True, capture_output=True)
    count = 220
    times = (np.arange(count, dtype=np.float32) * np.float32(0.137)).astype(np.float32)
    restated = oracle.poses(ani, times)
    host = oracle.poses(ani, times, on_host=True)
    J = restated[0].shape[1]
    out = str(tmp_path / "poses.bin")
    subprocess.run([REFDUMP, "--assets", os.path.join(REPO, "assets"), "--poses", ani, str(J), str(count), out], check=True, timeout=120)
    raw = np.fromfile(out, np.float32).reshape(count, J * 25)
    want_joints = raw[:, :J * 16].reshape(count, J, 4, 4)
    want_normals = raw[:, J * 16:].reshape(count, J, 3, 3)
    for got in (restated, host):
        assert np.array_equal(got[0].view(np.uint32), want_joints.view(np.uint32))
        assert np.array_equal(got[1].view(np.uint32), want_normals.view(np.uint32))


@pytest.mark.parametrize("file,usage", [
    ("helmet/helmet_basecolor.tga", 2), ("helmet/helmet_metalness.tga", 1), ("helmet/helmet_normal.tga", 1), ("helmet/helmet_emission.tga", 2),
    ("workshop/m1_px.hdr", 0), ("workshop/i_ny.hdr", 2), ("workshop/m3_pz.hdr", 2), ("common/brdf_lut.hdr", 3),
    ("common/diffuse.tga", 0), ("azura/body_diffuse.tga", 0), ("kgirl/kgirl_diffuse.tga", 0), ("drone/drone_specular.tga", 2)])
def test_loaded_textures_equal_the_unmodified_reference(oracle, file, usage, tmp_path):
    """texture_from_file of the UNMODIFIED reference (image decoding, bottom-up rows, and the conversion its usage asks
    for: byte / 255, pow(v, 2.2) on colour maps, ACES + pow(v, 1 / 2.2) on tone-mapped sky faces; texture.c:32-128)
    against this repository's loader: every float of texture->buffer, bit for bit.  usage: 0 LDR_COLOR, 1 LDR_DATA,
    2 HDR_COLOR, 3 HDR_DATA (texture.h:7-12)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_goldens import REFDUMP
    if not os.path.exists(os.path.join(REPO, "assets", file)):
        pytest.skip("the tree does not carry " + file)
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "ref"], check=True, capture_output=True)
    out = str(tmp_path / "texture.bin")
    subprocess.run([REFDUMP, "--assets", os.path.join(REPO, "assets"), "--texture", file, str(usage), out], check=True, timeout=120)
    w, h = np.fromfile(out, np.int32, 2)
    want = np.fromfile(out, np.float32, offset=8).reshape(h, w, 4)
    load = C.CFUNCTYPE(C.c_void_p, C.c_char_p, C.c_int)(("texture_from_file", oracle.lib))
    release = C.CFUNCTYPE(None, C.c_void_p)(("texture_release", oracle.lib))
    tex = load(file.encode(), usage)
    tw, th = C.cast(tex, C.POINTER(C.c_int))[0], C.cast(tex, C.POINTER(C.c_int))[1]
    assert (tw, th) == (w, h)
    buffer = C.cast(C.cast(tex + 8, C.POINTER(C.c_void_p))[0], C.POINTER(C.c_float))
    got = np.ctypeslib.as_array(buffer, shape=(h, w, 4)).copy()
    release(tex)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("file", ["helmet/helmet.obj", "kgirl/body.obj", "kgirl/hair.obj", "azura/hair.obj", "buster/buster10.obj",
                                  "dieselpunk/mech.obj", "common/sphere.obj", "common/box.obj", "drone/drone.obj"])
def test_loaded_meshes_equal_the_unmodified_reference(oracle, file, tmp_path):
    """mesh_load of the UNMODIFIED reference (core/mesh.c: OBJ faces de-indexed into the 80-byte vertex stream, with
    tangents, joints and weights) against this repository's loader: face count and every byte of the stream."""
    import ctypes as C
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_goldens import REFDUMP
    if not os.path.exists(os.path.join(REPO, "assets", file)):
        pytest.skip("the tree does not carry " + file)
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "ref"], check=True, capture_output=True)
    out = str(tmp_path / "mesh.bin")
    subprocess.run([REFDUMP, "--assets", os.path.join(REPO, "assets"), "--mesh", file, out], check=True, timeout=120)
    faces = int(np.fromfile(out, np.int32, 1)[0])
    want = np.fromfile(out, np.uint32, offset=4).reshape(faces * 3, 20)

    class Mesh(C.Structure):
        _fields_ = [("num_faces", C.c_int), ("vertices", C.POINTER(C.c_uint32)), ("center", C.c_float * 3)]
    load = C.CFUNCTYPE(C.POINTER(Mesh), C.c_char_p)(("rzh_mesh_load", oracle.lib))
    free = C.CFUNCTYPE(None, C.POINTER(Mesh))(("rzh_mesh_free", oracle.lib))
    mesh = load(file.encode())
    assert mesh.contents.num_faces == faces
    got = np.ctypeslib.as_array(mesh.contents.vertices, shape=(faces * 3, 20)).copy()
    free(mesh)
    assert np.array_equal(got, want)
