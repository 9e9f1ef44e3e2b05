"""CPU checks of the boundary and the host layer: the shared library exports
what include/mi355r.h declares, the public structs have the reference's
layout, and the asset / scene code (shared by the GPU library and the oracle)
reads files the way the reference does.  No compute on the GPU library here."""
import ctypes as C
import math
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import ORACLE_LIB, REPO, golden_cases, render_case
from renderer_amd.binding import GPU_LIB, SCENES

HEADER = os.path.join(REPO, "include", "mi355r.h")


def declared_functions():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = re.findall(r"^\s*(?!typedef)[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", src, flags=re.M)
    assert len(names) >= 40, names
    return names


@pytest.mark.parametrize("path", [GPU_LIB, ORACLE_LIB], ids=["libmi355r", "liboracle"])
def test_library_exports_every_declared_symbol(path, oracle):
    """Loading must work without a GPU (nothing touches the device before the first call)."""
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
    lib = C.CDLL(path)
    optional_in_oracle = {"mi355r_init", "mi355r_shutdown", "mi355r_stream", "framebuffer_read_device", "framebuffer_read_device_rgb", "framebuffers_read_device_rgb", "framebuffers_read_device_rgb_begin", "framebuffers_read_device_rgb_end",
                          "mi355r_get_counters", "mi355r_reset_counters", "mi355r_enable_counters",
                          "mi355r_enable_timing", "mi355r_get_timing", "mi355r_pack_texels",
                          "mi355r_set_scratch_estimate", "mi355r_passes_run_again", "mi355r_replayed_draws", "texture_sync_noalloc",
                          "mi355r_inspect_sky_first", "mi355r_last_raster_kernel"}
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    if path == ORACLE_LIB:
        missing = [n for n in missing if n not in optional_in_oracle]
    assert not missing, missing
    # the scene layer both libraries carry
    for n in ["rzs_scene_open", "rzs_scene_close", "rzs_scene_draw", "rzs_orbit_view", "rzs_set_per_triangle"]:
        assert hasattr(lib, n), n


def test_gpu_library_refuses_to_run_without_a_device():
    """No CPU fallback in the product: on a box without a GPU the first real call aborts."""
    code = ("import ctypes as C; l = C.CDLL(%r); l.framebuffer_create.restype = C.c_void_p; "
            "l.framebuffer_create(8, 8); print('created')" % GPU_LIB)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run(["python3", "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "created" not in r.stdout


LAYOUT_PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "mi355r.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
    S(rz_vertex); S(rz_blinn_attribs); S(rz_pbr_attribs); S(rz_skybox_attribs);
    S(rz_blinn_uniforms); S(rz_pbr_uniforms); S(rz_skybox_uniforms);
    S(rz_framebuffer); S(rz_texture); S(rz_cubemap);
    O(rz_blinn_uniforms, joint_matrices); O(rz_blinn_uniforms, shadow_map);
    O(rz_blinn_uniforms, diffuse_map); O(rz_blinn_uniforms, shadow_pass);
    O(rz_pbr_uniforms, basecolor_map); O(rz_pbr_uniforms, normal_map);
    O(rz_pbr_uniforms, ibldata); O(rz_pbr_uniforms, shadow_pass); O(rz_pbr_uniforms, layer_view);
    O(rz_framebuffer, color_buffer); O(rz_framebuffer, depth_buffer); O(rz_texture, buffer);
    return 0;
}
"""


def test_public_struct_layouts_match_the_reference(tmp_path):
    """Sizes and offsets of the reference's structs on x86-64 (blinn_shader.h:8-44,
    pbr_shader.h:20-76, skybox_shader.h:8-19, graphics.h:6-10, texture.h:14-21, mesh.h:8-15)."""
    src = tmp_path / "probe.c"
    src.write_text(LAYOUT_PROBE)
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c89", "-pedantic", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    want = {
        "rz_vertex": 80, "rz_blinn_attribs": 64, "rz_pbr_attribs": 80, "rz_skybox_attribs": 12,
        "rz_blinn_uniforms": 344, "rz_pbr_uniforms": 440, "rz_skybox_uniforms": 72,
        "rz_framebuffer": 24, "rz_texture": 16, "rz_cubemap": 48,
        "rz_blinn_uniforms.joint_matrices": 256, "rz_blinn_uniforms.shadow_map": 280,
        "rz_blinn_uniforms.diffuse_map": 312, "rz_blinn_uniforms.shadow_pass": 340,
        "rz_pbr_uniforms.basecolor_map": 312, "rz_pbr_uniforms.normal_map": 392,
        "rz_pbr_uniforms.ibldata": 416, "rz_pbr_uniforms.shadow_pass": 432, "rz_pbr_uniforms.layer_view": 436,
        "rz_framebuffer.color_buffer": 8, "rz_framebuffer.depth_buffer": 16, "rz_texture.buffer": 8,
    }
    assert {k: int(v) for k, v in got.items()} == want


class Texture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("buffer", C.POINTER(C.c_float))]


def write_tga(path, pixels, rle=False, top_origin=False):
    """pixels: (h, w, c) uint8 RGB(A) or (h, w) gray, row 0 = bottom (the renderer's convention)."""
    h, w = pixels.shape[:2]
    c = 1 if pixels.ndim == 2 else pixels.shape[2]
    rows = pixels.reshape(h, w, c)
    if c >= 3:
        rows = rows[..., [2, 1, 0] + ([3] if c == 4 else [])]       # stored BGR(A)
    if top_origin:
        rows = rows[::-1]
    kind = (3 if c == 1 else 2) + (8 if rle else 0)
    header = struct.pack("<BBBHHBHHHHBB", 0, 0, kind, 0, 0, 0, 0, 0, w, h, c * 8, 0x20 if top_origin else 0)
    flat = rows.reshape(-1, c)
    body = bytearray()
    if not rle:
        body += flat.tobytes()
    else:
        i = 0
        while i < len(flat):
            run = 1
            while i + run < len(flat) and run < 128 and np.array_equal(flat[i + run], flat[i]):
                run += 1
            if run > 1:
                body += bytes([0x80 | (run - 1)]) + flat[i].tobytes()
                i += run
            else:
                raw = 1
                while i + raw < len(flat) and raw < 128 and not np.array_equal(flat[i + raw], flat[i + raw - 1]):
                    raw += 1
                body += bytes([raw - 1]) + flat[i:i + raw].tobytes()
                i += raw
    open(path, "wb").write(header + bytes(body))


@pytest.mark.parametrize("channels,rle,top", [(3, False, False), (4, True, False), (3, True, True), (1, False, True)])
def test_tga_loader_and_ldr_texture_conversion(oracle, tmp_path, channels, rle, top):
    """image.c:172-265 (TGA) and texture.c:32-62: texel = byte / 255, gray replicated, alpha 1 when absent."""
    rng = np.random.default_rng(7)
    h, w = 9, 13
    shape = (h, w) if channels == 1 else (h, w, channels)
    px = rng.integers(0, 256, shape, dtype=np.uint8)
    px[2:5, 3:9] = px[2, 3]                              # some runs for the RLE encoder
    path = str(tmp_path / "t.tga")
    write_tga(path, px, rle, top)
    lib = oracle.lib
    lib.texture_from_file.restype = C.POINTER(Texture)
    lib.texture_from_file.argtypes = [C.c_char_p, C.c_int]
    lib.texture_release.argtypes = [C.POINTER(Texture)]
    tex = lib.texture_from_file(path.encode(), 1)        # RZ_USAGE_LDR_DATA: no colour-space change
    assert (tex.contents.width, tex.contents.height) == (w, h)
    got = np.ctypeslib.as_array(tex.contents.buffer, (h, w, 4)).copy()
    lib.texture_release(tex)
    f = px.astype(np.float32) / np.float32(255)
    want = np.ones((h, w, 4), np.float32)
    if channels == 1:
        want[..., :3] = f[..., None]
    else:
        want[..., :channels] = f
    assert np.array_equal(got, want)


def test_scene_catalogue_and_mesh_sizes(oracle):
    names = []
    while True:
        n = oracle.lib.rzs_catalogue(len(names))
        if not n:
            break
        names.append(n.decode())
    assert len(names) == 25 and {"azura", "helmet", "kgirl", "dieselpunk", "buster"} <= set(names)
    for name, obj in [("helmet", "helmet/helmet.obj")]:
        scene = oracle.open_scene(name)
        info = oracle.scene_info(scene)
        oracle.close_scene(scene)
        faces = sum(1 for line in open(os.path.join(REPO, "assets", obj)) if line.startswith("f "))
        assert info["faces"] == faces == 15452 and info["skybox"] == 1


def test_derived_scene_files_parse(oracle):
    """The .scn variants under scenes/ that turn on what BASELINE's configs name."""
    expect = {"azura_shadow.scn": ("azura", {"shadow_w": 512, "shadow_h": 512}),
              "dieselpunk_shadow.scn": ("dieselpunk", {"shadow_w": 2048, "shadow_h": 2048}),
              "kgirl_blend.scn": ("kgirl", {"skinned": 4}),
              "helmet_cutoff.scn": ("helmet", {"skybox": 1})}
    for scn, (like, fields) in expect.items():
        scene = oracle.open_scene(os.path.join(SCENES, scn), like)
        info = oracle.scene_info(scene)
        oracle.close_scene(scene)
        for k, v in fields.items():
            assert info[k] == v, (scn, info)
    scene = oracle.open_scene(os.path.join(SCENES, "kgirl_blend.scn"), "kgirl")
    assert oracle.scene_info(scene)["blended"] >= 1
    oracle.close_scene(scene)


def test_orbit_views(oracle):
    """eye = 1.5 * (sin a, 0.3, cos a), a = 2 pi i / N, frame_time = i / 30."""
    for i, n in [(0, 64), (5, 16), (63, 64)]:
        v = oracle.orbit_view(i, n)
        a = 2 * math.pi * i / n
        assert np.allclose(list(v.eye), [1.5 * math.sin(a), 0.45, 1.5 * math.cos(a)], atol=1e-6)
        assert list(v.target) == [0, 0, 0] and abs(v.frame_time - i / 30) < 1e-7 and v.layer_view == -1


@pytest.mark.parametrize("case", [c for c in golden_cases() if c["name"] in ("helmet_inside", "kgirl_blend", "azura_shadow")],
                         ids=lambda c: c["name"])
def test_per_triangle_submission_equals_mesh_submission(oracle, case):
    """graphics_draw_triangle per face (the reference's draw_model loop) and graphics_draw_mesh: same frames."""
    oracle.lib.rzs_set_per_triangle(1)
    try:
        a = render_case(oracle, case)
    finally:
        oracle.lib.rzs_set_per_triangle(0)
    b = render_case(oracle, case)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("name", ["kgirl/kgirl.ani", "buster/buster.ani", "phoenix/phoenix.ani", "drone/drone.ani", "drone/fire.ani"])
def test_skeleton_key_bisection_equals_linear_scan(oracle, name):
    """Host side of the path's caller: joint and normal matrices of a pose are the same bytes whether
    the key segment of a track is found by bisection (what the scene layer does) or by the
    reference's linear scan (skeleton.c:231-313), over times inside, between and beyond the keys."""
    lib = oracle.lib
    lib.rzh_acquire_skeleton.restype = C.c_void_p
    lib.rzh_acquire_skeleton.argtypes = [C.c_char_p]
    lib.rzh_release_skeleton.argtypes = [C.c_void_p]
    lib.rzh_skeleton_update.argtypes = [C.c_void_p, C.c_float]
    lib.rzh_skeleton_use_bisection.argtypes = [C.c_void_p, C.c_int]
    lib.rzh_skeleton_num_joints.argtypes = [C.c_void_p]
    lib.rzh_skeleton_joint_matrices.restype = C.POINTER(C.c_float)
    lib.rzh_skeleton_joint_matrices.argtypes = [C.c_void_p]
    lib.rzh_skeleton_normal_matrices.restype = C.POINTER(C.c_float)
    lib.rzh_skeleton_normal_matrices.argtypes = [C.c_void_p]
    sk = lib.rzh_acquire_skeleton(name.encode())
    n = lib.rzh_skeleton_num_joints(sk)
    rng = np.random.default_rng(7)
    times = np.concatenate([rng.uniform(0, 12, 150), np.arange(0, 4, 1 / 30.0), [0.0, 1e-6, 1000.25]]).astype(np.float32)

    def poses():
        out = []
        for t in times:
            lib.rzh_skeleton_update(sk, C.c_float(float(t)))
            out.append(np.ctypeslib.as_array(lib.rzh_skeleton_joint_matrices(sk), (n * 16,)).copy().view(np.uint32))
            out.append(np.ctypeslib.as_array(lib.rzh_skeleton_normal_matrices(sk), (n * 9,)).copy().view(np.uint32))
        return out
    lib.rzh_skeleton_use_bisection(sk, 1)
    fast = poses()
    lib.rzh_skeleton_use_bisection(sk, 0)
    slow = poses()
    lib.rzh_skeleton_use_bisection(sk, 1)
    lib.rzh_release_skeleton(sk)
    assert all(np.array_equal(a, b) for a, b in zip(fast, slow))
