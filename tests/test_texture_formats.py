"""Lossless device image formats: the host-side decision (mi355r_pack_texels, CPU) must pack an
image only when decoding gives back the very same floats, and frames rendered from packed images
must equal frames rendered from float4 images (GPU)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from renderer_amd.binding import GPU_LIB


@pytest.fixture(scope="module")
def pack():
    lib = C.CDLL(GPU_LIB)              # loading and calling this entry point needs no device
    lib.mi355r_pack_texels.restype = C.c_int
    lib.mi355r_pack_texels.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]

    def run(texels):
        texels = np.ascontiguousarray(texels, np.float32).reshape(-1, 4)
        out = np.zeros(len(texels), np.uint32)
        fmt = lib.mi355r_pack_texels(texels.ctypes.data_as(C.c_void_p), len(texels), out.ctypes.data_as(C.c_void_p))
        return fmt, out
    return run


UNORM = (np.arange(256, dtype=np.float32) / np.float32(255)).astype(np.float32)            # texture.c:40-61
SRGB = np.power(UNORM.astype(np.float64), 2.2).astype(np.float32)                           # maths.c:37-39


def decode(fmt, words):
    b = [(words >> s) & 0xff for s in (0, 8, 16, 24)]
    if fmt == 4:
        return np.stack([UNORM[b[0]], UNORM[b[1]], UNORM[b[2]], UNORM[b[3]]], axis=1)
    if fmt == 8:
        return np.stack([SRGB[b[0]], SRGB[b[1]], SRGB[b[2]], UNORM[b[3]]], axis=1)
    factor = np.ldexp(np.float32(1), b[3].astype(np.int32) - 136).astype(np.float32)       # image.c:329-347
    rgb = [(b[i].astype(np.float32) * factor).astype(np.float32) for i in range(3)]
    return np.stack(rgb + [np.ones(len(words), np.float32)], axis=1)


def test_eight_bit_images_pack_and_decode_exactly(pack):
    rng = np.random.default_rng(3)
    k = rng.integers(0, 256, (5000, 4))
    data = np.stack([UNORM[k[:, i]] for i in range(4)], axis=1)
    fmt, words = pack(data)
    assert fmt == 4 and np.array_equal(decode(fmt, words), data)
    colour = np.stack([SRGB[k[:, 0]], SRGB[k[:, 1]], SRGB[k[:, 2]], UNORM[k[:, 3]]], axis=1)
    fmt, words = pack(colour)
    assert fmt == 8 and np.array_equal(decode(fmt, words), colour)


def test_rgbe_images_pack_and_decode_exactly(pack):
    rng = np.random.default_rng(4)
    rgbe = rng.integers(0, 256, (5000, 4))
    rgbe[:50, 3] = 0                                       # exponent 0 decodes to black
    rgbe[50:60, 3] = rng.integers(1, 12, 10)               # tiny exponents: denormal floats
    rgbe[60:70, :3] = [[3, 0, 1]] * 10                     # unnormalised mantissas
    factor = np.ldexp(1.0 / 256, rgbe[:, 3] - 128).astype(np.float32)
    texels = np.ones((5000, 4), np.float32)
    for c in range(3):
        texels[:, c] = np.where(rgbe[:, 3] == 0, 0, rgbe[:, c].astype(np.float32) * factor)
    fmt, words = pack(texels)
    assert fmt == 16 and np.array_equal(decode(fmt, words), texels)


@pytest.mark.parametrize("spoil", ["value", "alpha", "nan", "negative", "huge"])
def test_images_that_do_not_fit_stay_float4(pack, spoil):
    rng = np.random.default_rng(5)
    k = rng.integers(0, 256, (1000, 4))
    data = np.stack([UNORM[k[:, i]] for i in range(4)], axis=1)
    bad = {"value": (7, 1, np.nextafter(np.float32(0.5), np.float32(1))), "alpha": (7, 3, np.float32(0.123)),
           "nan": (7, 0, np.float32("nan")), "negative": (7, 2, np.float32(-0.25)), "huge": (7, 0, np.float32(3e38))}[spoil]
    data[bad[0], bad[1]] = bad[2]
    fmt, _ = pack(data)
    assert fmt == 0
    # the same for an otherwise shared-exponent image
    hdr = np.ones((100, 4), np.float32)
    hdr[:, :3] = (rng.integers(0, 256, (100, 3)) * 2.0 ** -7).astype(np.float32)
    assert pack(hdr)[0] == 16
    hdr[3, 1] = np.float32(1 / 3)
    assert pack(hdr)[0] == 0


def test_every_loaded_scene_texture_round_trips(pack, oracle):
    """All images of the benchmark scene as the host layer builds them (TGA / HDR decode + usage
    conversion): whatever format is chosen decodes to the identical floats."""
    class Texture(C.Structure):
        _fields_ = [("width", C.c_int), ("height", C.c_int), ("buffer", C.POINTER(C.c_float))]
    lib = oracle.lib
    lib.texture_from_file.restype = C.POINTER(Texture)
    lib.texture_from_file.argtypes = [C.c_char_p, C.c_int]
    lib.texture_release.argtypes = [C.POINTER(Texture)]
    seen = set()
    for name, usage in [("helmet/helmet_basecolor.tga", 2), ("helmet/helmet_metalness.tga", 1), ("helmet/helmet_normal.tga", 1),
                        ("common/brdf_lut.hdr", 3), ("workshop/i_px.hdr", 2), ("workshop/m1_nz.hdr", 0)]:
        tex = lib.texture_from_file(name.encode(), usage)
        n = tex.contents.width * tex.contents.height
        texels = np.ctypeslib.as_array(tex.contents.buffer, (n, 4)).copy()
        lib.texture_release(tex)
        fmt, words = pack(texels)
        seen.add(fmt)
        if fmt:
            assert np.array_equal(decode(fmt, words), texels), name
    assert seen == {0, 4, 8, 16}       # tone-mapped sky face, data map, colour map, radiance maps


VARIANT_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from conftest import golden_cases, load_golden, render_case
import os
from renderer_amd.binding import Pipeline, GPU_LIB, SCENES
gpu = Pipeline(GPU_LIB)
bad = 0
for case in golden_cases():
    if case["name"] not in ("helmet_default", "helmet_cutoff", "azura_shadow", "kgirl_blend", "drone_specgloss", "dieselpunk_shadow", "synthetic_blinn"):
        continue
    want = load_golden(case["name"]); got = render_case(gpu, case)
    bad += int((got[0] != want[0]).sum()) + int((got[1] != want[1]).sum())
    if case["name"] in ("helmet_default", "azura_shadow"):
        # twice into one framebuffer: the second frame's strip height follows the first frame's lists
        scene = gpu.open_scene(os.path.join(SCENES, case["scene"]), case.get("like")) if case["scene"].endswith(".scn") else gpu.open_scene(case["scene"])
        view = gpu.make_view(case.get("eye"), case.get("target"), case.get("time", 0.0), case.get("layer", -1))
        fb = gpu.framebuffer(*case["size"])
        for _ in range(2):
            gpu.draw(scene, fb, view)
            got = gpu.read(fb)
            bad += int((got[0] != want[0]).sum()) + int((got[1] != want[1]).sum())
        gpu.release(fb); gpu.close_scene(scene)
print("DIFFS", bad)
sys.exit(0 if bad == 0 else 1)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"MI355R_PACK": "0"}, {"MI355R_LANES": "1", "MI355R_BATCH": "1"},
                                 {"MI355R_MICRO_AREA": "0"}, {"MI355R_MICRO_MIN_LIST": "0"}, {"MI355R_RECORD_COPIES": "0"},
                                 {"MI355R_STRIP": "4"}, {"MI355R_SKY_TABLE": "0"}, {"MI355R_FILL_WAVES": "4"}, {"MI355R_BUNDLE": "0"}, {"MI355R_SKY_SIDE": "0"}, {"MI355R_CROWD_MIN": "200"}, {"MI355R_CROWD_MIN": "300", "MI355R_CROWD_SPLIT": "2"}, {"MI355R_CROWD_MIN": "0"}, {"MI355R_SKY_BYTES": "0"}, {"MI355R_STRIP": "2"},
                                 {"MI355R_STRIP_LONG": "1"}, {"MI355R_STRIP_LONG": "600"}, {"MI355R_STRIP": "1", "MI355R_MICRO_MIN_LIST": "0"},
                                 {"MI355R_SPLIT_BLEND": "0"}, {"MI355R_SPLIT_BLEND": "1"},
                                 {"MI355R_MICRO_MIN_LIST": "0", "MI355R_MICRO_AREA": "128"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_gpu_alternative_code_paths_give_the_same_frames(gpu, env):
    """The default configuration is what every other GPU test checks against the goldens; here the
    same goldens through the library's other settings of its live code paths (chosen by environment
    variables read when it starts): float4 images only, no batching, the one-triangle-per-lane
    rasteriser for no triangle / in every bin / for every triangle a block can hold, a flush
    in front of every framebuffer -> texture copy, strips of a fixed height, strips whose height
    follows the framebuffer's last frame from a list length every scene / only azura reaches, sky boxes
    through the bins like any other draw, and visibility-first blending on and off."""
    r = subprocess.run([sys.executable, "-c", VARIANT_SCRIPT % (REPO, os.path.join(REPO, "tests"))],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DIFFS 0" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_gpu_images_ingested_from_file_bytes_decode_to_the_reference_floats(gpu, oracle):
    """texture_from_file on the GPU library hands the file's bytes to the device as they are
    (texture_create_packed) and never builds the float image; texture_sync then asks the device to decode
    it.  Those floats must be, bit for bit, what the reference's conversion (texture.c:32-128) produces
    from the same file, which the oracle's loader computes on the host: 8-bit data, colour and
    single-channel maps, Radiance maps, and a tone-mapped sky face (which has no packed form)."""
    class Texture(C.Structure):
        _fields_ = [("width", C.c_int), ("height", C.c_int), ("buffer", C.POINTER(C.c_float))]
    T = C.POINTER(Texture)
    cases = [("helmet/helmet_basecolor.tga", 2), ("helmet/helmet_metalness.tga", 1), ("helmet/helmet_normal.tga", 1),
             ("helmet/helmet_emission.tga", 2), ("common/brdf_lut.hdr", 3), ("workshop/i_px.hdr", 2), ("workshop/m0_py.hdr", 2),
             ("workshop/m1_nz.hdr", 0), ("azura/hair_diffuse.tga", 0), ("azura/body_diffuse.tga", 0), ("helmet/helmet_occlusion.tga", 1)]
    import os
    from renderer_amd.binding import ASSETS
    images = {}
    for pipe in (gpu, oracle):
        lib = pipe.lib
        lib.texture_from_file.restype = T
        lib.texture_from_file.argtypes = [C.c_char_p, C.c_int]
        lib.texture_release.argtypes = [T]
        lib.texture_sync.argtypes = [T]
        for name, usage in cases:
            if not os.path.exists(os.path.join(ASSETS, name)):
                continue
            tex = lib.texture_from_file(name.encode(), usage)
            if pipe is gpu and (usage != 0 or name.endswith(".tga")):
                assert not tex.contents.buffer            # no float image until somebody asks
            lib.texture_sync(tex)
            n = tex.contents.width * tex.contents.height
            images.setdefault((name, usage), []).append(np.ctypeslib.as_array(tex.contents.buffer, (n, 4)).copy())
            lib.texture_release(tex)
    assert len(images) >= 8
    for key, (got, want) in images.items():
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), key
