"""texture_sync_noalloc against texture_sync (GPU): the entry point that allocates nothing -- its staging
image lives in a fixed array inside the library's device image -- must leave the same bytes in
texture->buffer as the entry point that allocates and frees a staging buffer per call.  Two calls in a
row, the second with a smaller texture (the staging array still holds the first call's texels), through
the C ABI, with the device's free memory the same before and after; then the limits: a texture beyond the staging array's size and a copy that is still
recorded are refused with an error code and leave the host image alone."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Texture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("buffer", C.POINTER(C.c_float))]


def _api(lib):
    T = C.POINTER(_Texture)
    lib.texture_create.restype = T
    lib.texture_create.argtypes = [C.c_int, C.c_int]
    lib.texture_release.argtypes = [T]
    lib.texture_sync.argtypes = [T]
    lib.texture_sync_noalloc.argtypes = [T]
    lib.texture_sync_noalloc.restype = C.c_int
    lib.texture_from_colorbuffer.argtypes = [T, C.c_void_p]
    lib.texture_from_depthbuffer.argtypes = [T, C.c_void_p]


def _free_device_memory():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _texels(t):
    return np.ctypeslib.as_array(t.contents.buffer, (t.contents.height, t.contents.width, 4))


def test_texture_sync_without_allocations_equals_texture_sync(gpu):
    lib = gpu.lib
    _api(lib)
    scene = gpu.open_scene("helmet")
    for (w, h), depth in [((512, 256), True), ((96, 64), False), ((1024, 1024), True), ((33, 17), True)]:
        fb = gpu.framebuffer(w, h)
        gpu.draw(scene, fb, gpu.orbit_view(3, 16))
        frame = gpu.read(fb)                                        # nothing is recorded any more
        a, b = lib.texture_create(w, h), lib.texture_create(w, h)
        copy = lib.texture_from_depthbuffer if depth else lib.texture_from_colorbuffer
        copy(a, fb)
        copy(b, fb)
        _texels(b)[...] = -1.0                                      # every texel has to be written by the call
        lib.texture_sync(a)
        free_before = _free_device_memory()
        assert lib.texture_sync_noalloc(b) == 0
        assert _free_device_memory() == free_before             # nothing was allocated (and kept), nothing freed
        got, want = _texels(b).copy(), _texels(a).copy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, depth)
        if depth:
            assert np.array_equal(want[..., 0], frame[1]) and np.all(want[..., 3] == 1)
        else:
            assert np.array_equal(want, frame[0].astype(np.float32) / np.float32(255))
        assert lib.texture_sync_noalloc(b) == 0                     # the host image is current: nothing to do
        lib.texture_release(a)
        lib.texture_release(b)
        gpu.release(fb)
    # beyond the staging array: refused, host image untouched; texture_sync still serves it
    w, h = 1280, 1024
    fb = gpu.framebuffer(w, h)
    gpu.draw(scene, fb, gpu.orbit_view(5, 16))
    gpu.read(fb)
    big = lib.texture_create(w, h)
    lib.texture_from_depthbuffer(big, fb)
    _texels(big)[...] = -2.0
    assert lib.texture_sync_noalloc(big) == -1
    assert np.all(_texels(big) == -2.0)
    lib.texture_sync(big)
    assert np.all(_texels(big)[..., 3] == 1)
    lib.texture_release(big)
    gpu.release(fb)
    # a copy that is recorded behind draws but not submitted: refused until graphics_flush
    w, h = 128, 72
    fb = gpu.framebuffer(w, h)
    gpu.draw(scene, fb, gpu.orbit_view(7, 16))
    pending = lib.texture_create(w, h)
    lib.texture_from_depthbuffer(pending, fb)                       # recorded behind the frame's draws
    _texels(pending)[...] = -3.0
    assert lib.texture_sync_noalloc(pending) == -2
    assert np.all(_texels(pending) == -3.0)
    lib.graphics_flush()
    assert lib.texture_sync_noalloc(pending) == 0
    frame = gpu.read(fb)
    assert np.array_equal(_texels(pending)[..., 0], frame[1])
    lib.texture_release(pending)
    gpu.release(fb)
    gpu.close_scene(scene)
