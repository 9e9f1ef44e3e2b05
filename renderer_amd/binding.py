"""ctypes binding of the C ABI in include/mi355r.h plus the scene layer
(renderer_amd/host/scene.h).  The same binding drives the GPU library
(renderer_amd/libmi355r.so) and, in tests only, the CPU oracle
(oracle/liboracle.so): both export identical symbols."""
import ctypes as C
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_LIB = os.environ.get("MI355R_LIB") or os.path.join(REPO, "renderer_amd", "libmi355r.so")     # MI355R_LIB: another build, for A/B measurements
ASSETS = os.path.join(REPO, "assets")
SCENES = os.path.join(REPO, "scenes")


class Vec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Vec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Framebuffer(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int),
                ("color_buffer", C.POINTER(C.c_ubyte)), ("depth_buffer", C.POINTER(C.c_float))]


class Vertex(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("texcoord", C.c_float * 2), ("normal", C.c_float * 3),
                ("tangent", C.c_float * 4), ("joint", C.c_float * 4), ("weight", C.c_float * 4)]


class View(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("target", C.c_float * 3), ("light_dir", C.c_float * 3),
                ("frame_time", C.c_float), ("layer_view", C.c_int)]


class PassInfo(C.Structure):
    _fields_ = [("num_tris", C.c_int), ("width", C.c_int), ("height", C.c_int), ("bin_edge", C.c_int),
                ("bins_x", C.c_int), ("bins_y", C.c_int), ("bin_entries", C.c_uint)]


class Pipeline:
    """One loaded library (GPU or oracle)."""

    def __init__(self, path=GPU_LIB, assets=ASSETS):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no fallback path")
        self.path = path
        self.lib = lib = C.CDLL(path)
        P = C.c_void_p
        FB = C.POINTER(Framebuffer)
        lib.framebuffer_create.restype = FB
        lib.framebuffer_create.argtypes = [C.c_int, C.c_int]
        lib.framebuffer_release.argtypes = [FB]
        lib.framebuffer_clear_color.argtypes = [FB, Vec4]
        lib.framebuffer_clear_depth.argtypes = [FB, C.c_float]
        lib.framebuffer_sync.argtypes = [FB]
        lib.framebuffer_read.argtypes = [FB, P, P]
        lib.graphics_flush.argtypes = []
        lib.draw2d_draw_point.argtypes = [FB, Vec4, Vec2]
        lib.draw2d_draw_line.argtypes = [FB, Vec4, Vec2, Vec2]
        lib.draw2d_draw_triangle.argtypes = [FB, Vec4, Vec2, Vec2, Vec2]
        # function objects of the binding's own (a test may declare the library's attribute of the same name its way)
        self._draw_texture = C.CFUNCTYPE(None, FB, P, Vec2)(("draw2d_draw_texture", lib))
        self._texture_from_file = C.CFUNCTYPE(P, C.c_char_p, C.c_int)(("texture_from_file", lib))
        self._texture_release = C.CFUNCTYPE(None, P)(("texture_release", lib))
        lib.mi355r_backend.restype = C.c_char_p
        lib.rzs_set_assets_dir.argtypes = [C.c_char_p]
        lib.rzs_scene_open.restype = P
        lib.rzs_scene_open.argtypes = [C.c_char_p, C.c_char_p]
        lib.rzs_scene_close.argtypes = [P]
        lib.rzs_default_view.argtypes = [C.POINTER(View)]
        lib.rzs_orbit_view.argtypes = [C.POINTER(View), C.c_int, C.c_int]
        lib.rzs_scene_draw.argtypes = [P, FB, C.POINTER(View)]
        lib.rzs_scene_draw_no_clear.argtypes = [P, FB, C.POINTER(View)]
        lib.rzs_scene_info.argtypes = [P, C.POINTER(C.c_int)]
        lib.rzs_scene_shadow_buffer.restype = FB
        lib.rzs_scene_shadow_buffer.argtypes = [P]
        lib.rzs_scene_shadow_ring.restype = C.c_int
        lib.rzs_scene_shadow_ring.argtypes = [P]
        lib.rzs_evaluate_poses.restype = C.c_int
        lib.rzs_evaluate_poses.argtypes = [C.c_char_p, P, C.c_int, P, P, C.c_int]
        lib.rzs_set_device_poses.argtypes = [C.c_int]
        lib.rzs_catalogue.restype = C.c_char_p
        lib.rzs_catalogue.argtypes = [C.c_int]
        lib.mi355r_inspect_pass.argtypes = [C.POINTER(PassInfo)]
        lib.mi355r_inspect_pass.restype = C.c_int
        lib.mi355r_inspect_boxes.argtypes = [P, P]
        lib.mi355r_inspect_bins.argtypes = [P, P]
        self.backend = lib.mi355r_backend().decode()
        self.is_gpu = self.backend.startswith("hip")
        if self.is_gpu:
            lib.mi355r_init.argtypes = [C.c_int]
            lib.mi355r_init.restype = C.c_int
            lib.framebuffer_read_device.argtypes = [FB, P, P]
            lib.framebuffer_read_device_rgb.argtypes = [FB, P]
            lib.framebuffers_read_device_rgb.argtypes = [C.POINTER(FB), C.c_int, P, C.c_size_t]
            lib.framebuffers_read_device_rgb_begin.argtypes = [C.POINTER(FB), C.c_int, P, C.c_size_t]
            lib.framebuffers_read_device_rgb_end.argtypes = []
            lib.mi355r_get_counters.argtypes = [C.POINTER(C.c_ulonglong)]
            lib.mi355r_enable_counters.argtypes = [C.c_int]
            lib.mi355r_stream.restype = P
            lib.mi355r_reset_counters.argtypes = []
            lib.mi355r_enable_timing.argtypes = [C.c_int]
            lib.mi355r_get_timing.argtypes = [C.POINTER(C.c_double), C.c_int]
            lib.mi355r_get_timing.restype = C.c_ulonglong
            lib.mi355r_set_scratch_estimate.argtypes = [C.c_ulonglong, C.c_ulonglong]
            lib.mi355r_passes_run_again.restype = C.c_ulonglong
            lib.mi355r_last_raster_kernel.restype = C.c_char_p
        else:
            lib.oracle_get_stats.argtypes = [C.POINTER(C.c_ulonglong)]
        lib.rzs_set_assets_dir(assets.encode())

    # -- scenes -----------------------------------------------------------
    def open_scene(self, name, like=None):
        h = self.lib.rzs_scene_open(name.encode(), like.encode() if like else None)
        if not h:
            raise KeyError(name)
        return h

    def close_scene(self, scene):
        self.lib.rzs_scene_close(scene)

    def scene_info(self, scene):
        out = (C.c_int * 8)()
        self.lib.rzs_scene_info(scene, out)
        keys = ["models", "faces", "skybox", "shadow_w", "shadow_h", "blended", "skinned", "attached"]
        return dict(zip(keys, list(out)))

    def default_view(self):
        v = View()
        self.lib.rzs_default_view(C.byref(v))
        return v

    def orbit_view(self, index, count):
        v = View()
        self.lib.rzs_orbit_view(C.byref(v), index, count)
        return v

    def make_view(self, eye=None, target=None, time=0.0, layer=-1):
        v = self.default_view()
        if eye is not None:
            v.eye[:] = eye
        if target is not None:
            v.target[:] = target
        v.frame_time = time
        v.layer_view = layer
        return v

    def draw(self, scene, fb, view):
        self.lib.rzs_scene_draw(scene, fb, C.byref(view))

    def poses(self, ani, times, on_host=False):
        """(joint matrices [T, J, 4, 4], normal matrices [T, J, 3, 3]) of a .ani skeleton at the given times."""
        times = np.ascontiguousarray(times, np.float32)
        J = self.lib.rzs_evaluate_poses(ani.encode(), None, 0, None, None, 1 if on_host else 0)
        if J < 0:
            raise RuntimeError(f"{ani}: the pipeline does not take this skeleton")
        joints = np.zeros((len(times), J, 4, 4), np.float32)
        normals = np.zeros((len(times), J, 3, 3), np.float32)
        as_p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.rzs_evaluate_poses(ani.encode(), as_p(times), len(times), as_p(joints), as_p(normals), 1 if on_host else 0)
        return joints, normals

    # -- framebuffers -------------------------------------------------------
    def framebuffer(self, width, height):
        return self.lib.framebuffer_create(width, height)

    def release(self, fb):
        self.lib.framebuffer_release(fb)

    def read(self, fb):
        w, h = fb.contents.width, fb.contents.height
        color = np.empty((h, w, 4), np.uint8)
        depth = np.empty((h, w), np.float32)
        self.lib.framebuffer_read(fb, color.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p))
        return color, depth

    def overlay(self, fb, ops):
        """2-D overlays (renderer/core/draw2d.h) over what the framebuffer holds, in order:
        ["point", r, g, b, a, x, y] | ["line", r, g, b, a, x0, y0, x1, y1] |
        ["outline", r, g, b, a, x0, y0, x1, y1, x2, y2] | ["label", file, x, y] (an LDR colour image, added)."""
        for op in ops:
            kind, v = op[0], op[1:]
            if kind == "label":
                tex = self._texture_from_file(v[0].encode(), 0)            # USAGE_LDR_COLOR, texture.h:7
                self._draw_texture(fb, tex, Vec2(v[1], v[2]))
                self._texture_release(tex)
                continue
            color, pts = Vec4(*v[:4]), [Vec2(v[i], v[i + 1]) for i in range(4, len(v), 2)]
            {"point": self.lib.draw2d_draw_point, "line": self.lib.draw2d_draw_line,
             "outline": self.lib.draw2d_draw_triangle}[kind](fb, color, *pts)

    def render(self, scene, width, height, view=None, overlay=()):
        fb = self.framebuffer(width, height)
        self.draw(scene, fb, view if view is not None else self.default_view())
        self.overlay(fb, overlay)
        out = self.read(fb)
        self.release(fb)
        return out

    def inspect(self):
        """Geometry stage of the last pass: dict(info, boxes[T,2], fans[T], counts[bins], entries[E,2])."""
        info = PassInfo()
        if not self.lib.mi355r_inspect_pass(C.byref(info)):
            raise RuntimeError("no pass has been submitted")
        boxes = np.zeros((info.num_tris, 2), np.uint32)
        fans = np.zeros(info.num_tris, np.uint32)
        counts = np.zeros(info.bins_x * info.bins_y, np.uint32)
        entries = np.zeros((info.bin_entries, 2), np.uint32)
        as_p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.mi355r_inspect_boxes(as_p(boxes), as_p(fans))
        self.lib.mi355r_inspect_bins(as_p(counts), as_p(entries))
        return {"info": {k: getattr(info, k) for k, _ in PassInfo._fields_},
                "boxes": boxes, "fans": fans, "counts": counts, "entries": entries}

    def geometry(self, scene, width, height, view=None):
        """Draws one frame and returns inspect() of its (last) pass."""
        fb = self.framebuffer(width, height)
        self.draw(scene, fb, view if view is not None else self.default_view())
        out = self.inspect()
        self.release(fb)
        return out

    KERNELS = ["vertex", "setup", "bin", "raster", "shade_skybox", "shade_blinn", "shade_pbr"]

    def kernel_times(self, reset=True):
        """(passes, launch sequences, {kernel: summed ms}) since timing was enabled/reset."""
        out = (C.c_double * 8)()
        n = self.lib.mi355r_get_timing(out, 1 if reset else 0)
        return int(n), int(out[7]), dict(zip(self.KERNELS, list(out)[:7]))

    def counters(self):
        out = (C.c_ulonglong * 12)()
        if self.is_gpu:
            self.lib.mi355r_get_counters(out)
        else:
            self.lib.oracle_get_stats(out)
        return list(out)
