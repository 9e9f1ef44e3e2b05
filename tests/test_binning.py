"""Integer screen-bin lists: the CPU binning oracle (oracle/binning.c) against a
brute-force statement in numpy (CPU), and the GPU bin kernel against the oracle
(GPU).  Everything here is integer work: the bar is bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ORACLE_LIB, REPO


def brute_force_bins(boxes, fans, width, height, edge):
    """Per bin, every (triangle, fan index) whose box touches the bin, ascending."""
    bins_x, bins_y = (width + edge - 1) // edge, (height + edge - 1) // edge
    x0, x1 = (boxes[:, 0] & 0xffff).astype(int), (boxes[:, 0] >> 16).astype(int)
    y0, y1 = (boxes[:, 1] & 0xffff).astype(int), (boxes[:, 1] >> 16).astype(int)
    alive = (fans > 0) & (x0 <= x1) & (y0 <= y1)
    counts, entries = [], []
    for by in range(bins_y):
        for bx in range(bins_x):
            rx0, ry0 = bx * edge, by * edge
            hit = alive & (x0 <= rx0 + edge - 1) & (x1 >= rx0) & (y0 <= ry0 + edge - 1) & (y1 >= ry0)
            n = 0
            for t in np.nonzero(hit)[0]:
                for k in range(int(fans[t])):
                    entries.append((t, k))
                    n += 1
            counts.append(n)
    return np.array(counts, np.uint32), np.array(entries, np.uint32).reshape(-1, 2)


def oracle_bins(lib, boxes, fans, width, height, edge):
    lib.oracle_bin.restype = C.c_size_t
    lib.oracle_bin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    bins = ((width + edge - 1) // edge) * ((height + edge - 1) // edge)
    counts = np.zeros(bins, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = lib.oracle_bin(p(boxes), p(fans), len(fans), width, height, edge, p(counts), None)
    entries = np.zeros((n, 2), np.uint32)
    assert lib.oracle_bin(p(boxes), p(fans), len(fans), width, height, edge, p(counts), p(entries)) == n
    return counts, entries


def random_boxes(rng, n, width, height):
    x = np.sort(rng.integers(0, width, (n, 2)), axis=1)
    y = np.sort(rng.integers(0, height, (n, 2)), axis=1)
    small = rng.random(n) < 0.7                       # most triangles are a few pixels wide
    x[small, 1] = np.minimum(x[small, 0] + rng.integers(0, 12, small.sum()), width - 1)
    y[small, 1] = np.minimum(y[small, 0] + rng.integers(0, 12, small.sum()), height - 1)
    boxes = np.stack([x[:, 0] | (x[:, 1] << 16), y[:, 0] | (y[:, 1] << 16)], axis=1).astype(np.uint32)
    fans = rng.choice([0, 1, 1, 1, 1, 2, 3, 8], n).astype(np.uint32)
    boxes[fans == 0] = 0x0000ffff                     # culled: the empty box
    return boxes, fans


@pytest.mark.parametrize("size,edge,n", [((64, 64), 64, 1), ((65, 63), 64, 50), ((800, 600), 64, 3000),
                                          ((1920, 1080), 64, 5000), ((333, 77), 16, 700)])
def test_binning_oracle_matches_brute_force(oracle, size, edge, n):
    rng = np.random.default_rng(1234 + n)
    boxes, fans = random_boxes(rng, n, *size)
    want_counts, want_entries = brute_force_bins(boxes, fans, size[0], size[1], edge)
    counts, entries = oracle_bins(oracle.lib, boxes, fans, size[0], size[1], edge)
    assert np.array_equal(counts, want_counts)
    assert np.array_equal(entries, want_entries)


def test_binning_oracle_edge_cases(oracle):
    # no triangles; only culled triangles; one box covering the whole screen; boxes on bin borders
    w, h, e = 200, 130, 64
    empty = np.zeros((0, 2), np.uint32), np.zeros(0, np.uint32)
    counts, entries = oracle_bins(oracle.lib, *empty, w, h, e)
    assert counts.sum() == 0 and len(entries) == 0
    boxes = np.full((5, 2), 0x0000ffff, np.uint32)
    counts, entries = oracle_bins(oracle.lib, boxes, np.zeros(5, np.uint32), w, h, e)
    assert counts.sum() == 0
    full = np.array([[0 | (w - 1) << 16, 0 | (h - 1) << 16]], np.uint32)
    counts, entries = oracle_bins(oracle.lib, full, np.array([3], np.uint32), w, h, e)
    assert np.all(counts == 3) and len(entries) == 3 * 4 * 3
    border = np.array([[63 | 64 << 16, 63 | 64 << 16], [64 | 64 << 16, 0 | 0 << 16]], np.uint32)
    counts, entries = oracle_bins(oracle.lib, border, np.array([1, 1], np.uint32), w, h, e)
    assert counts.reshape(3, 4).tolist() == [[1, 2, 0, 0], [1, 1, 0, 0], [0, 0, 0, 0]]


def test_oracle_trace_is_consistent(oracle):
    """The oracle's own trace of a frame: fan counts add up to its rasterised-triangle count."""
    scene = oracle.open_scene("helmet")
    oracle.lib.oracle_reset_stats()
    g = oracle.geometry(scene, 320, 180)
    stats = oracle.counters()
    oracle.close_scene(scene)
    assert g["info"]["num_tris"] == stats[0]
    # every rasterised triangle is either in a fan with pixels or dropped for having none
    assert int(g["fans"].sum()) <= stats[1]
    assert g["counts"].sum() == len(g["entries"]) == g["info"]["bin_entries"]


GEOMETRY_CASES = [("helmet", (640, 360), None), ("helmet", (1920, 1080), (0.9, 0.4, 1.1)),
                  ("kgirl", (480, 270), None), ("dieselpunk", (800, 450), None),
                  ("buster", (480, 270), None), ("azura", (400, 300), None)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,size,eye", GEOMETRY_CASES)
def test_gpu_bins_match_cpu_binning_oracle(gpu, oracle, name, size, eye):
    """Setup (boxes, fan lengths) and binning (ordered lists) of a whole frame, bit-exact."""
    results = []
    for pipe in (gpu, oracle):
        scene = pipe.open_scene(name)
        view = pipe.make_view(eye, None, 0.25)
        pipe.render(scene, 64, 36, view)        # first use uploads textures, which splits the frame into two passes
        results.append(pipe.geometry(scene, size[0], size[1], view))
        pipe.close_scene(scene)
    got, want = results
    assert np.array_equal(got["fans"], want["fans"])
    assert np.array_equal(got["boxes"], want["boxes"])
    # a sky box drawn last in its pass is kept out of the library's bins (its triangles go on a table of their own):
    # the lists must be the oracle's without those triangles' entries
    sky_first = gpu.lib.mi355r_inspect_sky_first()
    assert want["info"]["num_tris"] - sky_first in ((0, 12) if name == "helmet" else (0,))      # one box of 12 triangles
    kept = want["entries"][:, 0] < sky_first
    edge = want["info"]["bin_edge"]
    if not kept.all():
        counts, entries = brute_force_bins(want["boxes"][:sky_first], want["fans"][:sky_first], size[0], size[1], edge)
        assert np.array_equal(entries, want["entries"][kept])
        want = dict(want, counts=counts, entries=entries, info=dict(want["info"], bin_entries=len(entries)))
    assert got["info"] == want["info"]
    assert np.array_equal(got["counts"], want["counts"])
    assert np.array_equal(got["entries"], want["entries"])
    want = results[1]
    # and the oracle's lists are what brute force gives for these boxes
    bc, be = brute_force_bins(want["boxes"], want["fans"], size[0], size[1], want["info"]["bin_edge"])
    assert np.array_equal(want["counts"], bc) and np.array_equal(want["entries"], be)
