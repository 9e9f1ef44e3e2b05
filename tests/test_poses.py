"""Skeletal poses (renderer/core/skeleton.c:231-341): the three evaluations this repository holds against
each other, for the five .ani files the scenes ship, at 64 times each (inside the animation, at its very
start and end, beyond its end where time wraps, and at key times exactly).
  * oracle/pose.c       the reference's algorithm restated (linear key scan), pinned by the golden frames
  * host/assets.c       the scene layer's per-frame evaluation (bisection), what per-triangle submission uses
  * rz_pose_kernel      the device (one wave per pose, all poses of a flush in one launch)
The device calls acos / sin in double like the reference; its library need not round the last bit of a
double like glibc, which after narrowing to float could show once in ~10^8 calls: the bar is bit-equality,
with at most 2 matrices per skeleton allowed to differ by an ulp (none do on the shipped files)."""
import numpy as np
import pytest

ANI = ["kgirl/kgirl.ani", "buster/buster.ani", "drone/drone.ani", "drone/fire.ani", "phoenix/phoenix.ani"]


def sample_times(seed):
    rng = np.random.default_rng(seed)
    t = np.concatenate([[0.0, 1e-6, 1.0 / 30, 0.5, 1.0, 2.0, 3.3333333, 16.866667, 25.0, 40.0, 123.456],
                        rng.uniform(0, 30, 53)]).astype(np.float32)
    assert len(t) == 64
    return t


@pytest.mark.parametrize("ani", ANI)
def test_host_pose_evaluation_equals_restated_reference(oracle, ani):
    times = sample_times(1)
    restated = oracle.poses(ani, times)                 # oracle/pose.c through the pose entry points
    host = oracle.poses(ani, times, on_host=True)       # host/assets.c
    assert np.array_equal(restated[0].view(np.uint32), host[0].view(np.uint32))
    assert np.array_equal(restated[1].view(np.uint32), host[1].view(np.uint32))
    assert np.isfinite(restated[0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("ani", ANI)
def test_gpu_pose_kernel_equals_restated_reference(gpu, oracle, ani):
    times = sample_times(2)
    got = gpu.poses(ani, times)
    want = oracle.poses(ani, times)
    for g, w, what in zip(got, want, ("joint", "normal")):
        differing = np.argwhere((g.view(np.uint32) != w.view(np.uint32)).any(axis=(2, 3)))
        print(f"{ani}: {what} matrices that differ: {len(differing)} of {g.shape[0] * g.shape[1]}")
        assert len(differing) <= 2, differing[:8]
        np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-30)


@pytest.mark.gpu
def test_gpu_frames_with_host_poses_equal_frames_with_device_poses(gpu):
    """The same animated frames with the pose evaluated per frame on the host (joint matrices copied with
    every draw) and on the device (graphics_draw_mesh_posed): identical bytes; skinned (kgirl) and
    node-attached (buster)."""
    import os
    from renderer_amd.binding import SCENES
    for name, like in ((os.path.join(SCENES, "kgirl_blend.scn"), "kgirl"), ("buster", None)):
        scene = gpu.open_scene(name, like)
        frames = []
        for on_device in (1, 0):
            gpu.lib.rzs_set_device_poses(on_device)
            try:
                frames.append([gpu.render(scene, 320, 180, gpu.orbit_view(i, 24)) for i in (0, 5, 17)])
            finally:
                gpu.lib.rzs_set_device_poses(1)
        gpu.close_scene(scene)
        for a, b in zip(*frames):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
