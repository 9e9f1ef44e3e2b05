"""One rank of the CPU rehearsal of the multi-GPU orbit: renders its shard of the views with the
CPU oracle (test infrastructure), gathers RGBA8 frames on rank 0 over gloo, and rank 0 checks
them against rendering every view itself.  Started by tests/test_orbit_gloo.py."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from renderer_amd import orbit                                  # noqa: E402
from renderer_amd.binding import Pipeline                       # noqa: E402


class HalvedCopies:
    """The library as FrameGather's lagged mode sees it, for frames in HOST memory (the oracle has no device):
    framebuffers_read_device_rgb_begin takes the frames as they are now -- later draws into the same
    framebuffers must not change what is delivered -- and _end puts them where the caller asked for them."""

    def __init__(self, pipe):
        self.real, self.pipe, self.held = pipe.lib, pipe, None

    def __getattr__(self, name):
        return getattr(self.real, name)

    def framebuffers_read_device_rgb_begin(self, handles, count, target, stride):
        assert self.held is None
        self.real.graphics_flush()
        frames = [np.ascontiguousarray(self.pipe.read(handles[i])[0][..., :3]) for i in range(count)]
        self.held = (frames, target.value, stride)

    def framebuffers_read_device_rgb_end(self):
        if self.held is None:
            return
        frames, target, stride = self.held
        self.held = None
        for i, frame in enumerate(frames):
            assert frame.nbytes <= stride
            C.memmove(target + i * stride, frame.ctypes.data, frame.nbytes)


def main():
    total, width, height = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    name = sys.argv[4] if len(sys.argv) > 4 else "helmet"
    peer_frames = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "-" else None    # given: the gather-aware split
    channels = int(sys.argv[6]) if len(sys.argv) > 6 else 4          # 3: frames travel without their alpha byte
    lagged = len(sys.argv) > 7 and sys.argv[7] == "lagged"             # the copy-out in two halves, as with frames in device memory
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe = Pipeline(os.path.join(REPO, "oracle", "liboracle.so"))
    scene = pipe.open_scene(name)
    if lagged:
        pipe.lib = HalvedCopies(pipe)
    if peer_frames is None:
        first, last = orbit.shard_views(total, rank, world)
        block = max(orbit.shard_views(total, r, world)[1] - orbit.shard_views(total, r, world)[0] for r in range(world))
        pipeline = orbit.FrameGather(block, height, width, "cpu", rank, world, channels=channels, lagged=lagged)
    else:
        first, last = orbit.shard_views_rooted(total, rank, world, peer_frames)
        block = last - first
        pipeline = orbit.FrameGather(peer_frames, height, width, "cpu", rank, world, root_frames=total - peer_frames * (world - 1), channels=channels,
                                     lagged=lagged)
    fbs = [pipe.framebuffer(width, height) for _ in range(block)]
    # the pipelined gather bench.py uses, three steps through two slots: the last step is the one checked.  Every step
    # shows the orbit one view further on, so frames that ended up in the wrong step's slot would be seen
    for step in range(3):
        views = [pipe.orbit_view((i + step) % total, total) for i in range(first, last)]
        orbit.render_views(pipe, scene, fbs, views)
        pipeline.submit(pipe, fbs, last - first)
    pipeline.drain()
    gathered = pipeline.received(2 % 2)
    # the timing reduction bench.py uses: MAX over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = True
    if rank == 0:
        if peer_frames is None:
            frames = orbit.assemble(gathered, total, world)
        else:
            frames = orbit.assemble_rooted(pipeline.own(2 % 2), gathered, total, world, peer_frames)
        ok = len(frames) == total and float(t.item()) == float(world)
        for i, frame in enumerate(frames):
            want, _ = pipe.render(scene, width, height, pipe.orbit_view((i + 2) % total, total))
            ok = ok and np.array_equal(frame.numpy(), want[..., :channels])
        print("ORBIT_OK" if ok else "ORBIT_MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
