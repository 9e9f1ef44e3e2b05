"""One rank, one GPU, the N > 1 code path: a world_size-1 "nccl" (RCCL) process group and FrameGather with
`force`, so that the slots, framebuffers_read_device_rgb_begin / _end, dist.gather(async_op=True) on device
tensors and the stream synchronisation behind it all execute on a real device.  Frames gathered to self
must equal what framebuffer_read hands out.  Started by tests/test_gpu_rccl_single.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from renderer_amd import orbit                                  # noqa: E402
from renderer_amd.binding import Pipeline, GPU_LIB              # noqa: E402


def main():
    width, height, block, total = 320, 180, 5, 16
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    gpu = Pipeline(GPU_LIB)
    gpu.lib.mi355r_init(0)
    ok = True
    for scene_name in ("helmet", "buster"):
        scene = gpu.open_scene(scene_name)
        fbs = [gpu.framebuffer(width, height) for _ in range(block)]
        # (channels, lagged, depth, rooted): the bench's own mode first
        for channels, lagged, depth, rooted in ((3, True, 2, False), (3, True, 1, False), (3, True, 2, True), (4, False, 2, False), (3, False, 2, False)):
            gather = orbit.FrameGather(block, height, width, "cuda", 0, 1, depth=depth, channels=channels, lagged=lagged,
                                       root_frames=block if rooted else None, force=True)
            want = None
            steps = 3
            for step in range(steps):
                views = [gpu.orbit_view((i + step) % total, total) for i in range(block)]
                orbit.render_views(gpu, scene, fbs, views)
                gather.submit(gpu, fbs, block)
                if step == steps - 1:
                    # read after the submit: the copies are in flight, the frames must not change under them
                    want = [gpu.read(fb)[0][..., :channels].copy() for fb in fbs]
            gather.drain()
            slot = (steps - 1) % depth
            got = gather.own(slot) if rooted else gather.received(slot)[0]
            got = got.cpu().numpy()
            for i in range(block):
                if not np.array_equal(got[i], want[i]):
                    ok = False
                    print(f"MISMATCH {scene_name} channels={channels} lagged={lagged} depth={depth} rooted={rooted} frame {i}", flush=True)
            del gather
        for fb in fbs:
            gpu.release(fb)
        gpu.close_scene(scene)
    t = torch.tensor([1.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)           # the timing reduction bench.py uses
    ok = ok and float(t.item()) == 1.0
    print(f"backend {dist.get_backend()} world {dist.get_world_size()}", flush=True)
    print("RCCL_OK" if ok else "RCCL_MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
