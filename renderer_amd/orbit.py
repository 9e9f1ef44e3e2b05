"""Frame-parallel orbit rendering: the only multi-GPU form of the path.

Frames of an orbit are independent (same immutable scene data, one camera and
animation time each), so rank r of W renders its own contiguous block of views
into its own framebuffers; nothing is exchanged while rendering.  The one
exchange step is the gather of finished RGBA8 frames on rank 0
(`gather_frames`), a torch.distributed collective: RCCL over xGMI for device
tensors under the "nccl" backend, gloo for host tensors in the CPU tests.
"""
import ctypes as C

import numpy as np


def shard_views(total, rank, world):
    """Contiguous block [first, last) of `total` views for `rank`; blocks differ by at most one view."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request: total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def render_views(pipe, scene, framebuffers, views):
    """Records every view into its framebuffer and submits them together (one flush)."""
    if len(framebuffers) < len(views):
        raise ValueError("fewer framebuffers than views")
    for fb, view in zip(framebuffers, views):
        pipe.draw(scene, fb, view)
    pipe.lib.graphics_flush()


def frames_to_tensor(pipe, framebuffers, count, out):
    """Copies `count` finished RGBA8 frames into the torch uint8 tensor `out` [>=count, H, W, 4].
    Device tensors are filled on the device (framebuffer_read_device), host tensors through
    framebuffer_read."""
    frame_bytes = out[0].numel()
    for i in range(count):
        dst = C.c_void_p(out.data_ptr() + i * frame_bytes)
        if out.is_cuda:
            pipe.lib.framebuffer_read_device(framebuffers[i], dst, None)
        else:
            pipe.lib.framebuffer_read(framebuffers[i], dst, None)
    return out


def gather_frames(local, rank, world, dst=0):
    """Gathers equally shaped per-rank frame tensors on `dst`; returns the list there, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [local]
    gathered = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, gathered, dst=dst)
    return gathered


def assemble(gathered, total, world):
    """Rank-major gathered blocks (each padded to the largest block) -> frames in view order."""
    frames = []
    for r, block in enumerate(gathered):
        first, last = shard_views(total, r, world)
        frames.extend(block[i] for i in range(last - first))
    return frames
