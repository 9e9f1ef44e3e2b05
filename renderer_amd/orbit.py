"""Frame-parallel orbit rendering: the only multi-GPU form of the path.

Frames of an orbit are independent (same immutable scene data, one camera and
animation time each), so rank r of W renders its own contiguous block of views
into its own framebuffers; nothing is exchanged while rendering.  The one
exchange step is the gather of finished RGBA8 frames on rank 0
(`gather_frames`), a torch.distributed collective: RCCL over xGMI for device
tensors under the "nccl" backend, gloo for host tensors in the CPU tests.
Blocks are either even (`shard_views`) or gather-aware (`shard_views_rooted`:
more views on rank 0, whose frames need not cross a link).
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


def shard_views_rooted(total, rank, world, peer_frames):
    """Blocks for a gather-aware split: every rank but 0 renders `peer_frames` views, rank 0 the rest.

    Rank 0 keeps its frames where they are, a peer's frames have to cross its xGMI link to rank 0;
    when a GPU renders faster than a link carries frames, the job is fastest with more views on
    rank 0 (see bench.py).  peer_frames = total // world gives the even split."""
    if not (0 <= rank < world) or total < 0 or peer_frames < 0 or peer_frames * (world - 1) > total:
        raise ValueError(f"bad shard request: total={total} rank={rank} world={world} peer_frames={peer_frames}")
    root = total - peer_frames * (world - 1)
    if rank == 0:
        return 0, root
    first = root + (rank - 1) * peer_frames
    return first, first + peer_frames


def render_views(pipe, scene, framebuffers, views):
    """Records every view into its framebuffer and submits them together (one flush)."""
    if len(framebuffers) < len(views):
        raise ValueError("fewer framebuffers than views")
    for fb, view in zip(framebuffers, views):
        pipe.draw(scene, fb, view)
    pipe.lib.graphics_flush()


def frames_to_tensor(pipe, framebuffers, count, out):
    """Copies `count` finished frames into the torch uint8 tensor `out` [>=count, H, W, 4 or 3]: RGBA8 as
    the framebuffer holds it, or RGB8 (the alpha byte is never written by a draw, graphics.c:466-468, so
    a gather can leave it behind).  Device tensors are filled on the device (framebuffer_read_device /
    framebuffer_read_device_rgb), host tensors through framebuffer_read."""
    frame_bytes = out[0].numel()
    channels = out.shape[-1]
    if out.is_cuda and channels == 3 and count > 0:
        # all of them behind the rendering on their own streams, one wait for the lot
        handles = (type(framebuffers[0]) * count)(*framebuffers[:count])
        pipe.lib.framebuffers_read_device_rgb(handles, count, C.c_void_p(out.data_ptr()), frame_bytes)
        return out
    for i in range(count):
        dst = C.c_void_p(out.data_ptr() + i * frame_bytes)
        if out.is_cuda and channels == 3:
            pipe.lib.framebuffer_read_device_rgb(framebuffers[i], dst)
        elif out.is_cuda:
            pipe.lib.framebuffer_read_device(framebuffers[i], dst, None)
        elif channels == 3:
            import torch
            h, w = out.shape[1], out.shape[2]
            rgba = np.empty((h, w, 4), np.uint8)
            pipe.lib.framebuffer_read(framebuffers[i], rgba.ctypes.data_as(C.c_void_p), None)
            out[i].copy_(torch.from_numpy(np.ascontiguousarray(rgba[..., :3])))
        else:
            pipe.lib.framebuffer_read(framebuffers[i], dst, None)
    return out


def split_for(render_s, delivery_s, frames, world):
    """Frames per peer and step for the gather-aware split (shard_views_rooted).  A step is frames * world
    views; rank 0 renders `frames` views in render_s, and `frames` views per peer arrive on rank 0 in
    delivery_s when all peers send at once.  With P views per peer rank 0 renders the other
    frames * world - (world - 1) * P, which takes as long as the delivery of P views from each peer when
        P = frames * world / (delivery_s / render_s + world - 1),
    clamped to [1, frames] (frames = the even split)."""
    if world < 2:
        return frames
    if not (render_s > 0 and delivery_s > 0):
        return frames
    peer = int(frames * world / (delivery_s / render_s + (world - 1)))
    return max(1, min(frames, peer))


def gather_frames(local, rank, world, dst=0):
    """Gathers equally shaped per-rank frame tensors on `dst`; returns the list there, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [local]
    gathered = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, gathered, dst=dst)
    return gathered


class FrameGather:
    """The gather as a pipeline of `depth` slots: a step's frames go into the next slot and their
    gather starts at once (async_op); the slot is only waited for when it comes round again, so
    the transfer of one step overlaps the rendering of the next.  Rank 0 ends up with
    `received(slot)` = list over ranks of that step's frames.  world == 1: nothing is moved, unless
    `force` is set -- then the one rank gathers to itself through the same slots, copies and collective
    calls (how the N > 1 code path is exercised on a single GPU).

    `lagged` (the default for RGB8 frames in device memory): the copy of a step's frames into the slot is only
    launched by `submit` (framebuffers_read_device_rgb_begin: behind the rendering, no wait), and waited for
    -- and the slot's gather started -- by the NEXT `submit` or by `drain`.  The host then records the next
    step while the GPU still renders and copies this one, instead of waiting for the copies first."""

    def __init__(self, frames, height, width, device, rank, world, depth=2, root_frames=None, channels=4, lagged=None, force=False):
        """frames: slot size of the collective (the largest block a rank sends).  root_frames: the
        gather-aware split (shard_views_rooted) -- rank 0's block has that many frames and does not
        go through the collective: rank 0 copies it into a buffer of its own per slot (`own(slot)`),
        and takes part in the gather with an unused slot."""
        import torch
        self.rank, self.world, self.depth, self.count = rank, world, depth, 0
        self.active = world > 1 or force
        self.rooted = root_frames is not None
        self.on_device = torch.device(device).type == "cuda"
        self.lagged = (self.on_device and channels == 3) if lagged is None else lagged
        self.pending = None             # (pipe, slot): copies launched, gather not started yet
        self.slots = self.inbox = self.work = self.kept = None
        if self.active:
            shape = (frames, height, width, channels)
            self.slots = [torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(depth)]
            self.inbox = [[torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
                          for _ in range(depth)]
            if self.rooted and rank == 0:
                self.kept = [torch.empty((root_frames, height, width, channels), dtype=torch.uint8, device=device) for _ in range(depth)]
            self.work = [None] * depth

    def _finish(self, slot):
        import torch
        if self.work[slot] is not None:
            self.work[slot].wait()
            if self.on_device:
                torch.cuda.current_stream().synchronize()   # wait() only orders streams; the slot is reused by other streams
            self.work[slot] = None

    def _start(self, slot):
        import torch.distributed as dist
        self.work[slot] = dist.gather(self.slots[slot], self.inbox[slot], dst=0, async_op=True)

    def _start_pending(self):
        """The step whose copies `submit` launched last: wait for them, then start its gather."""
        if self.pending is not None:
            pipe, slot = self.pending
            self.pending = None
            pipe.lib.framebuffers_read_device_rgb_end()
            self._start(slot)

    def submit(self, pipe, framebuffers, count):
        """Copies `count` finished frames into the next slot and starts gathering it (lagged: launches the
        copies, and starts gathering the step before)."""
        if not self.active:
            return
        slot = self.count % self.depth
        # the step before's gather is started first and only then is this slot waited for: with depth 1 they are
        # the same slot, and its gather must have finished before this step's copies overwrite it
        self._start_pending()
        self._finish(slot)
        target = self.kept[slot] if self.rooted and self.rank == 0 else self.slots[slot]
        if self.lagged:
            if count > 0:
                handles = (type(framebuffers[0]) * count)(*framebuffers[:count])
                pipe.lib.framebuffers_read_device_rgb_begin(handles, count, C.c_void_p(target.data_ptr()), target[0].numel())
            else:
                pipe.lib.graphics_flush()
            self.pending = (pipe, slot)
        else:
            # (frames_to_tensor returns when the copies are complete)
            frames_to_tensor(pipe, framebuffers, count, target)
            self._start(slot)
        self.count += 1

    def drain(self):
        if self.active:
            self._start_pending()
            for slot in range(self.depth):
                self._finish(slot)

    def received(self, slot):
        return self.inbox[slot]

    def own(self, slot):
        """Rank 0's own frames of that step under the gather-aware split."""
        return self.kept[slot]


def assemble(gathered, total, world):
    """Rank-major gathered blocks (each padded to the largest block) -> frames in view order."""
    frames = []
    for r, block in enumerate(gathered):
        first, last = shard_views(total, r, world)
        frames.extend(block[i] for i in range(last - first))
    return frames


def assemble_rooted(root_frames, gathered, total, world, peer_frames):
    """Frames in view order for shard_views_rooted: rank 0's own frames, then the peers' gathered blocks."""
    frames = list(root_frames[:total - peer_frames * (world - 1)])
    for r in range(1, world):
        frames.extend(gathered[r][i] for i in range(peer_frames))
    return frames
