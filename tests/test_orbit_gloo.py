"""The N > 1 path of the orbit batch on CPU: sharding arithmetic, and a world_size-2 gloo run
(one process per rank, as on the GPUs) whose gathered frames must equal a single-process render."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO
from renderer_amd.orbit import shard_views


def test_shards_partition_the_views():
    for total in [0, 1, 7, 64, 8192]:
        for world in [1, 2, 3, 4, 8]:
            blocks = [shard_views(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_views(8, 2, 2)


def test_rooted_shards_partition_the_views():
    from renderer_amd.orbit import shard_views_rooted
    for world in [1, 2, 4, 8]:
        for total, peer in [(8 * world, 8), (8 * world, 3), (8 * world, 0), (1024, 37)]:
            if peer * (world - 1) > total:
                continue
            blocks = [shard_views_rooted(total, r, world, peer) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            assert all(b - a == peer for a, b in blocks[1:])
    with pytest.raises(ValueError):
        shard_views_rooted(8, 0, 4, 3)


def test_gather_aware_split_arithmetic():
    """split_for with injected times: peers = frames * world / (delivery / render + world - 1), clamped."""
    from renderer_amd.orbit import split_for, shard_views_rooted
    assert split_for(1.0, 1.0, 128, 1) == 128                      # one rank: nothing to split
    assert split_for(1.0, 1.0, 128, 2) == 128                      # a link as fast as the renderer: even split
    assert split_for(1.0, 3.0, 128, 2) == 64                       # 256 / (3 + 1)
    assert split_for(1.0, 7.0, 128, 8) == 73                       # 1024 / (7 + 7) = 73.1
    assert split_for(1.0, 1e9, 128, 4) == 1                        # a dead link still gets one frame per peer
    assert split_for(1.0, 0.01, 128, 4) == 128                     # never more than the even split
    assert split_for(0.0, 1.0, 128, 4) == 128 and split_for(1.0, float("nan"), 128, 4) == 128
    for world in (2, 4, 8):
        for ratio in (0.5, 2.0, 5.5, 40.0):
            p = split_for(1.0, ratio, 128, world)
            root = 128 * world - p * (world - 1)
            first, last = shard_views_rooted(128 * world, 0, world, p)
            assert last - first == root >= p
            # rank 0's rendering and a peer's delivery take the same time to within one frame of either
            if 1 < p < 128:
                assert abs(root * 1.0 - p * ratio) <= max(1.0, ratio) * world


@pytest.mark.parametrize("world,total,scene,peer,channels,mode", [(2, 5, "helmet", None, 4, "-"), (2, 4, "helmet", None, 3, "-"), (2, 6, "buster", None, 4, "-"),
                                                                  (2, 6, "helmet", 2, 4, "-"), (3, 7, "helmet", 1, 3, "-"), (4, 11, "helmet", 2, 3, "-"),
                                                                  (4, 9, "buster", None, 3, "-"), (2, 6, "helmet", 2, 3, "lagged"),
                                                                  (4, 11, "helmet", 2, 3, "lagged"), (3, 7, "helmet", None, 3, "lagged")],
                         ids=["helmet-5", "helmet-4-rgb", "buster-animated-6", "helmet-6-rooted-2", "helmet-7-of-3-rooted-1-rgb",
                              "helmet-11-of-4-rooted-2-rgb", "buster-9-of-4-rgb", "helmet-6-rooted-2-rgb-lagged",
                              "helmet-11-of-4-rooted-2-rgb-lagged", "helmet-7-of-3-rgb-lagged"])
def test_gloo_orbit_gather(oracle, world, total, scene, peer, channels, mode):
    """helmet: the benchmark's scene; buster: the animated scene of the batch-render configuration
    (every view has its own frame time, so every rank evaluates its own poses).  rooted: the
    gather-aware split bench.py uses for N > 1 (peers render `peer` views each, rank 0 the rest and
    keeps its own outside the collective).  lagged: the copy-out in two halves as bench.py uses it with frames in device
    memory (a step's copies are launched by its submit and waited for, and its gather started, by the next)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "orbit_worker.py"), str(total), "96", "54", scene, str(peer) if peer is not None else "-", str(channels), mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), outs
    assert "ORBIT_OK" in outs[0], outs[0]
