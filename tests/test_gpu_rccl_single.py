"""The multi-GPU gather's device path on the one GPU a test box has: a child process builds a
world_size-1 RCCL process group and drives FrameGather with `force` (tests/rccl_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_frame_gather_over_rccl_with_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "tests", "rccl_worker.py")], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "RCCL_OK" in p.stdout and "backend nccl" in p.stdout, p.stdout[-3000:]
