"""Two ranks on the one GPU of the test box, real HIP engines, gloo process group (see tests/_dist_gpu_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_share_one_gpu_component_sharding_lockstep_and_gathers():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "_dist_gpu_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "RANK 0 OK" in res.stdout and "RANK 1 OK" in res.stdout
