"""Multi-GPU codec (SURVEY 8e "codec on 8 GPUs"): world_size 2 on one MI355X (both ranks on cuda:0, gloo
rendezvous on 127.0.0.1) — the sharded encode writes byte-identical files, the sharded decode returns bit-identical
parameters.  The worker is tests/_dist_codec_worker.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("N,version", [(7, 1), (40, 1), (2500, 1), (12000, 1), (40, 2), (12000, 2)])   # 7, 40: fewer streams than ranks
def test_sharded_codec_equals_single_process(tmp_path, N, version):
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_codec_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CGS_CONTAINER_VERSION=str(version))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), worker, str(tmp_path), str(N)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("sharded codec == single-process codec") == 2
