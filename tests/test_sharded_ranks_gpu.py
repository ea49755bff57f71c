"""The multi-RANK path on the HIP operators: two torch.distributed ranks over gloo
share the one GPU of the test box; their concatenated outputs (FIR direct and overlap-save, SOS,
resampler) must equal the single-handle run -- see tests/ranks_worker.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_and_match_the_single_handle():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_worker.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "RANKS_WORKER OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
