"""First contact of the multi-rank path with RCCL: torch.distributed backend "nccl" at world size 1
on the one GPU of the test box -- every collective / point-to-point call of libtsd_amd.sharding and bench.py on device
tensors (tests/rccl_worker.py), and bench.py itself through init_process_group("nccl") (TSDGPU_BENCH_FORCE_DIST=1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


@pytest.mark.gpu
def test_rccl_world1_collectives_and_overlapped_steps():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(_free_port())],
                       capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_WORKER OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["fir", "sos", "resample"])
def test_bench_through_rccl_with_one_rank(workload):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), TSDGPU_BENCH_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--workload", workload, "--no-cpu", "--log2n", "22"],
                       capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["backend"].startswith("nccl") and d["config"]["world_size"] == 1 and d["n_gpus"] == 1 and d["value"] > 0
