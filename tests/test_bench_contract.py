"""CPU-side guards of the driver contract: bench.py parses its flags without a GPU, and the bench
lines committed under profiles/ carry every key the contract names (metric, value, ..., roofline,
cpu_baseline)."""
import pytest
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_bench_help_lists_the_contract_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in r.stdout


def test_bench_reads_its_traffic_from_the_committed_pmc_file():
    sys.path.insert(0, ROOT)
    import bench
    for w in ("fir", "fft", "sos", "resample"):
        tr, src = bench.pmc_traffic(w)
        assert tr and src and os.path.exists(os.path.join(ROOT, src)), w
        # the kernels of the path move at least their algorithmic bytes and less than 2.5 x that
        alg = {"fir": 16.0 * 2 ** 26, "fft": 16.0 * 2 ** 28, "sos": 8.0 * 2 ** 26, "resample": (8 + 8 * 160 / 147) * 2 ** 27}[w]
        assert 0.98 * alg <= tr <= 2.5 * alg, (w, tr, alg)


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json")))
    assert files, "no bench line committed under profiles/"
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for k in KEYS:
            assert k in d, (f, k)
        assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
        assert "workload" in d["config"] and "model" not in d["config"]
        rf = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in rf, (f, k)
        assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
        if os.path.basename(f) >= "r2":
            # from round 2: both clocks named, traffic read from a committed PMC file (or null) with its source
            for k in ("traffic_source", "timer", "achieved_wall", "frac_wall"):
                assert k in rf, (f, k)
            assert abs(rf["frac_wall"] - rf["frac"]) <= 0.1 * rf["frac"]      # two clocks over two passes of the same steps
            assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
            if rf["traffic_source"]:
                assert os.path.exists(os.path.join(ROOT, rf["traffic_source"]))
        cb = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in cb, (f, k)
        assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1


def test_bench_launches_its_own_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2` the way the driver starts `--gpus 1` (no launcher in the command line, WORLD_SIZE unset):
    bench.py starts the two ranks as CHILD processes before anything touches a GPU, they rendezvous on 127.0.0.1 (gloo here,
    CPU tier), run the barrier and the max-over-ranks reduction, and the parent relays rank 0's one JSON line and the exit
    code.  --rendezvous-only stops there (no GPU in this tier); the same command without it is test_bench_two_ranks_rehearsal's
    self-launched case on the GPU box."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout       # stdout carries the ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["self_launched"] is True and d["value"] is None


def test_bench_self_launch_relays_a_failing_rank():
    """without a GPU the ranks fail loudly (no CPU fallback): the parent's exit code is the launcher's, and no JSON line appears"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_strong_scaling_sizes():
    """--scaling strong splits BASELINE.json's totals over the ranks (configs[4]: 2^30 inputs), weak keeps the per-GPU size"""
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    strong, weak = argparse.Namespace(scaling="strong"), argparse.Namespace(scaling="weak")
    assert bench._per_rank(1 << 30, 8, strong) == 1 << 27 and bench._per_rank(1 << 30, 1, strong) == 1 << 30
    assert bench._per_rank(1 << 26, 8, weak) == 1 << 26 and bench._per_rank(256, 4, strong) == 64
    with pytest.raises(AssertionError):
        bench._per_rank(256, 3, strong)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_self_launched(scaling):
    """`python3 bench.py --gpus 2` with NO launcher in the command line (how the driver starts --gpus 1): bench.py launches
    its two ranks itself; rehearsed on ONE GPU over gloo like the test below.  strong: the 2^26 samples are split in two."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TSDGPU_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling", scaling],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["backend"] == "gloo" and d["config"]["world_size"] == 2 and d["config"]["self_launched"] is True
    assert d["config"]["samples_per_gpu"] == (1 << 25 if scaling == "strong" else 1 << 26)


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["fir", "sos", "resample"])
def test_bench_two_ranks_rehearsal(workload):
    """The driver's multi-GPU command line with 2 ranks, rehearsed on ONE GPU: the ranks share the
    device and talk over gloo (TSDGPU_BENCH_BACKEND) instead of RCCL -- the values mean nothing,
    but the sharded path (seek / halo exchange / max over ranks / one JSON line) runs end to end."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSDGPU_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    # "did the collective library see N ranks" is answerable from the record
    assert d["config"]["backend"] == "gloo" and d["config"]["world_size"] == 2
