"""Time-based regression guards, kept OUT of the parity run: `pytest -m gpu` only reports them (conftest.perf_guard), this
file -- `pytest -m gpu_perf`, on the GPU box -- runs the tests that carry them with the guards enforced
(TSDGPU_PERF_ASSERTS=1).  A busy or down-clocked box can fail here without hiding a parity result."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARDED = ["tests/test_sos_gpu.py::test_sos_unaligned_device_views",
           "tests/test_sos_gpu.py::test_sos_small_and_ragged_blocks_are_not_a_cliff",
           "tests/test_sos_gpu.py::test_sos_long_memory_exact_carry",
           "tests/test_sos_gpu.py::test_sos_long_memory_forme_directe_1",
           "tests/test_sos_gpu.py::test_exponential_smoother_and_dc_blocker_long_memory",
           "tests/test_polyphase_gpu.py::test_filtre_rii_order6_2p26_under_2ms",
           "tests/test_polyphase_gpu.py::test_filtre_rii_complex_order6_2p24_under_1ms",
           "tests/test_polyphase_gpu.py::test_filtre_rii_high_order_is_not_a_cliff",
           "tests/test_resample_gpu.py::test_short_period_ratios_far_into_a_stream",
           "tests/test_fast_paths_gpu.py"]


@pytest.mark.gpu_perf
def test_perf_guards_enforced():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box")
    env = dict(os.environ, TSDGPU_PERF_ASSERTS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu"] + GUARDED, capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
