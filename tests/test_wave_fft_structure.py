"""CPU structural test of the in-wave 1024-point FFT (libtsd_amd/csrc/fft1024_wave.hpp): the
header's scalar flavour is plain C++, so g++ emulates the 64 lanes phase by phase and checks
forward() against a double-precision DFT and inverse(forward(x)) == N x
(tests/cpu/test_fft1024_wave.cc).  The packed (VOP3P) flavour is an opt-in build (-DOLS_SCALAR=0) of the
overlap-save kernel; the default build uses the scalar flavour tested here and on the GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_fft_header_on_cpu(tmp_path):
    exe = str(tmp_path / "t_w1024")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpu", "test_fft1024_wave.cc")],
                   check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_per_device_keying_on_cpu(tmp_path):
    """bounce blocks and plan reserves are keyed by device (tests/cpu/test_per_device.cc)."""
    exe = str(tmp_path / "t_per_device")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-o", exe,
                    os.path.join(ROOT, "tests", "cpu", "test_per_device.cc")], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
