"""C++ host layer (tsd:: / dsp:: mirror on the C ABI): builds on CPU, fails loudly without a
GPU, and passes its reference-style test-suite on the GPU (tests/cpp/test_host_api.cc)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "build", "test_host_api")


def build(*cibles):
    """the host library and the named targets of tests/cpp (default: the GPU test binaries; the sanitizer build of the design
    test is only made where it is run, in the CPU tier)"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "libtsd_amd", "host")], check=True, capture_output=True)
    cibles = cibles or ("build/test_host_api", "build/perf_host_api", "build/test_graph_capture")
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), *cibles], check=True, capture_output=True)


def test_host_layer_builds_and_refuses_cpu():
    build()
    r = subprocess.run([BIN, "--no-gpu"], capture_output=True, text=True)
    import libtsd_amd
    if libtsd_amd.device_count() == 0:
        assert r.returncode == 0, r.stdout + r.stderr
        assert "OK" in r.stdout


def test_design_layer_under_sanitizers():
    """the GPU-free part of the mirror (array type, windows, FIR / IIR designs incl. the elliptic prototype, polynomial roots,
    interpolator tables) built with -fsanitize=address,undefined: values against closed forms, no memory error, no UB"""
    build("build/test_design_cpu")
    exe = os.path.join(ROOT, "tests", "cpp", "build", "test_design_cpu")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")      # (the mirror keeps its allocation caches for the process lifetime)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "DESIGN LAYER OK" in r.stdout


def test_host_library_exports_factories():
    build()
    out = subprocess.run(["nm", "-DC", os.path.join(ROOT, "libtsd_amd", "lib", "libtsd_host.so")],
                         capture_output=True, text=True).stdout
    for sym in ["tsd::filtrage::filtre_rif<float, float>", "tsd::filtrage::filtre_rif<float, std::complex<float> >",
                "tsd::filtrage::filtre_rif<std::complex<float>, std::complex<float> >",
                "tsd::filtrage::filtre_rif_fft<float>", "tsd::filtrage::filtre_sois<float>",
                "tsd::filtrage::filtre_reechan<std::complex<float> >", "tsd::filtrage::filtre_itrp<float>",
                "tsd::filtrage::design_rif_fen", "tsd::filtrage::design_riia",
                "tsd::fourier::fftplan_defaut", "tsd::fourier::tfrplan_cr", "tsd::fourier::rtfrplan_cr"]:
        assert sym in out, f"{sym} not exported by libtsd_host.so"


@pytest.mark.gpu
def test_host_layer_gpu_suite():
    if not os.path.exists(BIN):
        build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "ALL C++ HOST TESTS OK" in r.stdout


@pytest.mark.gpu
def test_graph_capture_of_streaming_steps():
    """tests/cpp/test_graph_capture.hip: a FIR / SOS step captured once into a hipGraph and replayed per block
    reproduces the stream of ordinary steps bit for bit (tsdgpu_*_set_capturable)."""
    exe = os.path.join(ROOT, "tests", "cpp", "build", "test_graph_capture")
    if not os.path.exists(exe):
        build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GRAPH CAPTURE OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def _build_readme_example(tmp_path):
    exe = str(tmp_path / "readme_call_sites")
    libdir = os.path.join(ROOT, "libtsd_amd", "lib")
    subprocess.run(["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "libtsd_amd", "host", "include"),
                    os.path.join(ROOT, "examples", "readme_call_sites.cc"), "-L" + libdir, "-ltsd_host", "-ltsdgpu",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True)
    return exe


def test_reference_readme_call_sites_compile(tmp_path):
    """examples/readme_call_sites.cc: the filtering statements of libtsd's README (both APIs, `let` / `soit` spellings,
    umbrella headers) compile against the mirror as they stand."""
    build()
    _build_readme_example(tmp_path)


@pytest.mark.gpu
def test_reference_readme_call_sites_run(tmp_path):
    build()
    r = subprocess.run([_build_readme_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "README CALL SITES OK" in r.stdout, r.stdout + r.stderr
