"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/tsdgpu.h declares, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tsdgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tsdgpu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import libtsd_amd
    L = libtsd_amd.lib()
    names = declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/tsdgpu.h but not exported: {missing}"


def test_no_cpu_fallback():
    import libtsd_amd as t
    if t.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(t.TsdGpuError):
        t.Fir([1.0, 2.0, 3.0], t.F32)


def test_product_never_touches_oracle():
    """The product path must not import, link or call anything under oracle/."""
    bad = []
    for base in ("libtsd_amd",):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"(import|from)\s+oracle|pyoracle|liborc|tsd_oracle\.h|orc_[a-z]+\(", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
    out = os.popen(f"ldd {os.path.join(ROOT, 'libtsd_amd', 'lib', 'libtsdgpu.so')}").read()
    assert "liborc" not in out


def _build_c_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "fir_from_c")
    libdir = os.path.join(ROOT, "libtsd_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "fir_from_c.c"), "-L" + libdir, "-ltsdgpu", "-lm",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True)
    return exe


def test_plain_c_example_builds_and_refuses_cpu(tmp_path):
    """examples/fir_from_c.c: the ABI is usable from C99 with nothing but the header and the .so;
    without a GPU the program reports it and exits with status 2 (no CPU fallback)."""
    import subprocess
    import libtsd_amd
    exe = _build_c_example(tmp_path)
    if libtsd_amd.device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no GPU" in r.stderr


@pytest.mark.gpu
def test_plain_c_example_runs(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C ABI example OK" in r.stdout, r.stdout + r.stderr


def test_binding_refuses_wrong_dtypes_and_strided_views():
    """the C ABI reads packed float32 / complex64; the binding must refuse anything else instead of
    reinterpreting it (float64, integers, complex128, strided views)."""
    import numpy as np
    from libtsd_amd import capi
    assert capi._dtype_code(np.zeros(4, np.float32)) == capi.F32
    assert capi._dtype_code(np.zeros(4, np.complex64)) == capi.C64
    for bad in (np.zeros(4, np.float64), np.zeros(4, np.complex128), np.zeros(4, np.int32), np.zeros(4, np.float16)):
        with pytest.raises(capi.TsdGpuError):
            capi._dtype_code(bad)
        with pytest.raises(capi.TsdGpuError):
            capi._ptr(bad)
    with pytest.raises(capi.TsdGpuError):
        capi._ptr(np.zeros(8, np.float32)[::2])
    import torch
    with pytest.raises(capi.TsdGpuError):
        capi._ptr(torch.zeros(4, dtype=torch.float64))
    with pytest.raises(capi.TsdGpuError):
        capi._ptr(torch.zeros(8)[::2])
    assert capi._ptr(torch.zeros(8)) != 0
