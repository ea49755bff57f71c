"""Edge cases of every handle through the C ABI: empty calls in the middle of a stream (state
untouched), single-sample calls, NULL / bad arguments answered by a status code and a message
(never a crash), wrong element types refused by the binding."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


def relerr(y, ref):
    return float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1
    return t


def test_empty_calls_leave_the_stream_untouched(tg, orc):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    e = np.zeros(0, np.complex64)
    h = orc.design_rif_fen(127, "lp", 0.05)
    z, p, mn, md = orc.design_butter_lp(6, 0.2)
    co, gain, r1 = orc.SosChain(z, p, mn, md).coefs()
    cases = [
        (lambda: tg.Fir(h, tg.C64), lambda: orc.Fir(h)),
        (lambda: tg.Sos(co, gain, tg.C64, r1), lambda: orc.SosChain(z, p, mn, md)),
        (lambda: tg.Resampler(np.float32(1.3), tg.C64), lambda: orc.Resampler(np.float32(1.3))),
        (lambda: tg.PolyFir(tg.POLY_DECIM, tg.C64, orc.design_rif_fen(15, "lp", 0.2), 3), lambda: orc.PolyDecim(orc.design_rif_fen(15, "lp", 0.2), 3)),
        (lambda: tg.PolyFir(tg.POLY_UPS, tg.C64, orc.design_rif_fen(15, "lp", 0.2), 2), lambda: orc.PolyUps(orc.design_rif_fen(15, "lp", 0.2), 2)),
    ]
    for mk, mkref in cases:
        g, ref = mk(), mkref()
        got, exp = [], []
        for a, b in [(0, 0), (0, 1), (1, 1), (1, 2), (2, 1000), (1000, 1000), (1000, 5000), (5000, 5000)]:
            yg = g.step(x[a:b] if b > a else e)
            got.append(np.asarray(yg))
            if b > a:
                exp.append(ref.step(x[a:b]))
        y, r = np.concatenate(got), np.concatenate(exp)
        assert y.shape == r.shape and relerr(y, r) <= TOL, type(g).__name__


def test_null_and_bad_arguments_return_status(tg):
    L = tg.lib()
    h = C.c_void_p()
    assert L.tsdgpu_fir_create(None, tg.C64, tg.F32, None, 0, 0) != 0 and L.tsdgpu_last_error()
    taps = np.ones(3, np.float32)
    assert L.tsdgpu_fir_create(C.byref(h), 99, tg.F32, taps.ctypes.data, 3, 0) != 0          # bad data type
    assert L.tsdgpu_fir_create(C.byref(h), tg.C64, tg.F32, taps.ctypes.data, 0, 0) != 0       # no taps
    assert L.tsdgpu_fir_step(None, None, None, 10, None) != 0
    assert L.tsdgpu_fft_create(C.byref(h), 0, 1) != 0 and L.tsdgpu_fft_create(C.byref(h), -5, 1) != 0
    assert L.tsdgpu_fft_create(C.byref(h), (1 << 28) + 1, 1) != 0
    assert L.tsdgpu_fft_step(None, None, None, 1, 1, None) != 0
    assert L.tsdgpu_resampler_create(C.byref(h), tg.C64, C.c_float(0.0), taps.ctypes.data, 3, 1) != 0
    assert L.tsdgpu_resampler_create(C.byref(h), tg.C64, C.c_float(float("nan")), taps.ctypes.data, 3, 1) != 0
    assert L.tsdgpu_resampler_create_analytic(C.byref(h), tg.C64, C.c_float(1.5), 7, 1) != 0
    assert L.tsdgpu_resampler_create_analytic(C.byref(h), tg.C64, C.c_float(1.5), 2, 0) != 0   # Lagrange degree 0
    assert L.tsdgpu_sos_step(None, None, None, 1, None) != 0
    assert L.tsdgpu_ola_create(C.byref(h), 100, -1, None) != 0
    assert L.tsdgpu_welch(None, 10, 4, None, None, None, None) != 0
    # destroying NULL handles is allowed everywhere
    for fn in ("tsdgpu_fir_destroy", "tsdgpu_fft_destroy", "tsdgpu_rfft_destroy", "tsdgpu_sos_destroy", "tsdgpu_resampler_destroy",
               "tsdgpu_polyfir_destroy", "tsdgpu_rii_destroy", "tsdgpu_ola_destroy"):
        assert getattr(L, fn)(None) == 0
    # a valid handle with NULL buffers
    f = tg.Fir(taps, tg.F32)
    assert L.tsdgpu_fir_step(f._h, None, None, 5, None) != 0 and b"NULL" in L.tsdgpu_last_error()
    assert L.tsdgpu_fir_step(f._h, None, None, 0, None) == 0                                  # nothing to do
    assert L.tsdgpu_fir_step(f._h, taps.ctypes.data, taps.ctypes.data, -1, None) != 0


def test_binding_refuses_wrong_element_types(tg):
    f = tg.Fir(np.ones(3, np.float32), tg.C64)
    with pytest.raises(AssertionError):
        f.step(np.zeros(10, np.float32))
    with pytest.raises(AssertionError):
        tg.Sos(np.array([[1, 0, 0, 0, 0]], np.float32), 1.0, tg.F32).step(np.zeros(10, np.complex64))
