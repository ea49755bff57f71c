"""The fast paths that a handle is SUPPOSED to take, guarded by what they are worth: a silent fall-back to the general kernel is
a parity-green regression that only a clock sees.  Time guards (conftest.perf_guard): reported under `pytest -m gpu`, enforced by
tests/test_perf_guards.py (`-m gpu_perf`)."""
import numpy as np
import pytest

from conftest import perf_guard

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1
    return t


def _ms(fn, reps=5, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def test_windowed_ola_512_takes_the_in_wave_kernel(tg):
    """Ne = N = 512 with a window: olaw512_kernel (0.10 ms per 2^24 samples; the statement-by-statement run kernel 0.32)."""
    import torch
    from oracle import ola_oracle
    n = 1 << 24
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
    g = tg.Ola(512, 0, ola_oracle.fen_hann_periodique(512))
    H = (np.random.default_rng(0).standard_normal(512) + 1j * np.random.default_rng(1).standard_normal(512)).astype(np.complex64)
    g.set_response(H)
    y = torch.empty(n, dtype=torch.complex64, device="cuda")
    g.step(x, y)                                  # (the very first block gives no output: later calls are whole)
    ms = _ms(lambda: g.step(x, y))
    perf_guard(ms < 0.2, f"windowed OLA, Ne = N = 512: {ms:.3f} ms per 2^24 samples")


def test_long_interpolator_takes_the_register_window_kernel(tg):
    """127-tap sinc at 160/147: resample_long_kernel (0.43 ms per 2^24 inputs; one lane per output 1.15)."""
    import torch
    n = 1 << 24
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
    r = tg.Resampler(np.float32(160.0) / np.float32(147.0), tg.C64, K=127, fcut=0.4)
    y = torch.empty(r.out_count(n) + 4, dtype=x.dtype, device="cuda")

    def step():
        r.seek(0)
        r.step(x, y)
    ms = _ms(step)
    perf_guard(ms < 0.8, f"127-tap interpolator: {ms:.3f} ms per 2^24 inputs")


def test_mixed_radix_2000_is_one_kernel(tg):
    """n = 2000 = 125 x 16: both passes in fft_bluestein_kernel (0.18 ms per 2^24 points; two kernels 0.33)."""
    import torch
    n, batch = 2000, (1 << 24) // 2000
    x = torch.view_as_complex(torch.randn(batch * n, 2, device="cuda")).view(batch, n)
    p = tg.Fft(n, batch)
    y = torch.empty_like(x)
    ms = _ms(lambda: p.step(x, True, y))
    perf_guard(ms < 0.27, f"fft n = 2000: {ms:.3f} ms per 2^24 points")
    # (parity of the fused plan: tests/test_fft_gpu.py::test_fft_mixed_radix)
