"""Large HOST vectors through the C ABI's chunked H2D / kernel / D2H pipeline (csrc/common.hip,
pipelined_host_step: 16-MiB chunks on three streams) against the same operator on a resident copy: the
operator state is carried from chunk to chunk exactly as between two step() calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1
    return t


def crand(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def test_fir_host_pipeline_matches_resident(tg, orc):
    import torch
    n = (5 << 20) + 12345                      # 40 MiB of complex samples: three chunks, the last one ragged
    x = crand(n, 1)
    h31, h127 = orc.design_rif_fen(31, "lp", 0.2), orc.design_rif_fen(127, "lp", 0.02)
    for h, method, exact in ((h31, tg.FIR_DIRECT, True), (h127, tg.FIR_OVERLAP_SAVE, False)):
        yh = tg.Fir(h, tg.C64, method).step(x)                         # host numpy in, host numpy out: pipelined
        yd = tg.Fir(h, tg.C64, method).step(torch.from_numpy(x).cuda()).cpu().numpy()
        if exact:
            assert np.array_equal(yh, yd)
        assert np.abs(yh - yd).max() <= 2e-6 * np.abs(yd).max()
    # streaming across calls, in place, and the oracle on a slice across a chunk border (2 Mi samples per chunk)
    f = tg.Fir(h31, tg.C64, tg.FIR_DIRECT)
    y = x.copy()
    f.step(y[: 3 << 20], y[: 3 << 20])
    f.step(y[3 << 20:], y[3 << 20:])
    assert np.array_equal(y, yh if False else tg.Fir(h31, tg.C64, tg.FIR_DIRECT).step(x))
    lo = (2 << 20) - 500
    ref = orc.fir(h31, x[lo - 100: lo + 1000])[100:]
    assert np.abs(y[lo: lo + 1000] - ref).max() <= 1e-5 * np.abs(ref).max()


def test_sos_and_rii_host_pipeline(tg, orc):
    import torch
    from scipy.signal import butter
    n = (6 << 20) + 777                        # 24 MiB of float samples
    x = np.random.default_rng(2).standard_normal(n).astype(np.float32)
    sos = butter(12, 0.5, output="sos")
    co = np.array([[s[0], s[1], s[2], s[4], s[5]] for s in sos], np.float32)
    yh = tg.Sos(co, 1.0, tg.F32).step(x)
    yd = tg.Sos(co, 1.0, tg.F32).step(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(yh - yd).max() <= 1e-6 * np.abs(yd).max()
    de = np.real(np.poly([0.8 * np.exp(0.5j), 0.8 * np.exp(-0.5j), 0.6, -0.3])).astype(np.float32)
    nu = np.array([1.0, 0.4, 0.2, 0.1, 0.05], np.float32)
    f = tg.Rii(nu, de, tg.F32)
    assert f.path == 1
    yh = f.step(x)
    yd = tg.Rii(nu, de, tg.F32).step(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(yh - yd).max() <= 2e-6 * np.abs(yd).max()
    m = 4000
    ref = orc.Rii(nu, de).step(x[:m])
    assert np.abs(yh[:m] - ref).max() <= 1e-5 * np.abs(ref).max()
    # a long-memory smoother (exact carry of the state inside every chunk of the pipeline, the stream state between them)
    g = np.float32(3e-5)
    lis = (np.array([g, 0.0], np.float32), np.array([1.0, -(1.0 - g)], np.float32))
    xo = x + np.float32(2.0)
    yh = tg.Rii(*lis, tg.F32).step(xo)
    yd = tg.Rii(*lis, tg.F32).step(torch.from_numpy(xo).cuda()).cpu().numpy()
    from scipy.signal import lfilter
    ex = lfilter(lis[0].astype(np.float64), lis[1].astype(np.float64), xo.astype(np.float64))
    assert np.abs(yh - ex).max() <= 1e-5 * np.abs(ex).max() and np.abs(yd - ex).max() <= 1e-5 * np.abs(ex).max()


# variable output length: the resampler and the integer-rate stages report each chunk's output count on the host and
# the chunks' outputs are laid one behind the other (pipelined_host_step_var)
def test_resampler_host_pipeline_matches_resident(tg, orc):
    import torch
    n = (5 << 20) + 4321                       # 40 MiB of complex samples: three chunks
    x = crand(n, 3)
    for ratio in (160.0 / 147, 0.7311):
        yh = tg.Resampler(ratio, tg.C64).step(x)
        yd = tg.Resampler(ratio, tg.C64).step(torch.from_numpy(x).cuda()).cpu().numpy()
        assert yh.shape == yd.shape
        assert np.array_equal(yh, yd)          # the schedule and the per-output arithmetic do not depend on the cut
        # two calls (the second one short: plain staging), then the oracle across the first chunk border
        r = tg.Resampler(ratio, tg.C64)
        y2 = np.concatenate([r.step(x[: 4 << 20]), r.step(x[4 << 20:])])
        assert np.array_equal(y2, yd)
    # the oracle on the head of the stream (the resident run above was compared whole)
    ref = orc.Resampler(160.0 / 147).step(x[: 40000])
    yy = tg.Resampler(160.0 / 147, tg.C64).step(x)
    assert np.abs(yy[: ref.shape[0]] - ref).max() <= 1e-5 * np.abs(ref).max()
    # capacity is checked before anything is written
    r = tg.Resampler(160.0 / 147, tg.C64)
    small = np.empty(n, np.complex64)
    with pytest.raises(tg.TsdGpuError):
        r.step(x, small)
    assert r.out_offset == 0


@pytest.mark.parametrize("kind,R", [("decim", 3), ("decim", 8), ("halfband", 2), ("ups", 3), ("pick", 5)])
def test_polyfir_host_pipeline_matches_resident(tg, orc, kind, R):
    import torch
    n = (3 << 20) + 1001                       # 24 MiB of complex samples
    x = crand(n, 4)
    h = orc.design_rif_fen(31 if kind != "halfband" else 31, "lp", 0.4 / R).astype(np.float32)
    code = {"decim": tg.POLY_DECIM, "halfband": tg.POLY_HALFBAND, "ups": tg.POLY_UPS, "pick": tg.POLY_PICK}[kind]
    mk = lambda: tg.PolyFir(code, tg.C64, None if kind == "pick" else h, R)
    yh = mk().step(x)
    yd = mk().step(torch.from_numpy(x).cuda()).cpu().numpy()
    assert yh.shape == yd.shape
    assert np.array_equal(yh, yd)
    f = mk()
    y2 = np.concatenate([f.step(x[: (2 << 20) + 1]), f.step(x[(2 << 20) + 1:])])
    assert np.array_equal(y2, yd)


# page-locked host memory (what the C++ mirror's Vecteur holds from 1 MiB): asynchronous copies, one enqueueing thread;
# numpy arrays above are pageable and take the two-thread flavour
def test_pinned_host_memory_pipeline(tg, orc):
    import torch
    n = (5 << 20) + 99
    x = crand(n, 5)
    xp = torch.from_numpy(x).pin_memory()
    h = orc.design_rif_fen(63, "lp", 0.1)
    yp = torch.empty(n, dtype=torch.complex64).pin_memory()
    tg.Fir(h, tg.C64, tg.FIR_DIRECT).step(xp, yp)
    yd = tg.Fir(h, tg.C64, tg.FIR_DIRECT).step(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(yp.numpy(), yd)
    r = tg.Resampler(160.0 / 147, tg.C64)
    yo = torch.empty(r.out_count(n), dtype=torch.complex64).pin_memory()
    got = r.step(xp, yo)
    yd = tg.Resampler(160.0 / 147, tg.C64).step(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.shape[0] == yd.shape[0]
    assert np.array_equal(got.numpy(), yd)


def test_fft_rfft_and_ola_host_pipeline(tg):
    """the FFT, RFFT and OLA entry points chunk-pipeline large host vectors too (whole transforms /
    whole blocks per chunk); same results as the operator on a resident copy."""
    import torch
    # batched FFT: 3000 transforms of 4096 points (94 MiB), in place on the host vector; one transform per chunk at 2^22
    for n, batch in ((4096, 3000), (1 << 22, 3), (1000, 5000)):
        x = crand(n * batch, n).reshape(batch, n)
        p = tg.Fft(n, batch)
        yd = p.step(torch.from_numpy(x).cuda()).cpu().numpy()
        yh = p.step(x)
        assert np.abs(yh - yd).max() <= 1e-6 * np.abs(yd).max(), n
        z = x.copy()
        p.step(z, True, z)
        assert np.array_equal(z, yh), n
        p.close()
    xr = np.random.default_rng(3).standard_normal((6000, 2048)).astype(np.float32)
    pr = tg.Rfft(2048)
    yd = pr.step(torch.from_numpy(xr).cuda()).cpu().numpy()
    yh = pr.step(xr)
    assert yh.shape == yd.shape and np.abs(yh - yd).max() <= 1e-6 * np.abs(yd).max()
    # OLA engine (default geometry, a response): two calls, the first leaving samples waiting; windowed mode too
    n = (4 << 20) + 333
    x = crand(n, 9)
    for window in (None, np.hanning(512).astype(np.float32)):
        a, b = tg.Ola(512, 0, window), tg.Ola(512, 0, window)
        H = (np.exp(-np.arange(a.N) / 300.0) * np.exp(0.3j * np.arange(a.N))).astype(np.complex64)
        a.set_response(H)
        b.set_response(H)
        cut = (3 << 20) + 77
        yh = np.concatenate([a.step(x[:cut]), a.step(x[cut:])])
        xd = torch.from_numpy(x).cuda()
        yd = torch.cat([b.step(xd[:cut]), b.step(xd[cut:])]).cpu().numpy()
        assert yh.shape == yd.shape and np.abs(yh - yd).max() <= 2e-6 * np.abs(yd).max(), window is None
