"""GPU parity of the FIR path (through the C ABI) against the CPU oracle.
Tolerance for float data: max|y - y_ref| <= 1e-5 * max|y_ref|  (BASELINE.json north_star).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5


def relerr(y, ref):
    return float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))


def rand(n, cplx, seed):
    rng = np.random.default_rng(seed)
    if cplx:
        return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    return rng.standard_normal(n).astype(np.float32)


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1, "no GPU visible"
    return t


def methods(t):
    return [t.FIR_DIRECT, t.FIR_OVERLAP_SAVE]


# BASELINE configs[0]: 31-tap low-pass (design_rif_fen) on a 4096-sample real vector
@pytest.mark.parametrize("method", [1, 2])
def test_cfg1_31tap_real(tg, orc, method):
    h = orc.design_rif_fen(31, "lp", 0.25)
    x = rand(4096, False, 1)
    ref = orc.fir(h, x)
    y = tg.Fir(h, tg.F32, method).step(x)
    assert relerr(y, ref) <= TOL


# reference test_filtre_rif (test-filtres.cc:479-511): impulse -> taps
@pytest.mark.parametrize("method", [1, 2])
def test_impulse_response_is_taps(tg, method):
    nc, n = 31, 81
    h = np.linspace(1, nc, nc).astype(np.float32)
    x = np.zeros(n, np.float32)
    x[0] = 1
    y = tg.Fir(h, tg.F32, method).step(x)
    ref = np.concatenate([h, np.zeros(n - nc, np.float32)])
    assert np.abs(y - ref).max() <= (1e-7 if method == 1 else 1e-5 * nc)


@pytest.mark.parametrize("method", [1, 2])
@pytest.mark.parametrize("cplx_data,cplx_taps", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("K", [1, 2, 15, 31, 127, 128, 129, 500, 897, 961, 1200])
def test_fir_parity(tg, orc, method, cplx_data, cplx_taps, K):
    n = 20000
    h = rand(K, cplx_taps, K) / np.float32(np.sqrt(K))
    x = rand(n, cplx_data, K + 1)
    ref = orc.fir(h, x)
    y = tg.Fir(h, tg.C64 if cplx_data else tg.F32, method).step(x)
    assert y.shape == ref.shape
    assert relerr(y, ref) <= TOL


# long filters: the overlap-save plan on 4096..16384-point blocks (514..12289 taps), every block
# size and both ends of its tap range; AUTO must pick it; beyond 12289 taps the partitioned plan (8192-tap
# segments on the same engine) serves
@pytest.mark.parametrize("cplx_data,cplx_taps", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("K", [514, 898, 1025, 2048, 2049, 4097, 4098, 8193, 12289, 12290])
def test_fir_long_filters(tg, orc, cplx_data, cplx_taps, K):
    n = 40000 if K < 5000 else 30000
    h = rand(K, cplx_taps, K) / np.float32(np.sqrt(K))
    x = rand(n, cplx_data, K + 1)
    ref = orc.fir(h, x)
    f = tg.Fir(h, tg.C64 if cplx_data else tg.F32, tg.FIR_AUTO)
    assert f.method == tg.FIR_OVERLAP_SAVE
    # two ragged chunks: the second one starts from the carried history
    y = np.concatenate([f.step(x[:17001].copy()), f.step(x[17001:].copy())])
    assert relerr(y, ref) <= TOL


# filtre_par_bloc (test-filtres.cc:9-31): chunked streaming == one shot, ragged chunk sizes
@pytest.mark.parametrize("method", [1, 2])
@pytest.mark.parametrize("bs", [1000, 311, 80, 4, 1])
def test_fir_streaming_chunks(tg, orc, method, bs):
    h = orc.design_rif_fen(127, "lp", 0.02)
    n = 5000 if bs > 1 else 300
    x = rand(n, True, 3)
    ref = orc.fir(h, x)
    f = tg.Fir(h, tg.C64, method)
    y = np.concatenate([f.step(x[o:o + bs].copy()) for o in range(0, n, bs)])
    assert relerr(y, ref) <= TOL


@pytest.mark.parametrize("method", [1, 2])
def test_fir_empty_and_reset(tg, orc, method):
    h = orc.design_rif_fen(31, "lp", 0.25)
    f = tg.Fir(h, tg.F32, method)
    assert f.step(np.zeros(0, np.float32)).shape == (0,)
    x = rand(1000, False, 9)
    y1 = f.step(x)
    f.reset()
    y2 = f.step(x)
    assert np.array_equal(y1, y2)
    assert relerr(y1, orc.fir(h, x)) <= TOL


# device-resident path (torch tensors), in-place allowed (filtre-rt.cc:76-80), unaligned views
@pytest.mark.parametrize("method", [1, 2])
def test_fir_device_inplace_and_views(tg, orc, method):
    import torch
    h = orc.design_rif_fen(127, "lp", 0.02)
    x = rand(70001, True, 4)
    ref = orc.fir(h, x)
    xd = torch.from_numpy(x).cuda()
    f = tg.Fir(h, tg.C64, method)
    yd = f.step(xd)
    torch.cuda.synchronize()
    assert relerr(yd.cpu().numpy(), ref) <= TOL
    f.reset()
    f.step(xd, xd)                      # in place
    torch.cuda.synchronize()
    assert relerr(xd.cpu().numpy(), ref) <= TOL
    # odd offset view: 8-byte aligned only
    xo = torch.from_numpy(x).cuda()[1:]
    f.reset()
    yo = f.step(xo)
    torch.cuda.synchronize()
    assert relerr(yo.cpu().numpy(), orc.fir(h, x[1:])) <= TOL


# halo hook used by the multi-GPU sharding: chunk b seeded with chunk a's history
@pytest.mark.parametrize("method", [1, 2])
def test_fir_history_halo(tg, orc, method):
    h = orc.design_rif_fen(127, "lp", 0.02)
    x = rand(40000, True, 5)
    ref = orc.fir(h, x)
    fa, fb = tg.Fir(h, tg.C64, method), tg.Fir(h, tg.C64, method)
    ya = fa.step(x[:20000].copy())
    halo = fa.get_history(np.empty(126, np.complex64))
    assert np.array_equal(halo, x[20000 - 126:20000])
    fb.set_history(halo)
    yb = fb.step(x[20000:].copy())
    assert relerr(np.concatenate([ya, yb]), ref) <= TOL


# BASELINE configs[1] at full size: 127 taps on 2^26 complex samples; oracle on a slice,
# size-independent properties on the whole (linearity + block-shift invariance).
@pytest.mark.parametrize("method", [1, 2])
def test_cfg2_full_size_properties(tg, orc, method):
    import torch
    n = 1 << 26
    h = orc.design_rif_fen(127, "lp", 0.02)
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g))
    f = tg.Fir(h, tg.C64, method)
    y = f.step(x)
    torch.cuda.synchronize()
    # (a) oracle on two windows (start, and a far tile boundary region)
    for o in (0, n - 300000):
        lo = max(o - 126, 0)
        seg = x[lo:o + 200000].cpu().numpy()
        ref = orc.fir(h, seg)[o - lo:]
        assert relerr(y[o:o + 200000].cpu().numpy(), ref) <= TOL
    # (b) linearity: F(a*x) == a*F(x)
    f.reset()
    y2 = f.step(x * 2.0)
    torch.cuda.synchronize()
    assert float((y2 - 2.0 * y).abs().max()) <= 1e-5 * float(y.abs().max())
    # (c) time invariance: filtering x delayed by d equals y delayed by d
    d = 12345
    xs = torch.cat([torch.zeros(d, dtype=x.dtype, device="cuda"), x[:n - d]])
    f.reset()
    ys = f.step(xs)
    torch.cuda.synchronize()
    assert float((ys[d:] - y[:n - d]).abs().max()) <= 1e-5 * float(y.abs().max())


# tap counts beyond the long-filter plan (12289) used to create fine and fail at every step
# (the direct kernel's LDS need passes 160 KiB at K > 14312).  They are served by partitioned convolution:
# 8192-tap segments on the long-filter plan, partial sums added in segment order.
@pytest.mark.parametrize("K,cplx_taps", [(12290, False), (14313, False), (14313, True), (65536, False), (100001, True)])
def test_fir_very_long_filters(tg, orc, K, cplx_taps):
    rng = np.random.default_rng(K)
    h = rng.standard_normal(K).astype(np.float32) / np.float32(np.sqrt(K))
    if cplx_taps:
        h = (h + 1j * rng.standard_normal(K).astype(np.float32) / np.float32(np.sqrt(K))).astype(np.complex64)
    n = 150000
    x = rand(n, True, K + 1)
    # reference: exact convolution in double (an FFT), then the 1e-5 band; the oracle's K-tap time loop agrees with it
    full = np.fft.ifft(np.fft.fft(x.astype(np.complex128), 1 << 19) * np.fft.fft(h.astype(np.complex128), 1 << 19))[:n]
    f = tg.Fir(h, tg.C64)
    y = np.concatenate([f.step(x[:50001].copy()), f.step(x[50001:50002].copy()), f.step(x[50002:].copy())])
    assert relerr(y, full.astype(np.complex64)) <= TOL
    m = 3000
    assert relerr(y[:m], orc.fir(h, x[:m])) <= TOL
    # streaming history hooks still work on the partitioned plan
    g = tg.Fir(h, tg.C64)
    g.step(x[:70000].copy())
    hist = np.empty(K - 1, np.complex64)
    g.get_history(hist)
    assert np.array_equal(hist[-min(K - 1, 70000):], x[70000 - min(K - 1, 70000):70000])


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("K", [57, 127, 180, 250, 300, 513])
def test_ols_dynamic_handout_and_runs(tg, orc, monkeypatch, K, cplx):
    """Round 3 schedule of the overlap-save kernel, forced on a call that would keep the static partition (the switches are
    read at every step): blocks pulled from the counters (TSDGPU_OLS_DYN_MIN=0), walked in runs of 2 and 3 with the overlap rows
    reused from registers (overlaps of 1..4 rows have their own kernel, more rows the generic one), real data packed two
    blocks per transform; several steps on one handle (the counters are never reset: every launch starts from the base the
    host tracks), the counter count switched mid-stream, the static partition on the same handle."""
    if not cplx and K == 57:
        K = 63
    h = orc.design_rif_fen(K, "lp", 0.05)
    n = 700001
    x = rand(n, cplx, K)
    ref = orc.fir(h, x)
    f = tg.Fir(h, tg.C64 if cplx else tg.F32, tg.FIR_OVERLAP_SAVE)
    monkeypatch.setenv("TSDGPU_OLS_DYN_MIN", "0")
    cuts = [0, 200000, 200003, 470001, n]
    for run, nc in (("2", "16"), ("3", "8"), ("1", "32"), ("2", "0")):
        monkeypatch.setenv("TSDGPU_OLS_RUN", run)
        monkeypatch.setenv("TSDGPU_OLS_DYN", nc)
        f.reset()
        y = np.concatenate([f.step(x[a:b].copy()) for a, b in zip(cuts[:-1], cuts[1:])])
        assert relerr(y, ref) <= TOL, (run, nc, relerr(y, ref))


@pytest.mark.parametrize("K", [31, 127, 600, 2048])
@pytest.mark.parametrize("cplx", [True, False])
def test_fir_step_after_reads_its_delay_line_from_the_chunk(tg, orc, K, cplx):
    """tsdgpu_fir_step_after (the interior of a sharded chunk without the history copy) against set_history(x[lead - (K-1) : lead]) +
    step(x[lead:]) on every plan: bit for bit on the direct kernel; on the overlap-save plans the blocks also hold the row-alignment
    samples in front of the K - 1 that matter (the chunk's own here, the handle's older ones there), which moves the transforms'
    rounding, not the result.  The handle's stream state afterwards included (a second, ordinary step continues); and the oracle."""
    import torch
    n = 200000
    x = rand(n + 5000, cplx, K)
    h = np.hanning(K).astype(np.float32)
    h /= h.sum()
    dt = tg.C64 if cplx else tg.F32
    xd = torch.from_numpy(x).cuda()
    a, b = tg.Fir(h, dt), tg.Fir(h, dt)
    lead = a.lead
    assert lead >= K - 1
    for extra in (0, 64, 333):                                   # any lead >= the handle's
        ld = lead + extra
        ya = torch.zeros(n, dtype=xd.dtype, device="cuda")
        yb = torch.zeros(n, dtype=xd.dtype, device="cuda")
        a.step_after(xd[:n], ya, ld)
        b.set_history(xd[ld - (K - 1):ld].clone())
        b.step(xd[ld:n], yb[ld:])
        torch.cuda.synchronize()
        exact = a.method == tg.FIR_DIRECT
        close = lambda u, v: torch.equal(u, v) if exact else float((u - v).abs().max() / v.abs().max()) <= 2e-6
        assert close(ya, yb), (K, cplx, extra)
        assert float(ya[:ld].abs().max()) == 0.0                 # nothing before `lead` is written
        ya2, yb2 = a.step(xd[n:]), b.step(xd[n:])                # the stream goes on from the same state
        assert close(ya2, yb2)
    ref = orc.Fir(h).step(x[:n])
    assert relerr(ya[ld:].cpu().numpy(), ref[ld:]) <= TOL
    with pytest.raises(tg.TsdGpuError):
        a.step_after(xd[:n], ya, lead - 1)
    with pytest.raises(tg.TsdGpuError):
        a.step_after(xd[:n], xd[:n], lead)
