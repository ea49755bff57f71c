"""GPU parity of the fractional resampler (C ABI tsdgpu_resampler_*) against the CPU oracle
(restatement of AdaptationRythmeSimple / InterpolateurRIF / itrp_sinc, ra.cc:13-79).
Index results (output count, input index and LUT column of every output) must be BIT-EXACT;
sample values within 1e-5 relative."""
import numpy as np
import pytest

from conftest import perf_guard

pytestmark = pytest.mark.gpu
TOL = 1e-5
R160 = np.float32(160.0) / np.float32(147.0)


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
    assert t.device_count() >= 1
    return t


RATIOS = [R160, 1.5, 0.5, 1.2, 1.999, 0.7, float(np.pi / 2), 1.0000001, 147.0 / 160.0]


@pytest.mark.parametrize("ratio", RATIOS)
@pytest.mark.parametrize("cplx", [True, False])
def test_resample_values(tg, orc, ratio, cplx):
    n = 50000
    x = rand(n, cplx, 7)
    ref = orc.Resampler(ratio)
    yref = ref.step(x)
    g = tg.Resampler(ratio, tg.C64 if cplx else tg.F32, lut=ref.lut)     # identical LUT data on both sides
    y = g.step(x)
    assert len(y) == len(yref)                                           # exact output count
    assert relerr(y, yref) <= TOL


# other interpolator lengths through the generic kernel: the 4-tap cspline shape, K = 31/32 (LUT in
# LDS), the 127-tap sinc of the reference's test_ra_unit (test-ra.cc:154-156: LUT read from global
# memory) and the 256-tap limit; chunked so that the window history is exercised
@pytest.mark.parametrize("K,nph", [(4, 256), (2, 64), (31, 256), (32, 511), (127, 256), (256, 128), (15, 512), (15, 1024), (63, 8191)])
@pytest.mark.parametrize("ratio", [R160, 0.5, 1.999])
@pytest.mark.parametrize("cplx", [True, False])
def test_resample_other_lengths(tg, orc, K, nph, ratio, cplx):
    n = 30000
    x = rand(n, cplx, K)
    ref = orc.Resampler(ratio, K=K, nphases=nph, fcut=0.4)
    yref = ref.step(x)
    g = tg.Resampler(ratio, tg.C64 if cplx else tg.F32, K=K, nphases=nph, lut=ref.lut)
    y = np.concatenate([g.step(x[:12345].copy()), g.step(x[12345:].copy())])
    assert len(y) == len(yref)
    assert relerr(y, yref) <= TOL


# Bit-exact schedule: with a LUT that is 1 on the newest tap the output IS the input sample
# selected (x = ramp -> input index); with LUT[col][newest] = col and x = 1 the output IS the column.
@pytest.mark.parametrize("ratio", RATIOS)
def test_resample_schedule_bit_exact(tg, orc, ratio):
    n, K, nph = 70000, 15, 256
    ref = orc.Resampler(ratio)
    nout, idx, col = ref.schedule(n)
    lut_idx = np.zeros((nph + 1, K), np.float32)
    lut_idx[:, K - 1] = 1
    y = tg.Resampler(ratio, tg.F32, lut=lut_idx).step(np.arange(n, dtype=np.float32))
    assert len(y) == nout
    assert np.array_equal(y.astype(np.int64), idx)
    lut_col = np.zeros((nph + 1, K), np.float32)
    lut_col[:, K - 1] = np.arange(nph + 1)
    y = tg.Resampler(ratio, tg.F32, lut=lut_col).step(np.ones(n, np.float32))
    assert np.array_equal(y.astype(np.int32), col)


# streaming: ragged chunks carry phase, window history and output offset
@pytest.mark.parametrize("bs", [20000, 4096, 1000, 311, 1])
def test_resample_streaming(tg, orc, bs):
    n = 60000 if bs > 1 else 300
    x = rand(n, True, 8)
    ref = orc.Resampler(R160)
    yref = ref.step(x)
    g = tg.Resampler(R160, tg.C64, lut=ref.lut)
    y = np.concatenate([g.step(x[o:o + bs].copy()) for o in range(0, n, bs)])
    assert len(y) == len(yref) and relerr(y, yref) <= TOL
    g.reset()
    assert np.array_equal(g.step(x), y)


# the multi-GPU sharding hook: a shard seeks to its absolute position with its 14-sample halo
def test_resample_seek_shards(tg, orc):
    n, parts = 400000, 4
    x = rand(n, True, 9)
    ref = orc.Resampler(R160)
    yref = ref.step(x)
    out = np.zeros_like(yref)
    for p in range(parts):
        lo, hi = p * n // parts, (p + 1) * n // parts
        g = tg.Resampler(R160, tg.C64, lut=ref.lut)
        halo = np.zeros(14, np.complex64)
        if lo:
            halo[:] = x[lo - 14:lo]
        g.seek(lo, halo if lo else None)
        off = g.out_offset
        yp = g.step(x[lo:hi].copy())
        out[off:off + len(yp)] = yp
    assert relerr(out, yref) <= TOL


@pytest.mark.parametrize("K,ratio", [(127, R160), (31, 0.77), (63, 1.999)])
def test_resample_seek_shards_long_interpolators(tg, orc, K, ratio):
    """The same hook through resample_long_kernel: shards that seek to odd absolute positions with their K - 1 samples of halo
    reproduce the single stream bit for bit (the same multiply-adds whichever tile a sample falls in)."""
    n, cuts = 300000, [0, 77777, 150001, 262144, 300000]
    x = rand(n, True, K)
    ref = orc.Resampler(ratio, K=K, nphases=256, fcut=0.4)
    one = tg.Resampler(ratio, tg.C64, K=K, nphases=256, lut=ref.lut).step(x.copy())
    assert relerr(one, ref.step(x)) <= TOL
    out = np.zeros_like(one)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        g = tg.Resampler(ratio, tg.C64, K=K, nphases=256, lut=ref.lut)
        g.seek(lo, x[lo - (K - 1):lo].copy() if lo else None)
        off = g.out_offset
        yp = g.step(x[lo:hi].copy())
        out[off:off + len(yp)] = yp
    assert np.array_equal(out, one)


# known-answer counts of the float32 recurrence -- host schedule only
def test_resample_counts_known_answers(tg):
    g = tg.Resampler(R160, tg.C64)
    assert g.out_count(1 << 20) == 1141308
    assert g.out_count(1 << 26) == 73043660
    assert g.out_count(1 << 30) == 1168698548


def test_resample_device_large(tg, orc):
    import torch
    n = 1 << 24
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    xd = torch.view_as_complex(torch.randn(n, 2, device="cuda:0", generator=gen))
    ref = orc.Resampler(R160)
    g = tg.Resampler(R160, tg.C64, lut=ref.lut)
    yd = g.step(xd)
    torch.cuda.synchronize()
    yref = ref.step(xd.cpu().numpy())
    assert yd.shape[0] == len(yref)
    assert relerr(yd.cpu().numpy(), yref) <= TOL


def test_resample_bad_arguments(tg):
    with pytest.raises(tg.TsdGpuError):
        tg.Resampler(1.1, tg.F32, K=257, nphases=16, lut=np.zeros((17, 257), np.float32))
    with pytest.raises(tg.TsdGpuError):
        tg.Resampler(0.0, tg.F32)
    with pytest.raises(tg.TsdGpuError):
        tg.Resampler(1.5, tg.F32, K=15, nphases=8192, lut=np.zeros((8193, 15), np.float32))   # nphases <= 8191


# interpolators whose taps are a function of the float phase itself: itrp_lineaire and
# itrp_lagrange(d), the degrees of the reference's interpolator list (test-itrp.cc:63-72)
@pytest.mark.parametrize("analytic", [("lin", 0), ("lagrange", 1), ("lagrange", 2), ("lagrange", 3), ("lagrange", 5),
                                      ("lagrange", 6), ("lagrange", 7)])
@pytest.mark.parametrize("ratio", [1.5, 0.77, 160.0 / 147.0, 1.0, 3.1])
@pytest.mark.parametrize("cplx", [False, True])
def test_analytic_interpolators(tg, orc, analytic, ratio, cplx):
    x = rand(60000, cplx, 17)
    ref = orc.Resampler(ratio, analytic=analytic)
    g = tg.Resampler(ratio, tg.C64 if cplx else tg.F32, analytic=analytic)
    # ragged chunks: the window and the phase carry over
    got, exp = [], []
    for a, b in [(0, 1), (1, 4097), (4097, 4100), (4100, 60000)]:
        got.append(g.step(x[a:b]))
        exp.append(ref.step(x[a:b]))
    y, r = np.concatenate(got), np.concatenate(exp)
    assert y.shape == r.shape
    assert relerr(y, r) <= TOL


def test_linear_interpolator_on_a_ramp(tg):
    """Linear interpolation reproduces a ramp exactly (up to rounding): y_j = x(t_j - 1), t_j = j / ratio."""
    ratio = np.float32(1.25)
    x = np.arange(5000, dtype=np.float32)
    y = tg.Resampler(ratio, tg.F32, analytic=("lin", 0)).step(x)
    t = np.arange(len(y), dtype=np.float64) / float(ratio)
    assert np.max(np.abs(y[2:] - (t[2:] - 1.0))) < 2e-3


def test_shared_schedule_is_thread_safe(tg, orc):
    """Handles of one ratio share the host phase schedule (simulated once per process): several threads
    creating and stepping their own handles at the same time -- one of them far enough to trigger the
    cycle detection -- all get the single-threaded result."""
    import threading
    ratio = np.float32(0.8371)                      # a ratio no other test has used: the schedule starts empty
    x = rand(3_000_000, True, 23)
    ref = orc.Resampler(ratio).step(x[:400_000])
    out = [None] * 4

    def work(i):
        g = tg.Resampler(ratio, tg.C64)
        n = [400_000, 3_000_000, 400_000, 1_000_000][i]
        parts = [g.step(x[a:min(a + 250_000, n)]) for a in range(0, n, 250_000)]
        out[i] = np.concatenate(parts)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for q in th:
        q.start()
    for q in th:
        q.join()
    for i in range(4):
        assert out[i] is not None and np.array_equal(out[i][:len(ref)], out[0][:len(ref)]), i
    assert relerr(out[0], ref) <= TOL
    assert np.array_equal(out[1][:len(out[3])], out[3])


# a configuration no step could launch (LDS need of the generic kernel) is refused at creation
def test_resampler_create_refuses_what_no_step_can_launch(tg):
    lut = tg.itrp_sinc_lut(15, 256, 0.4)
    with pytest.raises(tg.TsdGpuError, match="LDS"):
        tg.Resampler(8.0, tg.C64, lut=lut)
    r = tg.Resampler(5.0, tg.F32, lut=lut)          # a large ratio that does fit keeps working
    x = np.ones(1000, np.float32)
    assert abs(len(r.step(x)) - 5000) <= 2


@pytest.mark.parametrize("ratio", [1.0, 0.5, 1.25, 1.5, 2.5])
def test_short_period_ratios_far_into_a_stream(tg, orc, ratio):
    """Ratios whose float32 phase recurrence has a SHORT period (1.0: one input, 0.5: two, 1.25: five ...) put millions of
    periods between the schedule table and a position far into a stream; the kernels used to fold the index back by
    repeated subtraction -- 178 ms per 4 M samples at ratio 1 (found by scripts/perf_resample_ratios.py).  Values against
    the oracle past 2^22 inputs, and a loose bound on the time."""
    import time
    import torch
    n1, n2 = (1 << 22) + 12345, 200000
    x = rand(n1 + n2, True, 77)
    ref = orc.Resampler(ratio, K=15, nphases=256, fcut=float(min(0.4, ratio / 2)))
    yref = ref.step(x)
    g = tg.Resampler(ratio, tg.C64, K=15, nphases=256, lut=ref.lut)
    xd = torch.from_numpy(x).cuda()
    y1 = g.step(xd[:n1])
    y2 = g.step(xd[n1:])
    torch.cuda.synchronize()
    assert y1.shape[0] + y2.shape[0] == yref.shape[0]
    tail = yref[y1.shape[0]:]
    assert relerr(y2.cpu().numpy(), tail) <= TOL
    t0 = time.perf_counter()
    for _ in range(5):
        g.step(xd[:n1])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    perf_guard(ms < 5.0, f"ratio {ratio}: {ms:.1f} ms per 4 M inputs far into the stream")


@pytest.mark.parametrize("cplx", [False, True])
def test_resampler_dynamic_tiles_forced(tg, orc, monkeypatch, cplx):
    """The fused K = 15 kernel with its tiles pulled from the counters on calls that would keep the static partition
    (TSDGPU_RS_DYN_MIN=0; the switches are read at every step): bit for bit the static partition's outputs, over several
    ragged steps of one handle (counters never reset) and with the counter count switched mid-stream."""
    ratio = np.float32(160.0) / np.float32(147.0)
    x = rand(300007, cplx, 21)
    dt = tg.C64 if cplx else tg.F32
    cuts = [0, 5, 100000, 100001, 222222, len(x)]
    monkeypatch.setenv("TSDGPU_RS_DYN", "0")
    ref = tg.Resampler(ratio, dt)
    y0 = np.concatenate([ref.step(x[a:b].copy()) for a, b in zip(cuts[:-1], cuts[1:])])
    monkeypatch.setenv("TSDGPU_RS_DYN_MIN", "0")
    g = tg.Resampler(ratio, dt)
    out = []
    for (a, b), nc in zip(zip(cuts[:-1], cuts[1:]), ("16", "16", "32", "8", "16")):
        monkeypatch.setenv("TSDGPU_RS_DYN", nc)
        out.append(g.step(x[a:b].copy()))
    assert np.array_equal(np.concatenate(out), y0)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("K,nph,ratio", [(127, 256, R160), (127, 256, 0.5), (127, 256, 1.999), (100, 64, 1.2), (31, 256, R160),
                                         (256, 128, 0.77), (24, 300, 1.0), (127, 256, 2.0), (64, 1024, 0.50001), (255, 128, 1.37),
                                         (40, 2047, 1.1), (25, 8191, 0.9)])
def test_long_interpolator_kernel_bit_for_bit(tg, orc, monkeypatch, K, nph, ratio, cplx):
    """resample_long_kernel (a lane's 8 inputs against a register window, second outputs of an input from a list, the table
    staged in tap slices): the same multiply-adds in the same order as resample_kernel, so the outputs must be IDENTICAL -- with
    one tap phase, with the default geometry and with three and five phases (partial sums stored and reloaded), over ragged
    steps of one handle; and within the tolerance of the oracle."""
    x = rand(70001, cplx, K + nph)
    dt = tg.C64 if cplx else tg.F32
    ref = orc.Resampler(ratio, K=K, nphases=nph, fcut=0.4)
    yref = ref.step(x)
    cuts = [0, 3, 20000, 20001, 51111, len(x)]

    def run():
        g = tg.Resampler(ratio, dt, K=K, nphases=nph, lut=ref.lut)
        return np.concatenate([g.step(x[a:b].copy()) for a, b in zip(cuts[:-1], cuts[1:])])

    monkeypatch.setenv("TSDGPU_RS_LONG", "0")
    y0 = run()
    monkeypatch.delenv("TSDGPU_RS_LONG")
    assert len(y0) == len(yref) and relerr(y0, yref) <= TOL
    for phases in (None, "1", "3", "5"):
        if phases is None:
            monkeypatch.delenv("TSDGPU_RS_LONG_PHASES", raising=False)
        else:
            monkeypatch.setenv("TSDGPU_RS_LONG_PHASES", phases)
        y = run()
        assert np.array_equal(y, y0), (phases, float(np.max(np.abs(y - y0))))
