"""Pins the CPU oracle against the known-answer checks the reference's own tests hold for
the hot path.  The reference has no golden files: every expectation
below is the one its test computes in-line, cited file:line (relative to
/root/reference/core/tests/).  The reference itself is unbuildable here (Eigen3 absent).
"""
import numpy as np
import pytest


def sigimp(n, p=0):
    x = np.zeros(n, np.float32)
    x[p] = 1
    return x


# test-filtres.cc:479-511  test_filtre_rif: impulse -> taps, err <= 1e-7
def test_filtre_rif_impulse(orc):
    nc, n = 31, 81
    h = np.linspace(1, nc, nc).astype(np.float32)
    y = orc.Fir(h).step(sigimp(n))
    assert len(y) == n
    ref = np.concatenate([h, np.zeros(n - nc, np.float32)])
    assert np.abs(y - ref).max() <= 1e-7


# test-filtres.cc:9-31,524-525  filtre_par_bloc: chunked step == one-shot
@pytest.mark.parametrize("bs", [1000, 311, 80, 4])
def test_fir_block_invariance(orc, bs):
    rng = np.random.default_rng(1)
    h = orc.design_rif_fen(127, "lp", 0.02)
    x = rng.standard_normal(5000).astype(np.float32)
    y1 = orc.Fir(h).step(x)
    f = orc.Fir(h)
    y2 = np.concatenate([f.step(x[o:o + bs]) for o in range(0, len(x), bs)])
    assert np.array_equal(y1, y2)


# test-filtres.cc:450-476  test_retard: boxcar d twice -> peak at 3+d-1
@pytest.mark.parametrize("d", [3, 4])
def test_retard(orc, d):
    h = np.ones(d, np.float32)
    y = orc.fir(h, sigimp(20, 3))
    y2 = orc.fir(h, y)
    assert int(np.argmax(y2)) == 3 + d - 1


# test-filtres.cc:33-71  test_design_rif_prod: cascade == product filter, err < 1e-5
@pytest.mark.parametrize("n1", [10, 11, 15, 20])
@pytest.mark.parametrize("n2", [10, 11, 15, 20])
def test_design_rif_prod(orc, n1, n2):
    rng = np.random.default_rng(n1 * 100 + n2)
    h1 = rng.standard_normal(n1).astype(np.float32)
    h2 = rng.standard_normal(n2).astype(np.float32)
    hp = np.convolve(h1.astype(np.float64), h2.astype(np.float64)).astype(np.float32)
    x = sigimp(n1 + n2)
    y2 = orc.fir(h2, orc.fir(h1, x))
    yp = orc.fir(hp, x)
    assert np.abs(y2 - yp).max() < 1e-5


# test-filtres.cc:556-606  test_filtre_rii: one-pole smoother vs closed recurrence <= 1e-6
def test_filtre_rii(orc):
    a = np.float32(0.1)
    # H(z^-1) = a / (1 - (1-a) z^-1)
    f = orc.Rii([a], [1.0, -(1 - a)])
    n = 20
    y = f.step(np.ones(n, np.float32))
    yref = np.empty(n, np.float32)
    yref[0] = a
    for i in range(1, n):
        yref[i] = yref[i - 1] + a * (1 - yref[i - 1])
    assert np.abs(yref - y).max() <= 1e-6


# test-fourier.cc:181-272  test_fft_valide: vs naive float DFT, err < 1e-2 (their bound);
# we also require 2e-5 relative, which the radix-2/Bluestein paths meet comfortably.
@pytest.mark.parametrize("n", [16, 1, 2, 3, 4, 5, 8, 10, 17, 128, 129, 1024])
@pytest.mark.parametrize("inv", [False, True])
def test_fft_valide(orc, n, inv):
    rng = np.random.default_rng(n)
    for x in (np.ones(n, np.complex64),
              (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)):
        X = orc.fft(x, not inv)
        k = np.arange(n)
        W = np.exp((1 if inv else -1) * 2j * np.pi * np.outer(k, k) / n)
        Y = (W @ x.astype(np.complex128)) / np.sqrt(n)
        err = np.abs(X - Y).max()
        assert err < 1e-2
        assert err <= 2e-5 * max(1.0, np.abs(Y).max())


@pytest.mark.parametrize("n", [16, 2, 4, 8, 10, 128, 1024, 17, 129, 5, 3])
def test_rfft_valide(orc, n):
    rng = np.random.default_rng(n + 7)
    x = rng.uniform(-1, 1, n).astype(np.float32)
    X = orc.rfft(x)
    Y = np.fft.fft(x.astype(np.float64)) / np.sqrt(n)
    assert np.abs(X - Y).max() < 1e-2
    assert np.abs(X - Y).max() <= 2e-5 * max(1.0, np.abs(Y).max())


# test-fourier.cc:275-312  test_fft: ifft(fft(x)) rms error <= 5e-6 at n = 1024
def test_fft_round_trip(orc):
    n = 1024
    x = np.cos(np.linspace(0, 8 * 2 * np.pi, n)).astype(np.float32)
    X = orc.fft(x.astype(np.complex64))
    x2 = orc.ifft(X).real
    assert np.sqrt(np.mean((x2 - x) ** 2)) <= 5e-6


# test-fourier.cc:6-37  test_fftplan: plan sizes incl. even-non-pow2 and odd
@pytest.mark.parametrize("n", [8, 16, 18, 19, 101])
def test_fftplan_sizes(orc, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    # odd n goes through Bluestein, whose chirp angle is formed in float32 up to ~n*pi rad
    # (fourier.cc:396-399): the reference's own accuracy there is ~1e-5..1e-4, and the
    # oracle must reproduce that, not improve on it.
    tol = 1e-5 if n % 2 == 0 else 2e-4
    Y = np.fft.fft(x.astype(np.complex128)) / np.sqrt(n)
    assert np.abs(orc.fft(x) - Y).max() <= tol * np.abs(Y).max() + 1e-6
    Yi = np.fft.ifft(x.astype(np.complex128)) * np.sqrt(n)
    assert np.abs(orc.ifft(x) - Yi).max() <= tol * np.abs(Yi).max() + 1e-6


# test-fourier.cc:39-72  test_fftshift: exact permutation, n = 15, 16
@pytest.mark.parametrize("n", [15, 16])
def test_fftshift(orc, n):
    x = np.arange(n).astype(np.complex64)
    y = orc.fftshift(x).real
    m = n // 2
    if n & 1:
        ref = np.concatenate([np.arange(m + 1, n), np.arange(0, m + 1)])
    else:
        ref = np.concatenate([np.arange(m, n), np.arange(0, m)])
    assert np.array_equal(y, ref.astype(np.float32))


# test-tsd.cc:248-258  pp2 known values
def test_next_pow2(orc):
    for i, r in [(1, 1), (2, 2), (3, 4), (4, 4), (5, 8), (1 << 16, 1 << 16), ((1 << 16) - 1, 1 << 16)]:
        assert orc.next_pow2(i) == r


# test-ra.cc:55-160  test_ra_unit: output count within 1 %, amplitude error < 10 %,
# spurious <= -50 dB on a 2 kHz sine at fe = 100 kHz; filtre_itrp / sinc{127,256,0.5,"hn"}
# (:150-153) and filtre_reechan's own sinc{15,256,fcut,"hn"} (ra.cc:149-152).
@pytest.mark.parametrize("ratio", [1.0, 1.5, 0.5, 2.0, 1.2, float(np.pi)])
@pytest.mark.parametrize("K,fc", [(127, 0.5), (15, None)])
def test_ra_unit(orc, ratio, K, fc):
    if K == 15:
        nd, nu, post, fcut = orc.reechan_config(ratio)
        if nd or nu or abs(post - 1) < 1e-6:
            pytest.skip("half-band stages / bypass are not the interpolator path")
        r = orc.Resampler(post, 15, 256, fcut)
        eff = post
    else:
        if ratio >= 2.5:
            pytest.skip("single-stage interpolator is used within [0.5,2) by filtre_reechan")
        r = orc.Resampler(ratio, K, 256, fc)
        eff = ratio
    fe, f2 = 100e3, 2e3
    t = np.arange(1000) / fe
    x = np.sin(t * 2 * np.pi * f2).astype(np.float32)
    y = r.step(x)
    assert 100.0 * abs((len(y) - eff * len(x)) / len(x)) < 1
    amp1, amp2 = x.max() - x.min(), y.max() - y.min()
    assert 100 * (amp1 - amp2) / amp1 < 10
    # spurious: remove the best-fit sine at f2/(eff*fe) from the steady-state part
    yy = y[int(K * eff) + 5:].astype(np.float64)
    n = np.arange(len(yy))
    w = 2 * np.pi * f2 / (eff * fe)
    A = np.stack([np.sin(w * n), np.cos(w * n)], 1)
    c, *_ = np.linalg.lstsq(A, yy, rcond=None)
    res = yy - A @ c
    win = np.hanning(len(res))
    S = np.abs(np.fft.rfft(res * win)) / (win.sum() / 2)
    assert 20 * np.log10(S.max() / np.hypot(*c) + 1e-30) <= -50


# Known-answer counts of the float32 phase recurrence measured in SURVEY.md section 7
# (ratio = float(160/147): period 3 853 516 inputs <-> 4 194 303 outputs).
def test_ra_counts_160_147(orc):
    r = orc.Resampler(np.float32(160.0) / np.float32(147.0))
    assert r.r.increment == np.float32(0.918749988)
    nout, _, _ = r.schedule(1 << 20, want=False)
    assert nout == 1141308
    r2 = orc.Resampler(np.float32(160.0) / np.float32(147.0))
    nout, _, _ = r2.schedule(1 << 26, want=False)
    assert nout == 73043660


def test_ra_schedule_matches_data_path(orc):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    ratio = np.float32(160.0) / np.float32(147.0)
    y = orc.Resampler(ratio).step(x)
    r = orc.Resampler(ratio)
    nout, idx, col = r.schedule(len(x))
    assert nout == len(y)
    lut = r.lut
    xp = np.concatenate([np.zeros(14, np.complex64), x])
    k = 1234
    win = xp[idx[k]: idx[k] + 15]
    acc = np.complex64(0)
    for t in range(15):
        acc = np.complex64(acc + lut[col[k], t] * win[t])
    assert abs(acc - y[k]) <= 1e-6 * max(1, abs(y[k]))


# design pins: README / test-filtres style sanity on the windowed-sinc designer
def test_design_rif_fen(orc):
    h = orc.design_rif_fen(31, "lp", 0.25)
    assert abs(h.sum() - 1) < 1e-6
    assert np.allclose(h, h[::-1], atol=1e-7)
    h = orc.design_rif_fen(127, "lp", 0.02)
    H = np.abs(np.fft.rfft(h, 4096))
    assert H[0] == pytest.approx(1, abs=1e-5)
    assert H[int(0.06 * 4096):].max() < 1e-2


# test-filtres.cc:668-679,327-404  test_riia: 12th-order Butterworth lp magnitude template
def test_riia_butterworth_template(orc):
    z, p, mn, md = orc.design_butter_lp(12, 0.25)
    ch = orc.SosChain(z, p, mn, md)
    assert ch.nsec == 6
    co, gain, r1 = ch.coefs()
    assert r1 is None
    w = np.exp(-2j * np.pi * np.linspace(0, 0.5, 513))
    H = np.ones_like(w) * gain
    for b0, b1, b2, a1, a2 in co:
        H *= (b0 + b1 * w + b2 * w * w) / (1 + a1 * w + a2 * w * w)
    mag = np.abs(H)
    assert mag[0] == pytest.approx(1, abs=1e-3)
    assert mag[256] == pytest.approx(np.sqrt(0.5), abs=2e-3)      # -3 dB at fcut
    assert mag[:200].min() > 0.97 and mag[330:].max() < 2e-2
    # impulse response through the chain follows the same transfer function
    x = np.zeros(4096, np.float32)
    x[0] = 1
    y = ch.step(x)
    # first-call state seed = x(0) (filtre-rt.cc:361-365) makes this differ from a
    # zero-state response, so only check stability / decay here
    assert np.isfinite(y).all() and np.abs(y[2000:]).max() < 1e-6


# The DATA PATH of the SOS chain tied to what test_riia holds (test-filtres.cc:668-679 -> test_design,
# :327-404): the template of design_riia(12, "lp", "butt", 0.25, 0.1, 60) sampled by frmag at 2048
# frequencies of [0, 0.5) -- pass band = the first 800 bins within 0.1 of gain 1, stop band = the last
# 800 bins within 0.1 of gain 0.  Here the bins are MEASURED through the recursion itself (DF2 sections,
# first-sample seed, final gain: filtre-rt.cc:361-381,562-571): a complex exponential of each bin
# frequency goes through the chain and the steady-state output amplitude is the gain.  (Exact sample
# values of this path stay "parity unpinned": no reference test holds one.)
RIIA_TEMPLATE_BINS_PASS = (0, 100, 400, 700, 799)
RIIA_TEMPLATE_BINS_STOP = (1248, 1300, 1500, 1800, 2047)


def riia_template_gain(step, bin_index, n=20000):
    f = 0.5 * bin_index / 2048.0
    x = np.exp(2j * np.pi * f * np.arange(n)).astype(np.complex64)
    y = step(x)
    return float(np.abs(y[n // 2:]).mean())


def butter12_gain(f, fc=0.25):
    return 1.0 / np.sqrt(1.0 + (np.tan(np.pi * f) / np.tan(np.pi * fc)) ** 24)


def test_riia_template_through_the_oracle_recursion(orc):
    z, p, mn, md = orc.design_butter_lp(12, 0.25)
    for forme in (2, 1):
        for k in RIIA_TEMPLATE_BINS_PASS + RIIA_TEMPLATE_BINS_STOP:
            g = riia_template_gain(lambda x: orc.SosChain(z, p, mn, md, forme).step(x), k)
            if k < 800:
                assert abs(g - 1.0) <= 0.1, (forme, k, g)          # test_design: emax_bp <= err_max = 0.1
            else:
                assert g <= 0.1, (forme, k, g)                     # emax_bc <= 0.1
            # and the Butterworth law itself, what "butt" designs: |H|^2 = 1 / (1 + (tan pi f / tan pi fc)^24)
            assert abs(g - butter12_gain(0.5 * k / 2048.0)) <= 2e-3, (forme, k, g)


# ---- integer-rate stages ------------------------------------------------------------------
# test-filtres.cc:186-200  test_decimateur: R = 3 on 0..89 in blocks of 4 -> exactly 0,3,...,87
def test_decimateur(orc):
    d = orc.Decimateur(3)
    x = np.arange(90, dtype=np.float32)
    y = np.concatenate([d.step(x[o:o + 4]) for o in range(0, 90, 4)])
    assert len(y) == 30 and np.array_equal(y, np.arange(0, 90, 3, dtype=np.float32))


def _ra_unit_checks(x, y, ratio, skip):
    """test_ra_unit's checks (test-ra.cc:126-143): count within 1 %, amplitude error < 10 %."""
    assert 100.0 * abs((len(y) - ratio * len(x)) / len(x)) < 1
    amp1, amp2 = x.max() - x.min(), y[skip:].max() - y[skip:].min()
    assert 100 * (amp1 - amp2) / amp1 < 10


# test-ra.cc:166-199  half-band, FIR-decim R in {2,3,4,5,8}, polyphase upsampler x2
def test_polyphase_stages(orc):
    fe, f2 = 100e3, 2e3
    x = np.sin(np.arange(1000) / fe * 2 * np.pi * f2).astype(np.float32)
    h = orc.design_rif_fen(15, "lp", 0.25)
    _ra_unit_checks(x, orc.PolyDecim(h, 2, 1).step(x), 0.5, 10)
    for R in (2, 3, 4, 5, 8):
        _ra_unit_checks(x, orc.PolyDecim(orc.design_rif_fen(15, "lp", 0.5 / R), R, 0).step(x), 1.0 / R, 10)
    _ra_unit_checks(x, orc.PolyUps(h, 2).step(x), 2.0, 20)
    # FiltreRIFDecim applies the taps un-reversed (polyphase.cc:223-229): equals the convolution
    # only for symmetric taps -- check with an asymmetric filter against an explicit correlation
    hh = np.array([1.0, 2.0, 3.0], np.float32)
    xx = np.arange(1, 13, dtype=np.float32)
    y = orc.PolyDecim(hh, 2, 0).step(xx)
    xp = np.concatenate([np.zeros(2, np.float32), xx])
    ref = np.array([np.dot(hh, xp[a:a + 3]) for a in range(1, 12, 2)], np.float32)
    assert np.array_equal(y, ref)


# ---- OLA engine / psd_welch restatements (oracle/ola_oracle.py): no reference test holds numbers for
# them (core/tests/test-filtre-fft.cc only plots), so they are pinned on what the reference code
# implies: identity processing = pure delay, FiltreFFTRIF's H = the direct FIR delayed by Ne - M
# (the contract core/tests/test-filtres.cc:514-554 checks through filtre_rif_fft), Welch = numpy.
def test_ola_oracle_identity_and_fir(orc):
    from oracle import ola_oracle
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(8192) + 1j * rng.standard_normal(8192)).astype(np.complex64)
    Ne = 512
    y = ola_oracle.Ola(Ne, 0, None, lambda X: X).step(x)
    assert np.abs(y[Ne:] - x[:-Ne]).max() < 1e-5 and np.abs(y[:Ne]).max() == 0
    yw = ola_oracle.Ola(Ne, 0, ola_oracle.fen_hann_periodique(Ne), lambda X: X).step(x)
    assert len(yw) == len(x) - Ne and np.abs(yw[Ne // 2:] - 0.5 * x[:len(yw) - Ne // 2]).max() < 1e-5
    M = 127
    h = orc.design_rif_fen(M, "lp", 0.02)                        # the design of test_rif_vs_rif_fft
    o = ola_oracle.Ola(Ne, M, None, None)
    h2 = np.zeros(o.N, np.complex64)
    h2[o.N - M:] = h
    H = orc.fft(h2, True) * np.float32(np.sqrt(o.N))             # fourier.cc:963-966
    o.cb = lambda X: X * H
    y = o.step(x)
    ref = orc.fir(h, x)
    d = Ne - M                                                   # = Nz - M for the reference's Ne = Nz = 512
    assert np.abs(y[d:] - ref[:len(y) - d]).max() <= 1e-5 * np.abs(ref).max()


def test_welch_oracle_against_numpy(orc):
    from oracle import ola_oracle
    rng = np.random.default_rng(2)
    for N in (64, 100, 129):
        x = (rng.standard_normal(20 * N + 3) + 1j * rng.standard_normal(20 * N + 3)).astype(np.complex64)
        w = ola_oracle.fen_hann_periodique(N)
        S, nseg = ola_oracle.psd_welch_sum(x, N, w)
        ref = np.zeros(N)
        k = 0
        i = 0
        while i + N < len(x):
            X = np.fft.fft(x[i:i + N].astype(np.complex128) * w) / np.sqrt(N)
            p = np.abs(X) ** 2
            h = N // 2
            ref += np.concatenate([p[N - h:], p[:N - h]])
            i += max(N // 2, 1)
            k += 1
        assert k == nseg and np.abs(S - ref).max() <= 2e-5 * ref.max()
