"""Device-side correlation tail and pattern detector against numpy
restatements of fourier.cc:489-597 / estimation-delais.cc:100-118 (oracle/ola_oracle.py, on the oracle's
FFT) and against the detector's definition."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def crand(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1
    return t


@pytest.mark.parametrize("n,m", [(64, -1), (1000, 10), (1000, 1000), (4096, 1), (777, 300)])
@pytest.mark.parametrize("unbiased", [False, True])
def test_xcorr_matches_the_restatement(tg, n, m, unbiased):
    from oracle import ola_oracle as oo
    x, y = crand(n, 1), crand(n, 2)
    ref = (oo.xcorr if unbiased else oo.xcorrb)(x, y, m)[1]
    got = tg.xcorr(x, y, m, unbiased)
    assert got.shape == ref.shape
    mm = n if m < 0 else m
    # the unbiased scaling n / (n - |lag|) amplifies the float rounding of the extreme lags by as much:
    # the band is the biased one times that factor
    w = (n / (n - np.abs(np.arange(-(mm - 1), mm)))) if unbiased else np.ones(2 * mm - 1)
    band = 1e-5 * np.abs(oo.xcorrb(x, y, m)[1]).max() * w
    assert (np.abs(got - ref) <= band).all()
    # autocorrelation, and device vectors in / out
    import torch
    ra = (oo.xcorr if unbiased else oo.xcorrb)(x, None, m)[1]
    ga = tg.xcorr(torch.from_numpy(x).cuda(), None, m, unbiased).cpu().numpy()
    assert (np.abs(ga - ra) <= 1e-5 * np.abs(oo.xcorrb(x, None, m)[1]).max() * w).all()


@pytest.mark.parametrize("d", [0, 1, -7, 100, -1000])
def test_delay_estimate(tg, d):
    from oracle import ola_oracle as oo
    n = 4096
    base = crand(n + 2048, 5)
    x = base[1024:1024 + n].copy()
    y = base[1024 - d:1024 - d + n].copy()           # y = x delayed by d samples
    delay, score = tg.delay_estimate(x, y)
    rd, rs = oo.estimation_delais(x, y)
    assert abs(delay - rd) <= 1e-3 and abs(score - rs) <= 1e-4
    assert abs(delay - d) <= 0.5 and score > 0.5


@pytest.mark.parametrize("mode", [0, 1])
def test_detector_scores_and_peaks(tg, mode):
    rng = np.random.default_rng(3)
    M, Ne, nblk = 200, 2048, 6
    pat = (rng.standard_normal(M) + 1j * rng.standard_normal(M)).astype(np.complex64)
    x = (0.01 * (rng.standard_normal(Ne * nblk) + 1j * rng.standard_normal(Ne * nblk))).astype(np.complex64)
    starts = [300, Ne - 100, 2 * Ne - 1, 3 * Ne, 4 * Ne + 1000]       # inside a block, across borders, on a border
    for k, s in enumerate(starts):
        x[s:s + M] += ((0.5 + 0.3 * k) * np.exp(0.4j * k)) * pat
    det = tg.Detector(pat, Ne, mode, threshold=0.8)
    found, scores = [], []
    for b in range(nblk):
        sc, pk = det.step(x[b * Ne:(b + 1) * Ne].copy())
        scores.append(sc)
        for p in pk:
            found.append((b * Ne + p.index - det.delay, p))
    # the definition of the score, on the host: correlation with the unit-energy pattern over the window
    # energy (the score at output index i belongs to the pattern starting at i - delay)
    pu = pat / np.sqrt(np.sum(np.abs(pat) ** 2))
    sc = np.concatenate(scores)
    for s in starts:
        seg = x[s:s + M]
        want = np.abs(np.vdot(pu, seg)) / np.sqrt(np.mean(np.abs(seg) ** 2)) / np.sqrt(M)
        i = s + det.delay
        if i < len(sc):
            assert abs(sc[i] - want) <= 2e-3, (s, sc[i], want)
            assert want > 0.95
    assert [f[0] for f in found] == [s for s in starts if s + det.delay + M < Ne * nblk]
    for pos, p in found:
        assert p.s0 >= p.s_m1 and p.s0 >= p.s_p1 and p.s0 > 0.8


@pytest.mark.parametrize("mode", [0, 1])
def test_detector_quiet_stream_start_and_pauses(tg, mode):
    """found by the fuzz sweep: where the stream is quiet -- its first samples (one tiny sample in the energy window), a pause
    of exact zeros -- the score divides by a near-zero energy average; with that average taken from an FFT convolution the
    rounding noise of the block's loud samples stood in its place and scores of 3e5 came out.  The average (and, in FIR
    mode, the correlation) now come from time-domain sums: the scores there stay small and no peak is reported."""
    rng = np.random.default_rng(9907)
    M, Ne, nblk = 200, 1024, 6
    pat = crand(M, 31)
    x = np.zeros(Ne * nblk, np.complex64)
    x[0] = 1e-4 + 2e-5j                                    # a nearly empty first window
    x[1:300] = (0.01 * crand(299, 32)).astype(np.complex64)
    starts = [700, 2 * Ne + 50, 4 * Ne - 30]               # patterns with pauses of exact zeros between them
    for s in starts:
        x[s:s + M] += pat
    det = tg.Detector(pat, Ne, mode, threshold=0.8)
    found, scores = [], []
    for b in range(nblk):
        sc, pk = det.step(x[b * Ne:(b + 1) * Ne].copy())
        scores.append(sc)
        found += [b * Ne + p.index - det.delay for p in pk]
    sc = np.concatenate(scores)
    assert np.isfinite(sc).all()
    assert found == starts, (found, starts)
    assert sc[det.delay - (M - 1): det.delay - (M - 1) + 250].max() < 0.8      # the first windows of the stream
    del rng
