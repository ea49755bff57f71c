"""rt_spectrum on the device (fourier.cc:1162-1342): tsdgpu_spectrum_* against the numpy restatement
oracle/ola_oracle.py::Spectrum (statement by statement, on the oracle's FFT).  No reference test holds a value for the
analyser ("parity unpinned"): the restatement and the definitional checks below are what pins it.
Tolerance: the sums of |X|^2 are float32 sums in another order than the reference's sequential accumulation -- 1e-5
relative on the linear spectrum, i.e. 4.4e-5 dB, plus the float32 log10."""
import numpy as np
import pytest

from oracle import ola_oracle

pytestmark = pytest.mark.gpu


def rand(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def check_db(got, ref, what):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    # dB of a linear value within 1e-5 relative; bins the mask or the sweep leaves empty are -inf-like (10 log10 FLT_MIN) in both
    lin_g, lin_r = 10.0 ** (got.astype(np.float64) / 10), 10.0 ** (ref.astype(np.float64) / 10)
    err = np.abs(lin_g - lin_r).max() / lin_r.max()
    assert err <= 1e-5, (what, err)
    quiet = ref < -300
    assert np.array_equal(quiet, got < -300), what


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1
    return t


@pytest.mark.parametrize("BS,nsubs,nmeans", [(1024, 1, 10), (4096, 4, 3), (512, 1, 1), (2048, 8, 2), (16384, 1, 2), (1000, 1, 4),
                                              (3000, 3, 2), (96, 2, 5), (32768, 2, 1), (15, 1, 2)])
def test_spectrum_plain_matches_oracle(tg, BS, nsubs, nmeans):
    Nf = BS // nsubs
    w = ola_oracle.fen_hann_periodique(Nf)
    ref = ola_oracle.Spectrum(BS, nmeans, nsubs, w)
    g = tg.Spectrum(BS, nsubs, nmeans, ref.f)
    assert g.Ns == ref.Ns == Nf
    nblocks = 3 * nmeans + 1
    x = rand(nblocks * BS, BS + nsubs) * np.float32(3.0)
    x[:BS] += np.exp(2j * np.pi * 0.123 * np.arange(BS)).astype(np.complex64) * 20          # a line above the noise
    want = [ref.step(x[b * BS:(b + 1) * BS]) for b in range(nblocks)]
    want = np.stack([v for v in want if len(v)])
    # one call for everything (3 spectra and one block left pending when nmeans > 1) ...
    got = g.step(x)
    assert got.shape == (nblocks // nmeans, Nf) and g.pending == nblocks % nmeans
    check_db(got, want, "one call")
    # ... block by block, like Spectrum::step (an empty result until the nmeans-th block)
    g2 = tg.Spectrum(BS, nsubs, nmeans, ref.f)
    rows = []
    for b in range(nblocks):
        r = g2.step(x[b * BS:(b + 1) * BS])
        assert r.shape[0] == (1 if (b + 1) % nmeans == 0 else 0)
        rows += list(r)
    check_db(np.stack(rows), want, "block by block")
    # ... and in ragged calls (the sums of an incomplete group wait on the device)
    g3 = tg.Spectrum(BS, nsubs, nmeans, ref.f)
    rows, b = [], 0
    for k in (1, 2, nmeans, 1, 2 * nmeans - 3 if 2 * nmeans > 3 else 1, 50):
        k = min(k, nblocks - b)
        if k <= 0:
            break
        rows += list(g3.step(x[b * BS:(b + k) * BS]))
        b += k
    check_db(np.stack(rows), want, "ragged calls")


@pytest.mark.parametrize("BS,nsubs,nmeans,step,bf,hf", [(4096, 4, 3, 512, 0, 0), (4096, 4, 2, 700, 3, 20), (2048, 8, 1, 256, 2, 0),
                                                         (3000, 3, 2, 400, 0, 10), (1024, 2, 4, 5000, 0, 0)])
def test_spectrum_sweep_matches_oracle(tg, BS, nsubs, nmeans, step, bf, hf):
    """sweep mode (fourier.cc:1196-1203,1262-1266,1285-1286): sub-block i lands step bins further, masks on the band edges and
    the centre, division by the number of contributions (a step wider than a sub-block leaves bins that nothing reaches)."""
    Nf = BS // nsubs
    w = ola_oracle.fen_hann_periodique(Nf)
    ref = ola_oracle.Spectrum(BS, nmeans, nsubs, w, sweep=(step, bf, hf))
    g = tg.Spectrum(BS, nsubs, nmeans, ref.f, sweep=(step, ref.masque))
    assert g.Ns == ref.Ns == Nf + (nsubs - 1) * step
    nblocks = 2 * nmeans + 1
    x = rand(nblocks * BS, 7 * BS + step)
    want = [ref.step(x[b * BS:(b + 1) * BS]) for b in range(nblocks)]
    want = np.stack([v for v in want if len(v)])
    check_db(g.step(x), want, "one call")
    g2 = tg.Spectrum(BS, nsubs, nmeans, ref.f, sweep=(step, ref.masque))
    rows = []
    for a, b in ((0, 1), (1, nmeans + 1), (nmeans + 1, nblocks)):
        rows += list(g2.step(x[a * BS:b * BS]))
    check_db(np.stack(rows), want, "three calls")


@pytest.mark.parametrize("BS,nsubs,nmeans,sweep", [(1024, 1, 2, (300, 2, 20)), (4099, 4, 2, None), (4099, 4, 3, (512, 0, 8)), (1030, 4, 1, None)])
def test_spectrum_one_sub_block_sweep_and_ragged_block_size(tg, BS, nsubs, nmeans, sweep):
    """(a) nsubs = 1 takes the reference's `sinon` branch (fourier.cc:1272-1277) -- no masque even with sweep.active and
    masque_hf > 0, so the masked bins hold the power, not 10 log10(FLT_MIN); (b) BS need not be a multiple of nsubs: Nf = BS / nsubs
    (:1150-1153) and the trailing samples of every block are ignored (`x.segment(i * Nf, Nf)`, :1254)."""
    Nf = BS // nsubs
    w = ola_oracle.fen_hann_periodique(Nf)
    ref = ola_oracle.Spectrum(BS, nmeans, nsubs, w, sweep=sweep)
    g = tg.Spectrum(BS, nsubs, nmeans, ref.f, sweep=None if sweep is None else (sweep[0], ref.masque))
    assert g.Ns == ref.Ns
    nblocks = 2 * nmeans + 1
    x = rand(nblocks * BS, 5 * BS + nsubs)
    want = [ref.step(x[b * BS:(b + 1) * BS]) for b in range(nblocks)]
    want = np.stack([v for v in want if len(v)])
    if nsubs == 1:
        assert want.min() > -200.0                   # (a masked bin would read 10 log10(FLT_MIN) = -379 dB)
    check_db(g.step(x), want, "one call")
    g2 = tg.Spectrum(BS, nsubs, nmeans, ref.f, sweep=None if sweep is None else (sweep[0], ref.masque))
    rows = []
    for b in range(nblocks):
        rows += list(g2.step(x[b * BS:(b + 1) * BS]))
    check_db(np.stack(rows), want, "block by block")
    import torch
    g3 = tg.Spectrum(BS, nsubs, nmeans, ref.f, sweep=None if sweep is None else (sweep[0], ref.masque))
    yd = g3.step(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    check_db(yd.cpu().numpy(), want, "resident, one call")


def test_spectrum_resident_input_many_blocks(tg):
    """2^22 resident samples in one call: 4096 blocks of 1024, nmeans = 16 -> 256 spectra; definition check on a tone
    (a unit tone at bin q through a rectangular window of energy Nf reads 0 dB at that bin: |X|^2 = Nf for the unitary
    transform, over nmeans nsubs Nf) and parity with the
    oracle on the first and last spectra."""
    import torch
    BS, nmeans = 1024, 16
    x = rand(4096 * BS, 99)
    w = np.ones(BS, np.float32)
    g = tg.Spectrum(BS, 1, nmeans, w)
    xd = torch.from_numpy(x).cuda()
    y = g.step(xd)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (256, BS) and y.is_cuda
    y = y.cpu().numpy()
    ref = ola_oracle.Spectrum(BS, nmeans, 1, w)
    first = [ref.step(x[b * BS:(b + 1) * BS]) for b in range(nmeans)][-1]
    check_db(y[:1], first[None], "first spectrum")
    ref2 = ola_oracle.Spectrum(BS, nmeans, 1, w)
    last = [ref2.step(x[b * BS:(b + 1) * BS]) for b in range(4096 - nmeans, 4096)][-1]
    check_db(y[-1:], last[None], "last spectrum")
    tone = np.exp(2j * np.pi * 100 * np.arange(BS) / BS).astype(np.complex64)
    g1 = tg.Spectrum(BS, 1, 1, w)
    yt = g1.step(tone)[0]
    assert abs(yt[BS // 2 + 100]) < 1e-3 and np.delete(yt, BS // 2 + 100).max() < -60


def test_spectrum_errors(tg):
    w = np.ones(100, np.float32)
    with pytest.raises(tg.TsdGpuError):
        tg.Spectrum(2, 3, 1, np.ones(1, np.float32))                # fewer samples than sub-blocks (Nf = 0)
    assert tg.Spectrum(1000, 3, 1, np.ones(333, np.float32)).Ns == 333       # BS not a multiple of nsubs: Nf = BS / nsubs, like the reference
    g = tg.Spectrum(100, 1, 2, w)
    assert g.step(np.zeros(0, np.complex64)).shape == (0, 100)
    y = g.step(np.zeros(200, np.complex64))
    assert y.shape == (1, 100) and np.all(y < -300)                 # pow2db(0 + FLT_MIN)
