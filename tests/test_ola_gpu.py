"""OLA frequency-domain engine (filtre_fft, fourier.cc:737-940) through the C ABI against the
numpy restatement oracle/ola_oracle.py, plus size-independent properties: identity processing =
the input delayed by one block (windowed mode: by half a block and halved, the case
core/tests/test-filtre-fft.cc plots), X *= H with FiltreFFTRIF's H (fourier.cc:963-966) = the
direct FIR delayed by Ne - M samples."""
import numpy as np
import pytest

import libtsd_amd as t
from oracle import ola_oracle, pyoracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))


def randc(rng, n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def response(N, rng, K=33):
    """FiltreFFTRIF's H: h2 = zeros(N), h2.tail(K) = h, H = fft(h2) * sqrt(N) (fourier.cc:963-966)."""
    h = (rng.standard_normal(K) * np.hanning(K)).astype(np.float32)
    h2 = np.zeros(N, np.complex64)
    h2[N - K:] = h
    return orc.fft(h2, True) * np.float32(np.sqrt(N)), h


@pytest.mark.parametrize("Ne,nz,windowed", [(512, 0, False), (512, 100, False), (256, 256, False), (1000, 24, False),
                                            (512, 0, True), (512, 127, True), (64, 64, True), (2048, 500, True)])
def test_builtin_response_matches_oracle(Ne, nz, windowed):
    rng = np.random.default_rng(Ne + nz + windowed)
    win = ola_oracle.fen_hann_periodique(Ne) if windowed else None
    g = t.Ola(Ne, nz, win)
    H, _ = response(g.N, rng)
    g.set_response(H)
    ref = ola_oracle.Ola(Ne, nz, win, lambda X: X * H)
    assert (g.N, g.Ne) == (ref.N, ref.Ne)
    # ragged call lengths: whole blocks, partial blocks, empty calls
    for n in [Ne, 3 * Ne, 17, 0, Ne - 17, 5 * Ne + 3, 1, 2 * Ne - 4]:
        x = randc(rng, n)
        y = g.step(x)
        yr = ref.step(x)
        assert y.shape == yr.shape, (n, y.shape, yr.shape)
        if len(yr):
            assert relerr(y, yr) <= TOL, (n, relerr(y, yr))


@pytest.mark.parametrize("windowed", [False, True])
def test_device_pointers_and_split_processing(windowed):
    """analyse -> caller-side edit of the device spectra -> synthese == the built-in product."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    Ne, nz = 512, 200
    win = ola_oracle.fen_hann_periodique(Ne) if windowed else None
    a, b = t.Ola(Ne, nz, win), t.Ola(Ne, nz, win)
    H, _ = response(a.N, rng)
    a.set_response(H)
    Hd = torch.from_numpy(H).to(dev)
    for n in [4 * Ne, Ne + 100, 3 * Ne - 100]:
        x = randc(rng, n)
        xd = torch.from_numpy(x).to(dev)
        ya = a.step(xd).cpu().numpy()
        sp, nf = b.analyse(xd)
        if nf:
            # a torch view over the handle's spectra buffer: the "device-side callback"
            S = _wrap(sp, nf * b.N, dev).view(nf, b.N)
            S.mul_(Hd)
        torch.cuda.synchronize()
        yb = b.synthese().cpu().numpy()
        assert ya.shape == yb.shape
        if len(ya):
            assert relerr(yb, ya) <= 1e-6


def _wrap(addr, count, dev):
    """complex64 torch tensor over device memory owned by the library (no copy)."""
    import torch

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (count,), "typestr": "<c8", "data": (addr, False), "version": 3, "strides": None}
    return torch.as_tensor(h, device=dev)


def test_identity_is_a_pure_delay():
    """No processing: the input comes back one block later; windowed: half a block later and
    halved (two Hann frames per block sum to 1, each weighted 1/2 -- fourier.cc:895,918), the first
    block yielding nothing."""
    rng = np.random.default_rng(3)
    Ne = 512
    x = randc(rng, 8 * Ne)
    g = t.Ola(Ne, 0, None)
    g.set_response(np.ones(g.N, np.complex64))
    y = g.step(x)
    assert len(y) == 8 * Ne and np.max(np.abs(y[:Ne])) == 0.0
    assert relerr(y[Ne:], x[:-Ne]) <= TOL
    gw = t.Ola(Ne, 0, ola_oracle.fen_hann_periodique(Ne))
    gw.set_response(np.ones(gw.N, np.complex64))
    y = gw.step(x)
    assert len(y) == 7 * Ne
    d = Ne // 2
    assert relerr(y[d:], 0.5 * x[:len(y) - d]) <= TOL


@pytest.mark.parametrize("Ne,K", [(512, 127), (512, 33), (1000, 24), (4096, 1025)])
def test_product_with_H_is_the_delayed_fir(Ne, K):
    """FiltreFFTRIF (fourier.cc:946-990): X *= H through the OLA engine = the direct FIR, Ne - K later."""
    rng = np.random.default_rng(5)
    g = t.Ola(Ne, K, None)
    H, h = response(g.N, rng, K)
    g.set_response(H)
    x = randc(rng, 16 * Ne)
    y = g.step(x)
    yr = orc.fir(h, x)
    d = Ne - K
    assert relerr(y[d:], yr[:len(y) - d]) <= TOL


def test_errors():
    with pytest.raises(t.TsdGpuError):
        t.Ola(100, 400, None)                      # Nz > Ne
    with pytest.raises(t.TsdGpuError):
        t.Ola(511, 0, np.ones(511, np.float32))   # odd block in the windowed mode
    g = t.Ola(512, 0, None)
    x = np.zeros(512, np.complex64)
    g.analyse(x)
    with pytest.raises(t.TsdGpuError):
        g.analyse(x)                               # analyse twice without synthese


# psd_welch (freqestim.cc:7-20) on the device: framing, one batched FFT, sum of the periodograms
# (16 ... 8192: one fused kernel on the LDS transform, 1024: on the in-wave transform; the others: framed transform + sums)
# (125, 250, 375, 1000, 1001, 2000, 2002, 3000: framed by the wave-level Bluestein kernel; 1025, 4000, 6000: framed transform + sums)
@pytest.mark.parametrize("N", [1, 2, 8, 16, 32, 64, 125, 128, 250, 256, 375, 512, 1000, 1001, 1024, 1025, 2000, 2002, 2048, 3000, 4000, 4096, 6000, 8192, 16384])
def test_welch_matches_oracle(N):
    rng = np.random.default_rng(N)
    w = ola_oracle.fen_hann_periodique(N) if N > 2 else np.ones(N, np.float32)
    for n in [0, N - 1, N, N + 1, 3 * N, 10 * N + 7, 100 * N + 3]:
        if n < 0:
            continue
        x = randc(rng, n)
        S, nseg = t.welch(x, N, w)
        ref, nref = ola_oracle.psd_welch_sum(x, N, w)
        assert nseg == nref, (N, n)
        assert np.max(np.abs(S - ref)) <= TOL * max(np.max(ref), 1e-30), (N, n)


@pytest.mark.parametrize("N", [64, 125, 256, 1000, 1024, 4096, 8192, 16384])
def test_welch_long_device_input(N):
    """2^22 samples resident on the device: runs of several segments per transform; white noise of variance 2 ->
    every bin sums to segments * (window energy) * 2 / N within the statistical spread; the first 50 segments and a
    stretch in the middle against the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    n = 1 << 22
    x = torch.view_as_complex(torch.randn(n, 2, device=dev))
    w = ola_oracle.fen_hann_periodique(N)
    S, nseg = t.welch(x, N, w)
    assert nseg == (n - N - 1) // (N // 2) + 1
    expect = nseg * float(np.sum(w.astype(np.float64) ** 2)) * 2.0 / N
    spread = 6.0 / np.sqrt(nseg / 2)
    assert abs(np.mean(S) / expect - 1) < 0.01 and np.max(np.abs(S / expect - 1)) < max(0.1, spread)
    for a, b in ((0, 50 * N + 1), (n // 2 + 17, n // 2 + 17 + 333 * (N // 2) + N + 1)):
        ref, _ = ola_oracle.psd_welch_sum(x[a:b].cpu().numpy(), N, w)
        S2, _ = t.welch(x[a:b], N, w)
        assert np.max(np.abs(S2 - ref)) <= TOL * np.max(ref)


def test_fused_default_geometry_long_runs():
    """Ne = 512 / N = 1024 without window runs as ONE kernel (ols.hip, ola1024_kernel): a wave keeps a run of up to 16
    consecutive blocks in registers and starts it by recomputing the block before.  Long calls (runs of 1, 4 and 16 blocks
    per wave), two calls in a row (the tail carried by the handle), in place, host arrays: against the device FIR that the
    product with H = FFT([0 .. 0 | h]) * sqrt(N) must equal, Ne - K later (fourier.cc:946-990)."""
    import torch
    rng = np.random.default_rng(11)
    K, Ne = 127, 512
    g = t.Ola(Ne, K, None)
    assert (g.N, g.Ne) == (1024, 512)
    H, h = response(g.N, rng, K)
    d = Ne - K
    for B in (3, 700, 9000, 40000):
        n = B * Ne
        x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(B)))
        ref = t.Fir(h, t.C64, t.FIR_DIRECT).step(x)
        g = t.Ola(Ne, K, None)
        g.set_response(H)
        y = g.step(x)
        assert y.shape[0] == n
        e = float((y[d:] - ref[:n - d]).abs().max() / ref.abs().max())
        assert e <= TOL, (B, e)
        # the same stream in two calls (the second one ragged: 100 samples wait for a next call), then in place
        g2 = t.Ola(Ne, K, None)
        g2.set_response(H)
        cut = (B // 2) * Ne
        y2 = torch.cat([g2.step(x[:cut]), g2.step(x[cut:n - Ne + 100])])
        assert torch.equal(y2, y[:y2.shape[0]])
        g3 = t.Ola(Ne, K, None)
        g3.set_response(H)
        xi = x.clone()
        assert torch.equal(g3.step(xi, xi), y)
    # host arrays through the same path
    xh = randc(rng, 64 * Ne)
    g = t.Ola(Ne, K, None)
    g.set_response(H)
    yh = g.step(xh)
    yr = orc.fir(h, xh)
    assert relerr(yh[d:], yr[:len(yh) - d]) <= TOL


@pytest.mark.parametrize("Ne,K", [(2048, 127), (3000, 500), (4096, 1025), (6000, 2000), (8192, 127), (64, 33), (100, 20), (16, 15), (900, 100)])
def test_fused_other_geometries_long_runs(Ne, K):
    """Every other geometry without window whose frame fits the LDS runs as ONE kernel too (ola.hip, ola_run_kernel):
    N/16 threads own a run of consecutive blocks, the carried block in registers (Ne = N/2) or in LDS.  Long calls
    (several blocks per run, the last run cut short), the stream in two ragged calls (waiting samples: block 0 is made
    contiguous on the host side), in place: against the device FIR, Ne - K later (fourier.cc:946-990)."""
    import torch
    rng = np.random.default_rng(12)
    g = t.Ola(Ne, K, None)
    H, h = response(g.N, rng, K)
    d = Ne - K
    for n in (3 * Ne, (1 << 22) // Ne * Ne + Ne, (13 << 20) // Ne * Ne):
        x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n)))
        ref = t.Fir(h, t.C64, t.FIR_DIRECT).step(x)
        g = t.Ola(Ne, K, None)
        g.set_response(H)
        y = g.step(x)
        assert y.shape[0] == n
        e = float((y[d:] - ref[:n - d]).abs().max() / ref.abs().max())
        assert e <= TOL, (n, e)
        g2 = t.Ola(Ne, K, None)
        g2.set_response(H)
        cut = (n // Ne // 2) * Ne + Ne // 3
        y2 = torch.cat([g2.step(x[:cut]), g2.step(x[cut:n - Ne + 7]), g2.step(x[n - Ne + 7:])])
        assert y2.shape[0] == n
        # (the runs fall elsewhere: same sums, another order inside the transforms' last bits only where a run starts)
        assert float((y2 - y).abs().max() / ref.abs().max()) <= 1e-6
        g3 = t.Ola(Ne, K, None)
        g3.set_response(H)
        xi = x.clone()
        assert torch.equal(g3.step(xi, xi), y)


@pytest.mark.parametrize("Ne,nz", [(512, 0), (512, 127), (64, 64), (2048, 500), (1000, 24), (4096, 0), (100, 20)])
def test_fused_windowed_long_runs(Ne, nz):
    """The windowed mode as ONE kernel (ola.hip, olaw_run_kernel: a run of blocks emulated statement by statement, runs past the
    first recompute two blocks of warm-up): long calls -- many runs, the last one cut short -- against the multi-kernel
    engine (TSDGPU_OLA_UNFUSED cannot be flipped inside a process: the analyse / synthese pair IS that engine), the stream in
    ragged calls and in place; and against the numpy restatement on a stream short enough for it."""
    import torch
    rng = np.random.default_rng(Ne + nz)
    win = ola_oracle.fen_hann_periodique(Ne)
    g = t.Ola(Ne, nz, win)
    H, _ = response(g.N, rng)
    Hd = torch.from_numpy(H).cuda()
    n = (3 << 20) // Ne * Ne + Ne
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n)))
    g.set_response(H)
    y = g.step(x)
    assert y.shape[0] == n - Ne                               # the very first block gives no output (fourier.cc:899-901)
    # the multi-kernel engine on the same stream: analyse -> spectra x H -> synthese
    m = t.Ola(Ne, nz, win)
    sp, nf = m.analyse(x)
    spectra = torch.empty((nf, m.N), dtype=torch.complex64, device="cuda")
    t.lib().tsdgpu_memcpy(spectra.data_ptr(), sp, nf * m.N * 8, None)
    spectra *= Hd
    t.lib().tsdgpu_memcpy(sp, spectra.data_ptr(), nf * m.N * 8, None)
    torch.cuda.synchronize()
    ym = m.synthese()
    assert ym.shape == y.shape
    assert float((y - ym).abs().max() / ym.abs().max()) <= 2e-6
    # ragged calls (waiting samples in front of a call, runs falling elsewhere) and in place
    g2 = t.Ola(Ne, nz, win)
    g2.set_response(H)
    cut = (n // Ne // 2) * Ne + Ne // 3
    y2 = torch.cat([g2.step(x[:cut]), g2.step(x[cut:n - Ne + 7]), g2.step(x[n - Ne + 7:])])
    assert y2.shape == y.shape and float((y2 - y).abs().max() / y.abs().max()) <= 1e-6
    g3 = t.Ola(Ne, nz, win)
    g3.set_response(H)
    xi = x.clone()
    y3 = g3.step(xi, xi)
    assert torch.equal(y3, y)
    # the numpy restatement on the first 40 blocks
    ref = ola_oracle.Ola(Ne, nz, win, lambda X: X * H)
    g4 = t.Ola(Ne, nz, win)
    g4.set_response(H)
    xs = x[:40 * Ne + 11].cpu().numpy()
    assert relerr(g4.step(xs), ref.step(xs)) <= TOL
