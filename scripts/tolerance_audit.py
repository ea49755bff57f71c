#!/usr/bin/env python3
"""Measured error behind every parity assertion of tests/test_*_gpu.py that is looser than north_star's 1e-5
(VERDICT r3, next #6).  For each class of case three figures, all max-abs relative to the reference's peak:

    gpu-vs-oracle   what the test asserts on (the HIP path against the float32 restatement of libtsd's arithmetic)
    oracle-vs-f64   how far libtsd's own float32 arithmetic sits from the float64 answer of the same definition
    gpu-vs-f64      the same for the HIP path

Where the oracle itself is 1e-5 or more from the float64 answer (float32 Bluestein chirps, float32 window / overlap sums)
a band of 1e-5 around the ORACLE is narrower than the reference's own rounding noise; the wider band is then stated in
INTEGRATION.md "Deliberate differences" with these figures.      python3 scripts/tolerance_audit.py > profiles/rN_tolerance_audit.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libtsd_amd as t                                   # noqa: E402
from oracle import ola_oracle as oo, pyoracle as orc     # noqa: E402

rng = np.random.default_rng(2024)


def crand(n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-30))


def line(what, g_o, o_64, g_64, asserted):
    print(f"{what:64s} gpu-vs-oracle {g_o:9.2e}   oracle-vs-f64 {o_64:9.2e}   gpu-vs-f64 {g_64:9.2e}   asserted {asserted}")


print("# tolerance audit: measured errors behind the parity assertions looser than 1e-5 (scripts/tolerance_audit.py)")
# ---- 1. FFT sizes whose plan goes through the reference's float32 Bluestein chirp (odd n; even n with an odd part > 31)
for n in (15, 125, 1001, 2187, 8191, 1000, 3000, 15360, 12345):
    worst = [0.0, 0.0, 0.0]
    for fwd in (True, False):
        x = crand(n)
        o = orc.fft(x, fwd)
        g = t.Fft(n).step(x.reshape(1, n), fwd)[0]
        x64 = x.astype(np.complex128)
        f = (np.fft.fft(x64) if fwd else np.fft.ifft(x64) * n) / np.sqrt(n)
        worst = [max(worst[0], rel(g, o)), max(worst[1], rel(o, f)), max(worst[2], rel(g, f))]
    line(f"fft n = {n} (odd part {n // (n & -n)})", *worst, "was 2e-5, now 1e-5 (test_fft_gpu, test_fft_sweep_gpu, test_fuzz_gpu)")

# ---- 2. OLA engine: windowed identity, FiltreFFTRIF through the engine
Ne = 512
x = crand(16 * Ne)
gw = t.Ola(Ne, 0, oo.fen_hann_periodique(Ne))
gw.set_response(np.ones(gw.N, np.complex64))
y = gw.step(x)
rw = oo.Ola(Ne, 0, oo.fen_hann_periodique(Ne), lambda X: X)
yo = rw.step(x)
d = Ne // 2
exact = 0.5 * x[:len(y) - d].astype(np.complex128)
line("ola windowed identity (Hann, Ne = 512): 0.5 x delayed", rel(y[d:], yo[d:]), rel(yo[d:], exact), rel(y[d:], exact), "was 2e-5, now 1e-5 (test_ola_gpu)")
for Ne, K in ((512, 127), (1000, 24), (4096, 1025)):
    g = t.Ola(Ne, K, None)
    h = (rng.standard_normal(K) * np.hanning(K)).astype(np.float32)
    h2 = np.zeros(g.N, np.complex64)
    h2[g.N - K:] = h
    H = orc.fft(h2, True) * np.float32(np.sqrt(g.N))
    g.set_response(H)
    x = crand(16 * Ne)
    y = g.step(x)
    ro = oo.Ola(Ne, K, None, lambda X: X * H)
    yo = ro.step(x)
    dd = Ne - K
    f64 = np.convolve(x.astype(np.complex128), h.astype(np.float64))[:len(y) - dd]
    line(f"filtre_rif_fft through the engine, Ne = {Ne}, K = {K}", rel(y[dd:], yo[dd:]), rel(yo[dd:], f64), rel(y[dd:], f64), "was 2e-5, now 1e-5 (test_ola_gpu)")

# ---- 3. rt_spectrum (linear scale)
for BS, nsubs, nmeans in ((1024, 1, 10), (4096, 4, 3), (3000, 3, 2)):
    Nf = BS // nsubs
    w = oo.fen_hann_periodique(Nf)
    ref = oo.Spectrum(BS, nmeans, nsubs, w)
    g = t.Spectrum(BS, nsubs, nmeans, ref.f)
    x = crand(nmeans * BS) * np.float32(3.0)
    want = [ref.step(x[b * BS:(b + 1) * BS]) for b in range(nmeans)][-1]
    got = g.step(x)[0]
    acc = np.zeros(Nf)
    f64w = ref.f.astype(np.float64)
    for b in range(nmeans):
        for i in range(nsubs):
            X = np.fft.fft(x[b * BS + i * Nf:b * BS + (i + 1) * Nf].astype(np.complex128) * f64w) / np.sqrt(Nf)
            acc += np.fft.fftshift(np.abs(X) ** 2)
    lin64 = acc / (nmeans * nsubs * Nf)
    lin = lambda db: 10.0 ** (db.astype(np.float64) / 10)
    line(f"rt_spectrum BS = {BS}, nsubs = {nsubs}, nmeans = {nmeans} (linear power)", rel(lin(got), lin(want)), rel(lin(want), lin64), rel(lin(got), lin64),
         "was 2e-5, now 1e-5 (test_spectrum_gpu)")

# ---- 4. xcorr
for n, m in ((1000, 1000), (4096, 1), (777, 300)):
    a, b = crand(n), crand(n)
    ro = oo.xcorrb(a, b, m)[1]
    gg = t.xcorr(a, b, m, False)
    a64, b64 = a.astype(np.complex128), b.astype(np.complex128)
    # the definition the oracle restates (fourier.cc:534-597): c(l) = sum_k x(k + l) conj(y(k)) / n over the overlapping part
    lags = np.arange(-(m - 1), m)
    f64 = np.array([np.sum(a64[max(0, l):n + min(0, l)] * np.conj(b64[max(0, -l):n - max(0, l)])) / n for l in lags])
    e_o = min(rel(ro, f64), rel(ro, np.conj(f64[::-1])))
    e_g = min(rel(gg, f64), rel(gg, np.conj(f64[::-1])))
    line(f"xcorrb n = {n}, m = {m}", rel(gg, ro), e_o, e_g, "was 2e-5, now 1e-5 (test_detect_gpu, test_fuzz_gpu)")

# ---- 5. analytic Lagrange interpolators (taps from the float phase; the K^2 constant divisions of itrp.cc:96-133 folded
#         into one host-computed reciprocal per tap on the device)
for deg in (3, 5, 7):
    for ratio in (160.0 / 147.0, 0.77):
        x = crand(60000)
        ref, g = orc.Resampler(ratio, analytic=("lagrange", deg)), t.Resampler(ratio, t.C64, analytic=("lagrange", deg))
        line(f"lagrange degree {deg}, ratio {ratio:.4f}", rel(g.step(x), ref.step(x)), float("nan"), float("nan"), "was 5e-5 (test_fuzz_gpu), now 1e-5 as in test_resample_gpu")

# ---- 6. long-memory first-order smoother through FiltreRII (exact carry)
from scipy.signal import lfilter      # noqa: E402
gam = np.float32(3e-5)
nu, de = np.array([gam, 0.0], np.float32), np.array([1.0, -(1.0 - gam)], np.float32)
x = rng.standard_normal(1 << 22).astype(np.float32) + np.float32(2.0)
yg = t.Rii(nu, de, t.F32).step(x)
yo = orc.Rii(nu, de).step(x)
ex = lfilter(nu.astype(np.float64), de.astype(np.float64), x.astype(np.float64))
line("filtre_rii first order, time constant 3e4 samples, 2^22 samples", rel(yg, yo), rel(yo, ex), rel(yg, ex), "2e-5 against float64 (test_host_pipeline_gpu)")

# ---- 7. SOS chain, slices far into a long stream (the > 2^31-sample run checks slices of a shorter replay)
z, p, mn, md = orc.design_butter_lp(12, 0.25)
ch = orc.SosChain(z, p, mn, md)
co, gain, r1 = ch.coefs()
x = rng.standard_normal(1 << 22).astype(np.float32)
yg = t.Sos(co, gain, t.F32, r1).step(x)
yo = ch.step(x)
y64 = ch.run_f64(x) if hasattr(ch, "run_f64") else None
line("sos Butterworth 12, fc 0.25, 2^22 samples", rel(yg, yo), rel(yo, y64) if y64 is not None else float("nan"),
     rel(yg, y64) if y64 is not None else float("nan"), "1e-5 (test_sos_gpu), 2e-5 on slices of the > 2^31 run (test_large_gpu)")
