"""CPU restatement of libtsd's OLA frequency-domain engine (OLA<cfloat>, behind filtre_fft):
core/src/fourier/fourier.cc:737-940.  TEST INFRASTRUCTURE ONLY (tests/ and smoke()); the
product path never imports it.  The transforms are the oracle's own restatement of the
reference's radix-2 plan (pyoracle.fft, fourier.cc:61-121); everything else follows the
reference statement by statement, in float32 / complex64.

Parity status: the reference's own test of this engine (core/tests/test-filtre-fft.cc) checks a
pass-through and a delayed FIR with tolerances; those checks are ported in
tests/cpp/test_host_api.cc (test_filtrage_ola_ref / test_filtre_fft) and in
tests/test_ola_gpu.py -- beyond them this restatement is "parity unpinned"."""
import numpy as np

from . import pyoracle as orc

c64 = np.complex64
f32 = np.float32


def fen_hann_periodique(n):
    """fenêtre("hn", n, false): fenetres.cc:16-56 (fen_inter) + :127-130 (a = 0.5)."""
    if n & 1:
        tmin, tmax = -(n // 2), n // 2 - (f32(n) - f32(1)) / f32(n)
    else:
        tmin, tmax = -(n // 2), (n - 1) // 2
    t = np.linspace(f32(tmin) / f32(n), f32(tmax) / f32(n), n).astype(f32)
    return (f32(0.5) + f32(0.5) * np.cos(f32(2 * np.pi) * t)).astype(f32)


class Ola:
    def __init__(self, Ne=0, nb_zeros_min=0, window=None, traitement_freq=None):
        self.Ne = Ne if Ne > 0 else 512                                   # :770-771
        self.N = orc.next_pow2(self.Ne + nb_zeros_min)                    # :776
        self.Nz = self.N - self.Ne
        self.cnt_ech = -(self.Ne // 2)                                    # :779
        self.padded = np.zeros(self.N, c64)
        self.last = np.zeros(self.Ne, c64)
        self.svg = np.zeros(self.Ne, c64)
        self.fen = None if window is None else np.asarray(window, f32)
        self.cb = traitement_freq
        self.tampon = np.zeros(0, c64)                                    # tampon_création(Ne, ...) (:806-812)

    def step(self, x):
        """OLA::step (:815-836): re-blocking into blocks of Ne, outputs concatenated."""
        x = np.asarray(x, c64)
        allx = np.concatenate([self.tampon, x])
        B = len(allx) // self.Ne
        self.tampon = allx[B * self.Ne:].copy()
        out = [self.step_interne(allx[b * self.Ne:(b + 1) * self.Ne]) for b in range(B)]
        return np.concatenate(out) if out else np.zeros(0, c64)

    def _tf(self, v):
        X = orc.fft(v, True)                                              # plan.step(padded, X)
        X = np.asarray(self.cb(X), c64)                                   # config.traitement_freq(X)
        return orc.fft(X, False)                                          # plan.step(X, x2, non)

    def step_interne(self, x):
        Ne, N, Nz, h = self.Ne, self.N, self.Nz, self.Ne // 2
        if self.fen is None:
            self.padded[N - Ne:] = x                                      # :850
            x2 = self._tf(self.padded)
            self.svg[Ne - Nz:] += x2[:Nz]                                 # :870
            y = self.svg.copy()
            self.svg = x2[N - Ne:].copy()                                 # :872
            self.cnt_ech += Ne
            return y
        fc = self.fen.astype(c64)
        self.padded[N - h:] = x[:h]                                       # :885
        self.padded[N - Ne:] *= fc                                        # :886
        x2 = self._tf(self.padded)
        self.svg[Ne - Nz:] += x2[:Nz]                                     # :892
        self.last[h:] += self.svg[:h] / f32(2)                            # :895
        y = self.last.copy() if self.cnt_ech >= 0 else np.zeros(0, c64)   # :896-899
        self.last[:h] = self.svg[h:] / f32(2)                             # :901
        self.last[h:] = 0                                                 # :902
        self.svg = x2[N - Ne:].copy()                                     # :905
        self.cnt_ech += h
        self.padded[N - Ne:] = x * fc                                     # :910
        x2 = self._tf(self.padded)
        self.svg[Ne - Nz:] += x2[:Nz]                                     # :917
        self.last += self.svg / f32(2)                                    # :918
        self.svg = x2[Nz:Nz + Ne].copy()                                  # :919
        self.cnt_ech += h
        self.padded[Nz:Nz + h] = x[h:]                                    # :926
        return y


def psd_welch_sum(x, N, window):
    """psd_welch before pow2db (freqestim.cc:7-20): S += fftshift(abs2(fft(x.segment(i, N) * f)))
    for i = 0, N/2, ... while i + N < len(x); fftshift as fourier.hpp:232-248."""
    x = np.asarray(x, c64)
    f = np.asarray(window, f32)
    S = np.zeros(N, f32)
    pas = max(N // 2, 1)
    i, nseg = 0, 0
    h = N // 2
    while i + N < len(x):
        X = orc.fft((x[i:i + N] * f).astype(c64), True)
        p = (X.real * X.real + X.imag * X.imag).astype(f32)
        S += np.concatenate([p[N - h:], p[:N - h]])
        i += pas
        nseg += 1
    return S, nseg


class Spectrum:
    """rt_spectrum / Spectrum (fourier.cc:1162-1342), statement by statement in float32.
    sweep = None or (step, masque_bf, masque_hf).  `window`: fenêtre(config.fenetre, Nf, non) BEFORE normalisation."""

    def __init__(self, BS, nmeans, nsubs, window, sweep=None):
        self.BS, self.nmeans, self.nsubs = BS, nmeans, nsubs
        Nf = self.Nf = BS // nsubs                                        # :1184
        self.Ns = Nf
        self.masque = np.ones(Nf, f32)                                    # :1187-1194
        self.sweep = sweep
        if sweep is not None:
            step, bf, hf = sweep
            if hf > 0:
                self.masque[:hf] = 0
                self.masque[Nf - hf:] = 0
            if bf > 0:
                self.masque[Nf // 2 - bf:Nf // 2 + bf] = 0
            self.Ns = Nf + (nsubs - 1) * step                             # :1198
            self.mag_cnt = np.zeros(self.Ns, f32)
            for i in range(nsubs):
                self.mag_cnt[i * step:i * step + Nf] += self.masque       # :1200-1201
            self.mag_cnt = np.maximum(self.mag_cnt, f32(1.0))             # :1203
        self.mag_moy = np.zeros(self.Ns, f32)
        f = np.asarray(window, f32)
        # f = sqrt(Nf / abs2(f).somme()) * f  (:1211; somme() accumulates in double, tableau.hpp:656-717)
        e = f32(np.sum((f * f).astype(f32).astype(np.float64)))
        self.f = (f32(np.sqrt(f32(Nf) / e)) * f).astype(f32)
        self.cntmag = 0

    def step(self, x):
        """one block of BS samples -> Ns floats (dB) every nmeans-th call, else an empty vector (:1236-1336)"""
        x = np.asarray(x, c64)
        assert len(x) == self.BS
        Nf, h = self.Nf, self.Nf // 2
        for i in range(self.nsubs):
            X = orc.fft((x[i * Nf:(i + 1) * Nf] * self.f).astype(c64), True)          # plan->step(x.segment(i * Nf, Nf) * f)
            p = (X.real * X.real + X.imag * X.imag).astype(f32)                       # abs2
            p = np.concatenate([p[Nf - h:], p[:Nf - h]])                              # fftshift (fourier.hpp:232-248)
            if self.sweep is not None and self.nsubs > 1:                             # (nsubs == 1: the `sinon` branch, no masque)
                step = self.sweep[0]
                self.mag_moy[i * step:i * step + Nf] += (p * self.masque).astype(f32)  # :1265
            else:
                self.mag_moy += p                                                      # :1270 / :1277
        self.cntmag += 1
        if self.cntmag < self.nmeans:
            return np.zeros(0, f32)
        m = (self.mag_moy / f32(self.nmeans * self.nsubs * Nf)).astype(f32)            # :1284
        if self.sweep is not None:
            m = (m / self.mag_cnt).astype(f32)                                        # :1286
        y = (f32(10) * np.log10(m + np.finfo(f32).tiny, dtype=f32)).astype(f32)        # pow2db(mag_moy + min) (:1287)
        self.mag_moy[:] = 0
        self.cntmag = 0
        return y


# ---- correlations and delay estimate: core/src/fourier/fourier.cc:489-597, estimation-delais.cc:9-14,100-118
def correlation_freq(X0, X1):
    """fourier.cc:489-503: Y(0) = X0(0) conj(X1(0)); Y.tail(n-1) = reversed tails multiplied; times sqrt(n)."""
    n = len(X0)
    Y = np.empty(n, c64)
    Y[0] = X0[0] * np.conj(X1[0])
    Y[1:] = (X0[1:][::-1] * np.conj(X1[1:][::-1])).astype(c64)
    return (Y * c64(np.sqrt(f32(n)))).astype(c64)


def correlateur_bloc(x0, x1):
    """TFRCorrelateurBloc::step (fourier.cc:505-531): two forward plans, correlation_freq, inverse plan."""
    return orc.fft(correlation_freq(orc.fft(x0), orc.fft(x1)), False)


def xcorrb(x, y=None, m=-1):
    """fourier.cc:534-560: biased cross-correlation, lags -(m-1) .. (m-1)."""
    x = np.asarray(x, c64)
    y = x if y is None else np.asarray(y, c64)
    n = len(x)
    if m < 0:
        m = n
    x2 = np.zeros(n + 2 * m, c64)
    y2 = np.zeros(n + 2 * m, c64)
    x2[m:m + n] = x
    y2[m:m + n] = y
    r = correlateur_bloc(x2, y2)
    res = np.empty(2 * m - 1, c64)
    res[m - 1:] = r[:m] / f32(n)
    res[:m - 1] = r[len(r) - (m - 1):] / f32(n)
    return np.linspace(-(m - 1), m - 1, 2 * m - 1).astype(f32), res


def xcorr(x, y=None, m=-1):
    """fourier.cc:562-597: unbiased version (the biased one divided by (n - |lag|) / n)."""
    n = len(x)
    if m < 0:
        m = n
    lags, zb = xcorrb(x, y, m)
    if m > 1:
        a = np.linspace(n - (m - 1), n - 1, m - 1).astype(f32) / f32(n)
        zb[:m - 1] = (zb[:m - 1] / a.astype(c64)).astype(c64)
        zb[len(zb) - (m - 1):] = (zb[len(zb) - (m - 1):] / a[::-1].astype(c64)).astype(c64)
    return lags, zb


def estimation_delais(x, y):
    """estimation-delais.cc:100-118 (vectors of one length): peak of |xcorrb| / (rms x . rms y), quadratic
    interpolation (:9-14) -> (delay, score)."""
    x = np.asarray(x, c64)
    y = np.asarray(y, c64)
    lags, corr = xcorrb(x, y)
    e1 = np.sqrt(f32(np.mean(np.abs(x.astype(np.complex128)) ** 2)))
    e2 = np.sqrt(f32(np.mean(np.abs(y.astype(np.complex128)) ** 2)))
    cn = (np.abs(corr) / (e1 * e2)).astype(f32)
    k = int(np.argmax(cn))
    d = 0.0
    if 0 < k < len(cn) - 1:
        d = float((cn[k + 1] - cn[k - 1]) / (2 * (2 * cn[k] - cn[k + 1] - cn[k - 1])))
        d = min(0.5, max(-0.5, d))
    return float(lags[k]) + d, float(cn[k])
