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
