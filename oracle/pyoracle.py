"""ctypes/numpy front-end of the CPU oracle (oracle/tsd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package libtsd_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c64 = np.complex64
f32 = np.float32


def build(force=False):
    """Compile both oracle variants with the committed Makefile."""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", _HERE, "-s"], check=True, capture_output=True)


def _cpu_has_v3():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = set(line.split())
                    return {"avx2", "fma", "bmi2"} <= fl
    except OSError:
        pass
    return False


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    name = "liborc_v3.so" if _cpu_has_v3() else "liborc_base.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    _declare(L)
    _LIB = L
    return L


class CF(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class Biquad(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("b0", "b1", "b2", "a1", "a2")]


class Sos(C.Structure):
    _fields_ = [("nsec", C.c_int), ("sec", Biquad * 32), ("avec_rii1", C.c_int),
                ("r_b0", C.c_float), ("r_b1", C.c_float), ("r_a1", C.c_float),
                ("gain", C.c_float), ("forme", C.c_int)]


class BqStateF(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("x1", "x2", "y0", "y1", "y2")] + [("premier_appel", C.c_int)]


class BqStateC(C.Structure):
    _fields_ = [(k, CF) for k in ("x1", "x2", "y0", "y1", "y2")] + [("premier_appel", C.c_int)]


class SosStateF(C.Structure):
    _fields_ = [("sec", BqStateF * 32), ("r_x1", C.c_float), ("r_y1", C.c_float)]


class SosStateC(C.Structure):
    _fields_ = [("sec", BqStateC * 32), ("r_x1", CF), ("r_y1", CF)]


class Ra(C.Structure):
    _fields_ = [("phase", C.c_float), ("ratio", C.c_float), ("increment", C.c_float),
                ("K", C.c_int), ("nphases", C.c_int), ("lut", C.c_void_p),
                ("fen_f", C.c_float * 256), ("fen_c", CF * 256), ("mode", C.c_int)]


def _declare(L):
    vp, i32, i64, fl = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.orc_fir_ff.argtypes = [vp, i32, vp, vp, vp, vp, i64]
    L.orc_fir_cf.argtypes = [vp, i32, vp, vp, vp, vp, i64]
    L.orc_fir_cc.argtypes = [vp, i32, vp, vp, vp, vp, i64]
    L.orc_rii_f.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i64]
    L.orc_rii_c.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i64]
    L.orc_sos_from_zpk.argtypes = [vp, vp, vp, i32, CF, CF, i32]
    L.orc_sos_from_zpk.restype = i32
    L.orc_sos_state_init_f.argtypes = [vp]
    L.orc_sos_state_init_c.argtypes = [vp]
    L.orc_sos_step_f.argtypes = [vp, vp, vp, vp, i64]
    L.orc_sos_step_c.argtypes = [vp, vp, vp, vp, i64]
    L.orc_sos_run_f64.argtypes = [vp, vp, vp, i64]
    L.orc_fft.argtypes = [vp, vp, i32, i32]
    L.orc_fft_twiddles.argtypes = [vp, i32]
    L.orc_rfft.argtypes = [vp, vp, i32]
    L.orc_fftshift_c.argtypes = [vp, vp, i32]
    L.orc_csym_force.argtypes = [vp, i32]
    L.orc_next_pow2.argtypes = [i32]
    L.orc_next_pow2.restype = i32
    L.orc_itrp_sinc_lut.argtypes = [i32, i32, fl, vp]
    L.orc_ra_init_analytic.argtypes = [vp, fl, i32, i32]
    L.orc_ra_init.argtypes = [vp, fl, i32, i32, vp]
    L.orc_ra_step_c.argtypes = [vp, vp, i64, vp]
    L.orc_ra_step_c.restype = i64
    L.orc_ra_step_f.argtypes = [vp, vp, i64, vp]
    L.orc_ra_step_f.restype = i64
    L.orc_ra_schedule.argtypes = [vp, i64, vp, vp, i64]
    L.orc_ra_schedule.restype = i64
    L.orc_reechan_config.argtypes = [fl, vp, vp, vp, vp]
    L.orc_decimateur_f.argtypes = [i32, vp, vp, i64, vp]
    L.orc_decimateur_f.restype = i64
    L.orc_polydecim_f.argtypes = [i32, vp, i32, i32, vp, vp, vp, vp, i64, vp]
    L.orc_polydecim_f.restype = i64
    L.orc_polydecim_c.argtypes = [i32, vp, i32, i32, vp, vp, vp, vp, i64, vp]
    L.orc_polydecim_c.restype = i64
    L.orc_ups_prepare.argtypes = [vp, i32, i32, vp]
    L.orc_ups_prepare.restype = i32
    L.orc_ups_f.argtypes = [vp, i32, i32, vp, vp, vp, i64, vp]
    L.orc_ups_f.restype = i64
    L.orc_ups_c.argtypes = [vp, i32, i32, vp, vp, vp, i64, vp]
    L.orc_ups_c.restype = i64
    L.orc_design_rif_fen_hann.argtypes = [i32, i32, fl, vp]
    L.orc_design_butter_lp.argtypes = [i32, fl, vp, vp, vp, vp]
    L.orc_sinc2.argtypes = [fl, fl]
    L.orc_sinc2.restype = fl
    L.orc_linspace.argtypes = [fl, fl, i32, vp]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- FIR
class Fir:
    """Stateful FiltreRIF<T,Tc> (filtre-rt.cc:53-109): data float32/complex64, taps f32/c64."""

    def __init__(self, taps):
        taps = np.ascontiguousarray(taps)
        self.cplx_taps = np.iscomplexobj(taps)
        self.taps = taps.astype(c64 if self.cplx_taps else f32)
        self.K = len(self.taps)
        self.index = C.c_int(0)
        self.fen = None

    def step(self, x):
        x = np.ascontiguousarray(x)
        cplx = np.iscomplexobj(x) or self.cplx_taps
        x = x.astype(c64 if cplx else f32)
        if self.fen is None:
            self.fen = np.zeros(self.K, dtype=x.dtype)
        y = np.empty_like(x)
        L = lib()
        if not cplx:
            L.orc_fir_ff(_p(self.taps), self.K, _p(self.fen), C.byref(self.index), _p(x), _p(y), len(x))
        elif not self.cplx_taps:
            L.orc_fir_cf(_p(self.taps), self.K, _p(self.fen), C.byref(self.index), _p(x), _p(y), len(x))
        else:
            L.orc_fir_cc(_p(self.taps), self.K, _p(self.fen), C.byref(self.index), _p(x), _p(y), len(x))
        return y


def fir(taps, x):
    return Fir(taps).step(x)


class Rii:
    """FiltreRII<float,float> (filtre-rt.cc:177-289), coefficient vectors in powers of z^-1."""

    def __init__(self, numer, denom):
        self.numer = np.ascontiguousarray(numer, dtype=f32)
        self.denom = np.ascontiguousarray(denom, dtype=f32)
        self.wndx = np.zeros(len(self.numer), f32)
        self.wndy = np.zeros(max(len(self.denom) - 1, 1), f32)
        self.ix = C.c_int(0)
        self.iy = C.c_int(0)

    def step(self, x):
        x = np.ascontiguousarray(x, dtype=f32)
        y = np.empty_like(x)
        lib().orc_rii_f(_p(self.numer), len(self.numer), _p(self.denom), len(self.denom) - 1,
                        _p(self.wndx), _p(self.wndy), C.byref(self.ix), C.byref(self.iy),
                        _p(x), _p(y), len(x))
        return y


class RiiC:
    """FiltreRII<cfloat,cfloat> (filtre-rt.cc:177-289,795): complex coefficients, complex data."""

    def __init__(self, numer, denom):
        self.numer = np.ascontiguousarray(numer, dtype=c64)
        self.denom = np.ascontiguousarray(denom, dtype=c64)
        self.wndx = np.zeros(len(self.numer), c64)
        self.wndy = np.zeros(max(len(self.denom) - 1, 1), c64)
        self.ix = C.c_int(0)
        self.iy = C.c_int(0)

    def step(self, x):
        x = np.ascontiguousarray(x, dtype=c64)
        y = np.empty_like(x)
        lib().orc_rii_c(_p(self.numer), len(self.numer), _p(self.denom), len(self.denom) - 1,
                        _p(self.wndx), _p(self.wndy), C.byref(self.ix), C.byref(self.iy),
                        _p(x), _p(y), len(x))
        return y


# --------------------------------------------------------------------------- SOS
class SosChain:
    """ChaineSOIS<T,T,T> (filtre-rt.cc:440-572) from zeros/poles/multipliers."""

    def __init__(self, z, p, mlt_num, mlt_den=1.0, forme=2):
        z = np.ascontiguousarray(z, dtype=c64)
        p = np.ascontiguousarray(p, dtype=c64)
        assert len(z) == len(p)
        self.s = Sos()
        mn, md = complex(mlt_num), complex(mlt_den)
        lib().orc_sos_from_zpk(C.byref(self.s), _p(z), _p(p), len(z),
                               CF(mn.real, mn.imag), CF(md.real, md.imag), forme)
        self.stf = None
        self.stc = None

    @property
    def nsec(self):
        return self.s.nsec

    def coefs(self):
        """[(b0,b1,b2,a1,a2)] * nsec as float32 array, plus (gain, rii1 or None)."""
        a = np.array([[b.b0, b.b1, b.b2, b.a1, b.a2] for b in self.s.sec[: self.s.nsec]], dtype=f32)
        r1 = (self.s.r_b0, self.s.r_b1, self.s.r_a1) if self.s.avec_rii1 else None
        return a.reshape(-1, 5), np.float32(self.s.gain), r1

    def run_f64(self, x):
        """Same recurrence in double (real data, one shot): conditioning yardstick."""
        x = np.ascontiguousarray(x, dtype=f32)
        y = np.empty(len(x), np.float64)
        lib().orc_sos_run_f64(C.byref(self.s), _p(x), _p(y), len(x))
        return y

    def step(self, x):
        x = np.ascontiguousarray(x)
        if np.iscomplexobj(x):
            x = x.astype(c64)
            if self.stc is None:
                self.stc = SosStateC()
                lib().orc_sos_state_init_c(C.byref(self.stc))
            y = np.empty_like(x)
            lib().orc_sos_step_c(C.byref(self.s), C.byref(self.stc), _p(x), _p(y), len(x))
            return y
        x = x.astype(f32)
        if self.stf is None:
            self.stf = SosStateF()
            lib().orc_sos_state_init_f(C.byref(self.stf))
        y = np.empty_like(x)
        lib().orc_sos_step_f(C.byref(self.s), C.byref(self.stf), _p(x), _p(y), len(x))
        return y


def design_butter_lp(n, fcut):
    """design_riia(n,"lp","butt",fcut) -> (zeros, poles, mlt_num, mlt_den)."""
    z = np.empty(n, c64)
    p = np.empty(n, c64)
    mn, md = CF(), CF()
    lib().orc_design_butter_lp(n, fcut, _p(z), _p(p), C.byref(mn), C.byref(md))
    return z, p, complex(mn.re, mn.im), complex(md.re, md.im)


def design_rif_fen(n, typ, fcut):
    """design_rif_fen(n, typ, fcut, "hn"); typ in {"lp","pb","hp"}."""
    h = np.empty(n, f32)
    lib().orc_design_rif_fen_hann(n, {"lp": 0, "pb": 1, "hp": 2}[typ], fcut, _p(h))
    return h


# --------------------------------------------------------------------------- FFT
def fft(x, avant=True):
    x = np.ascontiguousarray(x, dtype=c64)
    y = np.empty_like(x)
    lib().orc_fft(_p(x), _p(y), len(x), 1 if avant else 0)
    return y


def ifft(x):
    return fft(x, False)


def rfft(x):
    x = np.ascontiguousarray(x, dtype=f32)
    y = np.zeros(len(x), c64)
    lib().orc_rfft(_p(x), _p(y), len(x))
    return y


def fftshift(x):
    x = np.ascontiguousarray(x, dtype=c64)
    y = np.empty_like(x)
    lib().orc_fftshift_c(_p(x), _p(y), len(x))
    return y


def twiddles(n):
    r = np.empty(n, c64)
    lib().orc_fft_twiddles(_p(r), n)
    return r


def next_pow2(i):
    return lib().orc_next_pow2(i)


# --------------------------------------------------------------------------- resampler
def itrp_sinc_lut(K=15, nphases=256, fcut=0.4):
    """Column-major Tabf[K x (nphases+1)] -> returned as array [nphases+1, K] (row = phase)."""
    lut = np.empty((nphases + 1, K), f32)
    lib().orc_itrp_sinc_lut(K, nphases, fcut, _p(lut))
    return lut


def reechan_config(ratio):
    nd, nu, post, fc = C.c_int(), C.c_int(), C.c_float(), C.c_float()
    lib().orc_reechan_config(ratio, C.byref(nd), C.byref(nu), C.byref(post), C.byref(fc))
    return nd.value, nu.value, post.value, fc.value


class Resampler:
    """filtre_itrp<T>(ratio, itrp_sinc{K,nphases,fcut,"hn"}) (ra.cc:13-79)."""

    def __init__(self, ratio, K=15, nphases=256, fcut=None, analytic=None):
        ratio = float(np.float32(ratio))
        if analytic is not None:
            # ("lin", 0) = itrp_lineaire, ("lagrange", d) = itrp_lagrange(d)  (itrp.cc:80-133)
            kind, d = analytic
            self.r = Ra()
            lib().orc_ra_init_analytic(C.byref(self.r), ratio, 1 if kind == "lin" else 2, int(d))
            self.ratio = ratio
            return
        if fcut is None:
            fcut = min(np.float32(0.4), np.float32(ratio) / np.float32(2))
        self.lut = itrp_sinc_lut(K, nphases, float(fcut))
        self.r = Ra()
        lib().orc_ra_init(C.byref(self.r), ratio, K, nphases, _p(self.lut))
        self.ratio = ratio

    def step(self, x):
        x = np.ascontiguousarray(x)
        cap = int(np.ceil(self.ratio * len(x)) + 10)
        if np.iscomplexobj(x):
            x = x.astype(c64)
            y = np.empty(cap, c64)
            n = lib().orc_ra_step_c(C.byref(self.r), _p(x), len(x), _p(y))
        else:
            x = x.astype(f32)
            y = np.empty(cap, f32)
            n = lib().orc_ra_step_f(C.byref(self.r), _p(x), len(x), _p(y))
        return y[:n].copy()

    def schedule(self, n, want=True):
        cap = int(np.ceil(self.ratio * n) + 10) if want else 0
        idx = np.empty(cap, np.int64)
        col = np.empty(cap, np.int32)
        nout = lib().orc_ra_schedule(C.byref(self.r), n, _p(idx) if want else None,
                                     _p(col) if want else None, cap)
        return nout, idx[:nout], col[:nout]


# --------------------------------------------------------------------------- polyphase stages
class Decimateur:
    """decimateur<float>(R) (filtre-rt.cc:127-169)."""

    def __init__(self, R):
        self.R, self.cnt = R, C.c_int(0)

    def step(self, x):
        x = np.ascontiguousarray(x, dtype=f32)
        y = np.empty(len(x) // self.R + 2, f32)
        n = lib().orc_decimateur_f(self.R, C.byref(self.cnt), _p(x), len(x), _p(y))
        return y[:n].copy()


class PolyDecim:
    """filtre_rif_decim (kind 0) / filtre_rif_demi_bande (kind 1) (polyphase.cc:54-239)."""

    def __init__(self, coefs, R=2, kind=0):
        self.c = np.ascontiguousarray(coefs, dtype=f32)
        self.R, self.kind = (2 if kind == 1 else R), kind
        self.index, self.cnt, self.fen = C.c_int(0), C.c_int(0), None

    def step(self, x):
        x = np.ascontiguousarray(x)
        cplx = np.iscomplexobj(x)
        x = x.astype(c64 if cplx else f32)
        if self.fen is None:
            self.fen = np.zeros(len(self.c), x.dtype)
        y = np.empty(len(x) // self.R + 2, x.dtype)
        fn = lib().orc_polydecim_c if cplx else lib().orc_polydecim_f
        n = fn(self.kind, _p(self.c), len(self.c), self.R, _p(self.fen), C.byref(self.index), C.byref(self.cnt),
               _p(x), len(x), _p(y))
        return y[:n].copy()


class PolyUps:
    """filtre_rif_ups<float,T>(c, R) (polyphase.cc:246-341)."""

    def __init__(self, coefs, R):
        c = np.ascontiguousarray(coefs, dtype=f32)
        pad = np.zeros(len(c) + R, f32)
        self.K = lib().orc_ups_prepare(_p(c), len(c), R, _p(pad))
        self.c, self.R = pad[: self.K].copy(), R
        self.index, self.fen = C.c_int(0), None

    def step(self, x):
        x = np.ascontiguousarray(x)
        cplx = np.iscomplexobj(x)
        x = x.astype(c64 if cplx else f32)
        if self.fen is None:
            self.fen = np.zeros(self.K // self.R, x.dtype)
        y = np.empty(len(x) * self.R, x.dtype)
        fn = lib().orc_ups_c if cplx else lib().orc_ups_f
        n = fn(_p(self.c), self.K, self.R, _p(self.fen), C.byref(self.index), _p(x), len(x), _p(y))
        return y[:n]
