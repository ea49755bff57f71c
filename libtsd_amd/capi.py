"""ctypes binding of include/tsdgpu.h.  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
F32, C64 = 0, 1
FIR_AUTO, FIR_DIRECT, FIR_OVERLAP_SAVE = 0, 1, 2
_LIB = None


class TsdGpuError(RuntimeError):
    pass


def lib_path():
    # TSDGPU_LIB: developer switch -- load an experimental build of the same C ABI (scripts/build_variant.sh)
    return os.environ.get("TSDGPU_LIB") or os.path.join(_HERE, "lib", "libtsdgpu.so")


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise TsdGpuError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C libtsd_amd/csrc). There is no CPU fallback.")
        # PyTorch wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7, requested by
        # torch as "libamdhip64.so").  If ours (/opt/rocm) were loaded first the process
        # would end up with two HIP runtimes and torch would see no GPU; importing torch
        # first makes our DT_NEEDED libamdhip64.so.7 resolve to the copy already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(p)
        _declare(L)
        _LIB = L
    return _LIB


def _declare(L):
    vp, i32, i64, fl = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.tsdgpu_last_error.restype = C.c_char_p
    L.tsdgpu_version.restype = C.c_char_p
    L.tsdgpu_device_count.restype = i32
    L.tsdgpu_current_device.restype = i32
    L.tsdgpu_fir_create.argtypes = [C.POINTER(vp), i32, i32, vp, i32, i32]
    L.tsdgpu_fir_step.argtypes = [vp, vp, vp, i64, vp]
    L.tsdgpu_fir_reset.argtypes = [vp]
    L.tsdgpu_fir_step_after.argtypes = [vp, vp, vp, i64, i64, vp]
    L.tsdgpu_fir_lead.argtypes = [vp]
    L.tsdgpu_fir_lead.restype = i32
    L.tsdgpu_fir_reset_on.argtypes = [vp, vp]
    L.tsdgpu_sos_reset_on.argtypes = [vp, vp]
    L.tsdgpu_fir_get_history.argtypes = [vp, vp, vp]
    L.tsdgpu_fir_set_history.argtypes = [vp, vp, vp]
    L.tsdgpu_fir_method_used.argtypes = [vp]
    L.tsdgpu_fir_destroy.argtypes = [vp]
    L.tsdgpu_sos_create.argtypes = [C.POINTER(vp), i32, vp, i32, fl, vp, i32]
    L.tsdgpu_sos_state_floats.restype = i32
    L.tsdgpu_sos_get_state.argtypes = [vp, vp, vp]
    L.tsdgpu_sos_set_state.argtypes = [vp, vp, vp]
    L.tsdgpu_sos_propagate_state.argtypes = [vp, i64, vp, vp, vp]
    L.tsdgpu_sos_step.argtypes = [vp, vp, vp, i64, vp]
    L.tsdgpu_sos_step_skip.argtypes = [vp, vp, vp, i64, i64, vp]
    L.tsdgpu_sos_reset.argtypes = [vp]
    L.tsdgpu_sos_halo.argtypes = [vp]
    L.tsdgpu_sos_halo.restype = i64
    L.tsdgpu_sos_destroy.argtypes = [vp]
    L.tsdgpu_resampler_create.argtypes = [C.POINTER(vp), i32, fl, vp, i32, i32]
    L.tsdgpu_resampler_out_count.argtypes = [vp, i64]
    L.tsdgpu_resampler_out_count.restype = i64
    L.tsdgpu_resampler_create_analytic.argtypes = [C.POINTER(vp), i32, fl, i32, i32]
    L.tsdgpu_resampler_step.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), vp]
    L.tsdgpu_resampler_reset.argtypes = [vp]
    L.tsdgpu_resampler_seek.argtypes = [vp, i64, vp, vp]
    L.tsdgpu_resampler_out_offset.argtypes = [vp]
    L.tsdgpu_resampler_out_offset.restype = i64
    L.tsdgpu_resampler_destroy.argtypes = [vp]
    L.tsdgpu_polyfir_create.argtypes = [C.POINTER(vp), i32, i32, vp, i32, i32]
    L.tsdgpu_polyfir_out_count.argtypes = [vp, i64]
    L.tsdgpu_polyfir_out_count.restype = i64
    L.tsdgpu_polyfir_step.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), vp]
    L.tsdgpu_polyfir_reset.argtypes = [vp]
    L.tsdgpu_polyfir_destroy.argtypes = [vp]
    L.tsdgpu_rii_create.argtypes = [C.POINTER(vp), i32, vp, i32, vp, i32]
    L.tsdgpu_rii_create2.argtypes = [C.POINTER(vp), i32, i32, vp, i32, vp, i32]
    L.tsdgpu_rii_path.argtypes = [vp]
    L.tsdgpu_fir_sharded_create.argtypes = [C.POINTER(vp), i32, i32, vp, i32, i32, i32, vp]
    L.tsdgpu_sos_sharded_create.argtypes = [C.POINTER(vp), i32, vp, i32, fl, vp, i32, i32, vp]
    L.tsdgpu_resampler_sharded_create.argtypes = [C.POINTER(vp), i32, fl, vp, i32, i32, i32, vp]
    L.tsdgpu_sharded_count.argtypes = [vp]
    L.tsdgpu_sharded_halo.argtypes = [vp]
    L.tsdgpu_sharded_halo.restype = i64
    L.tsdgpu_sharded_device.argtypes = [vp, i32]
    L.tsdgpu_sharded_bounds.argtypes = [vp, i64, i32, C.POINTER(i64), C.POINTER(i64)]
    L.tsdgpu_sharded_bounds.restype = None
    L.tsdgpu_sharded_out_count.argtypes = [vp, i64]
    L.tsdgpu_sharded_out_count.restype = i64
    L.tsdgpu_sharded_step_host.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64)]
    L.tsdgpu_sharded_step_parts.argtypes = [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
    L.tsdgpu_sharded_step_parts_on.argtypes = [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(vp)]
    L.tsdgpu_sharded_reset.argtypes = [vp]
    L.tsdgpu_sharded_destroy.argtypes = [vp]
    L.tsdgpu_xcorr.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.tsdgpu_vec_op.argtypes = [i32, i32, vp, vp, vp, C.c_float, C.c_float, C.c_int64, vp]
    L.tsdgpu_vec_op.restype = i32
    L.tsdgpu_vec_reduce.argtypes = [i32, vp, C.c_int64, vp, vp, vp, vp]
    L.tsdgpu_vec_reduce.restype = i32
    L.tsdgpu_delay_estimate.argtypes = [vp, vp, i32, C.POINTER(fl), C.POINTER(fl), vp]
    L.tsdgpu_detector_create.argtypes = [C.POINTER(vp), vp, i32, i32, i32, fl]
    L.tsdgpu_detector_delay.argtypes = [vp]
    L.tsdgpu_detector_fft_size.argtypes = [vp]
    L.tsdgpu_detector_step.argtypes = [vp, vp, i64, vp, vp, i32, C.POINTER(i32), vp]
    L.tsdgpu_detector_destroy.argtypes = [vp]
    L.tsdgpu_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.tsdgpu_free.argtypes = [vp]
    L.tsdgpu_malloc_host.argtypes = [C.POINTER(vp), C.c_size_t]
    L.tsdgpu_free_host.argtypes = [vp]
    L.tsdgpu_memcpy.argtypes = [vp, vp, C.c_size_t, vp]
    L.tsdgpu_synchronize.argtypes = [vp]
    L.tsdgpu_is_device_pointer.argtypes = [vp]
    L.tsdgpu_rii_step.argtypes = [vp, vp, vp, i64, vp]
    L.tsdgpu_rii_destroy.argtypes = [vp]
    L.tsdgpu_fft_create.argtypes = [C.POINTER(vp), i32, i32]
    L.tsdgpu_fft_step.argtypes = [vp, vp, vp, i32, i32, vp]
    L.tsdgpu_fft_size.argtypes = [vp]
    L.tsdgpu_fft_destroy.argtypes = [vp]
    L.tsdgpu_rfft_create.argtypes = [C.POINTER(vp), i32]
    L.tsdgpu_rfft_step.argtypes = [vp, vp, vp, i32, vp]
    L.tsdgpu_rfft_destroy.argtypes = [vp]
    L.tsdgpu_fftshift.argtypes = [vp, vp, i32, i32, vp]
    L.tsdgpu_ola_create.argtypes = [C.POINTER(vp), i32, i32, vp]
    L.tsdgpu_ola_fft_size.argtypes = [vp]
    L.tsdgpu_ola_block_len.argtypes = [vp]
    L.tsdgpu_ola_set_response.argtypes = [vp, vp]
    L.tsdgpu_ola_max_out.argtypes = [vp, C.c_int64]
    L.tsdgpu_ola_max_out.restype = C.c_int64
    L.tsdgpu_ola_step.argtypes = [vp, vp, C.c_int64, vp, C.POINTER(C.c_int64), vp]
    L.tsdgpu_ola_analyse.argtypes = [vp, vp, C.c_int64, C.POINTER(vp), C.POINTER(i32), vp]
    L.tsdgpu_ola_synthese.argtypes = [vp, vp, C.POINTER(C.c_int64), vp]
    L.tsdgpu_ola_apply_response.argtypes = [vp, vp]
    L.tsdgpu_ola_read_spectra.argtypes = [vp, vp, vp]
    L.tsdgpu_ola_write_spectra.argtypes = [vp, vp, vp]
    L.tsdgpu_ola_destroy.argtypes = [vp]
    L.tsdgpu_welch.argtypes = [vp, C.c_int64, i32, vp, vp, C.POINTER(C.c_int64), vp]
    L.tsdgpu_spectrum_create.argtypes = [C.POINTER(vp), i32, i32, i32, vp, i32, i32, vp]
    L.tsdgpu_spectrum_bins.argtypes = [vp]
    L.tsdgpu_spectrum_pending.argtypes = [vp]
    L.tsdgpu_spectrum_step.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(C.c_int64), vp]
    L.tsdgpu_spectrum_reset.argtypes = [vp, vp]
    L.tsdgpu_spectrum_destroy.argtypes = [vp]


def device_count():
    return lib().tsdgpu_device_count()


def _check(rc):
    if rc != 0:
        raise TsdGpuError(f"tsdgpu status {rc}: {lib().tsdgpu_last_error().decode()}")


def _ptr(a):
    """Address of a numpy array (host) or a torch tensor (host or device).  The C ABI reads packed
    float32 / complex64 samples: anything else (float64, integers, a strided view such as x[::2])
    is refused here instead of being reinterpreted."""
    _dtype_code(a)
    if isinstance(a, np.ndarray):
        if not a.flags.c_contiguous:
            raise TsdGpuError("non-contiguous numpy array: pass np.ascontiguousarray(x)")
        return a.ctypes.data
    if not a.is_contiguous():
        raise TsdGpuError("non-contiguous tensor: pass x.contiguous()")
    return a.data_ptr()


def _dtype_code(a):
    if isinstance(a, np.ndarray):
        if a.dtype == np.float32:
            return F32
        if a.dtype == np.complex64:
            return C64
        raise TsdGpuError(f"dtype {a.dtype} is not served: the C ABI takes float32 or complex64 (convert with astype)")
    import torch
    if a.dtype == torch.float32:
        return F32
    if a.dtype == torch.complex64:
        return C64
    raise TsdGpuError(f"dtype {a.dtype} is not served: the C ABI takes float32 or complex64")


def _stream_of(a, stream):
    if stream is not None:
        return stream
    if not isinstance(a, np.ndarray) and a.is_cuda:
        import torch
        return torch.cuda.current_stream(a.device).cuda_stream
    return None


class Fir:
    """filtre_rif<Tc,T>(h) (filtre-rt.cc:171-175): stateful; step(x) filters one chunk."""

    def __init__(self, taps, data_type, method=FIR_AUTO):
        taps = np.ascontiguousarray(taps)
        tt = C64 if np.iscomplexobj(taps) else F32
        taps = taps.astype(np.complex64 if tt == C64 else np.float32)
        self.K = len(taps)
        self.data_type = data_type
        self._h = C.c_void_p()
        _check(lib().tsdgpu_fir_create(C.byref(self._h), data_type, tt, taps.ctypes.data, len(taps), method))

    @property
    def method(self):
        return lib().tsdgpu_fir_method_used(self._h)

    def step(self, x, y=None, stream=None):
        """x: numpy array (host path) or torch tensor (device path), float32/complex64."""
        assert _dtype_code(x) == self.data_type, "input dtype does not match the filter's data type"
        if y is None:
            y = np.empty_like(x) if isinstance(x, np.ndarray) else x.new_empty(x.shape)
        _check(lib().tsdgpu_fir_step(self._h, _ptr(x), _ptr(y), x.shape[0], _stream_of(x, stream)))
        return y

    def reset(self):
        _check(lib().tsdgpu_fir_reset(self._h))

    def reset_on(self, like=None, stream=None):
        """reset ordered on a stream (default: the current torch stream of `like`'s device): no host wait"""
        _check(lib().tsdgpu_fir_reset_on(self._h, _stream_of(like, stream) if like is not None else stream))

    def get_history(self, dst, stream=None):
        _check(lib().tsdgpu_fir_get_history(self._h, _ptr(dst), _stream_of(dst, stream)))
        return dst

    def set_history(self, src, stream=None):
        _check(lib().tsdgpu_fir_set_history(self._h, _ptr(src), _stream_of(src, stream)))

    @property
    def lead(self):
        """History length of the handle (>= K - 1): the smallest `lead` of step_after."""
        return int(lib().tsdgpu_fir_lead(self._h))

    def step_after(self, x, y, lead, stream=None):
        """Filters x[lead:] into y[lead:] with the delay line taken from x[lead - self.lead : lead] itself (device tensors, x is not y):
        set_history + step of the halo-free interior of a chunk without the copy."""
        assert _dtype_code(x) == self.data_type
        _check(lib().tsdgpu_fir_step_after(self._h, _ptr(x), _ptr(y), int(x.shape[0]), int(lead), _stream_of(x, stream)))
        return y

    def close(self):
        if self._h:
            lib().tsdgpu_fir_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Fft:
    """FFTPlan (fourier.hpp:19-32) of one size: step(x, forward) on [batch, n] complex64 data."""

    def __init__(self, n, batch_hint=1):
        self.n = int(n)
        self._h = C.c_void_p()
        _check(lib().tsdgpu_fft_create(C.byref(self._h), self.n, batch_hint))

    def step(self, x, forward=True, y=None, stream=None):
        assert _dtype_code(x) == C64
        total = int(np.prod(x.shape))
        assert total % self.n == 0
        if y is None:
            y = np.empty_like(x) if isinstance(x, np.ndarray) else x.new_empty(x.shape)
        _check(lib().tsdgpu_fft_step(self._h, _ptr(x), _ptr(y), total // self.n, 1 if forward else 0,
                                     _stream_of(x, stream)))
        return y

    def close(self):
        if self._h:
            lib().tsdgpu_fft_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fft(x, forward=True):
    """One-shot fft()/ifft() (fourier.hpp:163-205): plan per call, like the reference."""
    p = Fft(x.shape[-1])
    try:
        return p.step(x, forward)
    finally:
        p.close()


class Rfft:
    """RTFRPlan (fourier.cc:280-355): step(x) on [batch, n] float32 data -> [batch, n] complex64."""

    def __init__(self, n):
        self.n = int(n)
        self._h = C.c_void_p()
        _check(lib().tsdgpu_rfft_create(C.byref(self._h), self.n))

    def step(self, x, y=None, stream=None):
        assert _dtype_code(x) == F32
        total = int(np.prod(x.shape))
        assert total % self.n == 0
        if y is None:
            y = np.empty(x.shape, np.complex64) if isinstance(x, np.ndarray) else x.new_empty(x.shape, dtype=__import__("torch").complex64)
        _check(lib().tsdgpu_rfft_step(self._h, _ptr(x), _ptr(y), total // self.n, _stream_of(x, stream)))
        return y

    def close(self):
        if self._h:
            lib().tsdgpu_rfft_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ola:
    """OLA<cfloat> engine behind filtre_fft() (fourier.cc:737-940).  step(x) uses the built-in
    product with `response` (set_response); analyse()/synthese() bracket a caller-side edit of
    the spectra, which stay on the device (a torch view when the input is a device tensor)."""

    def __init__(self, block_len=0, min_zeros=0, window=None):
        self._h = C.c_void_p()
        w = None if window is None else np.ascontiguousarray(window, np.float32)
        _check(lib().tsdgpu_ola_create(C.byref(self._h), int(block_len), int(min_zeros), None if w is None else w.ctypes.data))
        self.N = lib().tsdgpu_ola_fft_size(self._h)
        self.Ne = lib().tsdgpu_ola_block_len(self._h)

    def set_response(self, H):
        if H is None:
            _check(lib().tsdgpu_ola_set_response(self._h, None))
            return
        if isinstance(H, np.ndarray):
            H = np.ascontiguousarray(H, np.complex64)
        assert int(np.prod(H.shape)) == self.N
        _check(lib().tsdgpu_ola_set_response(self._h, _ptr(H)))

    def _out(self, x, n):
        return np.empty(n, np.complex64) if isinstance(x, np.ndarray) else x.new_empty(n)

    def step(self, x, y=None, stream=None):
        assert _dtype_code(x) == C64
        n = int(x.shape[0])
        if y is None:
            y = self._out(x, max(1, lib().tsdgpu_ola_max_out(self._h, n)))
        nout = C.c_int64(0)
        _check(lib().tsdgpu_ola_step(self._h, _ptr(x) if n else None, n, _ptr(y), C.byref(nout), _stream_of(x, stream)))
        return y[:nout.value]

    def analyse(self, x, stream=None):
        """-> (device address of the spectra [frames][N] complex64, frames)."""
        assert _dtype_code(x) == C64
        sp, nf = C.c_void_p(), C.c_int(0)
        n = int(x.shape[0])
        self._last = x
        _check(lib().tsdgpu_ola_analyse(self._h, _ptr(x) if n else None, n, C.byref(sp), C.byref(nf), _stream_of(x, stream)))
        return sp.value, nf.value

    def synthese(self, stream=None):
        x = self._last
        # the pending block count is not visible here: size the output for the worst case
        y = self._out(x, max(1, self._pending_out(x)))
        nout = C.c_int64(0)
        _check(lib().tsdgpu_ola_synthese(self._h, _ptr(y), C.byref(nout), _stream_of(x, stream)))
        return y[:nout.value]

    def _pending_out(self, x):
        return (int(x.shape[0]) // self.Ne + 1) * self.Ne

    def close(self):
        if self._h:
            lib().tsdgpu_ola_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def welch(x, N, window):
    """psd_welch's sum of periodograms (freqestim.cc:7-20): -> (S[N] float32 linear, fftshift-ed; segments)."""
    assert _dtype_code(x) == C64
    w = np.ascontiguousarray(window, np.float32)
    assert w.shape[0] == N
    S = np.empty(N, np.float32)
    nseg = C.c_int64(0)
    _check(lib().tsdgpu_welch(_ptr(x), int(x.shape[0]), int(N), w.ctypes.data, S.ctypes.data, C.byref(nseg), _stream_of(x, None)))
    return S, nseg.value


class Spectrum:
    """rt_spectrum (fourier.cc:1162-1342): blocks of BS = nsubs x Nf samples -> one spectrum in dB per nmeans blocks.
    window: Nf values already normalised to energy Nf; sweep = (step, mask[Nf] or None) or None."""

    def __init__(self, BS, nsubs, nmeans, window, sweep=None):
        w = np.ascontiguousarray(window, np.float32)
        self.BS, self.nsubs, self.nmeans = BS, nsubs, nmeans
        self._h = C.c_void_p()
        if sweep is None:
            _check(lib().tsdgpu_spectrum_create(C.byref(self._h), BS, nsubs, nmeans, w.ctypes.data, 0, 0, None))
        else:
            step, mask = sweep
            m = None if mask is None else np.ascontiguousarray(mask, np.float32)
            _check(lib().tsdgpu_spectrum_create(C.byref(self._h), BS, nsubs, nmeans, w.ctypes.data, 1, int(step), None if m is None else m.ctypes.data))
        self.Ns = lib().tsdgpu_spectrum_bins(self._h)

    @property
    def pending(self):
        return lib().tsdgpu_spectrum_pending(self._h)

    def step(self, x, y=None, stream=None):
        """x: whole blocks (numpy / torch complex64) -> [spectra completed by this call, Ns] float32"""
        assert _dtype_code(x) == C64 and x.shape[0] % self.BS == 0
        B = x.shape[0] // self.BS
        nout = (self.pending + B) // self.nmeans
        if y is None:
            y = np.empty((nout, self.Ns), np.float32) if isinstance(x, np.ndarray) else x.new_empty((nout, self.Ns), dtype=_torch_f32())
        got = C.c_int64(0)
        _check(lib().tsdgpu_spectrum_step(self._h, _ptr(x) if B else None, B, _ptr(y) if nout else None, nout, C.byref(got), _stream_of(x, stream)))
        assert got.value == nout
        return y

    def reset(self):
        _check(lib().tsdgpu_spectrum_reset(self._h, None))

    def close(self):
        if self._h:
            lib().tsdgpu_spectrum_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch_f32():
    import torch
    return torch.float32


def rfft(x):
    """One-shot rfft() (fourier.hpp:116-122)."""
    p = Rfft(x.shape[-1])
    try:
        return p.step(x)
    finally:
        p.close()


def fftshift(x, stream=None):
    y = np.empty_like(x) if isinstance(x, np.ndarray) else x.new_empty(x.shape)
    _check(lib().tsdgpu_fftshift(_ptr(x), _ptr(y), x.shape[0], _dtype_code(x), _stream_of(x, stream)))
    return y


class Sos:
    """filtre_sois<T> (filtre-rt.cc:574-602) from already-paired sections:
    coefs [nsec,5] = (b0,b1,b2,a1,a2) normalised by a0; gain; optional first-order (b0,b1,a1)."""

    def __init__(self, coefs, gain, data_type, rii1=None, forme=2):
        coefs = np.ascontiguousarray(coefs, dtype=np.float32).reshape(-1, 5)
        self.data_type = data_type
        self._h = C.c_void_p()
        r1 = None if rii1 is None else np.ascontiguousarray(rii1, dtype=np.float32)
        _check(lib().tsdgpu_sos_create(C.byref(self._h), data_type, coefs.ctypes.data, coefs.shape[0],
                                       float(gain), None if r1 is None else r1.ctypes.data, forme))

    @property
    def halo(self):
        return lib().tsdgpu_sos_halo(self._h)

    def step(self, x, y=None, stream=None):
        assert _dtype_code(x) == self.data_type
        if y is None:
            y = np.empty_like(x) if isinstance(x, np.ndarray) else x.new_empty(x.shape)
        _check(lib().tsdgpu_sos_step(self._h, _ptr(x), _ptr(y), x.shape[0], _stream_of(x, stream)))
        return y

    def step_skip(self, x, y, skip, stream=None):
        """step() whose first `skip` outputs are not stored (y[:skip] untouched): device tensors, x is not y."""
        assert _dtype_code(x) == self.data_type
        _check(lib().tsdgpu_sos_step_skip(self._h, _ptr(x), _ptr(y), x.shape[0], int(skip), _stream_of(x, stream)))
        return y

    def reset(self):
        _check(lib().tsdgpu_sos_reset(self._h))

    def reset_on(self, like=None, stream=None):
        _check(lib().tsdgpu_sos_reset_on(self._h, _stream_of(like, stream) if like is not None else stream))

    # the carried memories as a host vector (see tsdgpu_sos_get_state): what the exact sharding of sharding.py exchanges
    def get_state(self, stream=None):
        st = np.zeros(lib().tsdgpu_sos_state_floats(), np.float32)
        _check(lib().tsdgpu_sos_get_state(self._h, st.ctypes.data, stream))
        return st

    def set_state(self, state, stream=None):
        st = np.ascontiguousarray(state, dtype=np.float32)
        assert st.size == lib().tsdgpu_sos_state_floats()
        _check(lib().tsdgpu_sos_set_state(self._h, st.ctypes.data, stream))

    def propagate_state(self, n_samples, state, end_state=None):
        a = np.ascontiguousarray(state, dtype=np.float32)
        e = None if end_state is None else np.ascontiguousarray(end_state, dtype=np.float32)
        out = np.zeros_like(a)
        _check(lib().tsdgpu_sos_propagate_state(self._h, int(n_samples), a.ctypes.data, None if e is None else e.ctypes.data, out.ctypes.data))
        return out

    def close(self):
        if self._h:
            lib().tsdgpu_sos_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def itrp_sinc_lut(K=15, nphases=256, fcut=0.4):
    """Host-side design helper: the LUT of itrp_sinc{K, nphases, fcut, "hn"} (itrp.cc:24-54),
    float32 arithmetic, returned phase-major [nphases+1, K]."""
    f = np.float32
    i = np.arange(K)
    step = (float((K - 1) // 2) - float(-(K // 2))) / (K - 1) if K > 1 else 0.0
    ls = np.array([f(-(K // 2))] + [f(-(K // 2) + step * j) for j in range(1, K)], dtype=f)   # linspace (tsd.hpp:916-931)
    lut = np.empty((nphases + 1, K), f)
    pi_f = f(np.pi)
    for j in range(nphases + 1):
        tau = f((1.0 * j) / nphases)
        t = (i - K // 2).astype(f) - tau
        a = pi_f * f(2 * f(fcut)) * t                                    # sinc(T, f) (divers.cc:6-12)
        with np.errstate(invalid="ignore", divide="ignore"):
            h = np.where(np.abs(a) < f(1e-7), f(2 * f(fcut)), np.sin(a, dtype=f) / (pi_f * t))
        w = f(0.5) + f(2) * f(0.25) * np.cos((ls - tau) * f(2 * np.pi / K), dtype=f)
        lut[j] = (h.astype(f) * w.astype(f)).astype(f)
    return lut


class Resampler:
    """filtre_itrp<T>(ratio, itrp_sinc{K,nphases,fcut,"hn"}) (ra.cc:13-79,185-188).  With the
    defaults this is the interpolator filtre_reechan configures for a ratio in [0.5,2)."""

    def __init__(self, ratio, data_type, K=15, nphases=256, fcut=None, lut=None, analytic=None):
        ratio = float(np.float32(ratio))
        if analytic is not None:          # ("lin", 0) = itrp_lineaire, ("lagrange", d) = itrp_lagrange(d)
            kind, d = analytic
            self.ratio, self.K, self.data_type = ratio, (2 if kind == "lin" else d + 1), data_type
            self._h = C.c_void_p()
            _check(lib().tsdgpu_resampler_create_analytic(C.byref(self._h), data_type, ratio, 1 if kind == "lin" else 2, int(d)))
            return
        if lut is None:
            if fcut is None:
                fcut = float(min(np.float32(0.4), np.float32(ratio) / np.float32(2)))      # ra.cc:149
            lut = itrp_sinc_lut(K, nphases, fcut)
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        assert lut.shape == (nphases + 1, K)
        self.ratio, self.K, self.data_type = ratio, K, data_type
        self._h = C.c_void_p()
        _check(lib().tsdgpu_resampler_create(C.byref(self._h), data_type, ratio, lut.ctypes.data, K, nphases))

    def out_count(self, n):
        return lib().tsdgpu_resampler_out_count(self._h, n)

    @property
    def out_offset(self):
        return lib().tsdgpu_resampler_out_offset(self._h)

    def step(self, x, y=None, stream=None):
        assert _dtype_code(x) == self.data_type
        n = x.shape[0]
        nout = self.out_count(n)
        if y is None:
            y = np.empty(nout, x.dtype) if isinstance(x, np.ndarray) else x.new_empty(nout)
        got = C.c_int64(0)
        _check(lib().tsdgpu_resampler_step(self._h, _ptr(x), n, _ptr(y), y.shape[0], C.byref(got),
                                           _stream_of(x, stream)))
        return y[: got.value]

    def reset(self):
        _check(lib().tsdgpu_resampler_reset(self._h))

    def seek(self, pos, hist=None, stream=None):
        _check(lib().tsdgpu_resampler_seek(self._h, pos, None if hist is None else _ptr(hist),
                                           None if hist is None else _stream_of(hist, stream)))

    def close(self):
        if self._h:
            lib().tsdgpu_resampler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


POLY_DECIM, POLY_HALFBAND, POLY_UPS, POLY_PICK = 0, 1, 2, 3


class PolyFir:
    """filtre_rif_decim / filtre_rif_demi_bande / filtre_rif_ups / decimateur
    (polyphase.cc:54-341, filtre-rt.cc:127-169)."""

    def __init__(self, kind, data_type, taps=None, R=2):
        self.data_type = data_type
        self._h = C.c_void_p()
        t = None if taps is None else np.ascontiguousarray(taps, dtype=np.float32)
        _check(lib().tsdgpu_polyfir_create(C.byref(self._h), kind, data_type, None if t is None else t.ctypes.data,
                                           0 if t is None else len(t), R))

    def step(self, x, stream=None):
        assert _dtype_code(x) == self.data_type
        n = x.shape[0]
        nout = lib().tsdgpu_polyfir_out_count(self._h, n)
        y = np.empty(nout, x.dtype) if isinstance(x, np.ndarray) else x.new_empty(nout)
        got = C.c_int64(0)
        _check(lib().tsdgpu_polyfir_step(self._h, _ptr(x), n, _ptr(y), nout, C.byref(got), _stream_of(x, stream)))
        return y[: got.value]

    def reset(self):
        _check(lib().tsdgpu_polyfir_reset(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().tsdgpu_polyfir_destroy(self._h)
        except Exception:
            pass


class Rii:
    """filtre_rii<Tc,T> (filtre-rt.cc:177-289): numer / denom in powers of z^-1, real or complex.
    path: 0 = block-parallel sections, 1 = FIR kernel + block-parallel sections, 2 = literal recursion."""

    def __init__(self, numer, denom, data_type):
        cplx = np.iscomplexobj(numer) or np.iscomplexobj(denom)
        ct = np.complex64 if cplx else np.float32
        nu = np.ascontiguousarray(numer, dtype=ct)
        de = np.ascontiguousarray(denom, dtype=ct)
        self.data_type = data_type
        self._h = C.c_void_p()
        _check(lib().tsdgpu_rii_create2(C.byref(self._h), data_type, C64 if cplx else F32, nu.ctypes.data, len(nu),
                                        de.ctypes.data, len(de)))

    @property
    def path(self):
        return lib().tsdgpu_rii_path(self._h)

    def step(self, x, y=None, stream=None):
        assert _dtype_code(x) == self.data_type
        if y is None:
            y = np.empty_like(x) if isinstance(x, np.ndarray) else x.new_empty(x.shape)
        _check(lib().tsdgpu_rii_step(self._h, _ptr(x), _ptr(y), x.shape[0], _stream_of(x, stream)))
        return y

    def __del__(self):
        try:
            if self._h:
                lib().tsdgpu_rii_destroy(self._h)
        except Exception:
            pass


class Sharded:
    """One process, several GPUs (tsdgpu_sharded_*): contiguous chunks + the operator's small halo.
    kind = "fir" (taps, method), "sos" (coefs [nsec,5], gain, rii1, forme) or "resampler" (ratio, lut).
    devices: one ordinal per shard (several shards may share a device)."""

    def __init__(self, kind, data_type, nshards, devices=None, **kw):
        self.kind, self.data_type, self.nshards = kind, data_type, nshards
        self._h = C.c_void_p()
        dv = None if devices is None else (C.c_int * nshards)(*devices)
        if kind == "fir":
            taps = np.ascontiguousarray(kw["taps"])
            tt = C64 if np.iscomplexobj(taps) else F32
            taps = taps.astype(np.complex64 if tt == C64 else np.float32)
            _check(lib().tsdgpu_fir_sharded_create(C.byref(self._h), data_type, tt, taps.ctypes.data, len(taps),
                                                   kw.get("method", FIR_AUTO), nshards, dv))
        elif kind == "sos":
            co = np.ascontiguousarray(kw["coefs"], dtype=np.float32).reshape(-1, 5)
            r1 = kw.get("rii1")
            r1 = None if r1 is None else np.ascontiguousarray(r1, dtype=np.float32)
            _check(lib().tsdgpu_sos_sharded_create(C.byref(self._h), data_type, co.ctypes.data, co.shape[0], float(kw.get("gain", 1.0)),
                                                   None if r1 is None else r1.ctypes.data, kw.get("forme", 2), nshards, dv))
        elif kind == "resampler":
            ratio = float(np.float32(kw["ratio"]))
            K, nph = kw.get("K", 15), kw.get("nphases", 256)
            lut = kw.get("lut")
            if lut is None:
                lut = itrp_sinc_lut(K, nph, float(min(np.float32(0.4), np.float32(ratio) / np.float32(2))))
            lut = np.ascontiguousarray(lut, dtype=np.float32)
            _check(lib().tsdgpu_resampler_sharded_create(C.byref(self._h), data_type, ratio, lut.ctypes.data, K, nph, nshards, dv))
        else:
            raise ValueError(kind)

    @property
    def halo(self):
        return lib().tsdgpu_sharded_halo(self._h)

    def bounds(self, n, g):
        lo, hi = C.c_int64(0), C.c_int64(0)
        lib().tsdgpu_sharded_bounds(self._h, n, g, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def out_count(self, n):
        return lib().tsdgpu_sharded_out_count(self._h, n)

    def step_host(self, x, y=None):
        """x: one host numpy vector -> y host numpy vector."""
        assert isinstance(x, np.ndarray) and _dtype_code(x) == self.data_type
        n = x.shape[0]
        cap = self.out_count(n)
        if y is None:
            y = np.empty(cap, x.dtype)
        got = C.c_int64(0)
        _check(lib().tsdgpu_sharded_step_host(self._h, _ptr(x) if n else None, n, _ptr(y) if cap else None, y.shape[0], C.byref(got)))
        return y[: got.value]

    def step_parts(self, xs, ys=None, capacities=None, device_sync=False):
        """xs: list of torch tensors, xs[g] resident on the device of shard g -> list of output tensors.  The shards wait
        for what torch's current streams hold (events); device_sync=True: the plain entry point, which waits for the devices."""
        N = self.nshards
        assert len(xs) == N
        cnt = (C.c_int64 * N)(*[int(t.shape[0]) for t in xs])
        if ys is None:
            if self.kind == "resampler":
                ys = [t.new_empty(int(c)) for t, c in zip(xs, capacities)]
            else:
                ys = [t.new_empty(t.shape) for t in xs]
        caps = (C.c_int64 * N)(*[int(t.shape[0]) for t in ys])
        got = (C.c_int64 * N)()
        xp = (C.c_void_p * N)(*[_ptr(t) if t.shape[0] else None for t in xs])
        yp = (C.c_void_p * N)(*[_ptr(t) if t.shape[0] else None for t in ys])
        if device_sync:
            _check(lib().tsdgpu_sharded_step_parts(self._h, xp, cnt, yp, caps, got))
        else:
            # ordered by events on the streams that produced the parts: torch's current stream of each part's device
            import torch
            sp = (C.c_void_p * N)(*[torch.cuda.current_stream(t.device).cuda_stream for t in xs])
            _check(lib().tsdgpu_sharded_step_parts_on(self._h, xp, cnt, yp, caps, got, sp))
        return [t[: got[g]] for g, t in enumerate(ys)]

    def reset(self):
        _check(lib().tsdgpu_sharded_reset(self._h))

    def close(self):
        if self._h:
            lib().tsdgpu_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def xcorr(x, y=None, m=-1, unbiased=True):
    """xcorr / xcorrb (fourier.cc:534-597) on the device: -> the 2m-1 complex lags -(m-1) .. (m-1)."""
    assert _dtype_code(x) == C64 and (y is None or _dtype_code(y) == C64)
    n = int(x.shape[0])
    if m < 0:
        m = n
    out = np.empty(2 * m - 1, np.complex64) if isinstance(x, np.ndarray) else x.new_empty(2 * m - 1)
    _check(lib().tsdgpu_xcorr(_ptr(x), None if y is None else _ptr(y), n, m, 1 if unbiased else 0, _ptr(out), _stream_of(x, None)))
    return out


VEC_OPS = {"reverse": 0, "scale": 1, "div": 2, "add": 3, "sub": 4, "mul": 5, "neg": 6, "abs": 7, "abs2": 8, "real": 9, "imag": 10,
           "to_complex": 11, "conj": 12}


def vec_op(op, a, b=None, scalar=0.0, out=None):
    """Element-wise arithmetic on RESIDENT torch tensors (tsdgpu_vec_op): reverse / scale / div / add / sub / mul / neg keep
    the element type, abs / abs2 / real / imag give float32, to_complex gives complex64."""
    import torch
    dt = _dtype_code(a)
    code = VEC_OPS[op]
    if out is None:
        odt = torch.float32 if code in (7, 8, 9, 10) else (torch.complex64 if code == 11 else a.dtype)
        out = torch.empty(a.shape[0], dtype=odt, device=a.device) if not isinstance(a, np.ndarray) else np.empty(a.shape[0])
    s = complex(scalar)
    _check(lib().tsdgpu_vec_op(code, dt, _ptr(out), _ptr(a), None if b is None else _ptr(b), s.real, s.imag, int(a.shape[0]),
                               _stream_of(a, None)))
    return out


def vec_reduce(a):
    """Reductions of a RESIDENT tensor (tsdgpu_vec_reduce): -> (sum as complex, max, min, index of the first max)."""
    s = (C.c_double * 2)()
    mm = (C.c_float * 2)()
    im = C.c_int64(-1)
    _check(lib().tsdgpu_vec_reduce(_dtype_code(a), _ptr(a), int(a.shape[0]), s, mm, C.byref(im), _stream_of(a, None)))
    return complex(s[0], s[1]), mm[0], mm[1], im.value


def delay_estimate(x, y):
    """estimation_délais (estimation-delais.cc:100-118): -> (delay in samples, score)."""
    assert _dtype_code(x) == C64 and _dtype_code(y) == C64 and x.shape[0] == y.shape[0]
    d, s = C.c_float(0), C.c_float(0)
    _check(lib().tsdgpu_delay_estimate(_ptr(x), _ptr(y), int(x.shape[0]), C.byref(d), C.byref(s), _stream_of(x, None)))
    return d.value, s.value


class Peak(C.Structure):
    _fields_ = [("index", C.c_int32), ("s_m1", C.c_float), ("s0", C.c_float), ("s_p1", C.c_float),
                ("c_m1", C.c_float * 2), ("c0", C.c_float * 2), ("c_p1", C.c_float * 2)]


class Detector:
    """Detecteur (detection.cc): score stream + peak records of a pattern detector; mode 0 = OLA engine, 1 = FIR."""

    def __init__(self, pattern, Ne, mode=0, threshold=0.5):
        p = np.ascontiguousarray(pattern, np.complex64)
        p = (p / np.float32(np.sqrt(np.sum(np.abs(p.astype(np.complex128)) ** 2)))).astype(np.complex64)
        self._h = C.c_void_p()
        _check(lib().tsdgpu_detector_create(C.byref(self._h), p.ctypes.data, len(p), int(Ne), int(mode), float(threshold)))
        self.delay = lib().tsdgpu_detector_delay(self._h)
        self.N = lib().tsdgpu_detector_fft_size(self._h)

    def step(self, x):
        assert _dtype_code(x) == C64
        n = int(x.shape[0])
        sc = np.empty(n, np.float32) if isinstance(x, np.ndarray) else x.new_empty(n, dtype=__import__("torch").float32)
        pk = (Peak * 256)()
        npk = C.c_int(0)
        _check(lib().tsdgpu_detector_step(self._h, _ptr(x), n, _ptr(sc), pk, 256, C.byref(npk), _stream_of(x, None)))
        return sc, [pk[i] for i in range(npk.value)]

    def __del__(self):
        try:
            if self._h:
                lib().tsdgpu_detector_destroy(self._h)
        except Exception:
            pass
