"""libtsd_amd -- MI355X-native (gfx950) streaming FIR / IIR(SOS) / FFT / resample path of
libtsd, behind a C ABI (include/tsdgpu.h).

The product is the shared library libtsd_amd/lib/libtsdgpu.so (hand-written HIP kernels +
extern "C" shim) and the C++ host layer under libtsd_amd/host/ that mirrors libtsd's
tsd:: / dsp:: interfaces.  This Python package is only the thin ctypes binding the tests
and bench.py drive the C ABI through; PyTorch supplies device memory, streams and
torch.distributed -- plumbing, not the product.
"""
from .capi import (Spectrum, Sharded, Detector, xcorr, delay_estimate, vec_op, vec_reduce, PolyFir, Rii, POLY_DECIM, POLY_HALFBAND, POLY_UPS, POLY_PICK, Fir, Sos, Resampler, itrp_sinc_lut, Fft, fft, Rfft, rfft, Ola, welch, fftshift, TsdGpuError, lib, lib_path, device_count, F32, C64,  # noqa: F401
                   FIR_AUTO, FIR_DIRECT, FIR_OVERLAP_SAVE)
