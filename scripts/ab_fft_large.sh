#!/bin/bash
# tile width / split experiments on the four-step plan beyond 2^20 (scripts/perf_fft_sizes.py, 2^26 points per call)
SZ="2097152 4194304 16777216"
echo "default"; python scripts/perf_fft_sizes.py $SZ 2>/dev/null
for ct in 4 2; do echo "CTMAX=$ct"; TSDGPU_FFT_CTMAX=$ct python scripts/perf_fft_sizes.py $SZ 2>/dev/null; done
for l1 in 10 12; do echo "LOGN1=$l1 (2^22 only)"; TSDGPU_FFT_LOGN1=$l1 python scripts/perf_fft_sizes.py 4194304 2>/dev/null; done
echo "LOGN1=10 CTMAX=4 (2^22)"; TSDGPU_FFT_LOGN1=10 TSDGPU_FFT_CTMAX=4 python scripts/perf_fft_sizes.py 4194304 2>/dev/null
