#!/bin/bash
# A/B of the 2^20 FFT's tile hand-out (TSDGPU_FFT_DYN=0: static partition), bench.py --workload fft, interleaved processes
for i in 1 2 3; do
  for v in 0 1; do
    TSDGPU_FFT_DYN=$v python bench.py --workload fft --steps 30 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fft dyn=$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
