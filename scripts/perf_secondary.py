#!/usr/bin/env python3
"""One pass over the secondary kernels (rfft, OLA engine, Welch, smooth / large / mixed FFT sizes, generic
resampler) -- run under `rocprofv3 --kernel-trace --stats` for profiles/r1_secondary_kernel_stats.csv."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for script, argv in [("perf_rfft.py", []), ("perf_ola.py", []), ("perf_fft_sizes.py", ["48", "1536", "3072", "15360", "1000", "16384", "4194304", "16777216"]),
                     ("perf_resample_generic.py", [])]:
    sys.argv = [script] + argv
    runpy.run_path(os.path.join(ROOT, "scripts", script), run_name="__main__")
