#!/usr/bin/env python3
"""OLA frequency-domain engine (filtre_fft) with the device-side response, data resident in HBM:
samples/s against the 16 B/sample of the in/out streams (the frames and spectra are internal)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libtsd_amd as t  # noqa: E402
from scripts.perf_configs import timeit  # noqa: E402

dev = torch.device("cuda", 0)
n = 1 << 24
x = torch.view_as_complex(torch.randn(n, 2, device=dev))
geoms = [(512, 127, False), (2048, 127, False), (4096, 1025, False), (8192, 127, False), (512, 0, True), (4096, 0, True)]
if "--all" in sys.argv:    # more geometries, whole-block and ragged: N/2-blocks (carry in registers) and others (carry in LDS)
    geoms += [(16, 15, False), (64, 33, False), (128, 127, False), (256, 255, False), (1024, 1023, False), (900, 100, False),
              (3000, 500, False), (1500, 400, False), (6000, 2000, False), (700, 200, False)]
for Ne, M, win in geoms:
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(Ne) / Ne)).astype(np.float32) if win else None
    g = t.Ola(Ne, M, w)
    g.set_response(np.ones(g.N, np.complex64))
    ms = timeit(lambda: g.step(x), 10, 3)
    print(json.dumps({"Ne": Ne, "zeros_min": M, "N": g.N, "windowed": win, "ms": round(ms, 3),
                      "Msamples_per_s": round(n / ms / 1e3, 1), "frac_of_8TBps_at_16B": round(16.0 * n / (ms * 1e-3) / 8e12, 4)}), flush=True)

if "--all" in sys.argv:
    sys.exit(0)

# psd_welch on the same resident data: segments of N with half overlap
for N in (256, 1024, 4096, 1000):
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
    ms = timeit(lambda: t.welch(x, N, w), 10, 3)
    print(json.dumps({"welch_N": N, "ms": round(ms, 3), "Msamples_per_s": round(n / ms / 1e3, 1)}), flush=True)
