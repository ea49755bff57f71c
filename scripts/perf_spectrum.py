#!/usr/bin/env python3
"""rt_spectrum on the device (tsdgpu_spectrum_*): 2^24 resident complex samples per call, spectra in dB back per nmeans blocks.
Algorithmic traffic: 8 B per sample in (the spectra are noise); plain and sweep mode, a few geometries."""
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
for BS, nsubs, nmeans, sweep in [(1024, 1, 10, None), (4096, 4, 10, None), (512, 1, 1, None), (16384, 8, 4, None), (8192, 8, 10, (64, None)),
                                 (1000, 1, 10, None), (8000, 8, 10, None)]:
    Nf = BS // nsubs
    w = np.hanning(Nf).astype(np.float32)
    w *= np.float32(np.sqrt(Nf / np.sum(w.astype(np.float64) ** 2)))
    s = t.Spectrum(BS, nsubs, nmeans, w, sweep)
    B = n // BS
    xs = x[:B * BS]
    y = torch.empty(((B + nmeans) // nmeans, s.Ns), dtype=torch.float32, device=dev)

    def step():
        s.reset()
        s.step(xs, y[:B // nmeans])
    ms = timeit(step, 8, 3)
    print(json.dumps({"BS": BS, "nsubs": nsubs, "Nf": Nf, "nmeans": nmeans, "sweep": sweep is not None, "ms": round(ms, 3),
                      "Gsamples_per_s": round(B * BS / (ms * 1e-3) / 1e9, 1), "frac_of_8TBps_at_8B": round(8.0 * B * BS / (ms * 1e-3) / 8e12, 4)}), flush=True)
