#!/usr/bin/env python3
"""Direct vs overlap-save FIR around the AUTO crossover, 2^26 samples (complex and real data)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libtsd_amd as t  # noqa: E402
from scripts.perf_configs import timeit, lowpass  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
n = 1 << 26
x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g))
y = torch.empty_like(x)
xr = torch.randn(n, device=dev, generator=g)
yr = torch.empty_like(xr)
Ks = [int(a) for a in sys.argv[1:]] or [7, 15, 31, 47, 63, 79, 95, 127]
for K in Ks:
    h = lowpass(K, 0.1)
    for nm, dt, xx, yy, bps in (("c64", t.C64, x, y, 16.0), ("f32", t.F32, xr, yr, 8.0)):
        row = {"K": K, "data": nm, "auto": {1: "direct", 2: "overlap-save"}[t.Fir(h, dt, t.FIR_AUTO).method]}
        for m, mn in ((t.FIR_DIRECT, "direct"), (t.FIR_OVERLAP_SAVE, "ols")):
            f = t.Fir(h, dt, m)
            ms = timeit(lambda: f.step(xx, yy), 20, 5)
            row[mn + "_ms"] = round(ms, 4)
            row[mn + "_frac"] = round(bps * n / (ms * 1e-3) / 8e12, 3)
        print(json.dumps(row), flush=True)
