#!/usr/bin/env python3
"""Generic resampler kernel (any table-driven or analytic interpolator) on 2^26 complex inputs, ratio 160/147."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libtsd_amd as t  # noqa: E402
from scripts.perf_configs import timeit  # noqa: E402

dev = torch.device("cuda", 0)
n = 1 << 26
x = torch.view_as_complex(torch.randn(n, 2, device=dev))
ratio = np.float32(160.0) / np.float32(147.0)
cases = [("sinc K=15 (fused kernel)", dict()), ("sinc K=7", dict(K=7, fcut=0.4)), ("sinc K=31", dict(K=31, fcut=0.4)),
         ("sinc K=127", dict(K=127, fcut=0.4)), ("linear", dict(analytic=("lin", 0))), ("lagrange 3", dict(analytic=("lagrange", 3))),
         ("lagrange 7", dict(analytic=("lagrange", 7)))]
for name, kw in cases:
    r = t.Resampler(ratio, t.C64, **kw)
    nout = r.out_count(n) + 4
    y = torch.empty(nout, dtype=x.dtype, device=dev)

    def step():
        r.seek(0)
        r.step(x, y)
    ms = timeit(step, 8, 3)
    byts = 8.0 * n * (1 + float(ratio))
    print(json.dumps({"interpolator": name, "ms": round(ms, 3), "frac_of_8TBps": round(byts / (ms * 1e-3) / 8e12, 4)}), flush=True)
