#!/usr/bin/env python3
"""The long-interpolator kernel alone (127-tap sinc, 2^26 complex inputs, ratio 160/147; argv: K, steps) -- for PMC passes."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libtsd_amd as t  # noqa: E402
from scripts.perf_configs import timeit  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 127
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
n = 1 << 26
x = torch.view_as_complex(torch.randn(n, 2, device=dev))
ratio = np.float32(160.0) / np.float32(147.0)
r = t.Resampler(ratio, t.C64, K=K, fcut=0.4)
y = torch.empty(r.out_count(n) + 4, dtype=x.dtype, device=dev)


def step():
    r.seek(0)
    r.step(x, y)


ms = timeit(step, steps, 2)
print(json.dumps({"K": K, "ms": round(ms, 3), "frac_of_8TBps": round(8.0 * n * (1 + float(ratio)) / (ms * 1e-3) / 8e12, 4)}), flush=True)
