#!/usr/bin/env python3
"""psd_welch alone on 2^24 resident complex samples (argv: the segment lengths; default 1000) -- for kernel traces."""
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
for N in [int(a) for a in sys.argv[1:]] or [1000]:
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
    wd = torch.from_numpy(w).to(dev)
    ms = timeit(lambda: t.welch(x, N, w), 10, 3)
    print(json.dumps({"welch_N": N, "ms": round(ms, 3), "Msamples_per_s": round(n / ms / 1e3, 1)}), flush=True)
