"""One-shot fft() cost (plan creation + transform + PCIe), the way fft(x) is used in libtsd user code."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libtsd_amd as t  # noqa: E402

rng = np.random.default_rng(0)
for n in (1024, 1000, 1 << 16, 1 << 20, 1 << 22, 12000, 1001):
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    y = np.empty_like(x)
    t.fft(x)
    t0 = time.perf_counter()
    for _ in range(5):
        p = t.Fft(n, 1)
        p.close()
    tc = (time.perf_counter() - t0) / 5
    p = t.Fft(n, 1)
    p.step(x, True, y)
    t0 = time.perf_counter()
    for _ in range(5):
        p.step(x, True, y)
    ts = (time.perf_counter() - t0) / 5
    print(f"n = {n}: plan create {tc*1e3:.3f} ms, step on host buffers {ts*1e3:.3f} ms")
