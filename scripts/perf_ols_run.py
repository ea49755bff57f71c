#!/usr/bin/env python3
"""A/B of the overlap-save kernel's block schedule, interleaved in ONE process on the headline shape (127 real taps, 2^26
cfloat, resident): TSDGPU_OLS_RUN = blocks a wave walks consecutively (overlap rows reused from registers),
TSDGPU_OLS_DYN = counters of the dynamic hand-out (0 = static partition).  usage: perf_ols_run.py "R:NC,R:NC,..." [K]
Prints ms per step (HIP events over 100 launches) per variant and round, then the medians."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libtsd_amd as t
from bench import design_lowpass

vals = [tuple(int(q) for q in v.split(":")) for v in (sys.argv[1] if len(sys.argv) > 1 else "1:0,2:0,1:16,2:16,4:16,2:8,2:32").split(",")]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 127
n = 1 << 26
dev = torch.device("cuda", 0)
x = torch.view_as_complex(torch.randn(n, 2, device=dev))
y = torch.empty_like(x)
f = t.Fir(design_lowpass(K, 0.02), t.C64, t.FIR_OVERLAP_SAVE)
for _ in range(150):
    f.step(x, y)
torch.cuda.synchronize()
res = {v: [] for v in vals}
for rnd in range(5):
    for v in vals:
        os.environ["TSDGPU_OLS_RUN"], os.environ["TSDGPU_OLS_DYN"] = str(v[0]), str(v[1])
        for _ in range(10):
            f.step(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f.step(x, y)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 100)
for v in vals:
    m = float(np.median(res[v]))
    print(f"K={K} run={v[0]:2d} counters={v[1]:2d}  " + " ".join(f"{q:.4f}" for q in res[v]) + f"   median {m:.4f} ms  = {16.0 * n / (m * 1e-3) / 8e12:.4f} of 8 TB/s")
