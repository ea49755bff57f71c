"""Does a small kernel between two long ones cost more than its own duration?  (127-tap FIR steps on 2^26 samples with and
without a set_history -- a 1-KiB copy kernel -- in front of each; one stream, one process.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libtsd_amd as t
import bench
dev = torch.device("cuda", 0)
n = 1 << 26
x = torch.view_as_complex(torch.randn(n, 2, device=dev)); y = torch.empty_like(x)
f = t.Fir(bench.design_lowpass(127, 0.02), t.C64, t.FIR_AUTO)
def run(with_copy, K=200):
    for _ in range(20): f.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        if with_copy: f.set_history(x[:126])
        f.step(x[126:] if with_copy else x, y[126:] if with_copy else y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for _ in range(2):
    print("plain", round(run(False), 4), "with a small copy kernel in front", round(run(True), 4), flush=True)
