#!/usr/bin/env python3
"""Pattern detector (tsdgpu_detector_*): time per block by block length, correlator mode (0: OLA engine, 1: FIR) and where the
block lives.  usage (GPU box): python3 scripts/perf_detector.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libtsd_amd as t  # noqa: E402


def med(fn, reps=100):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[reps // 2] * 1e6


def main():
    rng = np.random.default_rng(0)
    M = 127
    pat = (rng.standard_normal(M) + 1j * rng.standard_normal(M)).astype(np.complex64)
    for Ne in (512, 4096, 65536, 1 << 20):
        x = (0.1 * (rng.standard_normal(Ne) + 1j * rng.standard_normal(Ne))).astype(np.complex64)
        xd = torch.from_numpy(x).cuda()
        for mode in (0, 1):
            d = t.Detector(pat, Ne, mode, threshold=0.9)
            d.step(x)
            host = med(lambda: d.step(x), 50 if Ne > 100000 else 200)
            dev = med(lambda: d.step(xd), 50 if Ne > 100000 else 200)
            print(json.dumps({"Ne": Ne, "M": M, "mode": "ola" if mode == 0 else "fir", "us_per_block_host": round(host, 1),
                              "us_per_block_resident": round(dev, 1), "Msamples_s_resident": round(Ne / dev, 1)}))

    # long resident vectors through ONE step (what Detecteur::step(x) of the mirror does): many blocks of the
    # correlator per call
    for Ne in (512, 1024, 4096):
        n = (1 << 22) // Ne * Ne
        xd = torch.view_as_complex(0.1 * torch.randn(n, 2, device="cuda"))
        for mode in (0, 1):
            d = t.Detector(pat, Ne, mode, threshold=0.9)
            d.step(xd)
            us = med(lambda: d.step(xd), 30)
            print(json.dumps({"Ne": Ne, "M": M, "mode": "ola" if mode == 0 else "fir", "n_per_call": n, "us_per_call_resident": round(us, 1),
                              "Msamples_s_resident": round(n / us, 1)}))


if __name__ == "__main__":
    main()
