#!/usr/bin/env python3
"""Latency of small host-buffer calls (BASELINE configs[0] shape: 31 taps on 4096 real samples):
plan creation, one step with host pointers (H2D + kernel + D2H), destruction."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import libtsd_amd as t  # noqa: E402


def med(fn, reps=200):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[reps // 2] * 1e6


def main():
    h = np.hanning(33)[1:-1].astype(np.float32)
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32)
    y = np.empty_like(x)
    f = t.Fir(h, t.F32)
    f.step(x, y)

    def whole():
        g = t.Fir(h, t.F32)
        g.step(x, y)
        g.close()

    def create():
        t.Fir(h, t.F32).close()

    # host-block size sweep (float32 samples)
    sweep = {}
    for n in (512, 4096, 16384, 65536, 262144):
        xs = np.random.default_rng(1).standard_normal(n).astype(np.float32)
        ys = np.empty_like(xs)
        f.step(xs, ys)
        sweep[str(n)] = round(med(lambda: f.step(xs, ys)), 1)
    print(json.dumps({"fir31_step_host_us_by_block": sweep}))
    xc = x.astype(np.complex64)
    p = t.Fft(4096)
    p.step(xc)
    print(json.dumps({"fir_step_4096_host_us": round(med(lambda: f.step(x, y)), 1), "fir_create_destroy_us": round(med(create), 1),
                      "filtrer_like_create_step_destroy_us": round(med(whole), 1), "fft_4096_step_host_us": round(med(lambda: p.step(xc)), 1),
                      "fft_plan_create_destroy_us": round(med(lambda: t.Fft(4096).close()), 1)}))


if __name__ == "__main__":
    main()
