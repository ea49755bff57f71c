#!/usr/bin/env python3
"""Creation + destruction time of the operator handles (a one-shot filtrer() / rééchan() pays it per call).
usage (GPU box): python3 scripts/perf_create.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import libtsd_amd as t  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402  (designs only)


def ms(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[len(ts) // 2] * 1e3, 3)


def main():
    out = {}
    for K in (31, 127, 513, 1025, 5001, 12289, 20000, 100001):
        h = np.random.default_rng(K).standard_normal(K).astype(np.float32)
        out[f"fir K={K}"] = ms(lambda: t.Fir(h, t.C64).close(), 3 if K > 5000 else 5)
    z, p, mn, md = orc.design_butter_lp(12, 0.25)
    co, gain, r1 = orc.SosChain(z, p, mn, md).coefs()
    out["sos 6 sections"] = ms(lambda: t.Sos(co, gain, t.F32, r1))
    de = np.real(np.poly([0.8 * np.exp(0.5j), 0.8 * np.exp(-0.5j), 0.6, -0.3, 0.5 * np.exp(1j), 0.5 * np.exp(-1j)])).astype(np.float32)
    out["rii order 6"] = ms(lambda: t.Rii(np.array([1.0, 0.4, 0.2], np.float32), de, t.F32))
    out["resampler 160/147 (schedule cached after the first)"] = ms(lambda: t.Resampler(160 / 147, t.C64))
    for n in (1024, 4096, 1000, 65536, 1 << 20, 1000003):
        out[f"fft n={n}"] = ms(lambda: t.Fft(n).close())
    out["ola 512/127"] = ms(lambda: t.Ola(512, 127, None))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
