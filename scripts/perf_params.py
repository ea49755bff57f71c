#!/usr/bin/env python3
"""Throughput of the operators across their PARAMETER ranges (tap counts, section counts and decay, decimation rates, FFT sizes)
on resident data: a search for parameter-dependent cliffs.  usage (GPU box): python3 scripts/perf_params.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libtsd_amd as t  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402  (designs only)


def ms_per_call(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 3)


def main():
    n = 1 << 22
    xc = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
    xr = torch.randn(n, device="cuda")
    row = {}
    for K in (1, 2, 16, 47, 48, 127, 513, 514, 1025, 4097, 12289, 12290, 20000, 65536):
        h = np.random.default_rng(K).standard_normal(K).astype(np.float32)
        f = t.Fir(h, t.C64)
        row[str(K)] = ms_per_call(lambda: f.step(xc))
    print(json.dumps({"op": "fir complex data, real taps, 2^22 samples: ms by tap count", "ms": row}))
    row = {}
    for K in (1, 39, 40, 127, 513, 514, 4097):
        h = np.random.default_rng(K).standard_normal(K).astype(np.float32)
        f = t.Fir(h, t.F32)
        row[str(K)] = ms_per_call(lambda: f.step(xr))
    print(json.dumps({"op": "fir real data, 2^22 samples: ms by tap count", "ms": row}))
    row = {}
    for order, fc in ((2, 0.25), (12, 0.25), (24, 0.25), (40, 0.2), (12, 0.01), (4, 0.001), (4, 1e-4), (2, 1e-5)):
        z, p, mn, md = orc.design_butter_lp(order, fc)
        co, gain, r1 = orc.SosChain(z, p, mn, md).coefs()
        g = t.Sos(co, gain, t.F32, r1)
        row[f"order {order} fc {fc} (halo {g.halo})"] = ms_per_call(lambda: g.step(xr))
    print(json.dumps({"op": "sos real data, 2^22 samples: ms by order / cut-off", "ms": row}))
    row = {}
    h = orc.design_rif_fen(63, "lp", 0.05)
    for R in (2, 3, 8, 33, 62, 63, 100, 500):
        g = t.PolyFir(t.POLY_DECIM, t.C64, h, R)
        row[f"decim {R}"] = ms_per_call(lambda: g.step(xc))
    for K in (31, 63):
        g = t.PolyFir(t.POLY_HALFBAND, t.C64, orc.design_rif_fen(K, "lp", 0.25))
        row[f"half-band {K} taps"] = ms_per_call(lambda: g.step(xc))
    for R in (2, 3, 8, 33, 62, 63, 100):
        g = t.PolyFir(t.POLY_UPS, t.C64, h, R)
        xs = xc[: n // R]
        row[f"ups {R} (n/R in)"] = ms_per_call(lambda: g.step(xs))
    print(json.dumps({"op": "integer-rate stages, 63 taps, 2^22 complex samples: ms by rate", "ms": row}))
    row = {}
    for m in (1 << 22, (1 << 22) - 1, (1 << 22) + 1, 3 * (1 << 20), 5 ** 9, 4000000, 4194301, 1000003, 999983 * 4):
        try:
            p = t.Fft(m)
            x = xc[:m] if m <= n else torch.view_as_complex(torch.randn(m, 2, device="cuda"))
            row[str(m)] = ms_per_call(lambda: p.step(x), 4)
        except t.TsdGpuError as e:
            row[str(m)] = "refused: " + str(e)[-60:]
    print(json.dumps({"op": "fft, one transform of ~4 M points: ms by size", "ms": row}))


if __name__ == "__main__":
    main()
