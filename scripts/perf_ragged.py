#!/usr/bin/env python3
"""Per-call time of every streaming operator on SMALL and RAGGED resident blocks (the usual way libtsd call sites feed
a stream): a search for cliffs -- code paths that serve the awkward sizes sample by sample.  usage (GPU box):
python3 scripts/perf_ragged.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libtsd_amd as t  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402  (designs only)


def us_per_call(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e6, 1)


def main():
    sizes = [100, 512, 1000, 4097, 65536 + 777, (1 << 20) + 2047]
    z, p, mn, md = orc.design_butter_lp(12, 0.25)
    co, gain, r1 = orc.SosChain(z, p, mn, md).coefs()
    h31, h127, h1k, h5k = (orc.design_rif_fen(k, "lp", 0.1) for k in (31, 127, 1025, 5001))
    de = np.real(np.poly([0.8 * np.exp(0.5j), 0.8 * np.exp(-0.5j), 0.6, -0.3, 0.5 * np.exp(1j), 0.5 * np.exp(-1j)])).astype(np.float32)
    nu = np.array([1.0, 0.4, 0.2, 0.1, 0.05], np.float32)
    ops = {
        "fir31 (direct)": lambda: t.Fir(h31, t.C64),
        "fir127 (overlap-save)": lambda: t.Fir(h127, t.C64),
        "fir1025 (long blocks)": lambda: t.Fir(h1k, t.C64),
        "fir5001 (long blocks)": lambda: t.Fir(h5k, t.C64),
        "sos 6 sections": lambda: t.Sos(co, gain, t.C64, r1),
        "rii order 6 (factored)": lambda: t.Rii(nu, de, t.C64),
        "resampler 160/147": lambda: t.Resampler(160 / 147, t.C64),
        "resampler 0.73 lagrange3": lambda: t.Resampler(0.73, t.C64, analytic=("lagrange", 3)),
        "decim 3 (31 taps)": lambda: t.PolyFir(t.POLY_DECIM, t.C64, h31, 3),
        "halfband (31 taps)": lambda: t.PolyFir(t.POLY_HALFBAND, t.C64, h31),
        "ups 3 (31 taps)": lambda: t.PolyFir(t.POLY_UPS, t.C64, h31, 3),
        "ola 512/127 response": None,
    }
    for name, mk in ops.items():
        row = {}
        for n in sizes:
            x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
            if mk is None:
                g = t.Ola(512, 127, None)
                g.set_response(np.ones(g.N, np.complex64))
            else:
                g = mk()
            row[str(n)] = us_per_call(lambda: g.step(x))
        print(json.dumps({"op": name, "us_per_call": row}))
    # transforms: one call of n points
    for n in (100, 127, 1000, 1001, 4097, 30000, 65536 + 777):
        x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
        p = t.Fft(n)
        print(json.dumps({"op": "fft", "n": n, "us_per_call": us_per_call(lambda: p.step(x))}))


if __name__ == "__main__":
    main()
