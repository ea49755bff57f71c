#!/usr/bin/env python3
"""Throughput of the integer-rate stages (rows a11) and filtre_rii on 2^26 complex samples."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libtsd_amd as t  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]


def main():
    dev = torch.device("cuda", 0)
    n = 1 << 26
    x = torch.view_as_complex(torch.randn(n, 2, device=dev))
    k = np.arange(15) - 7
    h = (0.5 * np.sinc(0.5 * k) * np.hanning(17)[1:-1]).astype(np.float32)
    h /= h.sum()
    cases = [("decimateur R=2 (pick)", t.PolyFir(t.POLY_PICK, t.C64, None, 2), 8 + 4),
             ("filtre_rif_demi_bande (15 taps)", t.PolyFir(t.POLY_HALFBAND, t.C64, h, 2), 8 + 4),
             ("filtre_rif_decim R=2 (15 taps)", t.PolyFir(t.POLY_DECIM, t.C64, h, 2), 8 + 4),
             ("filtre_rif_decim R=4 (15 taps)", t.PolyFir(t.POLY_DECIM, t.C64, h, 4), 8 + 2),
             ("filtre_rif_ups R=2 (15 taps)", t.PolyFir(t.POLY_UPS, t.C64, h, 2), 8 + 16)]
    for name, f, bps in cases:
        ms = timeit(lambda: f.step(x))
        print(json.dumps({"stage": name, "ms": round(ms, 4), "Gsamples_in_per_s": round(n / ms / 1e6, 1),
                          "frac_of_8TBps": round(bps * n / (ms * 1e-3) / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    main()
