#!/usr/bin/env python3
"""Resampler throughput across the accepted ratio range (2^22 complex inputs, 15-tap sinc interpolator), after the one-time
schedule build of each ratio: a search for ratio-dependent cliffs.  usage (GPU box): python3 scripts/perf_resample_ratios.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libtsd_amd as t  # noqa: E402


def main():
    n = 1 << 22
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
    for ratio in (1 / 64, 0.02, 0.3, 0.5, 0.73, 1.0, 160 / 147, 1.25, 1.5, 1.99, 2.0, 2.5, 4.0, 5.0, 6.0):
        try:
            r = t.Resampler(ratio, t.C64)
        except t.TsdGpuError as e:
            print(json.dumps({"ratio": round(ratio, 4), "refused": str(e)[-90:]}))
            continue
        t0 = time.perf_counter()
        r.step(x)
        torch.cuda.synchronize()
        first = (time.perf_counter() - t0) * 1e3
        for _ in range(12):
            r.step(x)                      # (lets the schedule find its period: the stream position keeps advancing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = r.step(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(json.dumps({"ratio": round(ratio, 4), "first_call_ms": round(first, 2), "ms": round(ms, 3),
                          "GB_s": round((n * 8 + y.shape[0] * 8) / ms / 1e6, 1)}))


if __name__ == "__main__":
    main()
