#!/usr/bin/env python3
"""Batched complex FFT throughput over sizes, C ABI tsdgpu_fft_step.  argv: [--total LOG2] sizes...  (default 2^26 points per call;
cfg 3 of BASELINE.json is 2^28: the two launches of a four-step plan have their ramp and tail to amortise)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libtsd_amd as t  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    argv = sys.argv[1:]
    total = 1 << 26
    if argv and argv[0] == "--total":
        total = 1 << int(argv[1])
        argv = argv[2:]
    sizes = [int(a) for a in argv] or [1 << k for k in range(6, 25, 2)] + [1000, 3 * 1024, 15 * 1024, 1 << 11, 1 << 13, 1 << 21]
    for n in sizes:
        batch = max(1, total // n)
        x = torch.view_as_complex(torch.randn(batch * n, 2, device=dev)).reshape(batch, n)
        y = torch.empty_like(x)
        p = t.Fft(n, batch)
        for _ in range(3 if sizes.index(n) else 40):       # (the first size also brings the clocks up: ~30 ms of kernels)
            p.step(x, True, y)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        torch.cuda.synchronize()
        for a, b in evs:
            a.record(); p.step(x, True, y); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[5]
        gbs = 16.0 * n * batch / (ms * 1e-3) / 1e9
        print(json.dumps({"n": n, "batch": batch, "ms": round(ms, 4), "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 4)}), flush=True)
        del x, y, p


if __name__ == "__main__":
    main()
