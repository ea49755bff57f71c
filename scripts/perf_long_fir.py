#!/usr/bin/env python3
"""FIR throughput for long filters on 2^26 complex samples: overlap-save plans (1024-point wave
blocks up to 897 taps, 2048..16384-point Stockham blocks up to 12289) vs the direct kernel."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libtsd_amd as t  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n = 1 << 26
    x = torch.view_as_complex(torch.randn(n, 2, device=dev))
    y = torch.empty_like(x)
    Ks = [int(a) for a in sys.argv[1:]] or [127, 513, 897, 1024, 2048, 4096, 8192, 12289]
    rng = np.random.default_rng(0)
    for K in Ks:
        h = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
        for m, nm in ((t.FIR_AUTO, "auto"), (t.FIR_DIRECT, "direct")):
            if nm == "direct" and K > 2048:
                continue
            f = t.Fir(h, t.C64, m)
            reps = 3 if nm == "direct" and K > 500 else 10
            for _ in range(2):
                f.step(x, y)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            torch.cuda.synchronize()
            for a, b in evs:
                a.record(); f.step(x, y); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]
            print(json.dumps({"K": K, "requested": nm, "used": {1: "direct", 2: "overlap-save"}[f.method], "ms": round(ms, 4),
                              "Gsamples_per_s": round(n / ms / 1e6, 1), "frac_of_8TBps": round(16.0 * n / (ms * 1e-3) / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    main()
