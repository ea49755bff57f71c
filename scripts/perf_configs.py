#!/usr/bin/env python3
"""Per-config kernel timings (BASELINE.json configs 2-5) on one MI355X, data resident in HBM.
Prints one JSON line per config with the algorithmic HBM rate against the 8 TB/s roofline.
Secondary to bench.py (which carries the headline metric); numbers quoted in DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libtsd_amd as t  # noqa: E402

PEAK = 8000.0


def timeit(fn, steps=10, warmup=2):
    for _ in range(warmup):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]


def line(name, ms, alg_bytes, units, unit_name, extra=None):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    d = {"config": name, "ms": round(ms, 4), unit_name: round(units / (ms * 1e-3) / 1e6, 1),
         "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 4)}
    if extra:
        d.update(extra)
    print(json.dumps(d), flush=True)


def lowpass(n, fc):
    k = np.arange(n) - n // 2
    h = 2 * fc * np.sinc(2 * fc * k) * (0.5 + 0.5 * np.cos(2 * np.pi * np.linspace(-(n // 2) / n, (n // 2) / n, n)))
    return (h / h.sum()).astype(np.float32)


def main():
    which = sys.argv[1:] or ["fir", "fft", "sos", "resample"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device=dev).manual_seed(1)
    if "fir" in which:
        n = 1 << 26
        x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g))
        y = torch.empty_like(x)
        h = lowpass(127, 0.02)
        for m, nm in ((t.FIR_OVERLAP_SAVE, "overlap-save"), (t.FIR_DIRECT, "direct")):
            f = t.Fir(h, t.C64, m)
            line(f"cfg2 127-tap FIR 2^26 cfloat, real taps, {nm}", timeit(lambda: f.step(x, y)), 16.0 * n, n, "Msamples_per_s")
        hc = (h * np.exp(2j * np.pi * 0.1 * np.arange(127))).astype(np.complex64)
        for m, nm in ((t.FIR_OVERLAP_SAVE, "overlap-save"), (t.FIR_DIRECT, "direct")):
            f = t.Fir(hc, t.C64, m)
            line(f"cfg2 127-tap FIR 2^26 cfloat, complex taps, {nm}", timeit(lambda: f.step(x, y), 5, 1), 16.0 * n, n, "Msamples_per_s")
        xr = torch.randn(n, device=dev, generator=g)
        yr = torch.empty_like(xr)
        for m, nm in ((t.FIR_OVERLAP_SAVE, "overlap-save (block pairs packed re/im)"), (t.FIR_DIRECT, "direct")):
            f = t.Fir(h, t.F32, m)
            line(f"127-tap FIR 2^26 float (real data, real taps), {nm}", timeit(lambda: f.step(xr, yr), 5, 1), 8.0 * n, n, "Msamples_per_s")
        del x, y, xr, yr
    if "fft" in which:
        n, batch = 1 << 20, 256
        x = torch.view_as_complex(torch.randn(batch * n, 2, device=dev, generator=g)).reshape(batch, n)
        y = torch.empty_like(x)
        p = t.Fft(n, batch)
        line("cfg3 FFT 2^20 x 256 cfloat", timeit(lambda: p.step(x, True, y), 5, 1), 16.0 * n * batch, n * batch, "Mpoints_per_s")
        del x, y
    if "sos" in which:
        n = 1 << 26
        x = torch.randn(n, device=dev, generator=g)
        y = torch.empty_like(x)
        # 12th-order Butterworth lp 0.25 -> 6 DF2 sections (coefficients = bench input data)
        from scipy.signal import butter
        sos = butter(12, 0.5, output="sos")          # fcut 0.25 of fs == 0.5 of Nyquist
        co = np.array([[s[0], s[1], s[2], s[4], s[5]] for s in sos], np.float32)
        f = t.Sos(co, 1.0, t.F32)
        line("cfg4 6-section SOS 2^26 float", timeit(lambda: f.step(x, y)), 8.0 * n, n, "Msamples_per_s", {"halo": f.halo})
        del x, y
    if "resample" in which and hasattr(t, "Resampler"):
        n = 1 << 27
        x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g))
        r = t.Resampler(np.float32(160.0) / np.float32(147.0), t.C64)
        nout = r.out_count(n)
        y = torch.empty(nout, dtype=x.dtype, device=dev)
        ms = timeit(lambda: (r.reset(), r.step(x, y))[1], 5, 1)
        line("cfg5 resample 160/147 of 2^27 cfloat (one GPU's shard of 2^30)", ms, 8.0 * n + 8.0 * nout, n, "Msamples_in_per_s", {"n_out": int(nout)})


if __name__ == "__main__":
    main()
