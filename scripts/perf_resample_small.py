"""One-shot cost of a small resampling call through the C ABI (rééchan()-style use)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import libtsd_amd as t
x = (np.random.default_rng(0).standard_normal(4096) + 0j).astype(np.complex64)
lut = t.itrp_sinc_lut(15, 256, 0.4)          # (the table is the caller's: not part of the C-ABI cost)
for ratio in (160.0/147.0, 1.5, 0.77, 3.14159):
    t.Resampler(np.float32(ratio), t.C64, lut=lut).step(x)
    t0 = time.perf_counter()
    for _ in range(20):
        r = t.Resampler(np.float32(ratio), t.C64, lut=lut)
    tc = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        r = t.Resampler(np.float32(ratio), t.C64, lut=lut)
        y = r.step(x)
    dt = (time.perf_counter() - t0) / 20
    print(f"ratio {ratio:.5f}: create {tc*1e3:.3f} ms, create+step(4096) {dt*1e3:.3f} ms")
xb = (np.random.default_rng(0).standard_normal(1 << 22) + 0j).astype(np.complex64)
r = t.Resampler(np.float32(160.0/147.0), t.C64)
t0 = time.perf_counter(); r.step(xb); print("first 2^22 step", (time.perf_counter()-t0)*1e3, "ms")
t0 = time.perf_counter(); r.step(xb); print("second 2^22 step", (time.perf_counter()-t0)*1e3, "ms")
t0 = time.perf_counter(); r.step(xb); print("third 2^22 step (cycle known)", (time.perf_counter()-t0)*1e3, "ms")
r2 = t.Resampler(np.float32(160.0/147.0), t.C64)
t0 = time.perf_counter(); r2.step(xb); print("first 2^22 step of a NEW handle, same ratio (cached schedule)", (time.perf_counter()-t0)*1e3, "ms")
f = t.Fir(np.ones(31, np.float32) / 31, t.C64)
f.step(xb)
t0 = time.perf_counter(); yb = f.step(xb); print("(for scale) FIR 31 taps on the same 2^22 host buffer", (time.perf_counter()-t0)*1e3, "ms")
yb2 = np.empty_like(xb)
t0 = time.perf_counter(); f.step(xb, yb2); print("(for scale) same, into a caller-provided (untouched) buffer", (time.perf_counter()-t0)*1e3, "ms")
t0 = time.perf_counter(); f.step(xb, yb2); print("(for scale) same, buffer already touched", (time.perf_counter()-t0)*1e3, "ms")
