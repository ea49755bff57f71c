import sys, json, numpy as np, torch
sys.path.insert(0, "/root/repo")
import libtsd_amd as t
from scripts.perf_configs import timeit, lowpass
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
n = 1 << 26
x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g)); y = torch.empty_like(x)
xr = torch.randn(n, device=dev, generator=g); yr = torch.empty_like(xr)
for K in (7, 15, 31, 47, 63):
    h = lowpass(K, 0.1)
    for nm, dt, xx, yy, bps in (("c64", t.C64, x, y, 16.0), ("f32", t.F32, xr, yr, 8.0)):
        f = t.Fir(h, dt, t.FIR_AUTO)
        ms = timeit(lambda: f.step(xx, yy), 20, 5)
        print(json.dumps({"K": K, "data": nm, "method": f.method, "ms": round(ms, 4), "frac_8TBps": round(bps * n / (ms * 1e-3) / 8e12, 3)}))
