import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import libtsd_amd as t
from scripts.perf_configs import timeit
dev = torch.device("cuda", 0)
total = 1 << 26
for n in [256, 1024, 4096, 16384, 1000, 1 << 16]:
    b = total // n
    x = torch.randn(b, n, device=dev)
    y = torch.empty(b, n, dtype=torch.complex64, device=dev)
    p = t.Rfft(n)
    ms = timeit(lambda: p.step(x, y), 10, 3)
    print(json.dumps({"n": n, "batch": b, "ms": round(ms, 4), "frac_of_8TBps_at_12B": round(12.0 * total / (ms * 1e-3) / 8e12, 4)}), flush=True)
