#!/usr/bin/env python3
"""FFT 2^20 x 256 timing of the two column passes under the TSDGPU_FFT_* tuning hooks
(each setting runs in a fresh process: the hooks are read once)."""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import libtsd_amd as t
dev = torch.device("cuda", 0)
n, batch = 1 << 20, 256
x = torch.view_as_complex(torch.randn(batch * n, 2, device=dev)).reshape(batch, n)
y = torch.empty_like(x)
p = t.Fft(n, batch)
for _ in range(3): p.step(x, True, y)
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
torch.cuda.synchronize()
for a, b in evs:
    a.record(); p.step(x, True, y); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)
print("%s ms=%.4f" % (os.environ.get("TAG", ""), ms[len(ms) // 2]), flush=True)
'''

def main():
    settings = [dict(), dict(TSDGPU_FFT_GRID="128"), dict(TSDGPU_FFT_GRID="512")]
    for s in settings:
        env = dict(os.environ, **s, TAG=str(s))
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False, stderr=subprocess.DEVNULL)

if __name__ == "__main__":
    main()
