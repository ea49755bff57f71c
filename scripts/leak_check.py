"""Create / step / destroy every operator of the C ABI a few hundred times and compare the free device memory before and
after: a handle that leaks a buffer shows up as a delta that grows with the round count.  usage (GPU box): python3 scripts/leak_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libtsd_amd as t
from oracle import ola_oracle
torch.cuda.init()
def free(): 
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
rng = np.random.default_rng(0)
x = (rng.standard_normal(20000) + 1j*rng.standard_normal(20000)).astype(np.complex64)
xr = x.real.copy()
h = np.hanning(129)[1:-1].astype(np.float32)
from scipy.signal import butter
sos = butter(6, 0.3, output="sos"); co = np.array([[s[0], s[1], s[2], s[4], s[5]] for s in sos], np.float32)
def ops():
    f = t.Fir(h, t.C64); f.step(x); f.close() if hasattr(f, "close") else None
    f = t.Fir(np.hanning(3000).astype(np.float32), t.C64); f.step(x); f.close()
    s = t.Sos(co, 1.0, t.F32); s.step(xr); del s
    r = t.Resampler(160/147, t.C64); r.step(x); del r
    r = t.Resampler(0.7, t.F32, analytic=("lagrange", 3)); r.step(xr); del r
    p = t.Fft(4096); p.step(x[:4096]); p.close()
    p = t.Fft(1000); p.step(x[:1000]); p.close()
    p = t.Fft(1 << 16); p.step(np.tile(x[:16384], 4)); p.close()
    g = t.PolyFir(t.POLY_DECIM, t.C64, h, 3); g.step(x); del g
    g = t.PolyFir(t.POLY_DECIM, t.C64, h[:15].copy(), 2); g.step(x); del g              # decim_direct_kernel
    g = t.PolyFir(t.POLY_UPS, t.F32, h[:15].copy(), 2); g.step(xr); del g                # ups_direct_kernel
    p = t.Fft(1001); p.step(x[:1001]); p.close()                                          # wave-level Bluestein, two waves per transform
    p = t.Fft(1 << 15, 64); p.step(np.tile(x[:16384], 128).reshape(64, 1 << 15)); p.close()   # 1024 x C plan
    t.welch(x, 1000, ola_oracle.fen_hann_periodique(1000))
    q = t.Rii(np.array([1, .5, .2, .1], np.float32), np.array([1, -.5, .3, -.1, .05], np.float32), t.F32); q.step(xr); del q
    o = t.Ola(512, 100, None); o.set_response(np.ones(o.N, np.complex64)); o.step(x); del o
    d = t.Detector(x[:64].copy(), 1024, 0, threshold=0.8); d.step(x[:1024].copy()); del d
    t.xcorr(x[:1000], x[:1000], -1, False)
    t.welch(x, 256, ola_oracle.fen_hann_periodique(256))
    sh = t.Sharded("fir", t.C64, 3, taps=h, method=t.FIR_DIRECT); sh.step_host(x); del sh
import gc
for _ in range(20): ops()
gc.collect(); f0 = free()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300): ops()
gc.collect(); f1 = free()
print("free before", f0 >> 20, "MiB; after the rounds", f1 >> 20, "MiB; delta", (f0 - f1) >> 10, "KiB")
import resource
print("host maxrss MiB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10)
