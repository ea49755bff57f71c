"""Handle life cycle: filtrer()/fft()-style one-shot use creates and destroys an operator per
call (filtrage.hpp:1684-1711), so create -> step -> destroy must not leak device memory, for
every handle type and plan kind of the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def free_bytes(torch):
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_create_step_destroy_does_not_leak():
    import torch
    import libtsd_amd as t
    from oracle import pyoracle as orc
    rng = np.random.default_rng(0)
    xc = (rng.standard_normal(20000) + 1j * rng.standard_normal(20000)).astype(np.complex64)
    xr = rng.standard_normal(20000).astype(np.float32)
    z, p, mn, md = orc.design_butter_lp(6, 0.2)
    co, gain, r1 = orc.SosChain(z, p, mn, md).coefs()
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(256) / 256)).astype(np.float32)

    def cycle():
        for K in (7, 127, 600):                                   # direct, wave overlap-save, long overlap-save
            h = rng.standard_normal(K).astype(np.float32)
            t.Fir(h, t.C64).step(xc)
            t.Fir(h, t.F32).step(xr)
        t.Sos(co, gain, t.F32, r1).step(xr)
        t.Sos(co, gain, t.C64, r1).step(xc)
        t.Resampler(np.float32(160 / 147), t.C64).step(xc)
        t.Resampler(np.float32(0.77), t.F32, K=31, nphases=512, fcut=0.3).step(xr)
        for kind in (t.POLY_DECIM, t.POLY_HALFBAND, t.POLY_UPS):
            t.PolyFir(kind, t.C64, orc.design_rif_fen(15, "lp", 0.2), 2).step(xc)
        t.PolyFir(t.POLY_PICK, t.F32, None, 3).step(xr)
        t.Rii([0.1, 0.2], [1.0, -0.5, 0.2, -0.05], t.F32).step(xr)
        for n in (1, 16, 1024, 4096, 1 << 15, 96, 1000, 1001, 2 * 8191, 18):   # every plan kind
            m = max(1, 20000 // n)
            t.fft(xc[:n * m].reshape(m, n) if n * m <= 20000 else np.resize(xc, n).reshape(1, n))
        t.rfft(xr[:1000])
        t.rfft(xr[:999])
        g = t.Ola(256, 100, win)
        g.set_response(np.ones(g.N, np.complex64))
        g.step(xc)
        g.close()
        t.welch(xc, 256, win)

    cycle()                                                        # first pass: allocator pools, code objects
    cycle()
    before = free_bytes(torch)
    for _ in range(100):
        cycle()
    after = free_bytes(torch)
    # a leak of one 20000-sample buffer per cycle would be 100 x 160 kB = 16 MB
    print(f"free memory moved by {(before - after) / 1e6:.2f} MB over 100 cycles")
    assert before - after < (8 << 20), f"device memory shrank by {(before - after) / 1e6:.1f} MB over 100 cycles"
