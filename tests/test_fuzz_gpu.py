"""Seeded randomised parity sweep (GPU, through the C ABI, against the oracle): ragged sizes
around every tile / block / sub-tile boundary, random chunkings, unaligned device views."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5
# TSDGPU_FUZZ_SCALE=N multiplies the number of seeds of every sweep (default 1: seconds per sweep)
SCALE = int(os.environ.get("TSDGPU_FUZZ_SCALE", "1"))


def relerr(y, ref):
    return float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))


def rand(rng, n, cplx):
    if cplx:
        return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    return rng.standard_normal(n).astype(np.float32)


def random_chunks(rng, n):
    cuts, o = [], 0
    while o < n:
        c = int(rng.choice([1, 7, 63, 64, 65, 895, 896, 897, 1023, 1024, 2047, 2048, 2049, 5000, 20000]))
        cuts.append((o, min(n, o + c)))
        o += c
    return cuts


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1
    return t


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_fuzz_fir(tg, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    cplx = bool(rng.integers(2))
    ctaps = cplx and bool(rng.integers(2))
    K = int(rng.choice([1, 3, 16, 39, 40, 47, 48, 63, 64, 65, 127, 129, 193, 257, 513, 514, 640, 897, 898, 1000, 1025, 2500, 4097, 5000]))
    n = int(rng.choice([1, 100, 895, 896, 897, 1791, 1792, 1793, 4096, 30000, 70001]))
    h = rand(rng, K, ctaps) / np.float32(max(1.0, np.sqrt(K)))
    x = rand(rng, n + 5000, cplx)
    ref = orc.fir(h, x)
    for method in (tg.FIR_AUTO, tg.FIR_DIRECT, tg.FIR_OVERLAP_SAVE):
        f = tg.Fir(h, tg.C64 if cplx else tg.F32, method)
        y = np.concatenate([f.step(x[a:b].copy()) for a, b in random_chunks(rng, len(x))])
        assert relerr(y, ref) <= TOL, (seed, method, K, n, cplx, ctaps)


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_fuzz_sos(tg, orc, seed):
    rng = np.random.default_rng(2000 + seed)
    cplx = bool(rng.integers(2))
    order = int(rng.choice([1, 2, 3, 6, 12]))
    fc = float(rng.choice([0.05, 0.1, 0.25, 0.4]))
    forme = int(rng.choice([1, 2]))
    z, p, mn, md = orc.design_butter_lp(order, fc)
    ref = orc.SosChain(z, p, mn, md, forme=forme)
    co, gain, r1 = ref.coefs()
    g = tg.Sos(co, gain, tg.C64 if cplx else tg.F32, r1, forme=forme)
    x = rand(rng, int(rng.choice([1, 2047, 2049, 40000, 100001])), cplx)
    yref = ref.step(x)
    y = np.concatenate([g.step(x[a:b].copy()) for a, b in random_chunks(rng, len(x))])
    assert relerr(y, yref) <= TOL, (seed, order, fc, forme, cplx)


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_fuzz_resampler(tg, orc, seed):
    rng = np.random.default_rng(3000 + seed)
    cplx = bool(rng.integers(2))
    ratio = float(np.float32(rng.uniform(0.5, 1.999)))
    x = rand(rng, int(rng.choice([1, 15, 511, 512, 513, 2047, 2049, 33333])), cplx)
    K, nph = [(15, 256), (15, 256), (4, 256), (31, 128), (127, 256), (64, 1024)][int(rng.integers(6))]
    ref = orc.Resampler(ratio, K=K, nphases=nph, fcut=0.4)
    yref = ref.step(x)
    g = tg.Resampler(ratio, tg.C64 if cplx else tg.F32, K=K, nphases=nph, lut=ref.lut)
    parts = [g.step(x[a:b].copy()) for a, b in random_chunks(rng, len(x))]
    y = np.concatenate(parts) if parts else np.zeros(0, x.dtype)
    assert len(y) == len(yref), (seed, ratio)
    if len(y):
        assert relerr(y, yref) <= TOL, (seed, ratio, cplx, K, nph)


@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_fuzz_fft(tg, orc, seed):
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([6, 12, 16, 30, 32, 48, 64, 96, 255, 256, 600, 1024, 1536, 2048, 3000, 4096, 5120, 8192, 16384, 15360, 32768, 65536,
                        1 << 17, 3 << 15]))
    batch = int(rng.integers(1, 4))
    x = rand(rng, batch * n, True).reshape(batch, n)
    p = tg.Fft(n)
    for fwd in (True, False):
        y = p.step(x, fwd)
        for b in range(batch):
            assert relerr(y[b], orc.fft(x[b], fwd)) <= (TOL if n % 2 == 0 else 2e-5), (seed, n, fwd)


@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_fuzz_polyphase(tg, orc, seed):
    rng = np.random.default_rng(6000 + seed)
    cplx = bool(rng.integers(2))
    kind = int(rng.integers(3))
    R = int(rng.choice([2, 3, 4, 7, 16, 33]))
    K = int(rng.choice([3, 15, 16, 31, 64, 101]))
    h = orc.design_rif_fen(K, "lp", 0.45 / R)
    x = rand(rng, int(rng.choice([1, 100, 4095, 4096, 4097, 30011])), cplx)
    dt = tg.C64 if cplx else tg.F32
    if kind == 0:
        ref, g = orc.PolyDecim(h, R, 0), tg.PolyFir(tg.POLY_DECIM, dt, h, R)
    elif kind == 1:
        ref, g = orc.PolyDecim(h, 2, 1), tg.PolyFir(tg.POLY_HALFBAND, dt, h)
    else:
        ref, g = orc.PolyUps(h, R), tg.PolyFir(tg.POLY_UPS, dt, h, R)
    yref = ref.step(x)
    parts = [g.step(x[a:b].copy()) for a, b in random_chunks(rng, len(x))]
    y = np.concatenate(parts) if parts else np.zeros(0, x.dtype)
    assert len(y) == len(yref), (seed, kind, R, K)
    if len(y):
        assert relerr(y, yref) <= TOL, (seed, kind, R, K, cplx)


def test_device_views_unaligned(tg, orc):
    """device tensors that are only 4- or 8-byte aligned (odd offsets into a larger buffer)"""
    import torch
    rng = np.random.default_rng(5)
    x = rand(rng, 50003, False)
    xd = torch.from_numpy(x).cuda()
    h = orc.design_rif_fen(127, "lp", 0.05)
    for off in (1, 2, 3):
        f = tg.Fir(h, tg.F32, tg.FIR_AUTO)
        y = f.step(xd[off:])
        torch.cuda.synchronize()
        assert relerr(y.cpu().numpy(), orc.fir(h, x[off:])) <= TOL
    z, p, mn, md = orc.design_butter_lp(6, 0.2)
    ref = orc.SosChain(z, p, mn, md)
    co, gain, r1 = ref.coefs()
    g = tg.Sos(co, gain, tg.F32, r1)
    y = g.step(xd[3:])
    torch.cuda.synchronize()
    assert relerr(y.cpu().numpy(), ref.step(x[3:])) <= TOL
