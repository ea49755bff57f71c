"""GPU parity of the integer-rate polyphase stages, the plain decimator and the generic
direct-form-I IIR against the CPU oracle (polyphase.cc:54-341, filtre-rt.cc:127-289)."""
import numpy as np
import pytest

from conftest import perf_guard

pytestmark = pytest.mark.gpu
TOL = 1e-5


def relerr(y, ref):
    return float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))


def rand(n, cplx, seed):
    rng = np.random.default_rng(seed)
    if cplx:
        return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    return rng.standard_normal(n).astype(np.float32)


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1
    return t


def chunks(f, x, bs):
    return np.concatenate([f.step(x[o:o + bs].copy()) for o in range(0, len(x), bs)])


# test_decimateur (test-filtres.cc:186-200): R = 3 on 0..89 in blocks of 4 -> exactly 0,3,...,87
def test_decimateur_exact(tg, orc):
    x = np.arange(90, dtype=np.float32)
    y = chunks(tg.PolyFir(tg.POLY_PICK, tg.F32, None, 3), x, 4)
    assert len(y) == 30 and np.array_equal(y, np.arange(0, 90, 3, dtype=np.float32))
    for R in (2, 5, 7):
        xx = rand(1000, True, R)
        assert np.array_equal(chunks(tg.PolyFir(tg.POLY_PICK, tg.C64, None, R), xx, 33), xx[::R])


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("R", [2, 3, 4, 5, 8])
@pytest.mark.parametrize("bs", [100000, 1000, 37])
def test_rif_decim(tg, orc, cplx, R, bs):        # test_filtre_rif_decim (test-ra.cc:176-191)
    h = orc.design_rif_fen(15, "lp", 0.5 / R)
    x = rand(20000, cplx, R)
    ref = orc.PolyDecim(h, R, 0).step(x)
    y = chunks(tg.PolyFir(tg.POLY_DECIM, tg.C64 if cplx else tg.F32, h, R), x, bs)
    assert len(y) == len(ref) and relerr(y, ref) <= TOL


# decimators of rate 2 / 4 / 8 up to 64 taps run on decim_direct_kernel (round 4: the direct FIR kernel's scheme on the kept positions):
# tap counts around the 16- / 32-sample window chunks, streams cut in ragged calls (every phase of the decimation counter, tiles that
# start in the history), one large call (interior tiles, 16-B loads), and THE SAME BITS as the fused kernel it replaces below 32 taps
# (same products in the same order)
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("R,K", [(2, 1), (2, 7), (2, 15), (2, 16), (2, 17), (2, 33), (2, 64), (4, 15), (4, 31), (4, 64), (8, 15), (8, 47)])
def test_rif_decim_direct_kernel(tg, orc, cplx, R, K, monkeypatch):
    rng = np.random.default_rng(100 * R + K)
    h = rng.standard_normal(K).astype(np.float32) / K
    n = 300000 + 13
    x = rand(n, cplx, 7 * R + K)
    dt = tg.C64 if cplx else tg.F32
    ref = orc.PolyDecim(h, R, 0).step(x)
    f = tg.PolyFir(tg.POLY_DECIM, dt, h, R)
    cuts = [0, 1, 2, 3, 5, 100, 101, 4097, 4100, 70000, 70000 + R + 1, n]
    y = np.concatenate([f.step(x[a:b].copy()) for a, b in zip(cuts[:-1], cuts[1:])])
    assert len(y) == len(ref) and relerr(y, ref) <= TOL
    y1 = tg.PolyFir(tg.POLY_DECIM, dt, h, R).step(x)
    assert np.array_equal(y1.view(np.uint32), y.view(np.uint32))          # however the stream is cut
    monkeypatch.setenv("TSDGPU_POLY_NO_DIRECT", "1")
    y2 = tg.PolyFir(tg.POLY_DECIM, dt, h, R).step(x)
    monkeypatch.delenv("TSDGPU_POLY_NO_DIRECT")
    if K < 32:
        assert np.array_equal(y1.view(np.uint32), y2.view(np.uint32))     # the fused kernel's bits (same products, same order)
    else:
        assert relerr(y1, y2) <= 1e-6                                      # (32 taps and more used to go by polyphase rows: another order)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("K", [15, 17, 31])
def test_rif_demi_bande(tg, orc, cplx, K):       # test_filtre_rif_demi_bande (test-ra.cc:166-173)
    h = orc.design_rif_fen(K, "lp", 0.25)
    x = rand(30001, cplx, K)
    ref = orc.PolyDecim(h, 2, 1).step(x)
    y = chunks(tg.PolyFir(tg.POLY_HALFBAND, tg.C64 if cplx else tg.F32, h), x, 777)
    assert len(y) == len(ref) and relerr(y, ref) <= TOL


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("R", [2, 3, 4])
def test_rif_ups(tg, orc, cplx, R):              # test_filtre_rif_ups (test-ra.cc:193-199)
    h = orc.design_rif_fen(15, "lp", 0.5 / R)
    x = rand(10000, cplx, R + 10)
    ref = orc.PolyUps(h, R).step(x)
    y = chunks(tg.PolyFir(tg.POLY_UPS, tg.C64 if cplx else tg.F32, h, R), x, 999)
    assert len(y) == len(ref) == len(x) * R and relerr(y, ref) <= TOL


# upsamplers of rate 2 / 4 with branches of up to 32 taps run on ups_direct_kernel (round 4): tap counts that leave the last branch
# short, one tap per branch, the 32-tap limit and beyond it (fused kernel), ragged calls, one large call, the fused kernel's bits
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("R,K", [(2, 1), (2, 2), (2, 15), (2, 31), (2, 33), (2, 64), (2, 65), (4, 15), (4, 30), (4, 127), (4, 129)])
def test_rif_ups_direct_kernel(tg, orc, cplx, R, K, monkeypatch):
    rng = np.random.default_rng(10 * R + K)
    h = rng.standard_normal(K).astype(np.float32) / K
    n = 200000 + 7
    x = rand(n, cplx, 3 * R + K)
    dt = tg.C64 if cplx else tg.F32
    ref = orc.PolyUps(h, R).step(x)
    f = tg.PolyFir(tg.POLY_UPS, dt, h, R)
    cuts = [0, 1, 2, 5, 100, 2048, 2049, 6000, 70001, n]
    y = np.concatenate([f.step(x[a:b].copy()) for a, b in zip(cuts[:-1], cuts[1:])])
    assert len(y) == len(ref) == n * R and relerr(y, ref) <= TOL
    y1 = tg.PolyFir(tg.POLY_UPS, dt, h, R).step(x)
    assert np.array_equal(y1.view(np.uint32), y.view(np.uint32))          # however the stream is cut
    monkeypatch.setenv("TSDGPU_POLY_NO_DIRECT", "1")
    y2 = tg.PolyFir(tg.POLY_UPS, dt, h, R).step(x)
    monkeypatch.delenv("TSDGPU_POLY_NO_DIRECT")
    assert np.array_equal(y1.view(np.uint32), y2.view(np.uint32))         # the fused kernel's bits (same products, same order)


# rates / tap counts beyond the reference's tests: long decimation (fewer outputs per workgroup),
# a rate the fused kernel hands back to the composed path (R = 100), many-tap branches, tiny chunks
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("R,K", [(16, 63), (60, 15), (100, 31), (7, 200)])
def test_rif_decim_wide(tg, orc, cplx, R, K):
    h = orc.design_rif_fen(K, "lp", 0.5 / R)
    x = rand(50000, cplx, R + K)
    ref = orc.PolyDecim(h, R, 0).step(x)
    y = chunks(tg.PolyFir(tg.POLY_DECIM, tg.C64 if cplx else tg.F32, h, R), x, 12345)
    assert len(y) == len(ref) and relerr(y, ref) <= TOL


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("R,K,bs", [(8, 127, 5000), (2, 15, 1), (5, 33, 7)])
def test_rif_ups_wide(tg, orc, cplx, R, K, bs):
    h = orc.design_rif_fen(K, "lp", 0.5 / R)
    x = rand(4000 if bs > 1 else 100, cplx, R + K)
    ref = orc.PolyUps(h, R).step(x)
    y = chunks(tg.PolyFir(tg.POLY_UPS, tg.C64 if cplx else tg.F32, h, R), x, bs)
    assert len(y) == len(ref) == len(x) * R and relerr(y, ref) <= TOL


# test_filtre_rii (test-filtres.cc:556-606): one-pole smoother vs the closed recurrence
def test_filtre_rii(tg, orc):
    a = np.float32(0.1)
    n = 20
    y = tg.Rii([a], [1.0, -(1 - a)], tg.F32).step(np.ones(n, np.float32))
    yref = np.empty(n, np.float32)
    yref[0] = a
    for i in range(1, n):
        yref[i] = yref[i - 1] + a * (1 - yref[i - 1])
    assert np.abs(yref - y).max() <= 1e-6
    # low order (<= 2 poles, <= 3 zeros) runs on the block-parallel SOS kernel: long vector, chunks
    for nu, de in (([0.1], [1.0, -0.9]), ([0.2, 0.1, 0.05], [2.0, -1.2, 0.5]), ([1.0, -1.0], [1.0, -0.95])):
        x = rand(200000, False, 4)
        ref = orc.Rii(nu, de).step(x)
        assert relerr(chunks(tg.Rii(nu, de, tg.F32), x, 70001), ref) <= TOL
    # higher order, chunked, vs the oracle's FiltreRII
    nu, de = [0.2, 0.3, 0.1, -0.05], [1.0, -0.9, 0.5, -0.1]
    x = rand(5000, False, 3)
    ref = orc.Rii(nu, de).step(x)
    assert relerr(chunks(tg.Rii(nu, de, tg.F32), x, 613), ref) <= TOL


# FiltreRII of higher order (filtre-rt.cc:177-289): FIR kernel + the sequential recursive part,
# bit-identical operation order; orders in every register bucket (<= 4, 8, 16, 32) and beyond,
# real and complex data, ragged chunks.  The denominator is a product of stable real poles.
@pytest.mark.parametrize("order", [3, 5, 9, 20, 40])
@pytest.mark.parametrize("cplx", [False, True])
def test_filtre_rii_high_order(tg, orc, order, cplx):
    rng = np.random.default_rng(order)
    r = 0.6 if order < 20 else 0.3                            # (a direct form of order 20+ is badly conditioned: keep the poles small)
    poles = rng.uniform(-r, r, order)
    de = np.poly(poles).astype(np.float32)
    de = (de * np.float32(1.5)).astype(np.float32)            # denom[0] != 1: the division is exercised
    nu = rng.standard_normal(order // 2 + 2).astype(np.float32)
    n = 30000 if order <= 20 else 6000
    x = rand(n, cplx, order)
    if cplx:   # real coefficients on complex data act on the two components separately
        ref = (orc.Rii(nu, de).step(x.real.copy()) + 1j * orc.Rii(nu, de).step(x.imag.copy())).astype(np.complex64)
    else:
        ref = orc.Rii(nu, de).step(x)
    y = chunks(tg.Rii(nu, de, tg.C64 if cplx else tg.F32), x, 7001)
    assert relerr(y, ref) <= TOL


# Block-parallel FiltreRII (row a5): the denominator factored into zero-seeded DF1 sections on the SOS
# kernel whenever a create-time check shows the cascade reproduces the direct form; 2^22 samples
# against the oracle's literal recursion, ragged chunks, and the path actually taken.
@pytest.mark.parametrize("order,cplx", [(3, False), (4, True), (6, False), (6, True), (9, False), (12, False)])
def test_filtre_rii_block_parallel(tg, orc, order, cplx):
    rng = np.random.default_rng(100 + order)
    # conjugate pole pairs of radius 0.5..0.9 (+ a real pole for odd orders): what an IIR design gives
    ang = rng.uniform(0.2, 2.8, order // 2)
    rad = rng.uniform(0.5, 0.9, order // 2)
    poles = np.concatenate([rad * np.exp(1j * ang), rad * np.exp(-1j * ang), [0.7] if order & 1 else []])
    de = (np.real(np.poly(poles)) * 1.25).astype(np.float32)
    nu = rng.standard_normal(order // 2 + 2).astype(np.float32)
    n = 1 << 20
    x = rand(n, cplx, order)
    if cplx:
        ref = (orc.Rii(nu, de).step(x.real.copy()) + 1j * orc.Rii(nu, de).step(x.imag.copy())).astype(np.complex64)
    else:
        ref = orc.Rii(nu, de).step(x)
    f = tg.Rii(nu, de, tg.C64 if cplx else tg.F32)
    assert f.path in (0, 1), f"order {order}: the literal recursion was chosen (path {f.path})"
    assert relerr(chunks(f, x, 300007), ref) <= TOL


@pytest.mark.parametrize("order", [1, 3, 6, 11, 13])
def test_filtre_rii_block_parallel_complex_coefficients(tg, orc, order):
    """filtre_rii<cfloat, cfloat> (filtre-rt.cc:177-289, :795) block-parallel: one-sided complex poles --
    no conjugate partners -- as first-order complex sections (tsdgpu_rii_path 3); 2^20 samples against the oracle's literal
    recursion in ragged chunks that cut sub-tiles and chunk borders, in place, and an aligned / unaligned output view."""
    import torch
    rng = np.random.default_rng(300 + order)
    poles = rng.uniform(0.3, 0.9 if order <= 6 else 0.75, order) * np.exp(1j * rng.uniform(-3.0, 3.0, order))
    de = (np.poly(poles) * (1.25 - 0.5j)).astype(np.complex64)
    nu = (rng.standard_normal(order // 2 + 2) + 1j * rng.standard_normal(order // 2 + 2)).astype(np.complex64)
    n = 1 << 20
    x = rand(n, True, order)
    ref = orc.RiiC(nu, de).step(x)
    f = tg.Rii(nu, de, tg.C64)
    # (beyond order ~14 with poles up to 0.75 the complex direct form is so ill-conditioned in float32 that two orderings of
    # its own sums differ by 2e-5: the create-time check then keeps the literal recursion, which is not what this test is about)
    assert f.path == 3, f"order {order}: path {f.path}"
    assert relerr(chunks(f, x, 300007), ref) <= TOL
    f2 = tg.Rii(nu, de, tg.C64)
    y = x.copy()
    f2.step(y[:1000], y[:1000])                              # in place, a call shorter than a sub-tile
    f2.step(y[1000:], y[1000:])
    assert relerr(y, ref) <= TOL
    f3 = tg.Rii(nu, de, tg.C64)
    xd = torch.from_numpy(np.concatenate([np.zeros(1, np.complex64), x])).cuda()
    yd = torch.zeros(n + 1, dtype=torch.complex64, device="cuda")
    f3.step(xd[1:], yd[1:])                                  # views 8 bytes off a 16-B boundary
    assert relerr(yd[1:].cpu().numpy(), ref) <= TOL


def test_filtre_rii_complex_order6_2p24_under_1ms(tg):
    """filtre_rii<cfloat, cfloat> of order 6 on 2^24 samples in < 1 ms (the literal kernel: ~2.3 s)."""
    import torch
    rng = np.random.default_rng(5)
    poles = rng.uniform(0.5, 0.85, 6) * np.exp(1j * rng.uniform(-3.0, 3.0, 6))
    de = np.poly(poles).astype(np.complex64)
    f = tg.Rii(np.array([1.0, 0.5j, 0.25], np.complex64), de, tg.C64)
    assert f.path == 3
    x = torch.view_as_complex(torch.randn(1 << 24, 2, device="cuda"))
    y = torch.empty_like(x)
    for _ in range(3):
        f.step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f.step(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"filtre_rii<cfloat,cfloat> order 6, 2^24 samples: {ms:.3f} ms")
    perf_guard(ms < 1.0, f"filtre_rii<cfloat,cfloat> order 6, 2^24 samples: {ms:.3f} ms")


def test_filtre_rii_literal_fallback_and_env(tg, orc, monkeypatch):
    # an unstable denominator (a pole outside the unit circle) is left to the literal recursion
    de = np.poly([1.05, 0.5, -0.3]).astype(np.float32)
    f = tg.Rii([1.0, 0.2, 0.1, 0.3], de, tg.F32)
    assert f.path == 2
    x = rand(300, False, 5)
    assert relerr(f.step(x), orc.Rii([1.0, 0.2, 0.1, 0.3], de).step(x)) <= TOL


def test_filtre_rii_complex_coefficients(tg, orc):
    # filtre_rii<cfloat,cfloat>: one-sided (analytic) poles -> complex denominator: first-order complex sections
    poles = np.array([0.6 * np.exp(0.7j), 0.5 * np.exp(2.1j), 0.3 + 0j])
    de = (np.poly(poles) * (1.5 - 0.5j)).astype(np.complex64)
    nu = np.array([0.3 + 0.1j, -0.2j, 0.5, 0.1 - 0.4j], np.complex64)
    x = rand(20000, True, 7)
    ref = orc.RiiC(nu, de).step(x)
    f = tg.Rii(nu, de, tg.C64)
    assert f.path == 3
    assert relerr(chunks(f, x, 3001), ref) <= TOL
    # a complex pole outside the circle: the literal complex recursion
    de_u = np.poly(np.array([1.02 * np.exp(0.3j), 0.4j])).astype(np.complex64)
    f_u = tg.Rii(nu, de_u, tg.C64)
    assert f_u.path == 2
    assert relerr(f_u.step(x[:400]), orc.RiiC(nu, de_u).step(x[:400])) <= TOL
    # complex-typed coefficients whose imaginary parts vanish take the real (block-parallel) plan
    de_r = np.real(np.poly([0.6 * np.exp(0.7j), 0.6 * np.exp(-0.7j), 0.4])).astype(np.complex64)
    f2 = tg.Rii(nu.real.astype(np.complex64), de_r, tg.C64)
    assert f2.path in (0, 1)
    ref2 = orc.RiiC(nu.real.astype(np.complex64), de_r).step(x)
    assert relerr(f2.step(x), ref2) <= TOL


def test_filtre_rii_order6_2p26_under_2ms(tg):
    """order 6 on 2^26 floats in < 2 ms (the literal kernel needed ~9 s)."""
    import torch
    dev = torch.device("cuda", 0)
    ang, rad = np.array([0.5, 1.2, 2.0]), np.array([0.8, 0.7, 0.85])
    de = np.real(np.poly(np.concatenate([rad * np.exp(1j * ang), rad * np.exp(-1j * ang)]))).astype(np.float32)
    f = tg.Rii([1.0, 0.5, 0.25, 0.1], de, tg.F32)
    assert f.path == 1
    x = torch.randn(1 << 26, device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        f.step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f.step(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"filtre_rii order 6, 2^26 floats: {ms:.3f} ms")
    perf_guard(ms < 2.0, f"filtre_rii order 6, 2^26 floats: {ms:.3f} ms")


def test_filtre_rii_high_order_is_not_a_cliff(tg):
    """2^22 samples through a 6th-order direct-form recursion: seconds with the per-sample
    global-memory kernel, a fraction of a second with the tiled one."""
    import time
    import torch
    dev = torch.device("cuda", 0)
    de = np.poly([0.5, -0.4, 0.3, 0.2, -0.1, 0.6]).astype(np.float32)
    f = tg.Rii([1.0, 0.5], de, tg.F32)
    x = torch.randn(1 << 22, device=dev)
    f.step(x[:1000])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.step(x)
    torch.cuda.synchronize()
    perf_guard(time.perf_counter() - t0 < 1.5, "6th-order direct form on 2^22 samples took more than 1.5 s")
