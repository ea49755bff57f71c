"""Sizes past 2^31 elements (the domain's "maximum sizes" edge: libtsd caps a vector at 2^27
cfloat, tableau.cc:695-696, the GPU path does not): every operator is run ONCE on a vector of
more than 2^31 elements resident in HBM and checked (a) against the CPU oracle on slices --
head, around the 2^31 / 2^32-byte boundaries, tail -- and (b) through chunk invariance (one call
== the same handle stepped in pieces).  Catches 32-bit overflows in index arithmetic and grids."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5
BIG = (1 << 31) + 123_457          # elements


def relerr(y, ref):
    return float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture(scope="module")
def env():
    import torch
    import libtsd_amd as t
    assert t.device_count() >= 1
    free, _ = torch.cuda.mem_get_info(0)
    if free < 90 * (1 << 30):
        pytest.skip("needs 90 GiB of free HBM")
    return t, torch, torch.device("cuda", 0)


def fill(torch, dev, n, cplx, seed):
    """Deterministic pseudo-random data generated on the device in pieces (no 17 GiB temporaries)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.empty(n, dtype=torch.complex64 if cplx else torch.float32, device=dev)
    v = torch.view_as_real(x).reshape(-1) if cplx else x
    step = 1 << 28
    for i in range(0, v.shape[0], step):
        m = min(step, v.shape[0] - i)
        v[i:i + m] = torch.randn(m, device=dev, generator=g)
    return x


def slices(n, width):
    """Slice starts: head, the 2^31-element mark, 2^32-byte marks for 4/8-byte elements, tail."""
    marks = [0, (1 << 29) - width // 2, (1 << 30) - width // 2, (1 << 31) - width // 2, n - width]
    return [m for m in marks if 0 <= m <= n - width]


@pytest.mark.parametrize("method", [1, 2])
def test_fir_complex_past_2_31(env, orc, method):
    t, torch, dev = env
    K, W = 127, 20000
    h = orc.design_rif_fen(K, "lp", 0.02)
    x = fill(torch, dev, BIG, True, 11)
    y = torch.empty_like(x)
    t.Fir(h, t.C64, method).step(x, y)
    torch.cuda.synchronize()
    for s in slices(BIG, W):
        lo = max(0, s - (K - 1))
        xs = x[lo:s + W].cpu().numpy()
        ref = orc.fir(h, xs)[s - lo:]
        assert relerr(y[s:s + W].cpu().numpy(), ref) <= TOL, (method, s)
    # chunk invariance: the same stream in three ragged calls
    f = t.Fir(h, t.C64, method)
    y2 = torch.empty_like(x)
    cuts = [0, (1 << 30) + 77, (1 << 31) - 5, BIG]
    for a, b in zip(cuts[:-1], cuts[1:]):
        f.step(x[a:b], y2[a:b])
    torch.cuda.synchronize()
    for s in slices(BIG, W) + [c - W // 2 for c in cuts[1:-1]]:
        d = (y[s:s + W] - y2[s:s + W]).abs().max().item()
        assert d <= 1e-5 * 4.0, (method, s, d)


def test_sos_real_past_2_32_bytes(env, orc):
    t, torch, dev = env
    n, W, warm = BIG, 20000, 4096
    z, p, mn, md = orc.design_butter_lp(12, 0.25)
    co, gain, _ = orc.SosChain(z, p, mn, md).coefs()
    x = fill(torch, dev, n, False, 12)
    y = torch.empty_like(x)
    t.Sos(co, gain, t.F32).step(x, y)
    torch.cuda.synchronize()
    for s in slices(n, W):
        # a slice that does not start at 0 is warmed up on `warm` samples: the 12th-order
        # Butterworth's state (and the oracle's first-sample seed) decays below 1e-9 in 256
        lo = max(0, s - warm)
        ref = orc.SosChain(z, p, mn, md).step(x[lo:s + W].cpu().numpy())[s - lo:]
        assert relerr(y[s:s + W].cpu().numpy(), ref) <= TOL, s
    # chunk invariance (state carried exactly across ragged calls)
    f = t.Sos(co, gain, t.F32)
    y2 = torch.empty_like(x)
    cuts = [0, (1 << 30) + 77, (1 << 31) - 5, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        f.step(x[a:b], y2[a:b])
    torch.cuda.synchronize()
    for s in slices(n, W) + [c - W // 2 for c in cuts[1:-1]]:
        d = (y[s:s + W] - y2[s:s + W]).abs().max().item()
        assert d <= 1e-5 * 4.0, (s, d)


def test_resampler_past_2_31(env, orc):
    t, torch, dev = env
    ratio = np.float32(160.0) / np.float32(147.0)
    n = BIG
    x = fill(torch, dev, n, True, 13)
    r = t.Resampler(ratio, t.C64)
    y = r.step(x)
    torch.cuda.synchronize()
    nout = y.shape[0]
    assert nout > n
    # chunk invariance: the same stream in pieces on a fresh handle (64-bit positions and counts)
    r2 = t.Resampler(ratio, t.C64)
    cuts = [0, (1 << 29) + 3, (1 << 31) - 1, n]
    pos = 0
    W = 50000
    for a, b in zip(cuts[:-1], cuts[1:]):
        yp = r2.step(x[a:b])
        torch.cuda.synchronize()
        m = yp.shape[0]
        for s in [0, m // 2, m - W]:
            d = (yp[s:s + W] - y[pos + s:pos + s + W]).abs().max().item()
            assert d == 0.0, (a, s, d)
        pos += m
        del yp
    assert pos == nout
    # head against the oracle (sample-exact schedule + values)
    ro = orc.Resampler(float(ratio))
    ref = ro.step(x[:200000].cpu().numpy())
    assert relerr(y[:len(ref)].cpu().numpy(), ref) <= TOL
    # the float32 phase recurrence replayed by the oracle over ALL the inputs: exact output count,
    # then the tail values (the oracle's window is empty at the cut: skip its first K outputs)
    Wt = 100000
    ro = orc.Resampler(float(ratio))
    count0, _, _ = ro.schedule(n - Wt, want=False)
    ref = ro.step(x[n - Wt:].cpu().numpy())
    assert count0 + len(ref) == nout, (count0, len(ref), nout)
    skip = 40
    assert relerr(y[count0 + skip:].cpu().numpy(), ref[skip:]) <= TOL


def test_fft_batch_past_2_31(env, orc):
    t, torch, dev = env
    n = 1024
    batch = (1 << 21) + 3
    x = fill(torch, dev, n * batch, True, 14).reshape(batch, n)
    y = torch.empty_like(x)
    p = t.Fft(n, batch)
    p.step(x, True, y)
    torch.cuda.synchronize()
    for b in [0, 1, (1 << 20) - 1, 1 << 20, (1 << 21) - 1, 1 << 21, batch - 1]:
        ref = orc.fft(x[b].cpu().numpy())
        assert relerr(y[b].cpu().numpy(), ref) <= TOL, b
    # round trip on the whole batch (size-independent property)
    z = torch.empty_like(x)
    p.step(y, False, z)
    torch.cuda.synchronize()
    step = 1 << 18
    worst = 0.0
    for i in range(0, batch, step):
        worst = max(worst, (z[i:i + step] - x[i:i + step]).abs().max().item())
    assert worst <= 3e-5


@pytest.mark.parametrize("kind", ["decim", "halfband", "ups", "pick"])
def test_integer_rate_stages_past_2_31(env, orc, kind):
    """filtre_rif_decim / _demi_bande / _ups / decimateur on more than 2^31 complex samples:
    oracle on slices (started on a multiple of R, warmed up on 64 inputs) + chunk invariance."""
    t, torch, dev = env
    R = 2
    W = 20000
    n = BIG if kind != "ups" else (1 << 30) + 61_731        # the upsampler's OUTPUT passes 2^31
    x = fill(torch, dev, n, True, 21)
    if kind == "halfband":
        h = orc.design_rif_fen(15, "lp", 0.25)
    else:
        h = orc.design_rif_fen(15, "lp", 0.2)
    code = {"decim": t.POLY_DECIM, "halfband": t.POLY_HALFBAND, "ups": t.POLY_UPS, "pick": t.POLY_PICK}[kind]
    mk = lambda: t.PolyFir(code, t.C64, None if kind == "pick" else h, R)
    y = mk().step(x)
    torch.cuda.synchronize()
    nout = y.shape[0]
    if kind == "ups":
        assert nout == n * R
    elif kind == "pick":
        assert nout == (n - _pick_phase(t, R) + R - 1) // R     # every sample with index = phase (mod R)
    else:
        assert nout == n // R

    def oracle():
        if kind == "ups":
            return orc.PolyUps(h, R)
        if kind == "pick":
            return None
        return orc.PolyDecim(h, R, 1 if kind == "halfband" else 0)

    warm = 64
    for s in slices(n, W):
        s -= s % R
        lo = max(0, s - warm)
        xs = x[lo:s + W].cpu().numpy()
        o = oracle()
        if o is None:
            # decimateur keeps every R-th sample of the stream (filtre-rt.cc:127-169): bit-exact pick
            out_lo = (lo + R - 1) // R
            got = y[out_lo:out_lo + 100].cpu().numpy()
            phase = _pick_phase(t, R)
            exp = x[out_lo * R + phase: out_lo * R + phase + 100 * R: R].cpu().numpy()
            assert np.array_equal(got, exp), s
            continue
        ref = o.step(xs)
        if kind == "ups":
            a, b = s * R, (s + W) * R
            r = ref[(s - lo) * R:]
        else:
            a, b = s // R, (s + W) // R
            r = ref[(s - lo) // R:]
        g = y[a:b].cpu().numpy()
        m = min(len(g), len(r))
        assert relerr(g[:m], r[:m]) <= TOL, (kind, s)
    # chunk invariance with ragged cuts
    f = mk()
    cuts = [0, (1 << 29) + 3, n - 5, n]
    pos = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        yp = f.step(x[a:b])
        m = yp.shape[0]
        if m:
            w = min(W, m)
            for s in {0, max(0, m // 2 - w // 2), m - w}:
                d = (yp[s:s + w] - y[pos + s:pos + s + w]).abs().max().item()
                assert d == 0.0, (kind, a, s, d)
        pos += m
        del yp
    assert pos == nout


def _pick_phase(t, R):
    """Index (mod R) of the samples decimateur keeps, read off a tiny call."""
    x = np.arange(4 * R, dtype=np.float32)
    y = t.PolyFir(t.POLY_PICK, t.F32, None, R).step(x)
    return int(y[0])
