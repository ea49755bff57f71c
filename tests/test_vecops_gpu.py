"""Element-wise arithmetic on resident vectors (tsdgpu_vec_op, csrc/vecops.hip) against numpy: the same IEEE operations
as the host loops of the C++ layer -- exact for everything but abs (hypot of two libraries) and the complex quotient."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    import libtsd_amd as t
    assert t.device_count() >= 1
    return t


@pytest.mark.parametrize("n", [1, 255, 256, 257, 100003])
def test_vec_ops_match_numpy(tg, n):
    import torch
    rng = np.random.default_rng(n)
    xc = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    zc = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    xr, zr = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    g = lambda a: torch.from_numpy(a).cuda()
    h = lambda t_: t_.cpu().numpy()
    # numpy evaluates complex64 products as (ac - bd, ad + bc) in float32 too
    prod = lambda a, b: ((a.real * b.real - a.imag * b.imag) + 1j * (a.real * b.imag + a.imag * b.real)).astype(np.complex64)
    assert np.array_equal(h(tg.vec_op("reverse", g(xc))), xc[::-1])
    assert np.array_equal(h(tg.vec_op("reverse", g(xr))), xr[::-1])
    assert np.array_equal(h(tg.vec_op("add", g(xc), g(zc))), xc + zc)
    assert np.array_equal(h(tg.vec_op("sub", g(xr), g(zr))), xr - zr)
    assert np.array_equal(h(tg.vec_op("mul", g(xr), g(zr))), xr * zr)
    assert np.array_equal(h(tg.vec_op("mul", g(xc), g(zc))), prod(xc, zc))
    s = np.complex64(0.3 - 1.7j)
    assert np.array_equal(h(tg.vec_op("scale", g(xc), scalar=complex(s))), prod(xc, np.full(n, s, np.complex64)))
    assert np.array_equal(h(tg.vec_op("scale", g(xr), scalar=2.5)), xr * np.float32(2.5))
    assert np.array_equal(h(tg.vec_op("div", g(xr), scalar=3.0)), xr / np.float32(3.0))
    q = (xc.astype(np.complex128) / complex(s)).astype(np.complex64)
    assert np.abs(h(tg.vec_op("div", g(xc), scalar=complex(s))) - q).max() <= 1.2e-7 * np.abs(q).max()
    assert np.array_equal(h(tg.vec_op("neg", g(xc))), -xc)
    assert np.array_equal(h(tg.vec_op("abs2", g(xc))), xc.real * xc.real + xc.imag * xc.imag)
    assert np.abs(h(tg.vec_op("abs", g(xc))) - np.abs(xc)).max() <= 1.2e-7 * np.abs(xc).max()
    assert np.array_equal(h(tg.vec_op("abs", g(xr))), np.abs(xr))
    assert np.array_equal(h(tg.vec_op("real", g(xc))), xc.real)
    assert np.array_equal(h(tg.vec_op("imag", g(xc))), xc.imag)
    assert np.array_equal(h(tg.vec_op("to_complex", g(xr))), xr.astype(np.complex64))
    assert np.array_equal(h(tg.vec_op("conj", g(xc))), np.conj(xc))
    # in place
    a = g(xc)
    tg.vec_op("mul", a, g(zc), out=a)
    assert np.array_equal(h(a), prod(xc, zc))


def test_vec_op_refusals(tg):
    import torch
    x = torch.zeros(16, dtype=torch.complex64, device="cuda")
    with pytest.raises(tg.TsdGpuError):
        tg.vec_op("reverse", x, out=x)                         # not in place
    with pytest.raises(tg.TsdGpuError):
        tg.vec_op("add", x)                                    # second operand missing
    with pytest.raises(tg.TsdGpuError):
        tg.vec_op("real", torch.zeros(16, device="cuda"))      # real() of a real vector
    with pytest.raises(tg.TsdGpuError):
        tg.vec_op("add", x, torch.zeros(16, dtype=torch.complex64), out=x)   # host operand: resident vectors only


@pytest.mark.parametrize("n", [1, 255, 70000, (1 << 22) + 3])
def test_vec_reduce(tg, n):
    import torch
    rng = np.random.default_rng(n)
    xr = rng.standard_normal(n).astype(np.float32)
    xc = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    s, mx, mn, im = tg.vec_reduce(torch.from_numpy(xr).cuda())
    assert abs(s.real - xr.astype(np.float64).sum()) <= 1e-9 * max(1.0, np.abs(xr).sum())
    assert mx == xr.max() and mn == xr.min() and im == int(np.argmax(xr))
    s, _, _, _ = tg.vec_reduce(torch.from_numpy(xc).cuda())
    ref = xc.astype(np.complex128).sum()
    assert abs(s - ref) <= 1e-9 * max(1.0, np.abs(xc).sum())
    flat = np.zeros(5000, np.float32)
    flat[[1234, 4321]] = 2.0
    assert tg.vec_reduce(torch.from_numpy(flat).cuda())[3] == 1234          # ties: the first one
