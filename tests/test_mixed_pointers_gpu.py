"""Mixed residency in ONE call (host input -> device output and the reverse): the staging helpers treat the two sides
independently -- a host input is copied in (through the bounce buffer when small), a device output is written in place
and the call returns without synchronising -- and the results are those of the all-host / all-device calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def crand(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


@pytest.mark.parametrize("n", [700, 5000, 300001])
def test_host_in_device_out_and_back(orc, n):
    import torch
    import libtsd_amd as t
    from scipy.signal import butter
    h = orc.design_rif_fen(63, "lp", 0.1)
    sos = butter(6, 0.3, output="sos")
    co = np.array([[s[0], s[1], s[2], s[4], s[5]] for s in sos], np.float32)
    x = crand(n, n)
    xd = torch.from_numpy(x).cuda()
    for mk in (lambda: t.Fir(h, t.C64, t.FIR_DIRECT), lambda: t.Fir(h, t.C64, t.FIR_OVERLAP_SAVE), lambda: t.Sos(co, 1.0, t.C64)):
        ref = mk().step(x)                                   # host -> host
        a = mk()
        yd = torch.empty(n, dtype=torch.complex64, device="cuda")
        # several calls in a row without a synchronisation in between: the small input's bounce slot is reused safely
        for _ in range(6):
            a.reset() if hasattr(a, "reset") else None
            a.step(x, yd)                                    # host -> device (asynchronous on the output side)
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), ref)
        b = mk()
        yh = np.empty(n, np.complex64)
        b.step(xd, yh)                                       # device -> host
        assert np.array_equal(yh, ref)
    r_ref = t.Resampler(1.37, t.C64).step(x)
    r = t.Resampler(1.37, t.C64)
    yo = torch.empty(r.out_count(n), dtype=torch.complex64, device="cuda")
    got = r.step(x, yo)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), r_ref)
