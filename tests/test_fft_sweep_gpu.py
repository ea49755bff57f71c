"""A random sweep of FFT sizes -- the plan-selection boundaries of every round included -- against the oracle's restatement of the
reference's plan (radix-2, even / odd split, float32-chirp Bluestein): whatever kernel a size lands on, the result is the
reference's within the tolerance of the path (1e-5, the sizes that reproduce the reference's float32 chirp included)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOUNDARIES = [8176, 16368, 8190, 16380, 1023 * 2, 1023 * 4, 1023 * 8, 511 * 16, 255 * 16, 127 * 16, 63 * 16, 33 * 16, 33 * 8, 4095 * 2,
              2047 * 4, 2047 * 8, 2049 * 2, 4097 * 2, 8191 * 2, 12288, 24576, 96 * 3, 16 * 3, 32 * 3, 32 * 5, 32 * 7, 128 * 7, 256 * 9, 512 * 9,
              1024 * 9, 4096 * 3, 2048 * 5, 1024 * 5, 64 * 9]


def test_fft_random_sizes_against_the_oracle():
    import libtsd_amd as t
    from oracle import pyoracle as orc
    rng = np.random.default_rng(5)
    sizes = sorted(set(int(v) for v in rng.integers(2, 20000, 260)) | set(BOUNDARIES))
    bad = []
    for n in sizes:
        batch = 3
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
        p = t.Fft(n, batch)
        y = p.step(x, True)
        ref = np.stack([orc.fft(x[b], True) for b in range(batch)])
        e = float(np.abs(y - ref).max() / np.abs(ref).max())
        z = p.step(y, False)
        zr = np.stack([orc.fft(ref[b], False) for b in range(batch)])
        e2 = float(np.abs(z - zr).max() / np.abs(zr).max())
        if e > 1e-5 or e2 > 1e-5:
            bad.append((n, e, e2))
        p.close()
    assert not bad, bad
