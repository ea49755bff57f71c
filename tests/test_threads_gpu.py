"""Several host threads through the C ABI at once, each with its own handles: the per-thread bounce buffers (small host
blocks), the per-call host pipes (large host vectors) and the reduction scratch blocks are pooled -- results must not
depend on what the other threads are doing, and short-lived threads must leave nothing behind."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_threads_small_and_large_host_calls(orc):
    import torch
    import libtsd_amd as t
    h = orc.design_rif_fen(31, "lp", 0.2)
    rng = np.random.default_rng(0)
    xs = rng.standard_normal(4096).astype(np.float32)
    xl = (rng.standard_normal(3 << 20) + 1j * rng.standard_normal(3 << 20)).astype(np.complex64)      # 24 MiB: pipelined
    ref_s = orc.fir(h, xs)
    ref_l = t.Fir(h, t.C64, t.FIR_DIRECT).step(torch.from_numpy(xl).cuda()).cpu().numpy()
    errs = []

    def work(k):
        try:
            f = t.Fir(h, t.F32, t.FIR_DIRECT)
            for _ in range(200):                       # small host blocks: bounce buffers
                f.reset()
                y = f.step(xs)
                if np.abs(y - ref_s).max() > 1e-5 * np.abs(ref_s).max():
                    raise AssertionError(f"thread {k}: small step deviates")
            g = t.Fir(h, t.C64, t.FIR_DIRECT)          # a large host vector: a host pipe borrowed for the call
            yl = g.step(xl)
            if not np.array_equal(yl, ref_l):
                raise AssertionError(f"thread {k}: pipelined step deviates")
            s, mx, mn, im = t.vec_reduce(torch.from_numpy(xs).cuda())
            if mx != xs.max() or im != int(np.argmax(xs)):
                raise AssertionError(f"thread {k}: reduction deviates")
        except Exception as e:                          # noqa: BLE001
            errs.append(repr(e))

    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):                                  # three generations of short-lived threads
        th = [threading.Thread(target=work, args=(k,)) for k in range(5)]
        for q in th:
            q.start()
        for q in th:
            q.join()
    assert not errs, errs
    torch.cuda.synchronize()
    # pooled blocks are reused by the later generations: the device memory held does not grow with the thread count
    free1 = torch.cuda.mem_get_info()[0]
    for _ in range(2):
        th = [threading.Thread(target=work, args=(k,)) for k in range(5)]
        for q in th:
            q.start()
        for q in th:
            q.join()
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    assert not errs, errs
    # 25 threads have come and gone.  What stays is bounded by the CONCURRENCY (at most 5 calls at once: 5 host pipes of
    # 4 x 16 MiB, 5 reduction blocks), not by the number of threads: a per-thread leak of one pipe would be 1.6 GiB here
    assert free0 - free2 <= (600 << 20), f"device memory kept growing: {free0 >> 20} -> {free1 >> 20} -> {free2 >> 20} MiB"
