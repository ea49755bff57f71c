"""Worker of tests/test_sharded_ranks_gpu.py: launched by torch.distributed.run on 2 ranks that share
ONE GPU and talk over gloo (the N > 1 path of SURVEY.md section 8e with the HIP operators, not the
oracle).  Every rank filters ITS contiguous chunk of a common seeded stream with the halo it receives
from its left neighbour (libtsd_amd.sharding), rank 0 gathers the outputs and compares the
concatenation with the single-handle run on the whole stream."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import libtsd_amd as t
    from libtsd_amd import sharding
    # one device per rank where the box has them (a node: the ranks sit on distinct GPUs); they share device 0 on the one-GPU box
    dev = torch.device("cuda", rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(42)
    n = 600001
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    xr = np.ascontiguousarray(x.real)
    lo, hi = sharding.chunk_bounds(n, rank, world)
    fails = []

    def gather(y):
        """variable-length gather of device tensors to rank 0 through the host"""
        parts = [None] * world
        dist.gather_object(y.cpu().numpy(), parts if rank == 0 else None, dst=0)
        return np.concatenate(parts) if rank == 0 else None

    # ---- FIR, direct (bit-exact) and overlap-save: K-1-sample halo -> filter history
    h = np.hanning(127).astype(np.float32)
    h /= h.sum()
    for method in (t.FIR_DIRECT, t.FIR_OVERLAP_SAVE):
        f = t.Fir(h, t.C64, method)
        xc = torch.from_numpy(x[lo:hi].copy()).to(dev)
        halo = torch.zeros(126, dtype=torch.complex64, device=dev)
        sharding.exchange_left_halo(xc[-126:].clone(), halo, rank, world)
        if rank > 0:
            f.set_history(halo)
        got = gather(f.step(xc))
        if rank == 0:
            ref = t.Fir(h, t.C64, method).step(torch.from_numpy(x).to(dev)).cpu().numpy()
            if method == t.FIR_DIRECT and not np.array_equal(ref, got):
                fails.append("FIR direct: sharded output differs from the single handle")
            if np.abs(ref - got).max() > 2e-6 * np.abs(ref).max():
                fails.append(f"FIR method {method}: {np.abs(ref - got).max()}")

    # ---- the same with the exchange off the critical path: interior launched before the halo wait, edge behind it
    #      (sharding.OverlappedFir: what bench.py --gpus N runs) -- same outputs, bit for bit for the direct kernel
    for method in (t.FIR_DIRECT, t.FIR_OVERLAP_SAVE):
        ov = sharding.OverlappedFir(t, h, t.C64, method)
        xc = torch.from_numpy(x[lo:hi].copy()).to(dev)
        yc = torch.empty_like(xc)
        halo = torch.zeros(126, dtype=torch.complex64, device=dev)
        ex = sharding.start_halo_exchange(torch.view_as_real(xc[-126:].clone()), torch.view_as_real(halo), rank, world, result=halo)
        ov.step(xc, yc, ex, first=(rank == 0))
        got = gather(yc)
        if rank == 0:
            ref = t.Fir(h, t.C64, method).step(torch.from_numpy(x).to(dev)).cpu().numpy()
            if method == t.FIR_DIRECT and not np.array_equal(ref, got):
                fails.append("overlapped FIR direct: sharded output differs from the single handle")
            if np.abs(ref - got).max() > 2e-6 * np.abs(ref).max():
                fails.append(f"overlapped FIR method {method}: {np.abs(ref - got).max()}")

    # ---- SOS: warm-up halo
    from scipy.signal import butter
    sos = butter(12, 0.5, output="sos")
    co = np.array([[s[0], s[1], s[2], s[4], s[5]] for s in sos], np.float32)
    f = t.Sos(co, 1.0, t.F32)
    W = int(f.halo)
    xc = torch.from_numpy(xr[lo:hi].copy()).to(dev)
    halo = torch.zeros(W, dtype=torch.float32, device=dev)
    sharding.exchange_left_halo(xc[-W:].clone(), halo, rank, world)
    if rank > 0:
        f.reset()
        f.step(halo)
    got = gather(f.step(xc))
    if rank == 0:
        ref = t.Sos(co, 1.0, t.F32).step(torch.from_numpy(xr).to(dev)).cpu().numpy()
        if np.abs(ref - got).max() > 1e-6 * np.abs(ref).max():
            fails.append(f"SOS: {np.abs(ref - got).max()} (peak {np.abs(ref).max()})")

    ovs = sharding.OverlappedSos(t, co, 1.0, t.F32)
    yc = torch.empty_like(xc)
    halo = torch.zeros(W, dtype=torch.float32, device=dev)
    ex = sharding.start_halo_exchange(xc[-W:].clone(), halo, rank, world)
    ovs.step(xc, yc, ex, first=(rank == 0))
    got = gather(yc)
    if rank == 0 and np.abs(ref - got).max() > 1e-6 * np.abs(ref).max():
        fails.append(f"overlapped SOS: {np.abs(ref - got).max()} (peak {np.abs(ref).max()})")

    # ---- SOS with a long memory (first-order low-pass at fc = 1e-5: 5e5 samples; order 3 at 1e-4): no halo can warm it
    #      up -- exact exchange of the end states (ONE all_gather), two calls in a row (the stream state carried between)
    from oracle import pyoracle as orc
    for order, fc, tol in ((1, 1e-5, 2e-5), (3, 1e-4, 2e-3)):
        z, p, mn, md = orc.design_butter_lp(order, fc)
        co, gain, r1 = orc.SosChain(z, p, mn, md).coefs()
        f = t.Sos(co, gain, t.F32, r1)
        xo = xr + np.float32(0.5)
        got, etat = [], None
        for a, b in ((0, 250001), (250001, n)):
            lo2, hi2 = sharding.chunk_bounds(b - a, rank, world)
            xc = torch.from_numpy(xo[a + lo2:a + hi2].copy()).to(dev)
            yc, etat = sharding.sos_step_exact(f, xc, hi2 - lo2, rank, world, etat)
            got.append(gather(yc))
        if rank == 0:
            got = np.concatenate(got)
            ref = t.Sos(co, gain, t.F32, r1).step(torch.from_numpy(xo).to(dev)).cpu().numpy()
            if np.abs(ref - got).max() > tol * np.abs(ref).max():
                fails.append(f"SOS exact (order {order}, fc {fc}): {np.abs(ref - got).max() / np.abs(ref).max()}")

    # ---- resampler 160/147: seek to the chunk's stream position with the 14-sample window (bit-exact)
    ratio = np.float32(160.0) / np.float32(147.0)
    r = t.Resampler(ratio, t.C64)
    xc = torch.from_numpy(x[lo:hi].copy()).to(dev)
    halo = torch.zeros(14, dtype=torch.complex64, device=dev)
    sharding.exchange_left_halo(xc[-14:].clone(), halo, rank, world)
    r.seek(lo, halo if rank > 0 else None)
    got = gather(r.step(xc))
    if rank == 0:
        ref = t.Resampler(ratio, t.C64).step(torch.from_numpy(x).to(dev)).cpu().numpy()
        if len(ref) != len(got) or not np.array_equal(ref, got):
            fails.append(f"resampler: {len(got)} outputs vs {len(ref)}, equal = {len(ref) == len(got) and np.array_equal(ref, got)}")

    ovr = sharding.OverlappedResampler(t, ratio, t.C64)
    halo = torch.zeros(14, dtype=torch.complex64, device=dev)
    ex = sharding.start_halo_exchange(torch.view_as_real(xc[-14:].clone()), torch.view_as_real(halo), rank, world, result=halo)
    yc = torch.empty(int(ovr.main.out_count(hi - lo)) + 8, dtype=torch.complex64, device=dev)
    got = gather(ovr.step(xc, yc, lo, ex, first=(rank == 0)))
    if rank == 0 and (len(ref) != len(got) or not np.array_equal(ref, got)):
        fails.append(f"overlapped resampler: {len(got)} outputs vs {len(ref)}")

    ok = torch.tensor([0 if fails else 1])
    dist.broadcast(ok, 0)
    if rank == 0:
        print("RANKS_WORKER " + ("OK" if not fails else "FAILED: " + "; ".join(fails)), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(ok.item()) == 1 else 1)


if __name__ == "__main__":
    main()
