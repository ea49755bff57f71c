"""Worker of tests/test_rccl_world1_gpu.py: ONE rank, backend "nccl" (= RCCL on ROCm), on the one GPU of the test box.
Runs, on device tensors, every torch.distributed call the multi-rank path of libtsd_amd.sharding / bench.py makes, so that
librccl is loaded and each call signature, dtype (complex samples travel as their float32 view) and stream interaction has
executed once before a multi-GPU node ever sees it:
  * all_reduce(MAX) of a float64 scalar            (sharding.max_over_ranks: the bench's max-over-ranks time)
  * all_gather of an int64 length and of a float32 SOS state vector   (sharding.sos_step_exact)
  * batch_isend_irecv: a send and a receive of a 126-sample halo to / from itself in ONE group (sharding.start_halo_exchange)
  * barrier
and the overlapped steps (interior launched before the halo wait, edge after) against the single-handle result."""
import datetime
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29577")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    import libtsd_amd as t
    from libtsd_amd import sharding
    fails = []
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    # ---- all_reduce(MAX), barrier
    v = sharding.max_over_ranks(1.2345, dev, 1, force=True)
    if abs(v - 1.2345) > 1e-12:
        fails.append(f"all_reduce(MAX): {v}")
    dist.barrier()

    # ---- halo: self send / receive of 126 complex samples in one batched group, on a side stream too
    rng = np.random.default_rng(3)
    n, K = 300000, 127
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    xd = torch.from_numpy(x).to(dev)
    for use_side_stream in (False, True):
        st = torch.cuda.Stream(dev) if use_side_stream else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            tail = torch.view_as_real(xd[n - (K - 1):].clone())
            halo_c = torch.zeros(K - 1, dtype=torch.complex64, device=dev)
            got = sharding.exchange_left_halo(tail, torch.view_as_real(halo_c), 0, 1, ring=True)
            st.synchronize()
            if not np.array_equal(halo_c.cpu().numpy(), x[n - (K - 1):]) or got.data_ptr() != halo_c.data_ptr():
                fails.append(f"self send/recv of the halo (side stream: {use_side_stream})")

    # ---- overlapped FIR step on the ring (circular stream: the halo of the chunk is its own tail), both methods
    h = np.hanning(K).astype(np.float32)
    h /= h.sum()
    for method in (t.FIR_DIRECT, t.FIR_OVERLAP_SAVE):
        ref_f = t.Fir(h, t.C64, method)
        ref_f.set_history(xd[n - (K - 1):].clone())
        ref = ref_f.step(xd).cpu().numpy()
        ov = sharding.OverlappedFir(t, h, t.C64, method)
        y = torch.empty_like(xd)
        for _ in range(3):                                        # (several steps in flight: works and streams reused)
            tail = torch.view_as_real(xd[n - (K - 1):].clone())
            halo_c = torch.zeros(K - 1, dtype=torch.complex64, device=dev)
            ex = sharding.start_halo_exchange(tail, torch.view_as_real(halo_c), 0, 1, ring=True, result=halo_c)
            ov.step(xd, y, ex, first=False)
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        if method == t.FIR_DIRECT and not np.array_equal(got, ref):
            fails.append("overlapped FIR (direct): differs from the single handle")
        if np.abs(got - ref).max() > 2e-6 * np.abs(ref).max():
            fails.append(f"overlapped FIR method {method}: {np.abs(got - ref).max()}")

    # ---- overlapped resampler on the ring: bit for bit the single call from the same position and window
    ratio = np.float32(160.0) / np.float32(147.0)
    pos = 123457
    r0 = t.Resampler(ratio, t.C64)
    r0.seek(pos, xd[n - 14:].clone())
    ref = r0.step(xd).cpu().numpy()
    ovr = sharding.OverlappedResampler(t, ratio, t.C64)
    y = torch.empty(len(ref) + 8, dtype=torch.complex64, device=dev)
    tail = torch.view_as_real(xd[n - 14:].clone())
    halo_c = torch.zeros(14, dtype=torch.complex64, device=dev)
    ex = sharding.start_halo_exchange(tail, torch.view_as_real(halo_c), 0, 1, ring=True, result=halo_c)
    got = ovr.step(xd, y, pos, ex, first=False)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    if len(got) != len(ref) or not np.array_equal(got, ref):
        fails.append(f"overlapped resampler: {len(got)} outputs vs {len(ref)}")

    # ---- overlapped SOS (warm-up halo) on the ring
    from scipy.signal import butter
    sos = butter(12, 0.5, output="sos")
    co = np.array([[s[0], s[1], s[2], s[4], s[5]] for s in sos], np.float32)
    xr = torch.from_numpy(np.ascontiguousarray(x.real)).to(dev)
    f0 = t.Sos(co, 1.0, t.F32)
    W = int(f0.halo)
    scratch = torch.empty(W, dtype=torch.float32, device=dev)
    f0.step(xr[n - W:].clone(), scratch)
    ref = f0.step(xr).cpu().numpy()
    ovs = sharding.OverlappedSos(t, co, 1.0, t.F32)
    y = torch.empty_like(xr)
    halo = torch.zeros(W, dtype=torch.float32, device=dev)
    ex = sharding.start_halo_exchange(xr[n - W:].clone(), halo, 0, 1, ring=True)
    ovs.step(xr, y, ex, first=False)
    torch.cuda.synchronize()
    err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
    if err > 1e-6:
        fails.append(f"overlapped SOS: {err}")

    # ---- the same steps PIPELINED the way bench.py runs them at N > 1: the exchange on its own stream through two alternating halo
    # buffers (HaloPipe), posted after the interior launch, the edge on a stream of its own -- five steps in a row on a
    # non-default compute stream, every step's output against the single handle after wait_outputs()
    comp = torch.cuda.Stream(dev)
    with torch.cuda.stream(comp):
        ovp = sharding.OverlappedFir(t, h, t.C64, t.FIR_OVERLAP_SAVE, edge_stream=True)
        ovp.input_ready = True
        pipe = sharding.HaloPipe(torch.zeros(K - 1, 2, device=dev), 0, 1, ring=True, complex_view=True)
        if not pipe.piped:
            fails.append("HaloPipe did not take its own stream under nccl")
        xs = [torch.view_as_complex(torch.randn(n, 2, device=dev)) for _ in range(5)]
        tails = [torch.view_as_real(v[n - (K - 1):].clone()) for v in xs]
        ys = [torch.empty_like(v) for v in xs]
        torch.cuda.synchronize()
        for v, tl, yv in zip(xs, tails, ys):
            ovp.step(v, yv, lambda tl=tl: pipe.post(tl), first=False, consumed=pipe.consumed)
        ovp.wait_outputs()
        comp.synchronize()
        for i, (v, yv) in enumerate(zip(xs, ys)):
            one = t.Fir(h, t.C64, t.FIR_OVERLAP_SAVE)
            one.set_history(v[n - (K - 1):].clone())
            refp = one.step(v)
            e = float((yv - refp).abs().max() / refp.abs().max())
            if e > 2e-6:
                fails.append(f"pipelined FIR step {i} vs single handle: {e}")
        rp = sharding.OverlappedResampler(t, ratio, t.C64, edge_stream=True)
        rp.input_ready = True
        pipe14 = sharding.HaloPipe(torch.zeros(14, 2, device=dev), 0, 1, ring=True, complex_view=True)
        posp = 3 << 20
        outs = []
        for v in xs[:3]:
            yv = torch.empty(rp.counts(posp, n)[1], dtype=v.dtype, device=dev)
            t14 = torch.view_as_real(v[n - 14:].clone())
            rp.step(v, yv, posp, lambda t14=t14: pipe14.post(t14), first=False, consumed=pipe14.consumed)
            outs.append((v, yv))
        rp.wait_outputs()
        comp.synchronize()
        for i, (v, yv) in enumerate(outs):
            ro = t.Resampler(ratio, t.C64)
            ro.seek(posp, v[n - 14:].clone())
            refp = ro.step(v)
            if refp.shape != yv.shape or not torch.equal(refp, yv):
                fails.append(f"pipelined resampler step {i} differs from the single handle")
        sp = sharding.OverlappedSos(t, co, 1.0, t.F32, edge_stream=True)
        sp.input_ready = True
        pipeW = sharding.HaloPipe(torch.zeros(W, device=dev), 0, 1, ring=True)
        xrs = [torch.randn(n, device=dev) for _ in range(4)]
        yrs = [torch.empty_like(v) for v in xrs]
        torch.cuda.synchronize()
        for v, yv in zip(xrs, yrs):
            tw = v[n - W:].clone()
            sp.step(v, yv, lambda tw=tw: pipeW.post(tw), first=False, consumed=pipeW.consumed)
        sp.wait_outputs()
        comp.synchronize()
        for i, (v, yv) in enumerate(zip(xrs, yrs)):
            fo = t.Sos(co, 1.0, t.F32)
            fo.step(v[n - W:].clone(), scratch)
            refp = fo.step(v)
            e = float((yv - refp).abs().max() / refp.abs().max())
            if e > 1e-6:
                fails.append(f"pipelined SOS step {i}: {e}")
    torch.cuda.synchronize()

    # ---- exact SOS exchange: the two all_gathers (int64 length, float32 state) with one rank
    from oracle import pyoracle as orc
    z, p, mn, md = orc.design_butter_lp(1, 1e-5)
    co1, gain, r1 = orc.SosChain(z, p, mn, md).coefs()
    xo = xr + 0.5
    f1 = t.Sos(co1, gain, t.F32, r1)
    y1, etat = sharding.sos_step_exact(f1, xo, n, 0, 1, None, force_collective=True)
    ref1 = t.Sos(co1, gain, t.F32, r1).step(xo)
    torch.cuda.synchronize()
    if not torch.equal(y1, ref1):
        fails.append("sos_step_exact with forced all_gathers differs from the plain step")

    dist.barrier()
    print("RCCL_WORKER " + ("OK" if not fails else "FAILED: " + "; ".join(fails)), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if not fails else 1)


if __name__ == "__main__":
    main()
