#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native libtsd hot path.

Metric (BASELINE.json): Msamples/s for the 127-tap FIR on a 2^26-sample complex-float
stream, with the dominant kernel's HBM GB/s against the 8 TB/s roofline.

A "step" is one pass of the FIR (filtre_rif semantics, C ABI tsdgpu_fir_step) over one
2^26-sample Veccf batch already resident in HBM.  With N GPUs the stream is sharded by
contiguous chunk (weak scaling: 2^26 samples per GPU); every step posts the exchange of the
K-1-sample halo with the left neighbour over RCCL (torch.distributed send/recv), launches the
halo-free interior of the chunk at once and the K-1 edge outputs once the halo has arrived --
the only exchange the path needs, off the step's critical path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--workload fft|sos|resample` runs BASELINE.json configs[2..4] under the same contract (same
JSON line, same sharding rules of SURVEY.md 8e); the default `fir` is configs[1], the
configuration the metric is quoted on.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
LOG2N = 26
K_TAPS = 127
# HBM bytes per step: read at run time from the newest committed profiles/r*_pmc_traffic.txt (rocprofv3 --pmc
# FETCH_SIZE / WRITE_SIZE in separate passes, scripts/profile_round.sh; mean KB per launch of each kernel).
# gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B request on wide
# coalesced reads -> x2; WRITE_SIZE is exact.  The dominant kernels of each workload:
TRAFFIC_KERNELS = {
    "fir": ["ols_kernel<false"],
    "fft": ["fft1m_cols_kernel<1", "fft1m_cols_kernel<2"],
    "sos": ["sos_kernel"],
    "resample": ["resample15"],          # (resample15s_kernel; resample15_kernel in the files of rounds 1-3)
}


def pmc_traffic(workload):
    """-> (bytes per step or None, source file or None)"""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.txt")),
                   key=lambda f: int(re.search(r"r(\d+)_pmc_traffic", f).group(1)))
    if not files:
        return None, None
    src = files[-1]
    total, found = 0.0, 0
    for ln in open(src):
        m = re.match(r"pmc (.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=([0-9.e+]+)", ln)
        if not m:
            continue
        if any(("::" + k) in m.group(1) or m.group(1).startswith("void tsdgpu::" + k) for k in TRAFFIC_KERNELS[workload]):
            total += float(m.group(3)) * 1024.0 * (2.0 if m.group(2) == "FETCH_SIZE" else 1.0)
            found += 1
    if found != 2 * len(TRAFFIC_KERNELS[workload]):
        return None, None
    return total, os.path.relpath(src, ROOT)


def design_lowpass(n, fc):
    """Hann-windowed sinc low-pass normalised to unit DC gain (same design family as
    libtsd's design_rif_fen(n, "lp", fc); coefficients are bench input data)."""
    k = np.arange(n) - n // 2
    h = 2 * fc * np.sinc(2 * fc * k)
    w = 0.5 + 0.5 * np.cos(2 * np.pi * np.linspace(-(n // 2) / n, (n // 2) / n, n))
    h = h * w
    return (h / h.sum()).astype(np.float32)


def design_riia_butter_sos(order, fc):
    """Butterworth low-pass of even `order` as libtsd's design_riia(order, "lp", "butt", fc) -> filtre_sois builds it: analogue
    prototype poles on the unit circle, pre-warped bilinear transform, all zeros at -1; conjugate poles paired into DF2
    sections [1, 2, 1] / [1, a1, a2] (b0 = 1) and ONE gain = numer.mlt / denom.mlt applied to the last section's output
    (filtre-rt.cc:467-559).  Coefficients are bench input data (the oracle's design_butter_lp + SosChain.coefs() give the
    same table; the bench's GPU legs import nothing from oracle/).  -> ([nsec, 5] float32 (b0, b1, b2, a1, a2), gain)"""
    assert order % 2 == 0
    k = np.arange(order // 2)
    pa = np.exp(1j * np.pi * (2 * k + order + 1) / (2 * order))              # left half plane, upper half
    w = np.tan(np.pi * fc)
    pz = (1 + w * pa) / (1 - w * pa)                                        # bilinear transform, fs = 1
    co = np.array([[1.0, 2.0, 1.0, -2 * p.real, abs(p) ** 2] for p in pz], np.float64)
    gain = float(np.prod([(1 + c[3] + c[4]) / 4.0 for c in co]))            # unit gain at DC
    return co.astype(np.float32), gain


def _time_cpu(fn, units_per_call, unit, what, seconds_target=12.0):
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    reps = max(1, min(4096, int(seconds_target / max(dt, 1e-4))))   # ~12 s of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = time.perf_counter() - t0
    return {"value": round(reps * units_per_call / dt / 1e6, 3), "unit": unit, "cores": 1, "kind": "port",
            "sample": f"{reps} x {what}, 1 thread of {os.cpu_count()}"}


# ---------------------------------------------------------------------------------------
# Workloads.  Each provides: step() (N > 1: the neighbour exchange is posted inside it, before the
# halo-free interior is launched), units per step per GPU, algorithmic bytes per step per GPU, and a CPU leg on the oracle.
# ---------------------------------------------------------------------------------------
class FirWorkload:
    """configs[1]: 127-tap FIR on 2^26 cfloat per GPU; halo = K-1 samples from the left rank."""
    unit = "Msamples/s"

    def __init__(self, t, torch, dev, rank, world, args, method=None):
        self.t, self.rank, self.world = t, rank, world
        self.n = _per_rank(1 << args.log2n, world, args)
        self.h = design_lowpass(K_TAPS, 0.02)
        g = torch.Generator(device=dev).manual_seed(2 + rank)
        self.x = torch.view_as_complex(torch.randn(self.n, 2, device=dev, generator=g))
        self.y = torch.empty_like(self.x)
        self.halo_out = torch.view_as_real(self.x[self.n - (K_TAPS - 1):].clone())
        self.halo_in_c = torch.zeros(K_TAPS - 1, dtype=self.x.dtype, device=dev)
        self.halo_in = torch.view_as_real(self.halo_in_c)
        self.dist = world > 1 or args.force_dist
        if self.dist:
            # one rank per chunk: interior launched before the halo arrives, edge after (sharding.OverlappedFir)
            from libtsd_amd import sharding
            self.ov = sharding.OverlappedFir(t, self.h, t.C64, method, edge_stream=True)
            self.ov.input_ready = True            # (the bench's inputs are resident before the timed region)
            self.f = self.ov.main
        else:
            self.f = t.Fir(self.h, t.C64, t.FIR_AUTO if method is None else method)
        self.method = {t.FIR_DIRECT: "direct", t.FIR_OVERLAP_SAVE: "overlap-save"}[self.f.method]
        self.units = float(self.n)
        self.alg_bytes = 16.0 * self.n            # 8 B read + 8 B written per complex sample
        self.metric = "Msamples/s, 127-tap FIR on 2^%d cfloat stream" % args.log2n
        self.dtype = "f32 (complex64 data, real f32 taps)"
        self.config = {"workload": "configs[1]: 127-tap FIR (design_rif_fen lp 0.02, real taps via filtrer()) "
                                   "on %s, inputs resident in HBM" % _size_note("Veccf", 1 << args.log2n, world, args),
                       "method": self.method, "samples_per_gpu": self.n,
                       "sharding": "contiguous chunks, K-1 halo via RCCL send/recv posted before the interior launch, "
                                   "edge (first K-1 outputs) launched behind it" if self.dist else "single GPU"}
        self.ring = args.force_dist and world == 1
        self.pipe = None                          # sharding.HaloPipe, made at the first distributed step
        self.traffic_ok = self.method == "overlap-save" and self.n == 1 << LOG2N and not self.dist     # the shape the PMC passes were run on

    def step(self):
        if not self.dist:
            self.f.step(self.x, self.y)   # one GPU: no neighbour; launched on torch's current stream (passed through the C ABI)
            return
        from libtsd_amd import sharding
        if self.pipe is None:
            self.pipe = sharding.HaloPipe(self.halo_in, self.rank, self.world, ring=self.ring, complex_view=True)
        self.ov.step(self.x, self.y, lambda: self.pipe.post(self.halo_out), first=(self.rank == 0 and not self.ring), consumed=self.pipe.consumed)

    def cpu_baseline(self):
        from oracle import pyoracle as orc
        rng = np.random.default_rng(2)
        n0 = 1 << 20
        x = (rng.standard_normal(n0) + 1j * rng.standard_normal(n0)).astype(np.complex64)
        f = orc.Fir(self.h)
        res = _time_cpu(lambda: f.step(x), n0, "Msamples/s", "2^20 complex samples, 127 real taps, oracle orc_fir_cf")
        # the same port chunk-parallel over the host cores this process may use (libtsd itself is
        # single-threaded: one stateful filter per chunk) -- informative extra, not the baseline
        try:
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, len(os.sched_getaffinity(0)))      # every core this process may run on; the count is in `cores`
            firs = [orc.Fir(self.h) for _ in range(nthr)]
            xs = [x.copy() for _ in range(nthr)]
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(lambda i: firs[i].step(xs[i]), range(nthr)))
                t0 = time.perf_counter()
                reps = 8
                for _ in range(reps):
                    list(ex.map(lambda i: firs[i].step(xs[i]), range(nthr)))
                dt = time.perf_counter() - t0
            res["parallel"] = {"value": round(reps * nthr * n0 / dt / 1e6, 3), "unit": "Msamples/s", "cores": nthr,
                               "sample": f"{reps} x {nthr} chunks of 2^20 samples, one oracle filter per thread, "
                                         f"{nthr} threads = the affinity mask of this process ({os.cpu_count()} cores on the host)"}
        except Exception as e:      # never let the informative leg break the bench line
            res["parallel"] = {"error": str(e)}
        return res


class FftWorkload:
    """configs[2]: 2^20-point complex FFT, batch 256 per GPU; independent transforms, no exchange."""
    unit = "Mpoints/s"

    def __init__(self, t, torch, dev, rank, world, args):
        self.nfft, self.batch = 1 << 20, _per_rank(256, world, args)
        g = torch.Generator(device=dev).manual_seed(3 + rank)
        self.x = torch.view_as_complex(torch.randn(self.batch * self.nfft, 2, device=dev, generator=g)).reshape(self.batch, self.nfft)
        self.y = torch.empty_like(self.x)
        self.p = t.Fft(self.nfft, self.batch)
        self.units = float(self.nfft) * self.batch
        self.alg_bytes = 16.0 * self.units
        self.metric = "Mpoints/s, 2^20-point complex FFT, batch 256"
        self.dtype = "f32 (complex64)"
        self.config = {"workload": "configs[2]: fft() of %d x Veccf[2^20] per GPU, unitary scaling, inputs resident in HBM" % self.batch,
                       "sharding": "batch index split, no exchange" if world > 1 else "single GPU"}
        self.traffic_ok = self.batch == 256

    def step(self):
        self.p.step(self.x, True, self.y)

    def cpu_baseline(self):
        from oracle import pyoracle as orc
        rng = np.random.default_rng(3)
        x = (rng.standard_normal(self.nfft) + 1j * rng.standard_normal(self.nfft)).astype(np.complex64)
        return _time_cpu(lambda: orc.fft(x), self.nfft, "Mpoints/s", "one 2^20-point transform, oracle orc_fft (radix-2)")


class SosWorkload:
    """configs[3]: 6-section SOS (Butterworth 12, fc 0.25) on 2^26 float per GPU.  Sharding: every
    rank but the first warms its filter on the last `halo` samples of the left neighbour
    (state transition below 1e-9 after `halo` samples, computed by the library)."""
    unit = "Msamples/s"

    def __init__(self, t, torch, dev, rank, world, args):
        self.rank, self.world = rank, world
        self.n = _per_rank(1 << args.log2n, world, args)
        # design_riia(12, "lp", "butt", 0.25) -> filtre_sois: six DF2 sections with b0 = 1 and the gain applied at the end
        # (filtre-rt.cc:467-559) -- the form the reference builds, not scipy's (b0 != 1, gain folded into the first section)
        self.co, self.gain = design_riia_butter_sos(12, 0.25)
        g = torch.Generator(device=dev).manual_seed(4 + rank)
        self.x = torch.randn(self.n, device=dev, generator=g)
        self.y = torch.empty_like(self.x)
        self.dist = world > 1 or args.force_dist
        self.ring = args.force_dist and world == 1
        if self.dist:
            from libtsd_amd import sharding
            self.ov = sharding.OverlappedSos(t, self.co, self.gain, t.F32, edge_stream=True)
            self.ov.input_ready = True
            self.f = self.ov.main
        else:
            self.f = t.Sos(self.co, self.gain, t.F32)
        self.halo = int(self.f.halo)
        self.halo_out = self.x[self.n - self.halo:].clone()
        self.halo_in = torch.zeros(self.halo, dtype=self.x.dtype, device=dev)
        self.pipe = None
        self.units = float(self.n)
        self.alg_bytes = 8.0 * self.n
        self.metric = "Msamples/s, 6-section SOS IIR on 2^%d float stream" % args.log2n
        self.dtype = "f32"
        self.config = {"workload": "configs[3]: 6 DF2 biquads (Butterworth order 12, fc 0.25) on %s" % _size_note("Vecf", 1 << args.log2n, world, args),
                       "halo_samples": self.halo,
                       "sharding": "contiguous chunks, warm-up halo via RCCL send/recv posted before the interior launch" if self.dist else "single GPU"}
        self.traffic_ok = self.n == 1 << LOG2N and not self.dist

    def step(self):
        if not self.dist:
            self.f.step(self.x, self.y)
            return
        from libtsd_amd import sharding
        if self.pipe is None:
            self.pipe = sharding.HaloPipe(self.halo_in, self.rank, self.world, ring=self.ring)
        self.ov.step(self.x, self.y, lambda: self.pipe.post(self.halo_out), first=(self.rank == 0 and not self.ring), consumed=self.pipe.consumed)

    def cpu_baseline(self):
        from oracle import pyoracle as orc
        z, p, mn, md = orc.design_butter_lp(12, 0.25)     # design_riia(12,"lp","butt",0.25)
        f = orc.SosChain(z, p, mn, md)
        rng = np.random.default_rng(4)
        n0 = 1 << 20
        x = rng.standard_normal(n0).astype(np.float32)
        return _time_cpu(lambda: f.step(x), n0, "Msamples/s", "2^20 float samples, 6 sections, oracle orc_sos_step_f")


class ResampleWorkload:
    """configs[4]: 160/147 resampling of a 2^30 cfloat stream, 2^27 inputs per GPU.  Sharding: rank r
    seeks to stream position r * 2^27 and takes its K-1 = 14-sample window from the left rank."""
    unit = "Msamples/s"

    def __init__(self, t, torch, dev, rank, world, args):
        self.rank, self.world = rank, world
        self.n = _per_rank(1 << 30, world, args) if args.scaling == "strong" else 1 << 27
        from libtsd_amd import sharding
        self.dist = world > 1 or args.force_dist
        self.ring = args.force_dist and world == 1
        self.ov = sharding.OverlappedResampler(t, np.float32(160.0) / np.float32(147.0), t.C64, edge_stream=True) if self.dist else None
        if self.dist:
            self.ov.input_ready = True
        self.r = self.ov.main if self.dist else t.Resampler(np.float32(160.0) / np.float32(147.0), t.C64)
        g = torch.Generator(device=dev).manual_seed(5 + rank)
        self.x = torch.view_as_complex(torch.randn(self.n, 2, device=dev, generator=g))
        self.pos = rank * self.n
        self.r.seek(self.pos)
        self.nout = int(self.r.out_count(self.n))
        self.y = torch.empty(self.nout + 4, dtype=self.x.dtype, device=dev)   # (the count moves by one with the start phase)
        self.halo_out = torch.view_as_real(self.x[self.n - 14:].clone())
        self.halo_in_c = torch.zeros(14, dtype=self.x.dtype, device=dev)
        self.halo_in = torch.view_as_real(self.halo_in_c)
        self.pipe = None
        self.units = float(self.n)
        self.alg_bytes = 8.0 * self.n + 8.0 * self.nout
        self.metric = "Msamples/s (input), 160/147 resampling of a cfloat stream, %d inputs per GPU" % self.n if args.scaling == "strong" \
            else "Msamples/s (input), 160/147 resampling of a cfloat stream, 2^27 inputs per GPU"
        self.dtype = "f32 (complex64 data, f32 LUT taps)"
        self.config = {"workload": "configs[4]: filtre_reechan(160/147) interpolator on a %d-sample Veccf shard per GPU "
                                   "(2^30 over %s GPUs)" % (self.n, world if args.scaling == "strong" else 8), "outputs_per_gpu": self.nout,
                       "sharding": "contiguous input chunks, 14-sample halo via RCCL send/recv posted before the interior launch + seek" if self.dist else "single GPU"}
        self.traffic_ok = not self.dist and self.n == 1 << 27

    def step(self):
        if not self.dist:
            # every step resamples the same 2^27-sample shard from stream position `pos`
            self.r.seek(self.pos, None)
            self.r.step(self.x, self.y)
            return
        from libtsd_amd import sharding
        if self.pipe is None:
            self.pipe = sharding.HaloPipe(self.halo_in, self.rank, self.world, ring=self.ring, complex_view=True)
        self.ov.step(self.x, self.y, self.pos, lambda: self.pipe.post(self.halo_out), first=(self.rank == 0 and not self.ring), consumed=self.pipe.consumed)

    def cpu_baseline(self):
        from oracle import pyoracle as orc
        rng = np.random.default_rng(5)
        n0 = 1 << 20
        x = (rng.standard_normal(n0) + 1j * rng.standard_normal(n0)).astype(np.complex64)
        f = orc.Resampler(np.float32(160.0) / np.float32(147.0))
        return _time_cpu(lambda: f.step(x), n0, "Msamples/s", "2^20 complex input samples, oracle orc_ra_step")


WORKLOADS = {"fir": FirWorkload, "fft": FftWorkload, "sos": SosWorkload, "resample": ResampleWorkload}


def _per_rank(total, world, args):
    """weak: every rank takes `total`; strong: the total is split over the ranks (it must divide)"""
    if args.scaling != "strong":
        return total
    assert total % world == 0, f"--scaling strong: {total} units do not split over {world} ranks"
    return total // world


def _size_note(kind, total, world, args):
    if args.scaling == "strong":
        return "%d-sample %s in total, %d per GPU" % (total, kind, total // world)
    return "2^%d-sample %s per GPU" % (total.bit_length() - 1, kind)


def _self_launch(args):
    """`python3 bench.py --gpus N` started WITHOUT a launcher (WORLD_SIZE unset): start the N ranks as CHILD processes
    through torch.distributed.run, relay rank 0's JSON line and return the launcher's exit code.  Called before this
    process has imported torch or touched the GPU: nothing is exec'ed and no GPU-initialised process is relaunched."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    env["TSDGPU_BENCH_SELF_LAUNCHED"] = "1"
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    for ln in p.stdout:
        # stdout carries the ONE JSON line; anything else the launcher or a library wrote there goes to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln)
    sys.stdout.flush()
    return p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="fir")
    ap.add_argument("--log2n", type=int, default=LOG2N, help="fir/sos: samples per GPU = 2^log2n (default: the BASELINE size)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): the per-GPU size is fixed; strong: the TOTAL is fixed and split over the ranks "
                         "(resample: the 2^30 inputs of configs[4]; fir/sos: 2^log2n samples; fft: the 256 transforms)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher check for a box without N GPUs: the ranks rendezvous (gloo), run the barrier and the max-over-ranks "
                         "reduction of the bench and rank 0 prints a record with value null; no GPU work, nothing measured")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args))
    # TSDGPU_BENCH_FORCE_DIST=1: `--gpus 1` also goes through init_process_group(backend) and the multi-rank step (the halo
    # exchange becomes a self send / receive: a circular stream) -- how ONE GPU exercises the RCCL calls of the N > 1 path.
    # A diagnostic mode: the headline line is the default one.
    args.force_dist = os.environ.get("TSDGPU_BENCH_FORCE_DIST", "0") not in ("", "0")
    # stdout carries ONE line -- the JSON record: everything else this process or the libraries under it print there (RCCL writes
    # its version banner to stdout when NCCL_DEBUG=VERSION, which the GPU boxes export) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import libtsd_amd as t
    from libtsd_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = None
    if args.rendezvous_only:
        assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        t0 = time.perf_counter()
        dist.barrier()
        dt = sharding.max_over_ranks(time.perf_counter() - t0 + 1e-3 * rank, None, world, force=True)
        assert dt >= 1e-3 * (world - 1)       # the slowest rank's time is what every rank holds
        if rank == 0:
            os.write(json_fd, (json.dumps({"rendezvous_only": True, "value": None, "n_gpus": world, "scaling": args.scaling,
                                           "config": {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                      "self_launched": os.environ.get("TSDGPU_BENCH_SELF_LAUNCHED") == "1"}}) + "\n").encode())
        dist.destroy_process_group()
        return
    if world > 1 or args.gpus > 1 or args.force_dist:
        assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: start `python3 bench.py --gpus {args.gpus}` (it launches its own ranks) or torch.distributed.run --nproc-per-node {args.gpus}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # TSDGPU_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than
        # ranks (ranks share the devices, halos go through the host); the driver's runs use RCCL
        backend = os.environ.get("TSDGPU_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)
    if dist.is_initialized() and backend == "nccl":
        # the multi-rank steps run on a stream of their own instead of the legacy default stream: beside RCCL's transfer kernels
        # and the edge stream the default stream's implicit ordering cost 10-30 us per step (scripts/diag_dist_step.py)
        torch.cuda.set_stream(torch.cuda.Stream(dev))

    w = WORKLOADS[args.workload](t, torch, dev, rank, world, args)

    in_dist = dist.is_initialized()

    def barrier():
        if in_dist:
            dist.barrier()

    def run(wl, steps, warmup):
        """-> (wall seconds of the K timed steps, max over ranks; mean kernel ms of the same K steps replayed
        with one HIP-event pair per launch).  The timed region carries no per-step events: each record is a
        barrier packet between two launches (measured: ~6 us per step on the 0.22 ms FIR step), which belongs
        to the measurement, not to the path."""
        for _ in range(warmup):
            wl.step()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            wl.step()         # (N > 1: posts the halo exchange, launches the interior, waits for the halo, launches the edge)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        dt = sharding.max_over_ranks(dt, dev, world, force=in_dist)
        # kernel duration: the same steps again, every launch bracketed by events on its stream
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for i in range(steps):
            evs[i][0].record()
            wl.step()
            evs[i][1].record()
        torch.cuda.synchronize()
        barrier()
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        return dt, kern_ms

    # GPU clock ramp: the first ~20 ms of kernels after an idle period run at a lower clock
    # (measured: 0.27 ms vs 0.22 ms per FIR step).  A fixed untimed pre-run brings the device to
    # its sustained state whatever --warmup the caller picked; the W warm-up steps and the K
    # timed steps below are unchanged.
    PRE_WARM = 150 if args.workload in ("fir", "sos") else 20
    for _ in range(PRE_WARM):
        w.step()           # (every rank runs the same count: the halo exchanges pair up)
    torch.cuda.synchronize()
    dt, kern_ms = run(w, args.steps, args.warmup)
    value = w.units * world * args.steps / dt / 1e6
    # two clocks, both reported under their own names: HIP events around the operator on its stream
    # (the kernel's launch duration: `achieved` / `frac`, what the roofline is about) and the wall clock
    # of the K timed steps (`value`, `ms_per_step`; `achieved_wall` / `frac_wall`: launch gaps included)
    timer_note = ("achieved/frac: HIP events around every launch on the operator's stream (the K steps replayed right after the timed region); "
                  "achieved_wall/frac_wall: the wall clock of the timed region, the denominator of `value`")
    if getattr(w, "dist", False):
        # a multi-rank step spans three streams (interior, exchange, edge) and is pipelined across steps: an event pair around it on
        # one stream serialises it and adds two release fences -- the per-step figure is the wall clock's there
        kern_ms = dt / args.steps * 1e3
        timer_note = ("multi-rank step (interior / exchange / edge on their own streams, pipelined): achieved/frac and achieved_wall/frac_wall "
                      "are both the wall clock of the timed region")
    achieved = w.alg_bytes / (kern_ms * 1e-3) / 1e9
    achieved_wall = w.alg_bytes * args.steps / dt / 1e9
    traffic, traffic_src = pmc_traffic(args.workload) if w.traffic_ok else (None, None)

    out = None
    if rank == 0:
        cfg = dict(w.config)
        cfg["pre_warm_steps"] = PRE_WARM
        # "did the collective library see N ranks" is answerable from the record
        cfg["backend"] = (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if in_dist else "none (one process, one GPU)"
        cfg["world_size"] = dist.get_world_size() if in_dist else 1
        cfg["self_launched"] = os.environ.get("TSDGPU_BENCH_SELF_LAUNCHED") == "1"
        out = {
            "metric": w.metric, "value": round(value, 1), "unit": w.unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": w.dtype, "data": "synthetic", "config": cfg,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "timer": timer_note,
                         "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": w.alg_bytes,
                         "achieved_wall": round(achieved_wall, 1), "frac_wall": round(achieved_wall / HBM_PEAK_GBS, 4)},
        }
    if world == 1:
        if args.workload == "fir":
            # secondary line: the direct kernel on the same data (rank-local, N=1 only)
            # (same pre-warm and at least 20 timed steps: measured cold on 5 steps the line moved 25 % with --steps)
            args_d = argparse.Namespace(**dict(vars(args), force_dist=False))
            wd = FirWorkload(t, torch, dev, rank, world, args_d, method=t.FIR_DIRECT)
            sd = max(20, args.steps // 4)
            for _ in range(PRE_WARM // 2):
                wd.step()
            torch.cuda.synchronize()
            dt_d, kern_d = run(wd, sd, 5)
            out["direct"] = {"value": round(wd.units * sd / dt_d / 1e6, 1), "unit": "Msamples/s",
                             "kernel_ms": round(kern_d, 4),
                             "hbm_frac": round(wd.alg_bytes / (kern_d * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "valu_frac_of_157.3TF": round(4.0 * K_TAPS * wd.n / (kern_d * 1e-3) / 157.3e12, 4)}
        if not args.no_cpu:
            out["cpu_baseline"] = w.cpu_baseline()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if in_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
