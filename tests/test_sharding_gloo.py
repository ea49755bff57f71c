"""N > 1 path on CPU: two gloo ranks shard a stream by contiguous chunk, exchange the halo
with libtsd_amd.sharding (the code bench.py runs over RCCL), and each rank filters its chunk
with the oracle seeded by the received halo.  The concatenation must equal the one-process
result (to an ulp: the oracle's two-segment circular sum is split at a different index when
the delay line is re-seeded, which moves the last bit; the resampler is exactly equal; the
SOS chain is warmed on a 256-sample halo, after which its state transition is below 1e-9)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, K, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libtsd_amd import sharding
    from oracle import pyoracle as orc
    rng = np.random.default_rng(123)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)     # same stream on every rank
    h = orc.design_rif_fen(K, "lp", 0.02)
    lo, hi = sharding.chunk_bounds(n, rank, world)
    mine = x[lo:hi]
    # --- FIR: K-1 halo
    tail = torch.view_as_real(torch.from_numpy(mine[-(K - 1):].copy()))
    halo = torch.zeros(K - 1, 2)
    sharding.exchange_left_halo(tail, halo, rank, world)
    f = orc.Fir(h)
    halo_c = torch.view_as_complex(halo).numpy()
    if rank > 0:
        assert np.array_equal(halo_c, x[lo - (K - 1):lo])
        f.step(np.concatenate([np.zeros(1, np.complex64), halo_c]))       # load the delay line with the halo
    y = f.step(mine)
    # --- resampler: 14-sample window halo + absolute position (schedule restarts from the recurrence)
    r = orc.Resampler(np.float32(160.0) / np.float32(147.0))
    tail14 = torch.view_as_real(torch.from_numpy(mine[-14:].copy()))
    halo14 = torch.zeros(14, 2)
    sharding.exchange_left_halo(tail14, halo14, rank, world)
    pre = np.zeros(0, np.complex64)
    if rank > 0:
        pre = x[:lo]          # the oracle has no seek: replay the prefix (CPU test only)
        r.step(pre)
    yr = r.step(mine)
    # --- SOS chain: warm-up halo of W samples (W given: the library computes it per filter)
    W = 256
    xr = x.real.copy()
    mine_r = xr[lo:hi]
    tailw = torch.from_numpy(mine_r[-W:].copy())
    halow = torch.zeros(W)
    sharding.exchange_left_halo(tailw, halow, rank, world)
    z, p, mn, md = orc.design_butter_lp(12, 0.25)
    sc = orc.SosChain(z, p, mn, md)
    if rank > 0:
        assert np.array_equal(halow.numpy(), xr[lo - W:lo])
        sc.step(halow.numpy())
    ys = sc.step(mine_r)
    t = sharding.max_over_ranks(float(rank), torch.device("cpu"), world)
    q.put((rank, y, yr, t, ys))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fir_resampler_sos_match_single_process(orc, world):
    n, K = 30000, 127
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(123)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    h = orc.design_rif_fen(K, "lp", 0.02)
    yref = orc.fir(h, x)
    assert np.abs(np.concatenate([r[1] for r in res]) - yref).max() <= 2e-7 * np.abs(yref).max()
    yfull = orc.Resampler(np.float32(160.0) / np.float32(147.0)).step(x)
    assert np.array_equal(np.concatenate([r[2] for r in res]), yfull)
    assert all(r[3] == world - 1 for r in res)
    z, p, mn, md = orc.design_butter_lp(12, 0.25)
    ysref = orc.SosChain(z, p, mn, md).step(x.real.copy())
    assert np.abs(np.concatenate([r[4] for r in res]) - ysref).max() <= 1e-6 * np.abs(ysref).max()


# ---- exact sharding of a long-memory cascade: the all_gather orchestration of sharding.sos_step_exact, on CPU --------
class _ToySos:
    """The state-vector interface of capi.Sos (get_state / set_state / propagate_state / step) on a first-order smoother
    y[n] = a y[n-1] + (1 - a) x[n] in float64 -- stands in for the HIP cascade so that the exchange runs without a GPU."""

    def __init__(self, a):
        self.a, self.st = float(a), np.zeros(2, np.float32)        # [flag, y1]

    def get_state(self):
        return self.st.copy()

    def set_state(self, s):
        self.st = np.asarray(s, np.float32).copy()

    def propagate_state(self, n, state, end_state=None):
        out = np.array([max(state[0], 0.0 if end_state is None else end_state[0]), state[1] * self.a ** int(n)], np.float64)
        if end_state is not None:
            out[1] += end_state[1]
        return out.astype(np.float32)

    def step(self, x):
        x = np.asarray(x, np.float64)
        y = np.empty_like(x)
        prev = float(self.st[1]) if self.st[0] else float(x[0])      # first-sample seed, like SOIS
        for i, v in enumerate(x):
            prev = self.a * prev + (1.0 - self.a) * v
            y[i] = prev
        self.st = np.array([1.0, prev], np.float32)
        return y


def _worker_exact(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libtsd_amd import sharding
    rng = np.random.default_rng(7)
    x = rng.standard_normal(20011) + 0.5
    a = 1.0 - 1e-4
    f, etat, outs = _ToySos(a), None, []
    for lo, hi in ((0, 3), (3, 9000), (9000, 20011)):          # the first call has fewer samples than ranks
        l2, h2 = sharding.chunk_bounds(hi - lo, rank, world)
        y, etat = sharding.sos_step_exact(f, x[lo + l2:lo + h2], h2 - l2, rank, world, etat)
        parts = [None] * world
        dist.gather_object(np.asarray(y, np.float64)[:h2 - l2], parts if rank == 0 else None, dst=0)
        if rank == 0:
            outs.append(np.concatenate(parts))
    if rank == 0:
        ref = _ToySos(a).step(x)
        got = np.concatenate(outs)
        q.put(float(np.abs(got - ref).max() / np.abs(ref).max()))
    dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


class _ToySosFast(_ToySos):
    """The same stub with a vectorised step (scipy lfilter), for chunks of 2^24 samples and more."""

    def step(self, x):
        from scipy.signal import lfilter
        x = np.asarray(x, np.float64)
        prev = float(self.st[1]) if self.st[0] else float(x[0])
        y, zf = lfilter([1.0 - self.a], [1.0, -self.a], x, zi=[self.a * prev])
        self.st = np.array([1.0, y[-1]], np.float32)
        return y


def _worker_exact_long(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libtsd_amd import sharding
    # a pole at -(1 - 1e-9): the state after L samples is a^L ~ (-1)^L, so a chunk length rounded from 2^24 + 1 to 2^24
    # (what a float32 all_gather of the lengths did) flips the sign of the propagated start state
    a = -(1.0 - 1e-9)
    L = (1 << 24) + 1
    n = world * L
    lo, hi = sharding.chunk_bounds(n, rank, world)
    assert hi - lo == L
    x = np.random.default_rng(11).standard_normal(n)[lo:hi]         # same stream on every rank, own chunk kept
    f = _ToySosFast(a)
    y, etat = sharding.sos_step_exact(f, x, L, rank, world, None)
    if rank == world - 1:
        q.put((np.asarray(y[-1000:], np.float64), np.asarray(etat, np.float64)))
    dist.barrier()
    dist.destroy_process_group()


def test_sos_exact_exchange_chunks_longer_than_2p24():
    """the chunk lengths must travel as integers -- 2^24 + 1 samples per rank (the bench shape is 2^26).
    Three ranks: the start state of rank 2 is the first one that goes through propagate_state(len(chunk 1), ...)."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exact_long, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    tail, etat = q.get(timeout=240)
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    a = -(1.0 - 1e-9)
    x = np.random.default_rng(11).standard_normal(world * ((1 << 24) + 1))
    ref = _ToySosFast(a).step(x)
    assert np.abs(tail - ref[-1000:]).max() <= 1e-5 * np.abs(ref).max(), np.abs(tail - ref[-1000:]).max() / np.abs(ref).max()


@pytest.mark.parametrize("world", [2, 4])
def test_sos_exact_exchange_on_gloo(world):
    """sharding.sos_step_exact -- zero-state pass, ONE all_gather of the end states and chunk lengths, start states by
    propagation, second pass, stream state handed to the next call -- with a stub cascade, `world` CPU ranks, three
    calls (one of them shorter than the number of ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exact, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    err = q.get(timeout=5)
    assert err <= 1e-6, err              # (the stub keeps its state in float32 between passes)
