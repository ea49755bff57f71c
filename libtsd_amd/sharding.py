"""Multi-GPU sharding of the streaming path: contiguous chunk per rank plus ONE small
left-neighbour exchange.  No all-reduce anywhere: FIR needs the K-1
input samples before the chunk, the SOS chain a warm-up halo, the resampler its K-1-sample
window plus a stream position.  torch.distributed only (backend "nccl" == RCCL on the GPU
node, "gloo" in the CPU tests); plumbing around the C ABI, not part of it."""
import torch
import torch.distributed as dist


def chunk_bounds(n_total, rank, world):
    """Contiguous chunk [lo, hi) of rank `rank`."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


class HaloExchange:
    """One posted left-neighbour exchange: `finish()` makes the halo usable on the current stream.

    RCCL ("nccl"): the transfer runs on the process group's own stream, ordered after what the current stream held when
    it was posted; `finish()` is a stream dependency, not a host wait -- whatever is launched between `start` and
    `finish` (the halo-free interior of the step) overlaps with the exchange.  gloo (CPU tests, one-GPU rehearsals):
    no point-to-point on device tensors, so the tail is staged through the host and `finish()` blocks."""

    def __init__(self, works, halo_in, staged, result=None):
        self.works, self.halo_in, self.staged, self.result = works, halo_in, staged, result

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []
        if self.staged is not None:
            self.halo_in.copy_(self.staged)
            self.staged = None
        return self.halo_in if self.result is None else self.result


def start_halo_exchange(tail_out, halo_in, rank, world, group=None, ring=False, result=None, stream=None):
    """Posts the exchange of `exchange_left_halo` and returns at once (-> HaloExchange).  `ring=True` closes the chain
    (the last rank sends to rank 0, rank 0 receives): the layout of a circular stream, and with world == 1 a self
    send / receive -- how a single GPU exercises the RCCL point-to-point path.  `result`: what finish() returns instead
    of halo_in (complex samples travel as their float32 view: pass the complex view of the same storage here).
    `stream` (RCCL only): the torch stream the exchange is ordered after INSTEAD of the current one -- the process group's
    transfer kernel waits for what that stream held, not for the compute stream's queue (HaloPipe)."""
    if world == 1 and not (ring and dist.is_initialized()):
        return HaloExchange([], halo_in, None, result)
    # gloo has no point-to-point on device tensors: stage through the host there; RCCL sends device memory directly
    via_host = dist.get_backend(group) == "gloo" and tail_out.is_cuda
    src = tail_out.cpu() if via_host else tail_out
    dst = torch.empty_like(halo_in, device="cpu") if via_host else halo_in
    ops = []
    if rank + 1 < world or ring:
        ops.append(dist.P2POp(dist.isend, src, (rank + 1) % world, group))
    receives = rank > 0 or ring
    if receives:
        ops.append(dist.P2POp(dist.irecv, dst, (rank - 1) % world, group))
    if not ops:
        works = []
    elif stream is not None and not via_host and tail_out.is_cuda:
        with torch.cuda.stream(stream):
            works = dist.batch_isend_irecv(ops)
    else:
        works = dist.batch_isend_irecv(ops)
    return HaloExchange(works, halo_in, dst if (via_host and receives) else None, result)


class HaloPipe:
    """The per-step halo exchange of a rank that steps through a stream, kept OFF the compute stream's queue.

    Posted from the compute stream, RCCL's transfer kernel is ordered after everything that stream holds -- the previous
    step -- and then races the step's own interior launch for the chip: a persistent interior kernel that wins the race
    leaves the (tiny) transfer kernel no slot until it drains, and the edge launch behind it waits for both: measured at
    world size 1 (self send / receive), 127-tap FIR on 2^26 samples, 0.254 ms per step against 0.203 without an exchange --
    the exchange latency was back on the critical path.  Here the exchange of step k is posted on a stream of its own that
    waits only for the edge of step k - 2 (which read the halo buffer it is about to overwrite: two buffers alternate), so
    it is dispatched in the gaps the draining kernels of step k - 1 leave, a whole step before the edge of step k asks
    for it.  Valid when the tail a rank sends is ready when posted (a resident input, or a producer at least a step
    ahead); `after` takes the producer's event otherwise.  gloo / one rank: plain start_halo_exchange."""

    def __init__(self, halo_like, rank, world, group=None, ring=False, complex_view=False):
        self.rank, self.world, self.group, self.ring = rank, world, group, ring
        self.bufs = [torch.zeros_like(halo_like), torch.zeros_like(halo_like)]
        self.views = [torch.view_as_complex(b) for b in self.bufs] if complex_view else [None, None]
        self.k = 0
        self.piped = halo_like.is_cuda and dist.is_initialized() and dist.get_backend(group) != "gloo" and (world > 1 or ring)
        if self.piped:
            self.stream = torch.cuda.Stream(halo_like.device)
            self.done = [torch.cuda.Event(), torch.cuda.Event()]

    def post(self, tail_out, after=None):
        """-> HaloExchange of this step (finish() on the stream that reads the halo, then call consumed()).
        Pass `lambda: pipe.post(tail)` as the `exchange` of an Overlapped*.step to have it posted AFTER the interior launch:
        enqueueing a point-to-point pair was seen to hold the host until the process group's previous transfer kernel had
        finished -- and that kernel, sharing the chip with a persistent interior, finishes when the interior drains -- so a
        host that posts first launches every interior one launch latency (~19 us) late."""
        b = self.k & 1
        if self.piped:
            if self.k >= 2:
                self.stream.wait_event(self.done[b])           # the edge of step k - 2 has read this buffer
            if after is not None:
                self.stream.wait_event(after)
        return start_halo_exchange(tail_out, self.bufs[b], self.rank, self.world, self.group, self.ring, self.views[b],
                                   self.stream if self.piped else None)

    def consumed(self):
        """The step that used the last posted halo has been enqueued on the current stream."""
        if self.piped:
            self.done[self.k & 1].record(torch.cuda.current_stream(self.bufs[0].device))
        self.k += 1


def exchange_left_halo(tail_out, halo_in, rank, world, group=None, ring=False):
    """Every rank sends `tail_out` (the last samples of its chunk) to rank+1 and receives the
    tail of rank-1 into `halo_in`; rank 0's halo_in is left untouched (zeros = the reference's
    empty delay line, filtre-rt.cc:64).  One batched isend/irecv pair per rank."""
    return start_halo_exchange(tail_out, halo_in, rank, world, group, ring).finish()


# ---- a sharded step with the exchange OFF its critical path ("exchange once per call and overlap with the
# interior tile computation").  Only the first H outputs of a chunk depend on the neighbour's samples; everything behind
# them needs the chunk alone.  So a rank posts the exchange, launches the INTERIOR on its main handle -- primed with the
# chunk's own first H samples -- and only then waits for the halo and filters the EDGE (the first H samples) on a second,
# small handle of the same operator.  Two launches on the operator's stream, the halo wait between them.
class _EdgeStream:
    """The edge of an overlapped step on a stream of its own (opt-in: `edge_stream=True`).  On the operator's stream the edge
    -- a history copy and a launch of a few hundred outputs, behind the halo wait -- sits between two interiors and costs the
    step its launch latencies and the cross-queue wait (measured, 127-tap FIR on 2^26 samples with a self exchange at world
    size 1: 24 us of small kernels + 36 us of gaps on a 199-us interior).  On its own stream it runs in the slots the
    draining interior of the NEXT step leaves; the outputs of a step are complete once `wait_outputs()` has been called on the
    consumer's stream (or the device synchronised)."""

    def _init_edge_stream(self, on, device=None):
        self.edge_stream = None
        # True: x is known to be ready when step() is called (a resident input, a producer synchronised elsewhere) -- the edge
        # stream then takes NO event from the operator's stream.  An event recorded behind a kernel that wrote half a gigabyte
        # is not free on this platform (its release fence writes the caches back): ~22 us between that kernel's end and
        # the next launch in the kernel trace of the step, per event.
        self.input_ready = False
        if on and torch.cuda.is_available():
            self.edge_stream = torch.cuda.Stream(device)
            self._ready = torch.cuda.Event()
            self.edge_done = torch.cuda.Event()

    def _on_edge(self, x, fn, y=None):
        """Runs fn() -- the halo wait and the edge launches -- on the edge stream, ordered after what the current stream holds
        (the producer of x); without an edge stream: in place.  A chunk filtered IN PLACE (y is x) is always ordered behind
        the current stream, `input_ready` or not: the edge overwrites x[:H], which the interior's history copy -- enqueued
        on the current stream just before -- must have read first."""
        if self.edge_stream is None or not x.is_cuda:
            return fn()
        aliased = y is not None and getattr(y, "is_cuda", False) and y.data_ptr() == x.data_ptr()
        if not self.input_ready or aliased:
            cur = torch.cuda.current_stream(x.device)
            self._ready.record(cur)
            self.edge_stream.wait_event(self._ready)
        with torch.cuda.stream(self.edge_stream):
            r = fn()
            self.edge_done.record(self.edge_stream)
        return r

    def wait_outputs(self, x=None):
        """Makes the current stream wait for the edge of the last step (nothing to do without an edge stream)."""
        if self.edge_stream is not None:
            torch.cuda.current_stream(self.edge_stream.device).wait_event(self.edge_done)


class OverlappedFir(_EdgeStream):
    """FIR (tsdgpu_fir): H = K - 1.  Same outputs as set_history(halo) + step(x) on one handle: bit for bit with the
    direct kernel (an output's sum does not depend on where the call starts); the overlap-save blocks shift by H samples."""

    def __init__(self, t, taps, data_type, method=None, edge_stream=False):
        self.main = t.Fir(taps, data_type, t.FIR_AUTO if method is None else method)
        self.edge = t.Fir(taps, data_type, t.FIR_DIRECT)        # the edge's outputs: the direct kernel is one small launch
        self.H = self.main.K - 1                                 # samples of the neighbour a chunk needs (the halo)
        # the interior starts `lead` samples into the chunk and reads its delay line out of the chunk itself
        # (tsdgpu_fir_step_after: no history copy, no extra launch): lead = the handle's history length >= H
        # (lead < 0: the partitioned plan of more than 12289 taps has no step_after -- the copying form serves it)
        self.after = self.main.lead >= 0
        self.lead = max(self.main.lead, self.H)
        self._init_edge_stream(edge_stream)

    def interior(self, x, y):
        if not self.after or x.shape[0] <= self.lead or not getattr(x, "is_cuda", False) or x.data_ptr() == y.data_ptr():
            H = self.H                                           # host arrays / in place: the copying form, split at H
            if x.shape[0] <= H:
                return 0
            self.main.set_history(x[:H])
            self.main.step(x[H:], y[H:])
            return H
        self.main.step_after(x, y, self.lead)
        return self.lead

    def edge_step(self, x, y, halo, first, upto=None):
        """halo: the H samples before the chunk (ignored when `first`: the stream starts here, zero delay line); upto: outputs
        the edge owes (the interior's split point)."""
        H = min(self.H if upto is None else upto, x.shape[0])
        if first:
            self.edge.reset_on(x)
        else:
            self.edge.set_history(halo)          # (all K-1 samples of the delay line)
        if H > 0:
            self.edge.step(x[:H], y[:H])

    def step(self, x, y, exchange, first, consumed=None):
        """exchange: a posted HaloExchange (or None); consumed: called once the launches that read the halo are enqueued
        (HaloPipe.consumed).  Returns y."""
        split = self.interior(x, y)      # -> the first output the interior wrote (0: the chunk is no longer than the halo)
        if callable(exchange):
            exchange = exchange()        # (posted AFTER the interior launch: see HaloPipe.post)
        if split:
            def edge():
                halo = exchange.finish() if exchange is not None else None
                self.edge_step(x, y, halo, first, split)
                if consumed is not None:
                    consumed()
            self._on_edge(x, edge, y)
        else:                                # a chunk no longer than the halo: nothing to overlap
            halo = exchange.finish() if exchange is not None else None
            if first:
                self.main.reset_on(x)
            else:
                self.main.set_history(halo)
            self.main.step(x, y)
            if consumed is not None:
                consumed()
        return y


class OverlappedResampler(_EdgeStream):
    """Resampler (tsdgpu_resampler) at stream position `pos`: H = K - 1 window samples.  The edge produces the outputs of
    the chunk's first H inputs, the interior those of the rest -- disjoint output ranges whose border comes from the
    schedule (out_offset), so the result is bit for bit the single call's."""

    def __init__(self, t, ratio, data_type, edge_stream=False):
        self.main = t.Resampler(ratio, data_type)
        self.edge = t.Resampler(ratio, data_type)
        self.H = self.main.K - 1
        self._init_edge_stream(edge_stream)

    def counts(self, pos, n):
        """-> (outputs of the first min(H, n) inputs, outputs of all n) from stream position pos"""
        self.edge.seek(pos)
        o0 = self.edge.out_offset
        self.edge.seek(pos + min(self.H, n))
        o1 = self.edge.out_offset
        self.edge.seek(pos + n)
        return o1 - o0, self.edge.out_offset - o0

    def step(self, x, y, pos, exchange, first, consumed=None):
        n, H = x.shape[0], self.H
        c_edge, c_all = self.counts(pos, n)
        assert y.shape[0] >= c_all
        split = n > H
        if split:
            self.main.seek(pos + H, x[:H])
            self.main.step(x[H:], y[c_edge:])
        if callable(exchange):
            exchange = exchange()

        def edge():
            halo = exchange.finish() if exchange is not None else None
            self.edge.seek(pos, None if first else halo)
            if split:
                if c_edge > 0:
                    self.edge.step(x[:H], y[:c_edge])
            else:
                self.edge.step(x, y)
            if consumed is not None:
                consumed()
        if split:
            self._on_edge(x, edge)
        else:
            edge()
        return y[:c_all]


class OverlappedSos(_EdgeStream):
    """SOS cascade (tsdgpu_sos) sharded with a warm-up halo of W = Sos.halo samples (state transition below 1e-9 after W
    samples).  The interior is warmed up on the chunk's OWN first W samples, the edge -- the first W outputs -- on the
    neighbour's.  The first rank of a stream runs one plain step (first-sample seed, filtre-rt.cc:361-365)."""

    def __init__(self, t, coefs, gain, data_type, rii1=None, edge_stream=False):
        self.main = t.Sos(coefs, gain, data_type, rii1)
        self.edge = t.Sos(coefs, gain, data_type, rii1)
        self.W = int(self.main.halo)
        assert self.W >= 0, "this cascade does not decay: use sos_step_exact"
        self.scratch = None              # the interior's warm-up outputs (discarded)
        self.scratch_e = None            # the edge's (its own: the two may run on different streams)
        self._init_edge_stream(edge_stream)

    def step(self, x, y, exchange, first, consumed=None):
        W, n = self.W, x.shape[0]
        if first:
            if callable(exchange):
                exchange = exchange()
            if exchange is not None:
                exchange.finish()
            self.main.reset_on(x)
            self.main.step(x, y)
            if consumed is not None:
                consumed()
            return y
        if self.scratch is None or self.scratch.shape[0] < W:
            self.scratch = x.new_empty(max(W, 1))
            self.scratch_e = x.new_empty(max(W, 1))
        split = n > W
        if split:
            self.main.reset_on(x)
            on_dev = getattr(x, "is_cuda", False)
            if on_dev and x.data_ptr() != y.data_ptr() and (W * (2 if x.is_complex() else 1)) % 4 == 0:
                self.main.step_skip(x, y, W)                 # warm-up and interior in ONE launch (tsdgpu_sos_step_skip)
            else:
                self.main.step(x[:W], self.scratch[:W])
                self.main.step(x[W:], y[W:])
        if callable(exchange):
            exchange = exchange()

        def edge():
            halo = exchange.finish()
            self.edge.reset_on(x)
            self.edge.step(halo, self.scratch_e[:W])
            self.edge.step(x[:W] if split else x, y[:W] if split else y)
            if consumed is not None:
                consumed()
        if split:
            self._on_edge(x, edge, y)      # (in place: the edge overwrites x[:W], which the interior's warm-up reads)
        else:
            edge()
        return y


def max_over_ranks(value, device, world, force=False):
    """all_reduce(MAX) of a host scalar (`force`: also with one rank, so that the collective runs)"""
    if world == 1 and not force:
        return value
    t = torch.tensor([value], device="cpu" if dist.get_backend() == "gloo" else device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sos_step_exact(sos, x_chunk, chunk_len, rank, world, stream_state=None, group=None, force_collective=False):
    """One call of a cascade whose memory is too long for a warm-up halo (tsdgpu_sos_halo < 0, or longer than a
    chunk is worth), one rank per chunk: the path's one real exchange step, so the one place with a collective.

    The first rank WITH samples filters its chunk from the stream's state, every later one from zero memories; the
    end states E_r (a few floats per section, Sos.get_state) are ALL-GATHERED; rank r rebuilds its true start state
    S_r = propagate(L_{r-1}, S_{r-1}, E_{r-1}), S_{first+1} = E_first (Sos.propagate_state: the cascade's transition
    matrix over a chunk, in double, on the host) and filters its chunk again.  `stream_state`: the state of the stream
    before this call (None = a fresh stream: the first sample seeds the sections).  Returns (y_chunk, state of the
    stream after the call) -- the latter on every rank, for the next call."""
    import numpy as np
    # every state transfer of this call runs on the stream the cascade's step runs on (the chunk's current torch stream):
    # a null-stream copy would not wait for a step launched on a non-blocking stream and return the stale pre-step state
    st = None
    if hasattr(x_chunk, "is_cuda") and x_chunk.is_cuda:
        st = torch.cuda.current_stream(x_chunk.device).cuda_stream

    def get_state():
        return sos.get_state(st) if st is not None else sos.get_state()

    def set_state(v):
        return sos.set_state(v, st) if st is not None else sos.set_state(v)

    def step(xc):
        return sos.step(xc, None, st) if st is not None else sos.step(xc)

    nf = get_state().size
    fresh = np.zeros(nf, np.float32)
    before = fresh if stream_state is None else np.asarray(stream_state, np.float32)
    if world == 1 and not force_collective:
        if chunk_len <= 0:
            return x_chunk, before
        set_state(before)
        return step(x_chunk), get_state()
    on_dev = dist.get_backend(group) != "gloo"
    dev = x_chunk.device if (on_dev and hasattr(x_chunk, "device")) else "cpu"

    def all_gather(vec, dtype=np.float32):
        t_mine = torch.from_numpy(np.asarray(vec, dtype)).to(dev)
        parts = [torch.empty_like(t_mine) for _ in range(world)]
        dist.all_gather(parts, t_mine, group=group)          # RCCL on the GPU node, gloo in the rehearsals
        return [p.cpu().numpy() for p in parts]

    # the chunk lengths travel as int64: a float32 rounds any length beyond 2^24 samples (the bench shape is 2^26 per
    # rank), and propagate_state would then apply Phi^(L +- k) instead of Phi^L
    lens = [int(v[0]) for v in all_gather([int(chunk_len)], np.int64)]
    with_samples = [q for q in range(world) if lens[q] > 0]
    if not with_samples:
        return x_chunk, before
    first = with_samples[0]
    zero = np.zeros(nf, np.float32)
    zero[0] = 1.0                      # zero memories, first-sample seed spent
    y, end = x_chunk, np.zeros(nf, np.float32)
    if chunk_len > 0:
        set_state(before if rank == first else zero)
        y, end = step(x_chunk), get_state()
    ends = all_gather(end)
    cur = ends[first]                  # the true state after the first chunk
    for q in range(first + 1, world):
        if lens[q] <= 0:
            continue
        if q == rank:
            set_state(cur)
            y = step(x_chunk)
        cur = sos.propagate_state(lens[q], cur, ends[q])
    return y, cur
