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


def exchange_left_halo(tail_out, halo_in, rank, world, group=None):
    """Every rank sends `tail_out` (the last samples of its chunk) to rank+1 and receives the
    tail of rank-1 into `halo_in`; rank 0's halo_in is left untouched (zeros = the reference's
    empty delay line, filtre-rt.cc:64).  One batched isend/irecv pair per rank."""
    if world == 1:
        return halo_in
    # gloo (CPU tests, single-GPU rehearsals of the multi-rank path) has no point-to-point on device
    # tensors: stage through the host there; RCCL sends device memory directly
    via_host = dist.get_backend(group) == "gloo" and tail_out.is_cuda
    src = tail_out.cpu() if via_host else tail_out
    dst = torch.empty_like(halo_in, device="cpu") if via_host else halo_in
    ops = []
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, src, rank + 1, group))
    if rank > 0:
        ops.append(dist.P2POp(dist.irecv, dst, rank - 1, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if via_host and rank > 0:
        halo_in.copy_(dst)
    return halo_in


def max_over_ranks(value, device, world):
    if world == 1:
        return value
    t = torch.tensor([value], device="cpu" if dist.get_backend() == "gloo" else device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sos_step_exact(sos, x_chunk, chunk_len, rank, world, stream_state=None, group=None):
    """One call of a cascade whose memory is too long for a warm-up halo (tsdgpu_sos_halo < 0, or longer than a
    chunk is worth), one rank per chunk: the path's one real exchange step, so the one place with a collective.

    The first rank WITH samples filters its chunk from the stream's state, every later one from zero memories; the
    end states E_r (a few floats per section, Sos.get_state) are ALL-GATHERED; rank r rebuilds its true start state
    S_r = propagate(L_{r-1}, S_{r-1}, E_{r-1}), S_{first+1} = E_first (Sos.propagate_state: the cascade's transition
    matrix over a chunk, in double, on the host) and filters its chunk again.  `stream_state`: the state of the stream
    before this call (None = a fresh stream: the first sample seeds the sections).  Returns (y_chunk, state of the
    stream after the call) -- the latter on every rank, for the next call."""
    import numpy as np
    nf = sos.get_state().size
    fresh = np.zeros(nf, np.float32)
    before = fresh if stream_state is None else np.asarray(stream_state, np.float32)
    if world == 1:
        if chunk_len <= 0:
            return x_chunk, before
        sos.set_state(before)
        return sos.step(x_chunk), sos.get_state()
    on_dev = dist.get_backend(group) != "gloo"
    dev = x_chunk.device if (on_dev and hasattr(x_chunk, "device")) else "cpu"

    def all_gather(vec):
        t_mine = torch.from_numpy(np.asarray(vec, np.float32)).to(dev)
        parts = [torch.empty_like(t_mine) for _ in range(world)]
        dist.all_gather(parts, t_mine, group=group)          # RCCL on the GPU node, gloo in the rehearsals
        return [p.cpu().numpy() for p in parts]

    lens = [int(v[0]) for v in all_gather([float(chunk_len)])]
    with_samples = [q for q in range(world) if lens[q] > 0]
    if not with_samples:
        return x_chunk, before
    first = with_samples[0]
    zero = np.zeros(nf, np.float32)
    zero[0] = 1.0                      # zero memories, first-sample seed spent
    y, end = x_chunk, np.zeros(nf, np.float32)
    if chunk_len > 0:
        sos.set_state(before if rank == first else zero)
        y, end = sos.step(x_chunk), sos.get_state()
    ends = all_gather(end)
    cur = ends[first]                  # the true state after the first chunk
    for q in range(first + 1, world):
        if lens[q] <= 0:
            continue
        if q == rank:
            sos.set_state(cur)
            y = sos.step(x_chunk)
        cur = sos.propagate_state(lens[q], cur, ends[q])
    return y, cur
