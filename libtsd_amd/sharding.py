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
