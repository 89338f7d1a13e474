"""Splitting one frame over the GPUs of a node: one process per GPU.

The sample loop shards by sample index: rank g renders samples
[begin_g, begin_g + count_g) of EVERY pixel with the whole scene replicated in
its own HBM.  Sample s of pixel p always draws from RNG stream (seed, p, s),
so the set of radiance values that are summed does not depend on the number
of ranks; only the order of the f64 additions does.  Nothing is exchanged
while rendering.  The one collective is a sum-reduce of the per-rank
accumulation buffers ([H][W][3] f64) onto rank 0 at the end of the frame,
which is RCCL over xGMI when the tensors live on GPUs ("nccl" backend) and
gloo in the CPU tests.

The reference renders on one host with a rayon thread pool and has no
multi-device path; its equivalent of the reduce is the per-pixel
`.reduce(|| ZERO, |a, b| a + b)` of src/camera.rs:147-148.
"""
import torch
import torch.distributed as dist


def spp_range(rank, world, spp):
    """(begin, count) of the sample indices rank `rank` of `world` renders.

    The ranges are contiguous, disjoint, cover [0, spp) and differ in length
    by at most one: 500 over 8 ranks -> 63,63,63,63,62,62,62,62.
    """
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} out of range for world size {world}")
    if spp < 0:
        raise ValueError("spp must be >= 0")
    base, rem = divmod(spp, world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def reduce_accum(accum, world, dst=0):
    """Sums the per-rank accumulation buffers into rank `dst`'s buffer in
    place (the one collective of the frame).  No-op for a single rank."""
    if accum.dtype != torch.float64:
        raise TypeError("accumulation buffers are f64")
    if world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def collective_ranks(device=None):
    """How many ranks the frame's collective reaches: every rank contributes 1.0 to a one-element buffer that goes
    through the same sum-reduce as the accumulation buffers; the root's result is returned on every rank (broadcast).
    1 without a process group.  Also brings the communicator up before anything is timed."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    one = torch.ones(1, dtype=torch.float64, device=device)
    reduce_accum(one, dist.get_world_size())
    dist.broadcast(one, src=0)
    return int(round(float(one.item())))


def render_frame(render_accumulate, accum, rank, world, spp, scale=None):
    """One frame on `world` ranks.

    render_accumulate(begin, count) adds this rank's radiance sums into
    `accum` and returns its stats (or None); it is not called for an empty
    range.  After the reduce, rank 0 applies scale(accum) if given (the
    x 1/spp of src/camera.rs:146).  Returns (stats, (begin, count)).
    """
    begin, count = spp_range(rank, world, spp)
    stats = render_accumulate(begin, count) if count else None
    reduce_accum(accum, world)
    if rank == 0 and scale is not None:
        scale(accum)
    return stats, (begin, count)
