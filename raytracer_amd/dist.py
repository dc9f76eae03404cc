"""Multi-GPU: sample-index sharding + one all-reduce of the radiance sums.

Samples are independent and sample s of pixel (x, y) draws from a generator addressed by
(seed, s, x, y), so rank g of G simply renders its own slice of sample indices of every
pixel; the per-rank float32 sums are then added with a single collective (RCCL over xGMI
with backend "nccl", gloo on CPU).  No halo, no other exchange.
"""
import torch.distributed as dist


def shard_range(rank, world, total_spp):
    """[first, first + count) sample indices of `rank`: contiguous, disjoint, covering
    total_spp; the first (total_spp % world) ranks take one extra sample."""
    base, extra = divmod(total_spp, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def all_reduce_sums(accum, group=None):
    """Adds the per-rank radiance sums in place on every rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return accum


def render_sharded(renderer, total_spp, max_depth, rng_seed, first_sample=0, group=None):
    """Renders total_spp samples per pixel across the ranks of `group` into
    renderer.accum (every rank ends with the full sums) and returns total_spp."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = shard_range(rank, world, total_spp)
    renderer.reset()
    if count:
        renderer.draw(count, max_depth, rng_seed, first_sample=first_sample + first)
    all_reduce_sums(renderer.accum, group)
    renderer.samples = total_spp
    return total_spp
