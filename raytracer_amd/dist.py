"""Multi-GPU: sample-index sharding + one all-reduce of the radiance sums.

Samples are independent and sample s of pixel (x, y) draws from a generator addressed by
(seed, s, x, y), so rank g of G simply renders its own slice of sample indices of every
pixel; the per-rank float32 sums are then added with a single collective (RCCL over xGMI
with backend "nccl", gloo on CPU).  No halo, no other exchange.
"""
import os
import socket
import subprocess
import sys

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


# ---------------------------------------------------------------- image-space sharding
def band_rows(height, index, count):
    """Rows held by shard `index` of `count`: the image is cut into bands of 8 pixel rows and
    band b belongs to shard b % count (the rule of rt_band_rows / rt_render_bands)."""
    bands = (height + 7) // 8
    mine = (bands - index + count - 1) // count if index < bands else 0
    return 8 * mine


def assemble_bands(gathered, height):
    """[count, rows0, W, 3] equal-sized shard buffers (rows0 = band_rows(height, 0, count)) -> the
    [height, W, 3] image: band b sits in shard b % count at local band b // count, so ordering by
    (local band, shard) restores image order.  No arithmetic: every pixel is held exactly once."""
    count, rows0, width, ch = gathered.shape
    g = gathered.reshape(count, rows0 // 8, 8, width, ch).permute(1, 0, 2, 3, 4)
    return g.reshape(-1, width, ch)[:height]


def all_gather_bands(accum, height, group=None):
    """Gathers every rank's band buffer and returns the full image on every rank."""
    import torch
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return assemble_bands(accum.unsqueeze(0), height)
    # the shards concatenated along the rows (the layout every backend accepts), then seen per shard
    gathered = torch.empty((world * accum.shape[0],) + tuple(accum.shape[1:]), dtype=accum.dtype, device=accum.device)
    dist.all_gather_into_tensor(gathered, accum.contiguous(), group=group)
    return assemble_bands(gathered.reshape((world,) + tuple(accum.shape)), height)


# ---------------------------------------------------------------- launching the ranks
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(argv, world, extra_env=None, timeout=None):
    """Starts `world` copies of the command `argv` (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in their environment), waits for all of them and returns
    (exit code, rank 0's standard output).  The caller must not have touched the GPU: the
    children are fresh processes, nothing is exec'ed over an initialised one.  A rank that fails
    ends the others."""
    import tempfile
    import time
    port = free_port()
    procs = []
    with tempfile.TemporaryFile() as out0:
        for rank in range(world):
            env = dict(os.environ)
            env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
            env.update(extra_env or {})
            procs.append(subprocess.Popen(argv, env=env, stdout=out0 if rank == 0 else subprocess.DEVNULL))
        code = 0
        deadline = None if timeout is None else time.monotonic() + timeout
        try:
            while True:
                states = [p.poll() for p in procs]
                failed = [c for c in states if c not in (None, 0)]
                if failed:
                    code = failed[0]
                    break
                if all(c == 0 for c in states):
                    break
                if deadline is not None and time.monotonic() > deadline:
                    code = 124
                    break
                time.sleep(0.05)
        finally:
            for p in procs:             # the exact processes started above, nothing by pattern
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
        out0.seek(0)
        return code, out0.read().decode(errors="replace")
