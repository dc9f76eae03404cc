"""One rank of the multi-GPU exchange, rehearsed on the CPU: started by raytracer_amd.dist.launch_ranks
(the launcher bench.py uses for --gpus N), gloo as the collective backend, the CPU oracle standing in
for the device renderer.  usage: dist_worker.py spp|bands <out.npy>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_bridge as ob  # noqa: E402
from raytracer_amd import dist as rdist, scenes  # noqa: E402
from raytracer_amd.scene import RenderConfig  # noqa: E402

mode, out_path = sys.argv[1], sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
scene = scenes.simple_balls()
orc = ob.oracle_scene_from(scene)
W, H, SPP, DEPTH, SEED = 24, 21, 6, 8, 77
if mode == "spp":
    first, count = rdist.shard_range(rank, world, SPP)
    s, _, _ = orc.render(scene.camera, RenderConfig(count, DEPTH, 1, SEED, first), W, H, threads=1)
    acc = torch.from_numpy(s.astype(np.float32))
    image = rdist.all_reduce_sums(acc)
else:
    rows0 = rdist.band_rows(H, 0, world)
    acc = torch.zeros((rows0, W, 3), dtype=torch.float32)
    local = 0
    for band in range(rank, (H + 7) // 8, world):          # band b belongs to rank b % world
        y0, rows = 8 * band, min(8, H - 8 * band)
        s, _, _ = orc.render(scene.camera, RenderConfig(SPP, DEPTH, 1, SEED, 0), W, H, window=(0, y0, W, rows), threads=1)
        acc[8 * local:8 * local + rows] = torch.from_numpy(s.astype(np.float32))
        local += 1
    assert 8 * local == rdist.band_rows(H, rank, world)
    image = rdist.all_gather_bands(acc, H)
if rank == 0:
    np.save(out_path, image.numpy())
    print("world %d" % dist.get_world_size())
dist.barrier()
dist.destroy_process_group()
