"""The multi-GPU rule on CPU: world size 2 over gloo.  Each rank renders its slice of the
sample indices (with the CPU oracle standing in for the device renderer, which needs a
GPU) and the all-reduced sums must equal one rank rendering the whole range."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.dist import shard_range

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_ranges_cover_without_overlap():
    for world in (1, 2, 3, 8):
        for spp in (1, 7, 8, 1024, 8192):
            seen = []
            for r in range(world):
                first, count = shard_range(r, world, spp)
                seen += list(range(first, first + count))
            assert seen == list(range(spp)), (world, spp)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle_bridge as ob
    from raytracer_amd import scenes
    from raytracer_amd.dist import all_reduce_sums, shard_range
    from raytracer_amd.scene import RenderConfig
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = scenes.simple_balls()
    orc = ob.oracle_scene_from(scene)
    first, count = shard_range(rank, world, 6)
    s, _, _ = orc.render(scene.camera, RenderConfig(count, 8, 1, 77, first), 24, 16, threads=1)
    acc = torch.from_numpy(s.astype(np.float32))
    all_reduce_sums(acc)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    import oracle_bridge as ob
    from raytracer_amd import scenes
    from raytracer_amd.scene import RenderConfig
    out = str(tmp_path / "sum.npy")
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    scene = scenes.simple_balls()
    orc = ob.oracle_scene_from(scene)
    ref, _, _ = orc.render(scene.camera, RenderConfig(6, 8, 1, 77, 0), 24, 16, threads=1)
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_band_rows_and_assembly():
    from raytracer_amd.dist import assemble_bands, band_rows
    for height in (1, 8, 9, 21, 135 * 8, 1080, 2160):
        for count in (1, 2, 3, 8):
            rows = [band_rows(height, i, count) for i in range(count)]
            assert sum(rows) == 8 * ((height + 7) // 8) and rows[0] == max(rows)
            # every image row lands where assemble_bands looks for it
            image = torch.arange(height, dtype=torch.float32).reshape(height, 1, 1).expand(height, 2, 3)
            shards = torch.full((count, rows[0], 2, 3), -1.0)
            for band in range((height + 7) // 8):
                n = min(8, height - 8 * band)
                shards[band % count, 8 * (band // count):8 * (band // count) + n] = image[8 * band:8 * band + n]
            assert torch.equal(assemble_bands(shards, height), image)


def test_launcher_runs_both_exchanges_with_two_ranks(tmp_path):
    """bench.py --gpus N starts its ranks with raytracer_amd.dist.launch_ranks; here the same launcher
    starts two CPU ranks (gloo) for each sharding mode, and the exchanged image must be the one a
    single rank renders: sample-sharded sums added by all-reduce, band shards put together by
    all-gather."""
    import oracle_bridge as ob
    from raytracer_amd import scenes
    from raytracer_amd.dist import launch_ranks
    from raytracer_amd.scene import RenderConfig
    scene = scenes.simple_balls()
    orc = ob.oracle_scene_from(scene)
    ref, _, _ = orc.render(scene.camera, RenderConfig(6, 8, 1, 77, 0), 24, 21, threads=1)
    for mode in ("spp", "bands"):
        out = str(tmp_path / ("%s.npy" % mode))
        code, text = launch_ranks([sys.executable, os.path.join(HERE, "dist_worker.py"), mode, out], 2, timeout=300)
        assert code == 0, text
        assert "world 2" in text
        got = np.load(out)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), mode
    # a failing rank ends the job with its code instead of leaving the others waiting
    code, _ = launch_ranks([sys.executable, "-c", "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(60)"],
                           2, timeout=30)
    assert code == 7
