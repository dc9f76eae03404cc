#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path.

  python bench.py --gpus N --steps K --warmup W [--workload NAME] [--pipeline fused|wavefront] [--spp S]

One step = one pass of the render loop over the workload's image at its samples per
pixel (default workload: random spheres, 1920x1080, 1024 spp, depth 50, on one GPU), with
the scene and the accumulator resident in HBM before the timed region.  With N > 1 every
rank renders its own disjoint range of sample indices of the same image (weak scaling:
spp per GPU is fixed) and the per-GPU radiance sums are added with one RCCL all-reduce
inside the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import abi, roofline, scenes  # noqa: E402
from raytracer_amd.scene import RenderConfig  # noqa: E402

FP32_VALU_PEAK_TFLOPS = 157.3      # MI355X vector FP32 peak (= its FP32 matrix peak)


make_scene = scenes.make_workload


def cpu_baseline(name):
    """The CPU oracle (double precision restatement of the reference renderer) timed on a
    bounded sample of the same workload: the same scene at 960x540, 256 spp (10-25 s on 16 threads)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bridge as ob
    scene, w, h, spp, depth, seed = make_scene(name)
    if name.startswith("cfg5"):
        return None
    orc = ob.oracle_scene_from(scene)
    cw, ch, cspp = min(w, 960), min(h, 540), 256
    threads = min(16, os.cpu_count() or 1)
    orc.render(scene.camera, RenderConfig(1, depth, 1, seed, 0), cw, ch, threads=threads)      # warm
    t0 = time.perf_counter()
    _, used, _ = orc.render(scene.camera, RenderConfig(cspp, depth, 1, seed, 0), cw, ch, precision=ob.ORC_F64, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": cw * ch * cspp / dt / 1e6, "unit": "Msamples/s", "cores": used, "kind": "port",
            "sample": "%s scene at %dx%d, %d spp, depth %d, f64, %.1f s" % (name, cw, ch, cspp, depth, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2_random_spheres_1080p")
    ap.add_argument("--pipeline", default="fused", choices=["fused", "wavefront"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per step (0 = the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bvh", default="sah", choices=["sah", "median"])
    ap.add_argument("--refill", type=int, default=0)
    ap.add_argument("--node-min", type=int, default=-1)
    ap.add_argument("--nodes", default="auto", choices=["auto", "f32", "f16"])
    args = ap.parse_args()
    # a run that stops making progress ends with the Python stacks of all threads instead of hanging
    faulthandler.dump_traceback_later(int(os.environ.get("RT_BENCH_WATCHDOG", "1200")), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RT_DIST_BACKEND=gloo lets the multi-rank path be rehearsed where the ranks share one GPU
    backend = os.environ.get("RT_DIST_BACKEND", "nccl")
    local_rank %= max(1, torch.cuda.device_count())
    # RT_FORCE_DIST=1 takes the collective path with a single rank too (a check that RCCL initialises
    # and reduces next to the HIP library on a one-GPU box)
    use_dist = world > 1 or bool(os.environ.get("RT_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)

    from raytracer_amd.renderer import Renderer
    scene, w, h, spp, depth, seed = make_scene(args.workload)
    if args.workload.startswith("cfg5"):
        spp = spp // 8          # 1024 per GPU
    if args.spp:
        spp = args.spp
    pipeline = abi.RT_PIPELINE_FUSED if args.pipeline == "fused" else abi.RT_PIPELINE_WAVEFRONT
    r = Renderer(scene, w, h, pipeline=pipeline, device_ordinal=local_rank,
                 bvh=abi.RT_BVH_SAH if args.bvh == "sah" else abi.RT_BVH_MEDIAN_SPLIT, refill=args.refill or None, node_min=None if args.node_min < 0 else args.node_min,
                 nodes={"auto": abi.RT_NODES_AUTO, "f32": abi.RT_NODES_F32, "f16": abi.RT_NODES_F16}[args.nodes])

    trace = bool(os.environ.get("RT_BENCH_TRACE"))

    def step(i):
        # rank g renders sample indices [ (i*world + g)*spp, +spp ): disjoint over ranks and steps
        if use_dist:
            r.accum.zero_()      # each step reduces its own sums (a running total would be re-added)
        r.draw(spp, depth, seed, first_sample=(i * world + rank) * spp)
        if trace:
            torch.cuda.synchronize()
            print("rank %d step %d rendered" % (rank, i), file=sys.stderr, flush=True)
        if use_dist:
            if backend != "nccl":
                # gloo stages the tensor through the host from a helper thread; with two ranks
                # time-sliced on one GPU that wait was seen to stall behind the running kernel
                torch.cuda.synchronize()
            dist.all_reduce(r.accum, op=dist.ReduceOp.SUM)
        if trace:
            torch.cuda.synchronize()
            print("rank %d step %d reduced" % (rank, i), file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    r.reset()
    r.kernel_time(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    k_ms, k_n, k_name = r.kernel_time()

    if rank == 0:
        samples_step = w * h * spp * world
        value = samples_step * args.steps / dt / 1e6
        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": w, "height": h, "spp_per_gpu_per_step": spp,
                       "max_depth": depth, "seed": seed, "pipeline": args.pipeline,
                       "parallelism": "spp-sharded x%d, RCCL all-reduce of the f32 accumulator" % world if world > 1 else "single GPU"},
        }
        fixture = os.path.join(ROOT, "tests", "golden", "counters_%s.json" % args.workload)
        if os.path.exists(fixture) and k_n:
            cnt = json.load(open(fixture))["counters"]
            flops, nbytes, trace_flops = roofline.per_sample(cnt)
            samples_launch = w * h * spp * args.steps / k_n
            f_launch = (flops if k_name == "k_fused" else trace_flops) * samples_launch
            achieved = f_launch / (k_ms / k_n * 1e-3) / 1e12
            # HBM bytes per launch from the PMC passes kept under profiles/ (they cannot be collected in
            # this process); only quoted when they were measured at this launch size
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
            if os.path.exists(tfile) and k_name == "k_fused":
                measured = json.load(open(tfile))
                if measured["samples_per_launch"] == samples_launch:
                    traffic = measured["hbm_bytes_per_launch"]
            out["roofline"] = {"bound": "valu", "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / FP32_VALU_PEAK_TFLOPS, "traffic": traffic, "kernel": k_name,
                               "kernel_ms_avg": k_ms / k_n, "launches": k_n,
                               "flops_per_sample": flops, "algorithmic_bytes_per_sample": nbytes,
                               "algorithmic_GBps": nbytes * samples_launch / (k_ms / k_n * 1e-3) / 1e9}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out))
    faulthandler.cancel_dump_traceback_later()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
