#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path.

  python bench.py --gpus N --steps K --warmup W [--workload NAME] [--shard spp|bands]
                  [--pipeline fused|wavefront] [--spp S]

One step = one pass of the render loop over the workload's image at its samples per
pixel (default workload: random spheres, 1920x1080, 1024 spp, depth 50, on one GPU), with
the scene and the accumulator resident in HBM before the timed region.

With N > 1 the job is one process per GPU.  `python bench.py --gpus N` starts the N ranks
itself (fresh child processes, before this process touches a GPU); under torchrun the
ranks are already there (RANK / WORLD_SIZE in the environment) and --gpus must agree with
WORLD_SIZE.  Weak scaling: per-GPU work is fixed.
  --shard spp    every rank renders its own disjoint range of sample indices of the whole
                 image (N x spp samples per pixel in total); the per-GPU radiance sums are
                 added with one RCCL all-reduce inside the timed region.
  --shard bands  every rank renders N x spp samples per pixel of its own bands of 8 pixel
                 rows (band b belongs to rank b % N); the band buffers are put together
                 with one RCCL all-gather, no arithmetic.
Rank 0 prints one JSON line.
"""
import argparse
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3      # MI355X vector FP32 peak (= its FP32 matrix peak)


def host_threads():
    """Threads for the CPU baseline: the CPUs this process may run on, within the share the
    machine's operator set for one job (OMP_NUM_THREADS on a shared multi-GPU host, where
    os.cpu_count() reports every CPU of the machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_baseline(name):
    """The CPU oracle (double precision restatement of the reference renderer) timed on a
    bounded sample of the same workload: the same scene at 960x540, 256 spp."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bridge as ob
    from raytracer_amd import scenes
    from raytracer_amd.scene import RenderConfig
    if name.startswith("cfg5"):
        return None             # building the million-sphere oracle tree alone takes longer than the bench
    scene, w, h, spp, depth, seed = scenes.make_workload(name)
    orc = ob.oracle_scene_from(scene)
    cw, ch, cspp = min(w, 960), min(h, 540), 256
    threads = host_threads()
    orc.render(scene.camera, RenderConfig(1, depth, 1, seed, 0), cw, ch, threads=threads)      # warm
    t0 = time.perf_counter()
    _, used, _ = orc.render(scene.camera, RenderConfig(cspp, depth, 1, seed, 0), cw, ch, precision=ob.ORC_F64, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": cw * ch * cspp / dt / 1e6, "unit": "Msamples/s", "cores": used, "kind": "port",
            "host_cpus_visible": os.cpu_count(),
            "sample": "%s scene at %dx%d, %d spp, depth %d, f64, %.1f s on %d threads (the job's CPU share; the host shows %d CPUs)"
                      % (name, cw, ch, cspp, depth, dt, used, os.cpu_count() or 0)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2_random_spheres_1080p")
    ap.add_argument("--pipeline", default="fused", choices=["fused", "wavefront"])
    ap.add_argument("--shard", default="spp", choices=["spp", "bands"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per GPU per step (0 = the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bvh", default="sah", choices=["sah", "median"])
    ap.add_argument("--refill", type=int, default=0)
    ap.add_argument("--node-min", type=int, default=-1)
    ap.add_argument("--nodes", default="auto", choices=["auto", "f32", "f16"])
    ap.add_argument("--heavy-min", type=int, default=-1)
    ap.add_argument("--traversal", default="auto", choices=["auto", "tree", "list"])
    return ap.parse_args()


def start_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks and
    forwards rank 0's line.  Nothing here initialises a GPU (device_count does not)."""
    import torch
    from raytracer_amd import dist as rdist
    backend = os.environ.get("RT_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        print("bench.py: --gpus %d asked for but only %d GPU(s) are visible" % (args.gpus, have), file=sys.stderr)
        return 2
    code, out = rdist.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
    sys.stdout.write(out)
    sys.stdout.flush()
    return code


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(start_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d does not match WORLD_SIZE %d of the launcher" % (args.gpus, world))
    # a run that stops making progress ends with the Python stacks of all threads instead of hanging
    faulthandler.dump_traceback_later(int(os.environ.get("RT_BENCH_WATCHDOG", "1200")), exit=True)

    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, roofline, scenes
    from raytracer_amd import dist as rdist

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
        world = dist.get_world_size()       # what the collective library reports is what gets printed
    torch.cuda.set_device(local_rank)

    from raytracer_amd.renderer import Renderer
    scene, w, h, spp, depth, seed = scenes.make_workload(args.workload)
    if args.workload.startswith("cfg5"):
        spp = spp // 8          # the workload is quoted as 8192 spp over 8 GPUs: 1024 per GPU
    if args.spp:
        spp = args.spp
    bands = args.shard == "bands"
    if bands and args.pipeline != "fused":
        sys.exit("bench.py: --shard bands runs on the fused pipeline")
    pipeline = abi.RT_PIPELINE_FUSED if args.pipeline == "fused" else abi.RT_PIPELINE_WAVEFRONT
    r = Renderer(scene, w, h, pipeline=pipeline, device_ordinal=local_rank,
                 bvh=abi.RT_BVH_SAH if args.bvh == "sah" else abi.RT_BVH_MEDIAN_SPLIT, refill=args.refill or None,
                 node_min=None if args.node_min < 0 else args.node_min,
                 nodes={"auto": abi.RT_NODES_AUTO, "f32": abi.RT_NODES_F32, "f16": abi.RT_NODES_F16}[args.nodes],
                 timing=True, bands=(rank, world) if bands else None, heavy_min=None if args.heavy_min < 0 else args.heavy_min,
                 traversal={"auto": abi.RT_TRAVERSAL_AUTO, "tree": abi.RT_TRAVERSAL_TREE, "list": abi.RT_TRAVERSAL_LIST}[args.traversal])

    trace = bool(os.environ.get("RT_BENCH_TRACE"))
    image = [None]

    def step(i):
        if use_dist or bands:
            r.accum.zero_()      # each step exchanges its own sums (a running total would be re-added)
        if bands:
            # all ranks render the same world*spp sample indices, each of its own bands
            r.draw(spp * world, depth, seed, first_sample=i * world * spp)
        else:
            # rank g renders sample indices [ (i*world + g)*spp, +spp ): disjoint over ranks and steps
            r.draw(spp, depth, seed, first_sample=(i * world + rank) * spp)
        if trace:
            torch.cuda.synchronize()
            print("rank %d step %d rendered" % (rank, i), file=sys.stderr, flush=True)
        if use_dist and backend != "nccl":
            # gloo stages the tensor through the host from a helper thread; with two ranks
            # time-sliced on one GPU that wait was seen to stall behind the running kernel
            torch.cuda.synchronize()
        if bands:
            image[0] = rdist.all_gather_bands(r.accum, h)
        elif use_dist:
            dist.all_reduce(r.accum, op=dist.ReduceOp.SUM)
        if trace:
            torch.cuda.synchronize()
            print("rank %d step %d exchanged" % (rank, i), file=sys.stderr, flush=True)

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
        library = "RCCL" if backend == "nccl" else backend
        if world == 1:
            parallelism = "single GPU"
        elif bands:
            parallelism = "bands of 8 rows interleaved over %d ranks, %s all-gather of the band buffers" % (world, library)
        else:
            parallelism = "spp-sharded x%d, %s all-reduce of the f32 accumulator" % (world, library)
        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": w, "height": h, "spp_per_gpu_per_step": spp,
                       "max_depth": depth, "seed": seed, "pipeline": args.pipeline, "shard": args.shard if world > 1 else "none",
                       "parallelism": parallelism},
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
            roof = {"bound": "valu", "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP32_VALU_PEAK_TFLOPS, "traffic": traffic, "kernel": k_name,
                    "kernel_ms_avg": k_ms / k_n, "launches": k_n,
                    "flops_per_sample": flops, "algorithmic_bytes_per_sample": nbytes,
                    # bytes the reference's algorithm reads per sample x the sample rate: served by the
                    # caches (the scene is far smaller than the traffic), not an HBM figure
                    "cache_level_algorithmic_GBps": nbytes * samples_launch / (k_ms / k_n * 1e-3) / 1e9}
            # the same fraction priced at the work this kernel executes (its own tree and walk order),
            # from the phase counters of the diagnostic build kept under profiles/
            efile = os.path.join(ROOT, "profiles", "executed_%s.json" % args.workload)
            if os.path.exists(efile) and k_name == "k_fused":
                ex = json.load(open(efile))
                roof["executed_flops_per_sample"] = ex["executed_flops_per_sample"]
                roof["frac_executed"] = ex["executed_flops_per_sample"] * samples_launch / (k_ms / k_n * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
