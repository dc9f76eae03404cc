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


# Sample of each workload the CPU leg renders (and the GPU renders again after the timed region, for the
# PSNR): the whole image at a quarter of the pixels for the 1080p workloads; for the million-sphere workload
# a window of the 4K image at few samples, because the oracle walks the reference's median-split tree there
# (1535 box tests per sample) and builds it on one thread.
def baseline_sample(name, w, h):
    if name.startswith("cfg5"):
        return {"width": w, "height": h, "window": (w // 2 - 320, h // 2 - 180, 640, 360), "spp": 8}
    return {"width": min(w, 960), "height": min(h, 540), "window": None, "spp": 256}


def cpu_baseline(name):
    """The CPU oracle (double precision restatement of the reference renderer) timed on a
    bounded sample of the same workload.  Returns (record, mean image of the sample)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bridge as ob
    from raytracer_amd import scenes
    from raytracer_amd.scene import RenderConfig
    scene, w, h, spp, depth, seed = scenes.make_workload(name)
    smp = baseline_sample(name, w, h)
    t0 = time.perf_counter()
    orc = ob.oracle_scene_from(scene)
    build_s = time.perf_counter() - t0
    cw, ch, cspp, win = smp["width"], smp["height"], smp["spp"], smp["window"]
    threads = host_threads()
    orc.render(scene.camera, RenderConfig(1, depth, 1, seed, 0), cw, ch, window=(0, 0, 64, 36), threads=threads)      # warm
    t0 = time.perf_counter()
    img, used, _ = orc.render(scene.camera, RenderConfig(cspp, depth, 1, seed, 0), cw, ch, window=win, precision=ob.ORC_F64, threads=threads)
    dt = time.perf_counter() - t0
    n = (win[2] * win[3] if win else cw * ch) * cspp
    where = "a %dx%d window of the %dx%d image" % (win[2], win[3], cw, ch) if win else "%dx%d" % (cw, ch)
    rec = {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": used, "kind": "port",
           "host_cpus_visible": os.cpu_count(),
           "sample": "%s scene at %s, %d spp, depth %d, f64, %.1f s on %d threads (the job's CPU share; the host shows %d CPUs); "
                     "scene flattening and the reference's tree build took another %.1f s on one thread, not counted"
                     % (name, where, cspp, depth, dt, used, os.cpu_count() or 0, build_s)}
    return rec, img / cspp


def psnr_against(ref, r, name, w, h, depth, seed):
    """Mean radiance of the HIP path on the CPU leg's sample (same scene, same seed, same sample
    indices: both sides draw sample s of pixel (x, y) from the same stream) against the oracle's image."""
    import numpy as np
    import torch
    from raytracer_amd.renderer import Renderer
    smp = baseline_sample(name, w, h)
    g = Renderer(r.scene, smp["width"], smp["height"], native=r.native)
    g.draw(smp["spp"], depth, seed, first_sample=0)
    torch.cuda.synchronize()
    got = g.mean().cpu().numpy().astype(np.float64)
    if smp["window"]:
        x0, y0, ww, wh = smp["window"]
        got = got[y0:y0 + wh, x0:x0 + ww]
    a, b = np.minimum(got, 4.0), np.minimum(ref, 4.0)        # clipped like a display would
    mse = float(np.mean((a - b) ** 2))
    return {"psnr_db": 10 * np.log10(1.0 / max(mse, 1e-30)), "rmse": float(np.sqrt(mse)),
            "mean_rel_diff": float(abs(got.mean() - ref.mean()) / max(ref.mean(), 1e-12)),
            "pixels_equal_1e-3": float(np.mean(np.max(np.abs(got - ref), axis=2) <= 1e-3 * np.maximum(1.0, np.max(ref, axis=2)))),
            "against": "CPU oracle (f64) on the cpu_baseline sample, pre-gamma mean radiance clipped at 4, same seed and sample indices"}


def counter_record(name):
    """Lanes per vector instruction and vector-pipe occupancy of the dominant kernel, from the shader-engine counter
    passes kept under profiles/ (rocprofv3 --pmc cannot run inside this process)."""
    path = os.path.join(ROOT, "profiles", "pmc_sq_%s.json" % name)
    if not os.path.exists(path):
        return None
    c = json.load(open(path))
    try:
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4       # the counter is summed over the 8 XCDs
        return {"lanes_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
                # a wave64 instruction occupies the 16-lane pipe for 4 cycles
                "valu_busy": 4.0 * c["SQ_INSTS_VALU"] / simd_cycles,
                "wave_time_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                "spp_per_launch_of_the_counter_run": c.get("_spp"),
                "source": "profiles/pmc_sq_%s.json" % name}
    except (KeyError, ZeroDivisionError):
        return None


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
    forwards rank 0's line.  Nothing here asks the runtime anything about GPUs: each rank checks
    for itself that its device exists (exit code 2 when it does not)."""
    from raytracer_amd import dist as rdist
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
    have = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= have:
        print("bench.py: rank %d of --gpus %d has no GPU (%d visible)" % (rank, args.gpus, have), file=sys.stderr)
        sys.exit(2)
    local_rank %= max(1, have)
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
            cr = counter_record(args.workload)
            if cr and k_name == "k_fused":
                roof["counters"] = cr
            out["roofline"] = roof
        if not args.no_cpu_baseline:
            # after the timed region, on rank 0 only: the other ranks wait at the last barrier
            cb, ref = cpu_baseline(args.workload)
            out["cpu_baseline"] = cb
            par = psnr_against(ref, r, args.workload, w, h, depth, seed)
            out["psnr_db"] = par["psnr_db"]
            out["parity"] = par
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()
    if use_dist:
        dist.barrier()          # the other ranks stay until rank 0 has finished its CPU leg
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
