"""The collective path of bench.py next to the HIP library on one GPU (GPU needed): a single rank takes
the RCCL route (RT_FORCE_DIST=1) for both sharding rules - all-reduce of the sample-sharded sums, all-gather
of the band buffers - in a child process, as the driver's launcher starts it; and a job that asks for more
GPUs than the box has ends with exit code 2 from the rank that has none."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra, timeout=300):
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--spp", "4",
                        "--no-cpu-baseline"] + extra, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


@pytest.mark.timeout(400)
@pytest.mark.parametrize("shard", ["spp", "bands"])
def test_single_rank_takes_the_rccl_route(shard):
    code, out, err = run_bench(["--shard", shard], {"RT_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1",
                                                    "MASTER_PORT": str(29600 + os.getpid() % 300 + (1 if shard == "bands" else 0))})
    assert code == 0, err[-2000:]
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["spp_per_gpu_per_step"] == 4
    assert line["roofline"]["kernel"] == "k_fused" and line["roofline"]["launches"] == 1


@pytest.mark.timeout(400)
def test_more_ranks_than_gpus_is_refused_by_the_rank_without_one():
    import torch
    have = torch.cuda.device_count()
    code, out, err = run_bench(["--gpus", str(have + 1)], {"MASTER_PORT": str(29950 + os.getpid() % 40)})
    assert code == 2, (code, err[-2000:])
    assert "has no GPU" in err
