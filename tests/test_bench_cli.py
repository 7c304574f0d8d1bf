"""bench.py as a command: `python3 bench.py --gpus N` must start WITHOUT a launcher (the seam it serves is one
process, one thread: /root/reference/code/first.cpp:78-88, code/gfx/gfx.cpp:36) - through the native group in
this process, or through N fresh children it starts itself - and print exactly one JSON line.  Without the
GPUs it must say so, not ask for a launcher."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--config", "c2", "--width", "320", "--height", "192", "--spp", "2", "--steps", "2", "--warmup", "1", "--cpu-spp", "0", "--no-also"]


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env)


def _one_json_line(r):
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("transport", ["group", "procs"])
def test_without_the_gpus_it_says_so(transport):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs")
    r = _run(["--gpus", "2", "--transport", transport, *SMALL])
    assert r.returncode != 0
    msg = r.stdout + r.stderr
    assert "needs 2 GPU(s)" in msg and "torch.distributed.run" not in msg and "launch" not in msg.lower()


def test_a_launcher_with_the_wrong_world_is_refused():
    r = _run(["--gpus", "2", *SMALL], env_extra={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["group", "procs"])
def test_two_ranks_rehearsed_on_one_gpu_through_the_self_launch(transport):
    """RT_BENCH_SHARE_GPU=1: both ranks on GPU 0 (group: device copies instead of RCCL; procs: two fresh children
    under torch.distributed.run that the bench starts itself, gloo gather via the host).  One JSON line, the
    contract's keys, the rehearsal named in the workload, and the rays of the two ranks add up to the frame's."""
    one = _one_json_line(_run(["--gpus", "1", *SMALL]))
    two = _one_json_line(_run(["--gpus", "2", "--transport", transport, *SMALL], env_extra={"RT_BENCH_SHARE_GPU": "1"}))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in two, key
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "weak" and two["value"] > 0
    assert two["config"]["transport"] == transport and two["config"]["rccl_ranks"] == 0
    assert "REHEARSAL" in two["config"]["workload"]
    assert two["roofline"]["traffic"] is None                       # nothing measured at N > 1 is reported as measured
    # weak scaling: 2 spp per GPU -> a 4-spp frame split over two ranks; the 1-GPU run traced 2 spp
    assert two["config"]["paths_per_step"] == 2 * one["config"]["paths_per_step"]
    assert 1.9 < two["config"]["rays_per_step"] / one["config"]["rays_per_step"] < 2.1
