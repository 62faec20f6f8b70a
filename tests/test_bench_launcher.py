"""bench.py --gpus N must start N ranks itself (before anything touches the GPU) and shard a fixed total batch over
them; rehearsed here with gloo on the CPU (`--dry-run-cpu` skips every device call and claims no numbers)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu"] + list(flags), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    return json.loads(lines[0])


def test_self_launch_two_ranks_strong_scaling():
    line = _run("--gpus", "2", "--total-batch", "11")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["total_batch"] == 11 and line["batch_rank0"] == 6  # ragged split 6 + 5, gathered back in order


def test_weak_mode_and_single_rank():
    line = _run("--gpus", "2", "--batch", "3")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["total_batch"] == 6
    one = _run("--gpus", "1", "--total-batch", "5")
    assert one["n_gpus"] == 1 and one["batch_rank0"] == 5


def test_a_failing_rank_fails_the_launch():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    # without --dry-run-cpu the ranks need a GPU: on a CPU-only host each exits non-zero and so must the launcher
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
