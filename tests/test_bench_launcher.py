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


def test_run_configs_self_launches_ranks_and_gathers():
    """tools/run_configs.py --gpus N uses the same launcher (fresh ranks, nothing re-executed) and gathers per-circuit results"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_configs.py"), "--dry-run-cpu", "--gpus", "2", "--config", "4", "5",
                        "--batch", "7"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert [ln["config"] for ln in lines] == [4, 5]
    assert all(ln["n_gpus"] == 2 and ln["total_circuits"] == 7 and ln["circuits_rank0"] == 4 for ln in lines)


def test_gather_streams_ragged_blocks_in_rank_order():
    """three ranks, ragged blocks (one of them empty): the point-to-point gather restores the original order"""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from abc_amd.sharding import gather_results, shard_range\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "for total in (2, 7):\n"
        "    a, b = shard_range(total, r, w)\n"
        "    out = gather_results(torch.arange(a, b, dtype=torch.int64).reshape(-1, 1).repeat(1, 3), total)\n"
        "    if r == 0: assert out.shape == (total, 3) and bool((out[:, 2] == torch.arange(total)).all())\n"
        "    else: assert out is None\n"
        "dist.destroy_process_group()\n" % ROOT)
    from abc_amd.launcher import _free_port
    port = _free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stderr=subprocess.PIPE))
    for pr in procs:
        _, err = pr.communicate(timeout=300)
        assert pr.returncode == 0, err.decode()[-2000:]
