"""bench.py's launcher logic that needs no GPU: `--gpus N` without WORLD_SIZE must resolve to an answer (its own ranks, or a
clear error), never to "launch with torch.distributed.run" plumbing advice the driver cannot act on."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DESIRE_BENCH_ONE_GPU")}


def test_gpus_flag_without_enough_devices_is_a_clear_error():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        return
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "64"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 64" in (p.stdout + p.stderr) and "visible" in (p.stdout + p.stderr)


def test_mismatched_world_size_names_both_ways_to_launch():
    env = _env()
    env["WORLD_SIZE"] = "3"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stdout + p.stderr)
