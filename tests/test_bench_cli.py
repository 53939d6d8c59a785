"""CPU: `bench.py --gpus N` without a launcher never reports a one-GPU line under an N-GPU flag -- it drives N devices from
one process through the library's multi-device path, or exits non-zero when the host has fewer than N."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_without_a_launcher_runs_two_slots_or_exits_nonzero():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "cfg3-mini", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() >= 2:
        assert p.returncode == 0, p.stderr[-2000:]
        assert '"n_gpus": 2' in p.stdout
    else:
        assert p.returncode != 0
        assert '"n_gpus"' not in p.stdout
        assert "refusing" in p.stderr


def test_world_size_that_contradicts_gpus_is_refused():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "assert world == args.gpus or world == 1" not in src      # round 5's silent pass
    assert "--gpus %d but WORLD_SIZE is %d" in src
