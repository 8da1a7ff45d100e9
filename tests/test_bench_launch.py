"""bench.py's own launcher on a box without a GPU: `python bench.py --gpus 2` (no torch.distributed.run around it, the form
the driver's N = 1 command has) must start two ranks as child processes. Without a GPU each rank stops at "needs a GPU" -
the product path has no CPU fallback - and the parent leaves with a non-zero status and no JSON line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("covered on the GPU box by tests/test_gpu_bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stdout.strip() == ""  # no line is better than a wrong line
    assert "self-launch:" in r.stderr and "--nproc-per-node=2" in r.stderr
    # the ranks started and reached the device check (the launcher tears the second one down as soon as the first fails,
    # so one message is all that is guaranteed)
    assert r.stderr.count("bench.py needs a GPU") >= 1
