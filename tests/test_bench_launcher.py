"""bench.py's rank launcher, on the CPU (no GPU needed: every case ends before a HIP call or fails at
the first one).  `python bench.py --gpus N` must start N ranks itself when no outer launcher did, and
must never print a line whose `n_gpus` is not the number of ranks that ran."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True,
                          timeout=600, cwd=ROOT)


def test_world_size_must_equal_gpus():
    out = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and not out.stdout.strip()


def test_more_gpus_than_devices_is_refused():
    import torch

    if torch.cuda.device_count() >= 64:
        return
    out = _run(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert out.returncode == 2 and "refusing" in out.stderr and not out.stdout.strip()


def test_self_launch_starts_ranks_and_relays_their_exit_code():
    """Here (no HIP device) the two child ranks start, reach bench.py's own "needs a HIP device" error
    and the parent relays the launcher's non-zero exit code without printing a result line."""
    import torch

    if torch.cuda.device_count() > 0:
        return  # (on a GPU box the same invocation is a -m gpu test: tests/test_gpu_sharding.py)
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--shape", "tiny"],
               IRSPACK_AMD_BENCH_ONE_DEVICE="1", IRSPACK_AMD_BENCH_BACKEND="gloo")
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "HIP device" in out.stderr or "ChildFailedError" in out.stderr
