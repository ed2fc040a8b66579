"""bench.py's launch contract on a machine without GPUs: it never degrades to fewer ranks or to a CPU path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e, timeout=120)


def test_more_gpus_than_the_machine_has_is_an_error_not_a_smaller_run():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr and "{" not in r.stdout
    r = _run(["--gpus", "2"], {"ZDR_DIST_BACKEND": "gloo"})       # gloo does not conjure GPUs either
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_single_gpu_run_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    r = _run([])
    assert r.returncode != 0 and "no CPU back end" in r.stderr and "{" not in r.stdout
