"""bench.py's launch contract on a machine without GPUs: it never degrades to fewer ranks or to a CPU path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
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


def _fake_kfd(tmp_path, simd_counts):
    for i, n in enumerate(simd_counts):
        d = tmp_path / "kfd" / "topology" / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if n else 64}\nsimd_count {n}\nmem_banks_count 1\n")
    return str(tmp_path)


def test_the_spawning_parent_counts_gpus_from_the_kfd_topology(tmp_path, monkeypatch):
    """`python bench.py --gpus N` must not load a GPU runtime in the process that starts torchrun: GPUs = KFD topology nodes
    with SIMDs (CPU nodes have simd_count 0), narrowed by the *_VISIBLE_DEVICES variables."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("ZDR_KFD_ROOT", _fake_kfd(tmp_path, [0, 0, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024]))   # 2 CPU sockets + 8 GPUs
    assert bench.visible_gpu_count() == 8
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4")
    assert bench.visible_gpu_count() == 1
    monkeypatch.setenv("ZDR_KFD_ROOT", str(tmp_path / "nowhere"))
    assert bench.visible_gpu_count() == 0
    assert "torch" not in [m for m in sys.modules if m == "torch"] or True      # (this process has torch loaded by other tests; the parent itself never imports it:)
    src = open(os.path.join(ROOT, "bench.py")).read()
    parent = src[src.index("def spawn_ranks"):src.index("def algorithmic_bytes")]
    assert "import torch" not in parent and "torch.cuda" not in parent


def test_eight_rank_launch_is_refused_without_eight_gpus(tmp_path):
    r = _run(["--gpus", "8"], {"ZDR_KFD_ROOT": _fake_kfd(tmp_path, [0, 1024, 1024])})
    assert r.returncode != 0 and "needs 8 GPUs, this machine has 2" in r.stderr
