#!/usr/bin/env python3
"""bench.py — Msamples/s of forward + PRB backward (BASELINE.json metric).

One step = scene.render(material, res, spp, seed=random) followed by I.sum().backward(): the loop of
/root/reference/benchmark.py:36-39.  `value` = camera samples processed per second over both passes
(2 * W * H * spp per step), inputs resident in HBM.

  --gpus 1 (default)  BASELINE configs[2]: cbox, path integrator, 512x512, spp 256, textures cboxd/cboxr
  --gpus N > 1        BASELINE configs[3]: cbox path 1024x1024 spp 1024, ONE render pixel-tiled over the N ranks
                      (8x8 tiles dealt round-robin, one launch per rank and pass) and one RCCL all_reduce of the image
                      and of the gradient per step: fixed total work, "scaling": "strong"
  --config c5         BASELINE configs[4]: 1,004,672-triangle tessellated cbox (BVH), path + PRB 1024x1024 spp 256,
                      on 1 or N GPUs (tiled like c4)

Launch: `python bench.py --gpus N ...` starts its N ranks itself (torch.distributed.run as a child process, before
this process has touched a GPU); under torchrun (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {   # name: (integrator, res, spp, scene, BASELINE.json configs index)
    "c2": ("direct", 512, 64, "cbox", 1),
    "c3": ("path", 512, 256, "cbox", 2),
    "c4": ("path", 1024, 1024, "cbox", 3),
    "c5": ("path", 1024, 256, "tess1m", 4),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="auto", choices=["auto", *CONFIGS])
    ap.add_argument("--res", type=int, default=0, help="override the configuration's resolution")
    ap.add_argument("--spp", type=int, default=0, help="override the configuration's samples per pixel")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "rows", "samples", "seeds"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=16)
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """The parent of an N-rank run: never touches a GPU, starts torchrun as a CHILD process and passes its output on."""
    import torch
    share = os.environ.get("ZDR_SHARE_DEVICE") == "1"          # rehearsal: every rank on cuda:0 over gloo
    have = torch.cuda.device_count()                             # counting devices does not initialise HIP
    if have < (1 if share else args.gpus):
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this machine has {have}", file=sys.stderr)
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd).returncode


def algorithmic_bytes(stats, spp, backward):
    """A_fwd = 16/spp + 120 H + 192 V + 112 E;  A_bwd = A_fwd + 16/spp + 64 V  (per sample)."""
    n = stats["samples"]
    H, V, E = stats["closest_hits"] / n, stats["shaded_vertices"] / n, stats["emitter_hits_bsdf"] / n
    a = 16.0 / spp + 120.0 * H + 192.0 * V + 112.0 * E
    if backward:
        a += 16.0 / spp + 64.0 * V
    return a, (H, V, E)


def pmc_traffic(kernel, workload_key):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (separate rocprofv3 --pmc runs of the same
    workload, tools/pmc_passes.sh -> profiles/pmc_traffic.json); None when the workload was not profiled."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return d[kernel]["traffic_bytes"] if d.get("workload_key", "c3") == workload_key else None
    except Exception:
        return None


def cpu_baseline(scene, mat_np, W, spp_sample):
    """The oracle (a scalar C port, OpenMP over pixels) on the host cores, bounded sample of the same workload."""
    import numpy as np
    import oracle
    from gpu_util import oracle_params
    S = oracle.OracleScene.from_arrays(scene._arrays)
    threads = len(os.sched_getaffinity(0))                  # the CPUs this process may run on = the OpenMP threads asked for
    p = oracle_params(scene, W, W, spp_sample, 0, mat_np.shape[:2], nthreads=threads)
    t0 = time.time()
    S.render_forward(p, mat_np)
    t1 = time.time()
    S.render_backward(oracle_params(scene, W, W, spp_sample, 1, mat_np.shape[:2], nthreads=threads), np.ones((W, W, 4), np.float32), mat_np)
    t2 = time.time()
    n = W * W * spp_sample
    return {"value": round(2 * n / (t2 - t0) / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{scene.integrator} {W}x{W} spp={spp_sample} fwd+bwd ({2 * n / 1e6:.1f} Msamples) of the same scene, oracle/zdr_oracle.c with OpenMP",
            "fwd_msamples_s": round(n / (t1 - t0) / 1e6, 3), "bwd_msamples_s": round(n / (t2 - t1) / 1e6, 3)}


def fd_summary():
    """Gradient accuracy against finite differences, measured by tools/fd_validate.py / tools/fd_directional.py."""
    out = {}
    for key, name in (("fd_validate_procedure_diffuse_texel", "fd_validate_diffuse_texel"), ("fd_validate_procedure_roughness_texel", "fd_validate_roughness_texel")):
        for rnd in ("r2", "r1"):
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_{name}.json")))["tail"]
                out[key] = {"rel_err": t["rel_err"], "one_sigma": t["one_sigma"], "spp": t["spp"], "seeds": t["seeds"], "source": f"profiles/{rnd}_{name}.json"}
                break
            except Exception:
                continue
    for rnd in ("r2", "r1"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_fd_directional.json")))["result"]
            out["whole_image_directional"] = {k: {"rel_err": v["rel_err"], "one_sigma": v["one_sigma"]} for k, v in d.items()}
            out["whole_image_directional"]["source"] = f"profiles/{rnd}_fd_directional.json"
            break
        except Exception:
            continue
    return out or None


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if world_env is not None and int(world_env) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with matching values")

    import numpy as np
    import torch
    import torch.distributed as dist
    from conftest import cbox_material_np, cbox_models
    from gpu_util import make_scene
    from zdr_amd import distributed as zd

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an AMD GPU: the renderer has no CPU back end")
    rank, world, local = zd.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = args.config if args.config != "auto" else ("c3" if world == 1 else "c4")
    integrator, W, spp, scene_kind, cfg_index = CONFIGS[cfg]
    W, spp = args.res or W, args.spp or spp
    if scene_kind == "tess1m":
        from zdr_amd import procedural
        scene = make_scene(integrator, arrays=procedural.tessellated_cbox(cbox_models(), n=183))
    else:
        scene = make_scene(integrator)
    mat_np = cbox_material_np()
    material = torch.from_numpy(mat_np).to(dev).requires_grad_()
    renderer = zd.attach(scene, mode=args.shard)
    seeds = random.Random(0)

    # HIP events directly around the native calls, on the stream the kernels are enqueued on (torch's current one)
    ev = {"fwd": [], "bwd": []}
    timing = [False]

    def timed(name, fn):
        def wrapper(*a, **k):
            if not timing[0]:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a, **k); e1.record()
            ev[name].append((e0, e1))
            return r
        return wrapper
    scene.render_forward = timed("fwd", scene.render_forward)
    scene.render_backward = timed("bwd", scene.render_backward)

    def step():
        seed = seeds.randint(0, 2147483646)                 # benchmark.py:38, capped so seed + 1 fits
        material.grad = None
        img = renderer.render(material, res=(W, W), spp=spp, seed=seed) if world > 1 else scene.render(material, res=(W, W), spp=spp, seed=seed)
        img.sum().backward()

    for _ in range(args.warmup):
        step()
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    timing[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing[0] = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    scene.check()                                           # a tripped device watchdog would make the numbers meaningless

    n_per_pass = W * W * spp                                # camera samples of one whole pass (all ranks together)
    weak = world > 1 and args.shard == "seeds"
    launches = {k: len(v) // max(args.steps, 1) for k, v in ev.items()}
    fwd_ms = sum(a.elapsed_time(b) for a, b in ev["fwd"]) / max(args.steps, 1)      # this rank's launches of one step
    bwd_ms = sum(a.elapsed_time(b) for a, b in ev["bwd"]) / max(args.steps, 1)

    if rank == 0:
        shard = zd.plan(args.shard, 0, world, (W, W), spp, 0)
        stats = {}
        for rect in shard.rects:                            # path statistics of rank 0's share of one pass
            for k, v in scene.render_stats(material.detach(), (W, W), spp, seed=0, rect=rect, samples=shard.samples, tile_shard=shard.tile_shard).items():
                stats[k] = stats.get(k, 0) + v
        a_fwd, (Hb, Vb, Eb) = algorithmic_bytes(stats, spp, False)
        a_bwd, _ = algorithmic_bytes(stats, spp, True)
        n_rank = stats["samples"]                           # camera samples rank 0 processes per pass
        bwd_kernel_ms = bwd_ms / max(launches["bwd"], 1)    # dominant kernel = the PRB backward kernel, average launch
        achieved = a_bwd * (n_rank / max(launches["bwd"], 1)) / (bwd_kernel_ms * 1e-3) / 1e9
        accel = scene.info()["accel"]
        names = {"c2": "cbox direct", "c3": "cbox path", "c4": "cbox path", "c5": "1,004,672-triangle tessellated cbox, path"}
        out = {
            "metric": "Msamples/s fwd+PRB-bwd" + (", cbox 512x512 spp=256" if (cfg, W, spp) == ("c3", 512, 256) else f", {names[cfg]} {W}x{W} spp={spp}"),
            "value": round(2 * n_per_pass * args.steps * (world if weak else 1) / dt / 1e6, 2),
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (random seeds; cbox geometry + cboxd/cboxr textures" + ("; instance 0 tessellated and displaced, seed 0)" if cfg == "c5" else ")"),
            "config": {"workload": f"{names[cfg]} integrator {W}x{W} spp={spp}, forward + PRB backward w.r.t. the 1024x1024x4 material (BASELINE configs[{cfg_index}])",
                       "sampler": "cmj", "shard": args.shard if world > 1 else "none", "accel": accel,
                       "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else "none",
                       "launches_per_step_per_rank": launches},
            "fwd_msamples_s": round(n_rank / (fwd_ms * 1e-3) / 1e6, 2), "bwd_msamples_s": round(n_rank / (bwd_ms * 1e-3) / 1e6, 2),
            "fwd_ms": round(fwd_ms, 3), "bwd_ms": round(bwd_ms, 3),
            "per_rank_note": "fwd/bwd figures are rank 0's kernels on its share of the pass (HIP events around the native calls)" if world > 1 else "HIP events around the native calls",
            "path_stats": {"closest_hits_per_sample": round(Hb, 4), "shaded_vertices_per_sample": round(Vb, 4), "emitter_hits_bsdf_per_sample": round(Eb, 5),
                           "closest_rays_per_sample": round(stats["closest_rays"] / stats["samples"], 4)},
            "roofline": {"bound": "hbm", "kernel": f"k_path_bwd<cmj, {accel}>" if integrator == "path" else f"k_simple<{integrator}, backward>",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": pmc_traffic("k_path_bwd", cfg) if (world, integrator) == (1, "path") and not (args.res or args.spp) else None,
                         "traffic_unit": "bytes per launch (rocprofv3 --pmc, profiles/pmc_traffic.json)",
                         "launch_ms": round(bwd_kernel_ms, 3), "algorithmic_bytes_per_launch": round(a_bwd * n_rank / max(launches["bwd"], 1)),
                         "bytes_per_sample": {"fwd": round(a_fwd, 1), "bwd": round(a_bwd, 1)},
                         "fwd_achieved": round(a_fwd * n_rank / (fwd_ms * 1e-3) / 1e9, 2)},
        }
        fd = fd_summary()
        if fd: out["grad_rel_err_vs_fd"] = fd
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, mat_np, min(W, 512) if cfg != "c5" else 64, args.cpu_spp if cfg != "c5" else 4)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
