#!/usr/bin/env python3
"""bench.py — Msamples/s of forward + PRB backward on the Cornell box (BASELINE.json metric).

One step = scene.render(material, res=(512,512), spp=256, seed=random) followed by
I.sum().backward(): the loop of /root/reference/benchmark.py:36-39 on configs[2] of BASELINE.json
(cbox, path integrator, 512x512, spp 256, textures cboxd/cboxr), inputs resident in HBM.
`value` = camera samples processed per second, counting the forward and the backward pass
(2 * W * H * spp per step per GPU).  With N > 1 every rank renders its own sample set of the same
image (shard mode "seeds": N*spp samples per pixel in total) and the image and gradient tensors
are summed with one all_reduce each over RCCL — fixed work per GPU, i.e. weak scaling.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(stats, spp, backward):
    """A_fwd = 16/spp + 120 H + 192 V + 112 E;  A_bwd = A_fwd + 16/spp + 64 V  (per sample)."""
    n = stats["samples"]
    H, V, E = stats["closest_hits"] / n, stats["shaded_vertices"] / n, stats["emitter_hits_bsdf"] / n
    a = 16.0 / spp + 120.0 * H + 192.0 * V + 112.0 * E
    if backward:
        a += 16.0 / spp + 64.0 * V
    return a, (H, V, E)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC pass (separate rocprofv3 --pmc runs of
    the same workload, tools/pmc_passes.sh); None when the workload differs from the profiled one."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[kernel]["traffic_bytes"]
    except Exception:
        return None


def cpu_baseline(scene, mat_np, W, spp_sample):
    """The oracle (a scalar C port, OpenMP over pixels) on the host cores, bounded sample of the same workload."""
    import oracle
    from gpu_util import oracle_params
    S = oracle.OracleScene.from_arrays(scene._arrays)
    p = oracle_params(scene, W, W, spp_sample, 0, mat_np.shape[:2])
    t0 = time.time()
    S.render_forward(p, mat_np)
    t1 = time.time()
    S.render_backward(oracle_params(scene, W, W, spp_sample, 1, mat_np.shape[:2]), np.ones((W, W, 4), np.float32), mat_np)
    t2 = time.time()
    n = W * W * spp_sample
    return {"value": round(2 * n / (t2 - t0) / 1e6, 3), "unit": "Msamples/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"cbox path {W}x{W} spp={spp_sample} fwd+bwd ({2 * n / 1e6:.1f} Msamples), oracle/zdr_oracle.c with OpenMP",
            "fwd_msamples_s": round(n / (t1 - t0) / 1e6, 3), "bwd_msamples_s": round(n / (t2 - t1) / 1e6, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--integrator", default="path")
    ap.add_argument("--shard", default="seeds", choices=["seeds", "rows", "samples"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=16)
    args = ap.parse_args()

    import torch.distributed as dist
    from conftest import cbox_material_np
    from gpu_util import make_scene
    from zdr_amd import distributed as zd

    rank, world, local = zd.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    scene = make_scene(args.integrator)
    mat_np = cbox_material_np()
    material = torch.from_numpy(mat_np).to(dev).requires_grad_()
    W, spp = args.res, args.spp
    renderer = zd.attach(scene, mode=args.shard)
    seeds = random.Random(0)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(k=None):
        seed = seeds.randint(0, 2147483646)                 # benchmark.py:38, capped so seed + 1 fits
        material.grad = None
        if k is not None: ev[k][0].record()
        if world > 1:
            img = renderer.render(material, res=(W, W), spp=spp, seed=seed)
        else:
            img = scene.render(material, res=(W, W), spp=spp, seed=seed)
        if k is not None: ev[k][1].record()
        img.sum().backward()
        if k is not None: ev[k][2].record()

    for _ in range(args.warmup):
        step()
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_per_pass = W * W * spp
    weak = world == 1 or args.shard == "seeds"
    fwd_ms = float(np.mean([ev[k][0].elapsed_time(ev[k][1]) for k in range(args.steps)]))
    bwd_ms = float(np.mean([ev[k][1].elapsed_time(ev[k][2]) for k in range(args.steps)]))

    if rank == 0:
        stats = scene.render_stats(material.detach(), (W, W), spp, seed=0)
        a_fwd, (Hb, Vb, Eb) = algorithmic_bytes(stats, spp, False)
        a_bwd, _ = algorithmic_bytes(stats, spp, True)
        # dominant kernel = the PRB backward kernel (k_path<..., BWD>): one launch per step
        # its duration: HIP events around the backward launch on torch's current stream (includes the
        # zero-fill of the gradient tensor and the tiny sum() backward, both < 1 % of the kernel)
        achieved = a_bwd * n_per_pass / (bwd_ms * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s fwd+PRB-bwd, cbox 512x512 spp=256",
            # "seeds": every rank renders its own W*H*spp sample set (weak); "rows"/"samples" split ONE
            # render of W*H*spp samples over the ranks (strong)
            "value": round(2 * n_per_pass * args.steps * (world if weak else 1) / dt / 1e6, 2),
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (random seeds; cbox geometry + cboxd/cboxr textures)",
            "config": {"workload": f"cbox {args.integrator} integrator {W}x{W} spp={spp}, forward + PRB backward w.r.t. the 1024x1024x4 material (BASELINE configs[2])",
                       "sampler": "cmj", "shard": args.shard if world > 1 else "none", "accel": scene.info()["accel"]},
            "fwd_msamples_s": round(n_per_pass / (fwd_ms * 1e-3) / 1e6, 2),
            "bwd_msamples_s": round(n_per_pass / (bwd_ms * 1e-3) / 1e6, 2),
            "fwd_ms": round(fwd_ms, 3), "bwd_ms": round(bwd_ms, 3),
            "path_stats": {"closest_hits_per_sample": round(Hb, 4), "shaded_vertices_per_sample": round(Vb, 4), "emitter_hits_bsdf_per_sample": round(Eb, 5),
                           "closest_rays_per_sample": round(stats["closest_rays"] / stats["samples"], 4)},
            "roofline": {"bound": "hbm", "kernel": "k_path<cmj, brute, backward>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic("k_path_bwd") if (W, spp, args.integrator) == (512, 256, "path") else None, "traffic_unit": "bytes per launch (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json)",
                         "algorithmic_bytes_per_launch": round(a_bwd * n_per_pass),
                         "bytes_per_sample": {"fwd": round(a_fwd, 1), "bwd": round(a_bwd, 1)},
                         "fwd_achieved": round(a_fwd * n_per_pass / (fwd_ms * 1e-3) / 1e9, 2)},
        }
        try:   # gradient accuracy against finite differences: measured by tools/fd_validate.py / tools/fd_directional.py
            fdv = json.load(open(os.path.join(ROOT, "profiles", "r1_fd_validate_diffuse_texel.json")))["tail"]
            fdd = json.load(open(os.path.join(ROOT, "profiles", "r1_fd_directional.json")))["result"]
            out["grad_rel_err_vs_fd"] = {
                "fd_validate_procedure_single_pixel_texel": {"rel_err": fdv["rel_err"], "one_sigma": fdv["one_sigma"], "spp": fdv["spp"], "seeds": fdv["seeds"]},
                "whole_image_directional": {k: {"rel_err": v["rel_err"], "one_sigma": v["one_sigma"]} for k, v in fdd.items()},
                "source": "profiles/r1_fd_validate_diffuse_texel.json, profiles/r1_fd_directional.json (FD eps 0.01, same-seed renders)"}
        except Exception:
            pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, mat_np, W, args.cpu_spp)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
