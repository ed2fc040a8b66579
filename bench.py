#!/usr/bin/env python3
"""bench.py — Msamples/s of forward + PRB backward (BASELINE.json metric).

One step = scene.render(material, res, spp, seed=random) followed by I.sum().backward(): the loop of
/root/reference/benchmark.py:36-39.  `value` = camera samples processed per second over both passes
(2 * W * H * spp per step), inputs resident in HBM.

  --gpus 1 (default)  BASELINE configs[2]: cbox, path integrator, 512x512, spp 256, textures cboxd/cboxr — the headline
                      line — and, in the same JSON line under "configs", the other configurations that fit one GPU, each
                      timed the same way on its own scene: c2 (direct 512^2 spp 64), c4_one_gpu (path 1024^2 spp 1024, the
                      8-GPU workload on one GPU) and c5 (1,004,672-triangle tessellated cbox, path 1024^2 spp 256)
  --gpus N > 1        BASELINE configs[3]: cbox path 1024x1024 spp 1024, ONE render pixel-tiled over the N ranks
                      (8x8 tiles dealt round-robin, one launch per rank and pass) and one RCCL all_reduce of the image
                      and of the gradient per step: fixed total work, "scaling": "strong"
  --config c5         BASELINE configs[4] as the headline, on 1 or N GPUs (tiled like c4)

Launch: `python bench.py --gpus N ...` starts its N ranks itself (torch.distributed.run as a child process; the parent
never loads torch or touches a GPU); under torchrun (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import gc
import glob
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NAMES = {"c2": "cbox direct", "c3": "cbox path", "c4": "cbox path", "c5": "1,004,672-triangle tessellated cbox, path"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="auto", choices=["auto", "c2", "c3", "c4", "c5"])
    ap.add_argument("--res", type=int, default=0, help="override the configuration's resolution")
    ap.add_argument("--spp", type=int, default=0, help="override the configuration's samples per pixel")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "rows", "samples", "seeds"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="headline line only (skip the c2 / c4_one_gpu / c5 legs of the default run)")
    ap.add_argument("--cpu-spp", type=int, default=16)
    return ap.parse_args()


def visible_gpu_count():
    """GPUs this process tree may use, WITHOUT loading a GPU runtime: the KFD topology lists every agent and a GPU is a
    node with SIMDs; HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES narrow it.  None when the topology cannot be read (the ranks
    then fail on their own if a device is missing)."""
    n = 0
    root = os.environ.get("ZDR_KFD_ROOT", "/sys/class/kfd")       # (the override exists for tests/test_bench_cli.py)
    if not os.path.isdir(root):
        return 0                                                # no amdgpu compute driver at all
    props = glob.glob(os.path.join(root, "kfd/topology/nodes/*/properties"))
    if not props:
        return None
    for path in props:
        try:
            for line in open(path):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except OSError:
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args) -> int:
    """The parent of an N-rank run: loads neither torch nor HIP, starts torchrun as a CHILD process and hands its exit
    code on.  A process that has touched the GPU is never re-executed or restarted."""
    share = os.environ.get("ZDR_SHARE_DEVICE") == "1"          # rehearsal: every rank on cuda:0 over gloo
    have = visible_gpu_count()
    if have is not None and have < (1 if share else args.gpus):
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


def pmc_record(kernel, workload_key):
    """Counter figures per launch of `kernel` from the committed PMC passes (separate rocprofv3 --pmc runs of the same
    workload, tools/pmc_passes.sh -> profiles/pmc_traffic.json).  The file carries the hash of the kernel sources it was
    measured on: returns (record or None, stale) — stale when the sources in this tree hash differently."""
    from zdr_amd import build as hip_build
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None, False
    stale = d.get("csrc_sha256") != hip_build.source_hash()
    if workload_key != d.get("workload_key", "c3"):          # the other workloads of the file sit under their own key ("c5")
        d = d.get(workload_key) or {}
    if kernel not in d:
        return None, False
    return d[kernel], stale


def cpu_baseline(scene, mat_np, W, spp_sample):
    """The oracle (a scalar C port, OpenMP over pixels) on the host cores, bounded sample of the same workload."""
    import numpy as np
    import oracle
    S = oracle.OracleScene.from_arrays(scene._arrays)
    threads = len(os.sched_getaffinity(0))                  # the CPUs this process may run on = the OpenMP threads asked for
    cam = scene.camera
    def params(seed):
        return oracle.make_params(scene.integrator, W, W, spp_sample, seed, (cam.fov, tuple(cam.origin), tuple(cam.target), tuple(cam.up)), mat_np.shape[:2],
                                  use_tent=scene.use_tent_filter, max_depth=scene.max_depth, rr_depth=scene.rr_depth, nthreads=threads)
    t0 = time.time()
    S.render_forward(params(0), mat_np)
    t1 = time.time()
    S.render_backward(params(1), np.ones((W, W, 4), np.float32), mat_np)
    t2 = time.time()
    n = W * W * spp_sample
    return {"value": float(f"{2 * n / (t2 - t0) / 1e6:.4g}"), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{scene.integrator} {W}x{W} spp={spp_sample} fwd+bwd ({2 * n} camera samples) of the same scene, oracle/zdr_oracle.c with OpenMP"
                      + (" (rays searched through the oracle's own binary BVH)" if scene._arrays.tris.shape[0] > 256 else ""),
            "samples": 2 * n, "seconds": round(t2 - t0, 3),
            "fwd_msamples_s": float(f"{n / (t1 - t0) / 1e6:.4g}"), "bwd_msamples_s": float(f"{n / (t2 - t1) / 1e6:.4g}")}


def fd_summary():
    """Gradient accuracy against finite differences, measured by tools/fd_validate.py / tools/fd_directional.py.  Every file says
    which kernel sources it was measured on (csrc_sha256 = zdr_amd.build.source_hash()): a figure from other sources than this
    tree's — or from a file of an earlier round that carries no hash — is printed with "fd_stale": true, never silently."""
    from zdr_amd import build as hip_build
    here = hip_build.source_hash()
    out = {}
    def load(name, key):
        for rnd in ("r4", "r3", "r2", "r1"):
            path = os.path.join(ROOT, "profiles", f"{rnd}_{name}.json")
            try:
                d = json.load(open(path))
                return d, d[key], f"profiles/{rnd}_{name}.json", d.get("csrc_sha256") != here
            except Exception:
                continue
        return None
    for key, name in (("fd_validate_procedure_diffuse_texel", "fd_validate_diffuse_texel"), ("fd_validate_procedure_roughness_texel", "fd_validate_roughness_texel")):
        r = load(name, "tail")
        if r:
            _, t, src, stale = r
            out[key] = {"rel_err": t["rel_err"], "one_sigma": t["one_sigma"], "spp": t["spp"], "seeds": t["seeds"], "source": src, "fd_stale": stale}
    r = load("fd_directional", "result")
    if r:
        _, d, src, stale = r
        out["whole_image_directional"] = {k: {"rel_err": v["rel_err"], "one_sigma": v["one_sigma"]} for k, v in d.items()}
        out["whole_image_directional"].update({"source": src, "fd_stale": stale})
    if out:
        out["fd_stale"] = any(v.get("fd_stale", False) for v in out.values() if isinstance(v, dict))
    return out or None


def shard_union_check(scene, renderer, material, rank, world, W=256, spp=16, seed=12345):
    """An N-rank run checks itself BEFORE its timed loop: one small frame rendered sharded over the ranks (tiles dealt round-robin,
    one all_reduce of the image and one of the gradient — the run's own exchange) and, on rank 0, once more unsharded on its GPU
    alone.  The two must be the same image (bit for bit in `tiles` mode when both cut the sample range into the same chunks,
    which at this size they do) and the same gradient up to the order the float atomics land in: the line then says whether
    RCCL summed the right tensors, not only how fast.  Collective calls: every rank must take part."""
    import torch
    res = (W, W)
    m = material.detach().clone().requires_grad_()
    cot = torch.rand((W, W, 4), device=material.device, generator=torch.Generator(device=material.device).manual_seed(7)) + 0.5
    img = renderer.render(m, res=res, spp=spp, seed=seed)
    (img * cot).sum().backward()
    torch.cuda.synchronize()
    if rank != 0:
        return None
    m1 = material.detach().clone().requires_grad_()
    ref = scene.render(m1, res=res, spp=spp, seed=seed)
    (ref * cot).sum().backward()
    torch.cuda.synchronize()
    g, gref = m.grad.double(), m1.grad.double()
    return {"workload": f"path {W}x{W} spp={spp}, {world} ranks ({renderer.mode}) vs rank 0 alone",
            "image_max_abs": float((img.detach() - ref.detach()).abs().max()), "image_bit_identical": bool(torch.equal(img.detach(), ref.detach())),
            "image_mean": float(ref.detach()[..., :3].mean()),
            "grad_rel_l1": float((g - gref).abs().sum() / gref.abs().sum().clamp_min(1e-30)),
            "grad_nnz": [int((g != 0).sum()), int((gref != 0).sum())]}


class Leg:
    """One timed workload: `steps` steps of render + backward on one scene, HIP events around the native calls."""

    def __init__(self, scene, material, W, spp, renderer=None, world=1):
        import torch
        self.torch, self.scene, self.material, self.W, self.spp, self.renderer, self.world = torch, scene, material, W, spp, renderer, world
        self.ev = {"fwd": [], "bwd": []}
        self.timing = False
        self.seeds = random.Random(0)
        scene.render_forward = self._timed("fwd", scene.render_forward)
        scene.render_backward = self._timed("bwd", scene.render_backward)

    def _timed(self, name, fn):
        # HIP events directly around the native calls, on the stream the kernels are enqueued on (torch's current one)
        def wrapper(*a, **k):
            if not self.timing:
                return fn(*a, **k)
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a, **k); e1.record()
            self.ev[name].append((e0, e1))
            return r
        return wrapper

    def step(self):
        seed = self.seeds.randint(0, 2147483646)                # benchmark.py:38, capped so seed + 1 fits
        self.material.grad = None
        res = (self.W, self.W)
        img = self.renderer.render(self.material, res=res, spp=self.spp, seed=seed) if self.world > 1 else self.scene.render(self.material, res=res, spp=self.spp, seed=seed)
        img.sum().backward()

    def run(self, steps, warmup):
        import torch.distributed as dist
        torch = self.torch
        for _ in range(warmup):
            self.step()
        if self.world > 1: dist.barrier()
        torch.cuda.synchronize()
        self.timing = True
        if self.renderer is not None: self.renderer.reduce_events = []
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        if self.world > 1: dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.timing = False
        self.scene.check()                                      # a tripped device watchdog would make the numbers meaningless
        self.steps = steps
        self.dt = dt
        self.launches = {k: len(v) // max(steps, 1) for k, v in self.ev.items()}
        self.fwd_ms = sum(a.elapsed_time(b) for a, b in self.ev["fwd"]) / max(steps, 1)      # this rank's launches of one step
        self.bwd_ms = sum(a.elapsed_time(b) for a, b in self.ev["bwd"]) / max(steps, 1)
        return dt

    def stats(self, shard):
        """Path statistics of this rank's share of one pass (the counting kernel variant, zdr_render_stats)."""
        st = {}
        for rect in shard.rects:
            for k, v in self.scene.render_stats(self.material.detach(), (self.W, self.W), self.spp, seed=0, rect=rect, samples=shard.samples, tile_shard=shard.tile_shard).items():
                st[k] = st.get(k, 0) + v
        return st

    def roofline(self, stats, integrator, accel):
        a_fwd, (Hb, Vb, Eb) = algorithmic_bytes(stats, self.spp, False)
        a_bwd, _ = algorithmic_bytes(stats, self.spp, True)
        n_rank = stats["samples"]                               # camera samples this rank processes per pass
        nb = max(self.launches["bwd"], 1)
        bwd_kernel_ms = self.bwd_ms / nb                        # dominant kernel = the backward kernel, average launch
        achieved = a_bwd * (n_rank / nb) / (bwd_kernel_ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": f"k_path_bwd<cmj, {accel}>" if integrator == "path" else f"k_simple<{integrator}, backward>",
             "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
             "launch_ms": round(bwd_kernel_ms, 3), "algorithmic_bytes_per_launch": round(a_bwd * n_rank / nb),
             "bytes_per_sample": {"fwd": round(a_fwd, 1), "bwd": round(a_bwd, 1)},
             "fwd_achieved": round(a_fwd * n_rank / (self.fwd_ms * 1e-3) / 1e9, 2), "fwd_frac": round(a_fwd * n_rank / (self.fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        ps = {"closest_hits_per_sample": round(Hb, 4), "shaded_vertices_per_sample": round(Vb, 4), "emitter_hits_bsdf_per_sample": round(Eb, 5),
              "closest_rays_per_sample": round(stats["closest_rays"] / stats["samples"], 4)}
        return r, ps, n_rank


def pmc_fields(roof, kernel, cfg, stats, active):
    """What actually bounds the kernel, from the committed counter passes: `bound` stays the nominal HBM roofline the
    contract asks for; VALU issue, lane utilisation and atomic requests per shaded vertex say what the kernel is limited by."""
    rec, stale = pmc_record(kernel, cfg) if active else (None, False)
    roof["traffic"] = None if (rec is None or stale) else rec.get("traffic_bytes")
    roof["traffic_unit"] = "bytes per launch (rocprofv3 --pmc, profiles/pmc_traffic.json)"
    if rec is not None:
        roof["traffic_stale"] = bool(stale)
        if not stale:
            cyc = rec.get("gpu_cycles_per_xcd")
            if rec.get("SQ_ACTIVE_INST_VALU") and cyc:
                roof["valu_issue_frac"] = round(rec["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc), 4)    # busy cycles summed over the SIMDs: x4 per wave64 instruction, 1024 SIMDs
            if rec.get("valu_lane_utilisation"):
                roof["lane_utilisation"] = round(rec["valu_lane_utilisation"], 4)
            if rec.get("SQ_WAVES"):                            # persistent kernel: the launch IS the resident set, one wave per workgroup
                roof["waves_per_simd"] = round(rec["SQ_WAVES"] / 1024.0, 2)
                roof["wavefront_occupancy"] = round(rec["SQ_WAVES"] / 1024.0 / 8.0, 3)   # of the 8 wave slots per SIMD of gfx950
            if rec.get("atomic_requests") and stats.get("shaded_vertices"):
                roof["atomic_requests_per_vertex"] = round(rec["atomic_requests"] / stats["shaded_vertices"], 4)
            roof["measured_bound"] = rec.get("measured_bound", "valu_issue")
    return roof


def extra_config(name, cfg_key, dev, mat_np, steps, warmup):
    """One more BASELINE configuration on this GPU, timed like the headline: its own scene, its own path statistics."""
    import torch
    from zdr_amd import distributed as zd
    from zdr_amd import scenes
    t_build = time.perf_counter()
    scene, W, spp = scenes.config_scene(cfg_key)
    t_build = time.perf_counter() - t_build
    material = torch.from_numpy(mat_np).to(dev).requires_grad_()
    leg = Leg(scene, material, W, spp)
    dt = leg.run(steps, warmup)
    stats = leg.stats(zd.plan("tiles", 0, 1, (W, W), spp, 0))
    integrator = scenes.CONFIGS[cfg_key][0]
    accel = scene.info()["accel"]
    roof, ps, n = leg.roofline(stats, integrator, accel)
    out = {"workload": f"{NAMES[cfg_key]} integrator {W}x{W} spp={spp} (BASELINE configs[{scenes.CONFIGS[cfg_key][4]}])" + (" on ONE GPU" if cfg_key == "c4" else ""),
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "msamples_s": round(2 * n * steps / dt / 1e6, 2),
           "fwd_ms": round(leg.fwd_ms, 3), "bwd_ms": round(leg.bwd_ms, 3),
           "fwd_msamples_s": round(n / (leg.fwd_ms * 1e-3) / 1e6, 2), "bwd_msamples_s": round(n / (leg.bwd_ms * 1e-3) / 1e6, 2),
           "accel": accel, "scene_build_s": round(t_build, 2), "path_stats": ps,
           "bytes_per_sample": roof["bytes_per_sample"],
           "roofline": {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_ms", "fwd_achieved", "fwd_frac")}}
    if accel == "bvh":
        out["triangles"] = scene.info()["ntris"]
    if integrator == "path":   # counter traffic of this leg's dominant kernel (tools/refresh_profiles.sh), when it was measured on these sources
        rec, stale = pmc_record("k_path_bwd", cfg_key)
        if rec is not None:
            ok = not stale and rec.get("traffic_bytes") is not None
            out["roofline"]["traffic"] = rec["traffic_bytes"] if ok else None
            out["roofline"]["traffic_stale"] = bool(stale)
            if ok:
                alg = roof["bytes_per_sample"]["bwd"] * n
                out["roofline"]["algorithmic_bytes_per_launch"] = round(alg)
                out["roofline"]["traffic_over_algorithmic"] = round(rec["traffic_bytes"] / alg, 3)
                for k in ("l2_hit_rate", "valu_issue_frac", "valu_lane_utilisation", "wait_any_frac"):
                    if rec.get(k) is not None: out["roofline"][k] = round(rec[k], 4)
    scene.render_forward = scene.render_backward = None     # break the cycle scene -> timing wrapper -> leg -> scene
    del leg, scene, material
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if world_env is not None and int(world_env) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with matching values")

    import torch
    import torch.distributed as dist
    from zdr_amd import distributed as zd
    from zdr_amd import scenes

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an AMD GPU: the renderer has no CPU back end")
    if world_env is not None and os.environ.get("ZDR_SHARE_DEVICE") != "1" and torch.cuda.device_count() < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this rank sees {torch.cuda.device_count()}")
    rank, world, local = zd.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = args.config if args.config != "auto" else ("c3" if world == 1 else "c4")
    integrator, W, spp, scene_kind, cfg_index = scenes.CONFIGS[cfg]
    W, spp = args.res or W, args.spp or spp
    scene = scenes.make_scene(integrator, arrays=scenes.tess1m_arrays()) if scene_kind == "tess1m" else scenes.make_scene(integrator)
    mat_np = scenes.cbox_material_np()
    material = torch.from_numpy(mat_np).to(dev).requires_grad_()
    renderer = zd.attach(scene, mode=args.shard)
    union = shard_union_check(scene, renderer, material, rank, world) if (world > 1 and integrator == "path" and args.shard != "seeds") else None
    leg = Leg(scene, material, W, spp, renderer, world)
    dt_rank = leg.run(args.steps, args.warmup)
    dt = dt_rank
    per_rank = None
    if world > 1:
        t = torch.tensor([dt_rank], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # outside the timed region: what every rank measured, so that a shortfall can be attributed to imbalance
        # (kernel time differs between ranks) or to communication (all-reduce time)
        red = renderer.reduce_ms()
        mine = torch.tensor([dt_rank / args.steps * 1e3, leg.fwd_ms, leg.bwd_ms, red["image"] / args.steps, red["gradient"] / args.steps], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu().numpy()

    n_per_pass = W * W * spp                                # camera samples of one whole pass (all ranks together)
    weak = world > 1 and args.shard == "seeds"

    if rank == 0:
        stats = leg.stats(zd.plan(args.shard, 0, world, (W, W), spp, 0))
        accel = scene.info()["accel"]
        roof, ps, n_rank = leg.roofline(stats, integrator, accel)
        roof = pmc_fields(roof, "k_path_bwd", cfg, stats, (world, integrator) == (1, "path") and not (args.res or args.spp))
        out = {
            "metric": "Msamples/s fwd+PRB-bwd" + (", cbox 512x512 spp=256" if (cfg, W, spp) == ("c3", 512, 256) else f", {NAMES[cfg]} {W}x{W} spp={spp}"),
            "value": round(2 * n_per_pass * args.steps * (world if weak else 1) / dt / 1e6, 2),
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (random seeds; cbox geometry + cboxd/cboxr textures" + ("; instance 0 tessellated and displaced, seed 0)" if cfg == "c5" else ")"),
            "config": {"workload": f"{NAMES[cfg]} integrator {W}x{W} spp={spp}, forward + PRB backward w.r.t. the 1024x1024x4 material (BASELINE configs[{cfg_index}])",
                       "sampler": "cmj", "shard": args.shard if world > 1 else "none", "accel": accel,
                       "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else "none",
                       "launches_per_step_per_rank": leg.launches},
            "fwd_msamples_s": round(n_rank / (leg.fwd_ms * 1e-3) / 1e6, 2), "bwd_msamples_s": round(n_rank / (leg.bwd_ms * 1e-3) / 1e6, 2),
            "fwd_ms": round(leg.fwd_ms, 3), "bwd_ms": round(leg.bwd_ms, 3),
            "per_rank_note": "fwd/bwd figures are rank 0's kernels on its share of the pass (HIP events around the native calls)" if world > 1 else "HIP events around the native calls",
            "path_stats": ps, "roofline": roof,
        }
        if per_rank is not None:
            col = lambda i: {"min": round(float(per_rank[:, i].min()), 3), "max": round(float(per_rank[:, i].max()), 3)}
            out["per_rank_ms"] = {"step": col(0), "fwd_kernels": col(1), "bwd_kernels": col(2),
                                  "note": "per step; step = wall clock of the timed loop / steps on each rank, kernels = HIP events around the native calls"}
            out["allreduce_ms"] = {"image": col(3), "gradient": col(4), "bytes": {"image": W * W * 16, "gradient": int(material.numel()) * 4},
                                   "note": "per step, HIP events on the compute stream around torch.distributed.all_reduce (includes waiting for the slowest rank)"}
        if union is not None:
            out["shard_union_check"] = union
        fd = fd_summary()
        if fd: out["grad_rel_err_vs_fd"] = fd
        if world == 1 and cfg == "c3" and not (args.res or args.spp) and not args.no_extra_configs:
            # the other single-GPU BASELINE configurations, each on its own scene with its own statistics (c5: BVH built once, 2 steps)
            scene.render_forward = scene.render_backward = None
            del leg, renderer, scene
            gc.collect()
            torch.cuda.empty_cache()
            out["configs"] = {}
            for name, key, steps, warm in (("c2", "c2", 10, 2), ("c4_one_gpu", "c4", 2, 1), ("c5", "c5", 2, 1)):
                try:                                            # a leg that fails is reported as such; the headline above stands on its own
                    out["configs"][name] = extra_config(name, key, dev, mat_np, steps, warm)
                except Exception as e:                          # noqa: BLE001
                    out["configs"][name] = {"error": f"{type(e).__name__}: {e}"}
            scene = scenes.make_scene(integrator)
        if world == 1 and not args.no_cpu_baseline:
            # only now is the CPU oracle loaded — after every timed region
            try:
                out["cpu_baseline"] = cpu_baseline(scene, mat_np, min(W, 512), args.cpu_spp)
            except Exception as e:                              # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
