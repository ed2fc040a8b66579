"""-m gpu: the acceleration structures (brute-force slot loop, BVH2 with LDS stack) against the
oracle's brute-force Moeller-Trumbore, as LuisaCompute's Accel.trace_closest / trace_any."""
import numpy as np
import pytest
import torch

import oracle
from conftest import cbox_models
from gpu_util import make_scene, random_rays, terrain_arrays

pytestmark = pytest.mark.gpu


def check_closest(scene, S, rays, what):
    ip, bt = scene.trace_closest(torch.from_numpy(rays).cuda())
    ip, bt = ip.cpu().numpy(), bt.cpu().numpy()
    rip, rbt = S.trace_closest(rays)
    hit_g, hit_r = ip[:, 0] >= 0, rip[:, 0] >= 0
    agree = hit_g == hit_r
    print(f"[trace] {what}: hits {hit_r.mean():.3f}, hit/miss disagreements {(~agree).sum()} of {len(agree)}")
    assert (~agree).mean() < 2e-4
    both = hit_g & hit_r
    same_prim = (ip[both] == rip[both]).all(axis=1)
    assert (~same_prim).mean() < 2e-4          # exact ties on shared edges may pick the neighbour
    ok = both.copy(); ok[both] = same_prim
    # t = (n.p0 - n.o) / (n.d): the numerator cancels, so its error is ABSOLUTE — about one ulp of the
    # scene coordinates (|p| ~ 10 -> 1e-6) — on top of the 1-ulp v_rcp_f32.  Bound the hit-point error
    # accordingly: |dt| <= 1e-5 |t| + 5e-6 for >= 99.98 % of the rays, 50x that for every ray.
    terr = np.abs(bt[ok, 2] - rbt[ok, 2]) / (1e-5 * np.abs(rbt[ok, 2]) + 5e-6)
    berr = np.abs(bt[ok, :2] - rbt[ok, :2]).max(axis=1)
    assert (terr > 1).mean() < 2e-4 and terr.max() < 50, (terr.max(), (terr > 1).mean())
    assert (berr > 1e-4).mean() < 2e-4 and berr.max() < 5e-3, (berr.max(), (berr > 1e-4).mean())


@pytest.mark.parametrize("accel", ["brute", "bvh"])
def test_cbox_closest_and_any(accel, cbox_arrays, cbox_oracle):
    scene = make_scene("path", accel=accel)
    assert scene.info()["accel"] == accel
    rays = random_rays(100000, (-3, 0, -5.5), (2.5, 5.2, 6), seed=1)
    check_closest(scene, cbox_oracle, rays, f"cbox/{accel}")
    # occlusion rays of bounded length
    rays[:, 3] = 1e-4; rays[:, 7] = np.random.default_rng(2).uniform(0.1, 6.0, rays.shape[0]).astype(np.float32)
    occ = scene.trace_any(torch.from_numpy(rays).cuda()).cpu().numpy()
    rocc = cbox_oracle.trace_any(rays)
    assert (occ != rocc).mean() < 2e-4


def test_bvh_equals_brute_on_the_gpu(monkeypatch):
    # Same plane-form triangle records and the same fmaf order in both accels.  With the quad merge switched off
    # (ZDR_NO_QUADS: one primitive per triangle) only the culling differs and the results are identical bit for bit;
    # with quads (the default) a ray through the second triangle of a quad takes t from the first one's plane and the
    # barycentrics refer to rotated corners: same primitive, t and barycentrics equal to float32 rounding.
    b = make_scene("path", accel="bvh")
    rays = torch.from_numpy(random_rays(200000, (-3, 0, -5.5), (2.5, 5.2, 6), seed=3)).cuda()
    ipb, btb = b.trace_closest(rays)
    a = make_scene("path", accel="brute")
    ipa, bta = a.trace_closest(rays)
    same = (ipa == ipb).all(dim=1)
    assert (~same).float().mean().item() < 1e-4
    ga, gb = bta[same].cpu().numpy(), btb[same].cpu().numpy()
    hit = (ipa[same][:, 0] >= 0).cpu().numpy()
    terr = np.abs(ga[hit, 2] - gb[hit, 2]) / (1e-5 * np.abs(gb[hit, 2]) + 5e-6)       # the bounds of check_closest
    berr = np.abs(ga[hit, :2] - gb[hit, :2]).max(axis=1)
    print(f"[trace] quads vs per-triangle BVH: t err max {terr.max():.3f} (> 1: {(terr > 1).mean():.2e}), bary err max {berr.max():.2e}, equal bit for bit: {(ga == gb).all(axis=1).mean():.3f}")
    assert (terr > 1).mean() < 2e-4 and terr.max() < 50 and (berr > 1e-4).mean() < 2e-4 and berr.max() < 5e-3
    assert np.array_equal(ga[~hit], gb[~hit])
    monkeypatch.setenv("ZDR_NO_QUADS", "1")
    a = make_scene("path", accel="brute")
    ipa, bta = a.trace_closest(rays)
    same = (ipa == ipb).all(dim=1)
    assert (~same).float().mean().item() < 1e-4
    assert torch.equal(bta[same], btb[same])


def test_terrain_bvh_matches_oracle_brute_force():
    A = terrain_arrays(n=48)                       # 4.6k triangles: oracle brute force stays in seconds
    scene = make_scene("path", arrays=A)
    info = scene.info()
    assert info["accel"] == "bvh" and info["bvh_nodes"] > 500
    S = oracle.OracleScene.from_arrays(A)
    rays = random_rays(30000, (-3, -0.5, -3), (3, 3.5, 3), seed=5)
    check_closest(scene, S, rays, "terrain/bvh")
    rays[:, 3] = 1e-4; rays[:, 7] = 2.5
    occ = scene.trace_any(torch.from_numpy(rays).cuda()).cpu().numpy()
    assert (occ != S.trace_any(rays)).mean() < 2e-4


def test_large_bvh_any_is_consistent_with_closest():
    # size-independent property at a scale the oracle cannot brute-force (131k triangles)
    A = terrain_arrays(n=256)
    scene = make_scene("path", arrays=A)
    assert scene.info()["ntris"] == 2 * 256 * 256 + 2
    rays = random_rays(500000, (-3, -0.5, -3), (3, 3.5, 3), seed=6)
    tmax = np.random.default_rng(7).uniform(0.05, 5.0, rays.shape[0]).astype(np.float32)
    r = torch.from_numpy(rays).cuda()
    ip, bt = scene.trace_closest(r)
    rays2 = rays.copy(); rays2[:, 7] = tmax
    occ = scene.trace_any(torch.from_numpy(rays2).cuda())
    t = bt[:, 2]; hit = ip[:, 0] >= 0
    expect = hit & (t < torch.from_numpy(tmax).cuda())
    margin = (t - torch.from_numpy(tmax).cuda()).abs() > 1e-4          # ignore hits right at tmax
    assert ((occ != 0) == expect)[margin].all()


def test_million_triangle_tessellated_cbox():
    """BASELINE configs[4] at full size: size-independent properties (the oracle cannot brute-force 1M
    triangles): any-hit agrees with closest-hit, the closed room has no holes, and the displaced,
    tessellated room renders to (statistically) the same image as the 32-triangle room."""
    from conftest import cbox_material_np
    from zdr_amd import procedural
    A = procedural.tessellated_cbox(cbox_models(), n=183, amplitude=0.004)
    scene = make_scene("path", arrays=A)
    info = scene.info()
    assert info["ntris"] == 30 * 183 * 183 + 2 and info["accel"] == "bvh" and info["bvh_stack_entries"] <= 48
    rng = np.random.default_rng(11)
    n = 400000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.uniform((-2.5, 0.3, -5.3), (2.0, 4.8, -0.8), (n, 3))
    d = rng.standard_normal((n, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 7] = 1e30
    r = torch.from_numpy(rays).cuda()
    ip, bt = scene.trace_closest(r)
    # the room is closed except for the open front (+z) and the hairline gaps of the original mesh (its
    # walls do not meet exactly): interior rays heading away from the opening hit as often as they do in
    # the 32-triangle room, i.e. tessellation + BVH open no new holes
    inward = torch.from_numpy(rays[:, 6] < -0.2).cuda()
    ip32, _ = make_scene("path").trace_closest(r)
    hit_big, hit_small = (ip[inward, 0] >= 0).float().mean().item(), (ip32[inward, 0] >= 0).float().mean().item()
    assert hit_big > 0.999 and abs(hit_big - hit_small) < 3e-4, (hit_big, hit_small)
    tmax = torch.from_numpy(rng.uniform(0.05, 6.0, n).astype(np.float32)).cuda()
    r2 = r.clone(); r2[:, 7] = tmax
    occ = scene.trace_any(r2)
    expect = (ip[:, 0] >= 0) & (bt[:, 2] < tmax)
    margin = (bt[:, 2] - tmax).abs() > 1e-3
    assert ((occ != 0) == expect)[margin].all()
    m = torch.from_numpy(cbox_material_np()).cuda()
    big = scene.render(m, res=(128, 128), spp=64, seed=1)[..., :3]
    small = make_scene("path").render(m, res=(128, 128), spp=64, seed=1)[..., :3]
    assert abs(big.mean().item() - small.mean().item()) / small.mean().item() < 0.03
    assert not torch.isnan(big).any()


@pytest.mark.parametrize("accel", ["bvh"])
def test_reference_pinned_sphere(accel):
    """sphere.obj as the REFERENCE's loader reads it (tests/golden/obj_fixtures.npz: 559 vertices, 960 triangles,
    /root/reference/load_obj.py:1-68) under a small light: BVH closest / any hit against the oracle's brute force."""
    import os
    from conftest import ASSETS, GOLDEN
    from zdr_amd import geometry
    fx = np.load(os.path.join(GOLDEN, "obj_fixtures.npz"))
    A = geometry.assemble([(os.path.join(ASSETS, "sphere.obj"), None, 0.0), (os.path.join(ASSETS, "cbox-light.obj"), None, 20.0)])
    assert np.array_equal(A.tris[:960].reshape(-1), fx["sphere_triangles"])
    np.testing.assert_array_equal(A.verts[:559, :5], fx["sphere_vertices"][:, :5].astype(np.float32))
    scene = make_scene("path", arrays=A, accel=accel)
    assert scene.info()["accel"] == "bvh" and scene.info()["ntris"] == 962
    S = oracle.OracleScene.from_arrays(A)
    lo, hi = A.verts[:559, :3].min(0) - 0.5, A.verts[:559, :3].max(0) + 0.5
    rays = random_rays(60000, lo, hi, seed=9)
    check_closest(scene, S, rays, "sphere.obj/bvh")
    rays[:, 3] = 1e-4; rays[:, 7] = np.random.default_rng(10).uniform(0.05, 2.0, rays.shape[0]).astype(np.float32)
    occ = scene.trace_any(torch.from_numpy(rays).cuda()).cpu().numpy()
    assert (occ != S.trace_any(rays)).mean() < 2e-4
