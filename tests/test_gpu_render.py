"""-m gpu: forward images and gradients of the HIP path (called through the C-ABI) against the CPU
oracle on the same seeded inputs; shard unions; the autograd boundary (render.py:201-241)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import CBOX_CAMERA, cbox_material_np, fd_material_np
from zdr_amd.mathtypes import Camera, float3
from gpu_util import (TERRAIN_CAMERA, Flips, assert_grad_parity, assert_image_parity, make_scene, oracle_params, terrain_arrays)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mat_a():
    return cbox_material_np()


@pytest.fixture(scope="module")
def mat_b():
    return fd_material_np(256, 0)


@pytest.fixture(scope="module")
def cbox_oracle_fma(cbox_arrays):
    # fp32 noise-floor calibration only (see gpu_util.assert_image_parity); never the reference
    return oracle.OracleScene.from_arrays(cbox_arrays, variant="fma")


@pytest.mark.parametrize("integrator,W,spp", [("collocated", 256, 1), ("direct", 128, 16), ("path", 128, 16), ("path", 64, 64)])
@pytest.mark.parametrize("accel", ["brute", "bvh"])
def test_forward_matches_oracle(integrator, W, spp, accel, cbox_oracle, mat_a):
    scene = make_scene(integrator, accel=accel)
    m = torch.from_numpy(mat_a).cuda()
    img = scene.render(m, res=(W, W), spp=spp, seed=0).cpu().numpy()
    ref = cbox_oracle.render_forward(oracle_params(scene, W, W, spp, 0, mat_a.shape[:2]), mat_a)
    assert (img[..., 3] == 1.0).all()
    assert_image_parity(img[..., :3], ref[..., :3], f"{integrator}/{accel} {W}x{W} spp{spp}")


@pytest.mark.parametrize("integrator", ["direct", "path"])
def test_forward_glossy_material_box_filter_other_seed(integrator, cbox_oracle, cbox_oracle_fma, mat_b):
    scene = make_scene(integrator)
    scene.use_tent_filter = False
    m = torch.from_numpy(mat_b).cuda()
    img = scene.render(m, res=(96, 96), spp=16, seed=12345).cpu().numpy()
    p = oracle_params(scene, 96, 96, 16, 12345, mat_b.shape[:2])
    ref = cbox_oracle.render_forward(p, mat_b)
    floor = cbox_oracle_fma.render_forward(p, mat_b)
    assert_image_parity(img[..., :3], ref[..., :3], f"{integrator} glossy box-filter seed 12345", floor=floor[..., :3])


@pytest.mark.parametrize("integrator", ["collocated", "direct", "path"])
@pytest.mark.parametrize("material", ["A", "B"])
def test_backward_matches_oracle(integrator, material, cbox_oracle, cbox_oracle_fma, mat_a, mat_b):
    mat = mat_a if material == "A" else mat_b
    scene = make_scene(integrator)
    W, spp, seed = 96, 16, 5
    rng = np.random.default_rng(1)
    cot = rng.uniform(0.5, 1.5, (W, W, 4)).astype(np.float32)
    m = torch.from_numpy(mat).cuda().requires_grad_()
    img = scene.render(m, res=(W, W), spp=spp, seed=seed)
    (img * torch.from_numpy(cot).cuda()).sum().backward()
    # the reference renders the backward pass with seed + 1 (render.py:196)
    p = oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2])
    ref = cbox_oracle.render_backward(p, cot, mat)
    floor = cbox_oracle_fma.render_backward(p, cot, mat) if material == "B" else None
    assert_grad_parity(m.grad.cpu().numpy(), ref, f"backward {integrator} material {material}", floor=floor)


def test_backward_bvh_path(cbox_oracle, cbox_oracle_fma, mat_b):
    scene = make_scene("path", accel="bvh")
    W, spp, seed = 64, 16, 9
    m = torch.from_numpy(mat_b).cuda().requires_grad_()
    scene.render(m, res=(W, W), spp=spp, seed=seed).sum().backward()
    p = oracle_params(scene, W, W, spp, seed + 1, mat_b.shape[:2])
    ones = np.ones((W, W, 4), np.float32)
    ref = cbox_oracle.render_backward(p, ones, mat_b)
    assert_grad_parity(m.grad.cpu().numpy(), ref, "backward path/bvh", floor=cbox_oracle_fma.render_backward(p, ones, mat_b))


def test_terrain_scene_forward_and_backward():
    A = terrain_arrays(n=40)
    S = oracle.OracleScene.from_arrays(A)
    Sf = oracle.OracleScene.from_arrays(A, variant="fma")
    scene = make_scene("path", arrays=A)
    scene.camera = TERRAIN_CAMERA
    mat = fd_material_np(128, 3)
    W, spp = 64, 16
    m = torch.from_numpy(mat).cuda().requires_grad_()
    img = scene.render(m, res=(W, W), spp=spp, seed=2)
    ref = S.render_forward(oracle_params(scene, W, W, spp, 2, mat.shape[:2]), mat)
    assert ref[..., :3].mean() > 0.01
    floor = Sf.render_forward(oracle_params(scene, W, W, spp, 2, mat.shape[:2]), mat)
    assert_image_parity(img.detach().cpu().numpy()[..., :3], ref[..., :3], "terrain path forward", floor=floor[..., :3])
    img.sum().backward()
    pb = oracle_params(scene, W, W, spp, 3, mat.shape[:2])
    ones = np.ones((W, W, 4), np.float32)
    gref = S.render_backward(pb, ones, mat)
    assert_grad_parity(m.grad.cpu().numpy(), gref, "terrain path backward", floor=Sf.render_backward(pb, ones, mat))


def test_shard_unions(mat_a):
    scene = make_scene("path")
    m = torch.from_numpy(mat_a).cuda()
    W, spp = 64, 64
    full = scene.render_forward(m, (W, W), spp, 4)
    tiles = torch.zeros_like(full)
    for rect in [(0, 0, 40, 64), (40, 0, 64, 24), (40, 24, 64, 64)]:
        scene.render_forward(m, (W, W), spp, 4, rect=rect, out=tiles)
    assert torch.equal(tiles, full)                              # pixel tiles: bit-for-bit
    acc = torch.zeros_like(full)
    for s in [(0, 16), (16, 48), (48, 64)]:
        acc += scene.render_forward(m, (W, W), spp, 4, samples=s)
    torch.testing.assert_close(acc, full, rtol=1e-5, atol=1e-6)  # sample ranges: re-association only
    g_full = torch.zeros_like(m); g_acc = torch.zeros_like(m)
    ones = torch.ones_like(full)
    scene.render_backward(ones, g_full, m, (W, W), spp, 4)
    for rect in [(0, 0, 64, 32), (0, 32, 64, 64)]:
        scene.render_backward(ones, g_acc, m, (W, W), spp, 4, rect=rect)
    torch.testing.assert_close(g_acc, g_full, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("integrator,accel", [("path", "brute"), ("path", "bvh"), ("direct", "brute")])
@pytest.mark.parametrize("count", [2, 3, 8])
def test_interleaved_tile_shards_union(integrator, accel, count, mat_a):
    """BASELINE configs[3]: pixel tiles dealt round-robin to `count` ranks along diagonals (zdr_render_params.tile_shard_*), one launch
    each.  A pixel's samples do not depend on who renders it: the union of the shards is the unsharded image — bit for bit
    here, where shard and whole frame cut the sample range into the same chunks (at other sizes up to the re-association
    of the per-pixel sum, tools/shard_balance.py: 7e-7 at 1024^2 spp 1024) — the gradients add up to the unsharded gradient,
    the counters add up exactly; a shard touches no other pixel."""
    scene = make_scene(integrator, accel=accel)
    m = torch.from_numpy(mat_a).cuda()
    W, H, spp = 77, 52, 32                                       # 10 x 7 tiles, ragged right and bottom edges
    full = scene.render_forward(m, (W, H), spp, 6)
    parts = torch.full_like(full, -1.0)
    owner = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    for r in range(count):
        one = scene.render_forward(m, (W, H), spp, 6, tile_shard=(r, count), out=torch.full_like(full, -1.0))
        mine = one[..., 3] >= 0
        owner += mine.int()
        parts = torch.where(mine[..., None], one, parts)
        ty, tx = torch.meshgrid(torch.arange(H, device="cuda") // 8, torch.arange(W, device="cuda") // 8, indexing="ij")
        tiles_x = (W + 7) // 8                                    # zdr.h: row ty is numbered from column ty on -> a shard's tiles run along diagonals
        assert torch.equal(mine, (ty * tiles_x + (tx - ty) % tiles_x) % count == r)
        from zdr_amd import distributed as zd                     # the host-side mirror the gloo tests and rehearsals use
        host = torch.zeros((H, W), dtype=torch.bool)
        for (x0, y0, x1, y1) in zd.shard_tiles((0, 0, W, H), r, count): host[y0:y1, x0:x1] = True
        assert torch.equal(mine.cpu(), host)
    assert (owner == 1).all() and torch.equal(parts, full)
    cot = torch.from_numpy(np.random.default_rng(2).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)).cuda()
    g_full = torch.zeros_like(m); g_parts = torch.zeros_like(m)
    scene.render_backward(cot, g_full, m, (W, H), spp, 6)
    stats = {}
    for r in range(count):
        scene.render_backward(cot, g_parts, m, (W, H), spp, 6, tile_shard=(r, count))
        for k, v in scene.render_stats(m, (W, H), spp, 6, tile_shard=(r, count)).items(): stats[k] = stats.get(k, 0) + v
    torch.testing.assert_close(g_parts, g_full, rtol=1e-4, atol=1e-6 * float(g_full.abs().max()))
    assert stats == scene.render_stats(m, (W, H), spp, 6)


def test_a_shard_without_tiles_leaves_no_stale_tile_masks(mat_a):
    """More shards than tiles: the last shard owns nothing and launches nothing — in particular not k_tile_masks.  The
    handle must not remember masks it never built (the next call on the same view would have read a
    freshly allocated, uninitialised mask buffer and silently skipped primitives)."""
    m = torch.from_numpy(mat_a).cuda()
    W, H, spp = 24, 16, 8                                        # 3 x 2 = 6 tiles
    want = make_scene("path").render_forward(m, (W, H), spp, 5, tile_shard=(0, 7))
    scene = make_scene("path")                                   # a fresh handle: its mask buffer does not exist yet
    empty = scene.render_forward(m, (W, H), spp, 5, tile_shard=(6, 7))
    assert (empty == 0).all()
    g = torch.zeros_like(m)
    scene.render_backward(torch.ones((H, W, 4), device="cuda"), g, m, (W, H), spp, 5, tile_shard=(6, 7))
    assert (g == 0).all()
    got = scene.render_forward(m, (W, H), spp, 5, tile_shard=(0, 7))
    assert torch.equal(got, want)
    scene.check()


def test_tile_shards_inside_a_rectangle_and_for_uvgrad(mat_a):
    """Tile shards are numbered from the rectangle's own corner, so they compose with rectangle shards; and the
    render_duvdxy kernel (which writes the image directly, one chunk) honours them too."""
    scene = make_scene("path")
    m = torch.from_numpy(mat_a).cuda()
    W, H, spp = 61, 45, 16
    rect = (13, 7, 58, 41)
    whole = scene.render_forward(m, (W, H), spp, 3, rect=rect, out=torch.full((H, W, 4), -1.0, device="cuda"))
    parts = torch.full_like(whole, -1.0)
    for r in range(3):
        scene.render_forward(m, (W, H), spp, 3, rect=rect, tile_shard=(r, 3), out=parts)
    assert torch.equal(parts, whole)
    assert (whole[:7] == -1).all() and (whole[:, :13] == -1).all() and (whole[41:] == -1).all() and (whole[:, 58:] == -1).all()
    from zdr_amd import _native as N
    uv = scene.render_forward(m, (W, H), spp, 3, kernel=N.UVGRAD)
    uparts = torch.zeros_like(uv)
    for r in range(4):
        scene.render_forward(m, (W, H), spp, 3, kernel=N.UVGRAD, tile_shard=(r, 4), out=uparts)
    assert torch.equal(uparts, uv)


@pytest.mark.parametrize("integrator", ["collocated", "direct", "path"])
def test_tile_masks_cull_nothing_that_can_be_hit(integrator, mat_a, monkeypatch):
    """Camera rays of a tile test only the triangle pairs in the tile's mask (k_tile_masks).  The mask may keep
    too much, never too little: images are bit-identical (gradients up to atomic ordering) with the masks off, for
    cameras inside the box, tilted, with a wide field of view, and with the tent filter's half-pixel reach."""
    m = torch.from_numpy(mat_a).cuda()
    cams = [(CBOX_CAMERA[0], CBOX_CAMERA[1], CBOX_CAMERA[2], CBOX_CAMERA[3]),
            (1.9, (0.3, 1.2, -1.0), (-0.4, 2.0, -4.0), (0.2, 1.0, 0.1)),          # inside the box, rolled, wide
            (0.35, (0.0, 2.7, 6.0), (1.5, 0.3, -3.0), (0.0, 1.0, 0.0)),           # narrow, looking at a box corner
            (1.2, (0.0, 2.7, -3.0), (0.0, 2.7, 5.0), (0.0, 1.0, 0.0))]            # looking out of the open side
    for k, (fov, o, t, up) in enumerate(cams):
        for tent in (False, True):
            scene = make_scene(integrator, accel="brute")
            scene.camera = Camera(fov=fov, origin=float3(*o), target=float3(*t), up=float3(*up))
            scene.use_tent_filter = tent
            W, H, spp = 72, 56, 4
            ones = torch.ones((H, W, 4), device="cuda")
            monkeypatch.delenv("ZDR_NO_TILE_MASKS", raising=False)
            img = scene.render_forward(m, (W, H), spp, 10 + k)
            g = torch.zeros_like(m); scene.render_backward(ones, g, m, (W, H), spp, 10 + k)
            monkeypatch.setenv("ZDR_NO_TILE_MASKS", "1")
            img0 = scene.render_forward(m, (W, H), spp, 10 + k)
            g0 = torch.zeros_like(m); scene.render_backward(ones, g0, m, (W, H), spp, 10 + k)
            assert torch.equal(img, img0), (integrator, k, tent)
            torch.testing.assert_close(g, g0, rtol=1e-4, atol=1e-6)   # float atomics: same terms, free order
            if k == 0: assert float(img[..., :3].sum()) > 0.0


@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_shadow_pair_mask_changes_no_answer(integrator, mat_a, mat_b, monkeypatch):
    """The shadow walk of the brute-force accel skips the pairs whose primitives can never lie between a surface point and a point
    of a light (zdr_api.cpp never_occluders: 3 of the Cornell box's 9 pairs — back wall, floor, side walls).  Same answers, bit for
    bit: images equal, per-path traces (every vertex's light-sample decision, radiance and gradient) equal, gradient textures equal
    up to the order of the float atomics — for the stock lights, after update_lights has made a box a second light, and for a camera
    inside the box.  ZDR_NO_SHADOW_MASK=1 (read when the lights are set) keeps every pair in the walk."""
    from path_trace import all_queries
    def scenes():
        monkeypatch.delenv("ZDR_NO_SHADOW_MASK", raising=False)
        a = make_scene(integrator, accel="brute")
        monkeypatch.setenv("ZDR_NO_SHADOW_MASK", "1")
        b = make_scene(integrator, accel="brute")
        monkeypatch.delenv("ZDR_NO_SHADOW_MASK", raising=False)
        return a, b
    W, H, spp = 72, 56, 8
    cot = torch.rand((H, W, 4), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) + 0.5
    for case, mat in (("stock", mat_a), ("glossy", mat_b), ("inside", mat_b)):
        m = torch.from_numpy(mat).cuda()
        a, b = scenes()
        if case == "inside":
            for sc in (a, b): sc.camera = Camera(fov=1.9, origin=float3(0.3, 1.2, -1.0), target=float3(-0.4, 2.0, -4.0), up=float3(0.2, 1.0, 0.1))
        for seed in (1, 2):
            ia, ib = a.render_forward(m, (W, H), spp, seed), b.render_forward(m, (W, H), spp, seed)
            assert torch.equal(ia, ib), (case, seed)
            ga, gb = torch.zeros_like(m), torch.zeros_like(m)
            a.render_backward(cot, ga, m, (W, H), spp, seed); b.render_backward(cot, gb, m, (W, H), spp, seed)
            torch.testing.assert_close(ga, gb, rtol=1e-4, atol=1e-6 * float(gb.abs().max()))   # float atomics: same terms, free order
            if integrator == "path":
                q = torch.from_numpy(all_queries(W, H, spp)).cuda()
                assert torch.equal(a.path_dump(m, q, (W, H), spp, seed + 1, d_image=cot).view(torch.int32), b.path_dump(m, q, (W, H), spp, seed + 1, d_image=cot).view(torch.int32)), (case, seed)
        assert float(ia[..., :3].sum()) > 0.0
    # another set of lights: the room itself (instance 0) glows as well -> the mask is rebuilt with the lights (every plane now carries a light: nothing is ruled out)
    m = torch.from_numpy(mat_a).cuda()
    monkeypatch.delenv("ZDR_NO_SHADOW_MASK", raising=False)
    a = make_scene(integrator, accel="brute"); a.update_lights([float3(2.0, 1.0, 0.5), float3(17, 12, 4)])
    monkeypatch.setenv("ZDR_NO_SHADOW_MASK", "1")
    b = make_scene(integrator, accel="brute"); b.update_lights([float3(2.0, 1.0, 0.5), float3(17, 12, 4)])
    monkeypatch.delenv("ZDR_NO_SHADOW_MASK", raising=False)
    assert torch.equal(a.render_forward(m, (W, H), spp, 5), b.render_forward(m, (W, H), spp, 5))


@pytest.mark.parametrize("W,H,spp,max_depth,rr_depth", [
    (1, 1, 1, 16, 2),        # one pixel, one sample: a wave with one valid lane and a one-entry FIFO
    (5, 3, 3, 16, 2),        # fewer camera samples than one refill batch
    (24, 16, 32, 1, 2),      # max_depth 1: every path stops after its first vertex
    (24, 16, 32, 16, 0),     # Russian roulette from the first vertex: every record carries its RR fields
    (24, 16, 32, 16, 16),    # no Russian roulette: paths run to max_depth, up to 15 records per lane — the wave's pool of 106 LDS slots overflows and most records (and their links) go through scratch
    (24, 16, 32, 3, 1),
])
def test_edge_configurations_match_oracle(W, H, spp, max_depth, rr_depth, cbox_oracle, cbox_oracle_fma, mat_a):
    """Path forward + backward at the corners of the configuration space of prb.py:15-16 and of the
    persistent kernels' machinery (tiny shards, tiny sample counts, depth limits).  Material A (rough):
    with a few dozen pixels the floor-calibrated statistics of a glossy material mean nothing."""
    m = torch.from_numpy(mat_a).cuda()
    scene = make_scene("path")
    scene.max_depth, scene.rr_depth = max_depth, rr_depth
    p = oracle_params(scene, W, H, spp, 77, mat_a.shape[:2])
    img = scene.render_forward(m, (W, H), spp, 77).cpu().numpy()
    ref = cbox_oracle.render_forward(p, mat_a)
    tag = f"path {W}x{H} spp{spp} depth{max_depth}/{rr_depth}"
    assert_image_parity(img[..., :3], ref[..., :3], tag)
    cot = np.random.default_rng(3).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)
    g = torch.zeros_like(m)
    scene.render_backward(torch.from_numpy(cot).cuda(), g, m, (W, H), spp, 77)
    pb = oracle_params(scene, W, H, spp, 78, mat_a.shape[:2])    # backward renders with seed + 1 (render.py:196)
    assert_grad_parity(g.cpu().numpy(), cbox_oracle.render_backward(pb, cot, mat_a), tag + " backward",
                       floor=cbox_oracle_fma.render_backward(pb, cot, mat_a))   # 16-vertex paths accumulate rounding


@pytest.mark.parametrize("accel", ["brute", "bvh"])
def test_work_item_granularity_only_reassociates(accel, mat_b, monkeypatch):
    """The path kernels' persistent waves draw (tile x sample-chunk) items and overlap consecutive items in two LDS
    banks.  How the samples are cut into items must not matter beyond the order of float additions: one item per
    tile (a wave finishes a tile before the next), the default, and 4-sample items (every wave juggles many items
    and both banks all the time) give the same image and gradient."""
    m = torch.from_numpy(mat_b).cuda()
    scene = make_scene("path", accel=accel)
    W, H, spp = 88, 40, 64                                       # 55 tiles, the last column half empty
    cot = torch.from_numpy(np.random.default_rng(5).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)).cuda()
    out = []
    for env in ({"ZDR_TARGET_WAVES": "1"}, {}, {"ZDR_TARGET_WAVES": "100000", "ZDR_MIN_CHUNK": "4"}):
        for k in ("ZDR_TARGET_WAVES", "ZDR_MIN_CHUNK"): monkeypatch.delenv(k, raising=False)
        for k, v in env.items(): monkeypatch.setenv(k, v)
        img = scene.render_forward(m, (W, H), spp, 21)
        g = torch.zeros_like(m); scene.render_backward(cot, g, m, (W, H), spp, 21)
        st = scene.render_stats(m, (W, H), spp, 21)
        out.append((img, g, st))
    for img, g, st in out[1:]:
        torch.testing.assert_close(img, out[0][0], rtol=2e-5, atol=1e-6)
        torch.testing.assert_close(g, out[0][1], rtol=1e-4, atol=1e-6 * float(out[0][1].abs().max()))
        assert st == out[0][2]                                   # the very same paths: identical counters


@pytest.mark.parametrize("integrator", ["direct", "path"])
def test_shards_at_odd_offsets(integrator, mat_a):
    """Shard rectangles need not sit on the 8x8 tile grid of the full image: tiles are laid out from the
    rectangle's own corner, so the same pixel lands in different tiles (and lanes) — only the order of the
    float additions may change."""
    scene = make_scene(integrator)
    m = torch.from_numpy(mat_a).cuda()
    W, H, spp = 61, 45, 32
    full = scene.render_forward(m, (W, H), spp, 8)
    parts = torch.zeros_like(full)
    for rect in [(0, 0, 13, 45), (13, 0, 61, 7), (13, 7, 38, 45), (38, 7, 61, 45)]:
        scene.render_forward(m, (W, H), spp, 8, rect=rect, out=parts)
    torch.testing.assert_close(parts, full, rtol=2e-5, atol=1e-6)
    cot = torch.from_numpy(np.random.default_rng(2).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)).cuda()
    g_full = torch.zeros_like(m); g_parts = torch.zeros_like(m)
    scene.render_backward(cot, g_full, m, (W, H), spp, 8)
    for rect in [(0, 0, 61, 11), (0, 11, 29, 45), (29, 11, 61, 45)]:
        scene.render_backward(cot, g_parts, m, (W, H), spp, 8, rect=rect)
    torch.testing.assert_close(g_parts, g_full, rtol=1e-4, atol=1e-6 * float(g_full.abs().max()))


def test_limits_are_reported_before_anything_is_launched(mat_a):
    """INTEGRATION.md, Limits: the path kernels pack (pixel, bank, sample index) into 32 bits."""
    from zdr_amd._native import ZdrError
    scene = make_scene("path")
    m = torch.from_numpy(mat_a).cuda()
    with pytest.raises(ZdrError, match="spp above 2\\^25"):
        scene.render_forward(m, (8, 8), (1 << 25) + 1, 0, samples=(0, 4))
    img = scene.render_forward(m, (8, 8), 1 << 25, 0, samples=(0, 4))      # at the limit: fine (4 of 2^25 samples rendered)
    assert torch.isfinite(img).all()


def test_stats_match_oracle_counters(cbox_oracle, mat_a):
    scene = make_scene("path")
    m = torch.from_numpy(mat_a).cuda()
    got = scene.render_stats(m, (64, 64), 16, seed=0)
    _, ref = cbox_oracle.render_forward(oracle_params(scene, 64, 64, 16, 0, mat_a.shape[:2]), mat_a, counters=True)
    for k in ("samples", "closest_rays", "closest_hits", "shadow_rays", "shaded_vertices", "emitter_hits_bsdf"):
        assert abs(got[k] - ref[k]) <= max(3, 1e-3 * ref[k]), (k, got[k], ref[k])
    assert got["samples"] == 64 * 64 * 16


def test_autograd_boundary_semantics(mat_b):
    scene = make_scene("path")
    m = torch.from_numpy(mat_b).cuda().requires_grad_()
    img = scene.render(m, res=(32, 32), spp=4)                   # seed defaults to 0
    assert img.shape == (32, 32, 4) and img.dtype == torch.float32 and img.is_cuda
    cam0 = scene.camera
    scene.camera = type(cam0)(fov=0.3, origin=cam0.origin, target=cam0.target, up=cam0.up)   # changed after forward
    img.sum().backward()
    g1 = m.grad.clone(); m.grad = None
    assert scene.camera.fov == pytest.approx(0.3)                # restored to the user's camera
    scene.camera = cam0
    scene.render(m, res=(32, 32), spp=4).sum().backward()
    # backward used the camera snapshot of its own forward (render.py:216-222): same gradient up to atomics order
    torch.testing.assert_close(g1, m.grad, rtol=1e-4, atol=1e-6)
    with pytest.raises(AssertionError):
        scene.render(torch.rand((8, 8, 3), device="cuda"), res=(8, 8), spp=1)
    with pytest.raises(KeyError):
        make_scene("bidirectional")


def test_update_lights(mat_a):
    scene = make_scene("direct")
    m = torch.from_numpy(mat_a).cuda()
    lit = scene.render(m, res=(32, 32), spp=4)
    scene.update_lights([None, 0])
    dark = scene.render(m, res=(32, 32), spp=4)
    scene.update_lights([None, 20.0])
    again = scene.render(m, res=(32, 32), spp=4)
    assert lit[..., :3].max() > 1 and dark[..., :3].max() == 0 and torch.equal(lit, again)


def test_instance_transforms(mat_b):
    """per-instance 4x4 transforms (render.py:83-84,109; interaction.py:19-28).  The native
    side flattens instances to world space at build time, the oracle transforms at interaction time."""
    import math
    from conftest import cbox_models
    from zdr_amd import float4x4, geometry
    c, s = math.cos(0.15), math.sin(0.15)
    room = float4x4.from_rows([[1.0, 0, 0, 0.1], [0, 0.9, 0, 0.2], [0, 0, 1.1, -0.3], [0, 0, 0, 1]])          # non-uniform scale: exercises the inverse-transpose
    light = float4x4.from_rows([[c, 0, -s, 0.0], [0, 1, 0, -0.6], [s, 0, c, -0.2], [0, 0, 0, 1]])
    models = [(cbox_models()[0][0], room, 0.0), (cbox_models()[1][0], light, 25.0)]
    A = geometry.assemble(models)
    S = oracle.OracleScene.from_arrays(A); Sf = oracle.OracleScene.from_arrays(A, variant="fma")
    scene = make_scene("path", models=models)
    W, spp = 64, 16
    m = torch.from_numpy(mat_b).cuda().requires_grad_()
    img = scene.render(m, res=(W, W), spp=spp, seed=3)
    p = oracle_params(scene, W, W, spp, 3, mat_b.shape[:2])
    ref = S.render_forward(p, mat_b)
    assert ref[..., :3].mean() > 0.02
    assert_image_parity(img.detach().cpu().numpy()[..., :3], ref[..., :3], "transformed instances forward", floor=Sf.render_forward(p, mat_b)[..., :3],
                        flips=Flips(scene, S, Sf, mat_b, (W, W), spp, 3, what="transformed instances forward"))
    img.sum().backward()
    pb = oracle_params(scene, W, W, spp, 4, mat_b.shape[:2])
    ones = np.ones((W, W, 4), np.float32)
    assert_grad_parity(m.grad.cpu().numpy(), S.render_backward(pb, ones, mat_b), "transformed instances backward", floor=Sf.render_backward(pb, ones, mat_b),
                       flips=Flips(scene, S, Sf, mat_b, (W, W), spp, 4, cot=ones, what="transformed instances backward"))


def test_render_duvdxy_matches_oracle(cbox_oracle, mat_a):
    scene = make_scene("direct")
    m = torch.from_numpy(mat_a).cuda()
    got = scene.render_duvdxy(m, res=(128, 128), spp=16, seed=3).cpu().numpy()
    p = oracle_params(scene, 128, 128, 16, 3, mat_a.shape[:2])
    p.integrator = oracle.UVGRAD
    ref = cbox_oracle.render_forward(p, mat_a)
    assert np.abs(ref).max() > 1e-3
    d = np.abs(got - ref)
    # pixels whose primary ray grazes a triangle edge may pick the neighbouring triangle
    assert (d > 1e-5 + 1e-3 * np.abs(ref)).mean() < 2e-3, (d.max(), (d > 1e-5 + 1e-3 * np.abs(ref)).mean())


def test_a_tripped_watchdog_is_reported_not_swallowed(mat_a, monkeypatch):
    """The BVH walk and the persistent path loop carry watchdogs that end work instead of spinning.  If one trips,
    the device error word makes the next check fail (zdr_scene_check, zdr_render_stats, or every call under
    ZDR_CHECK=1) — an incomplete image never comes back as ZDR_OK with nothing said.  The walk budget is forced to
    three iterations here (ZDR_DEBUG_BVH_BUDGET); a normal scene reports nothing."""
    from zdr_amd._native import ZdrError
    m = torch.from_numpy(mat_a).cuda()
    good = make_scene("path", accel="bvh")
    good.render_forward(m, (32, 32), 4, 0)
    good.check()                                                  # nothing tripped
    monkeypatch.setenv("ZDR_DEBUG_BVH_BUDGET", "3")
    bad = make_scene("path", accel="bvh")
    monkeypatch.delenv("ZDR_DEBUG_BVH_BUDGET")
    import os
    if os.environ.get("ZDR_CHECK", "0") not in ("", "0"):          # the suite itself runs under ZDR_CHECK=1: the render call reports it
        with pytest.raises(ZdrError, match="BVH walk exceeded its iteration budget"):
            bad.render_forward(m, (32, 32), 4, 0)
        bad.check()                                               # reading the word cleared it
        return
    img = bad.render_forward(m, (32, 32), 4, 0)
    assert torch.isfinite(img).all()                              # zero-filled, never uninitialised memory
    with pytest.raises(ZdrError, match="BVH walk exceeded its iteration budget"):
        bad.check()
    bad.check()                                                   # reading the word clears it
    g = torch.zeros_like(m)
    bad.render_backward(torch.ones((32, 32, 4), device="cuda"), g, m, (32, 32), 4, 0)
    with pytest.raises(ZdrError, match="incomplete"):
        bad.render_stats(m, (32, 32), 4)                          # the stats call checks by itself


@pytest.mark.parametrize("tex", [(1, 1), (2, 2), (4, 4), (5, 3), (8, 8), (16, 16), (40, 24)])
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_few_texels_gradient_matches_oracle(tex, integrator, cbox_oracle, cbox_oracle_fma):
    """README.md:21 of the reference: gradients that concentrate on few texels.  The kernels then keep the whole
    staging-cell array in LDS (<= 28 cells) or add into replicated cell arrays (scene.h); both are re-associations of
    the same sum.  (tex_h, tex_w) from a constant material (1 x 1) over non-square ones; float32 accumulators that
    receive every term of the launch are where the unreduced scatter lost 40 % of the sum."""
    th, tw = tex
    rng = np.random.default_rng(th * 100 + tw)
    mat = np.empty((th, tw, 4), np.float32)
    mat[..., :3] = rng.uniform(0.2, 0.8, (th, tw, 3)); mat[..., 3] = rng.uniform(0.6, 1.0, (th, tw))
    scene = make_scene(integrator)
    W, spp, seed = 96, 16, 4
    cot = rng.uniform(0.5, 1.5, (W, W, 4)).astype(np.float32)
    m = torch.from_numpy(mat).cuda()
    g = torch.zeros_like(m)
    scene.render_backward(torch.from_numpy(cot).cuda(), g, m, (W, W), spp, seed)
    scene.check()
    ref = cbox_oracle.render_backward(oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2]), cot, mat)
    got = g.cpu().numpy()
    # every texel holds thousands of contributions: compare texel by texel (the larger textures may show a flipped path)
    if th * tw <= 64:
        bad = np.abs(got - ref) > 2e-3 * np.abs(ref) + 2e-4 * np.abs(ref).max()
        assert bad.sum() == 0, (int(bad.sum()), np.abs(got - ref).max())
    else:
        assert_grad_parity(got, ref, f"few texels {th}x{tw} {integrator}",
                           flips=Flips(scene, cbox_oracle, cbox_oracle_fma, mat, (W, W), spp, seed + 1, cot=cot, what=f"few texels {th}x{tw}") if integrator == "path" else None)
    assert abs(got.sum() - ref.sum()) <= 3e-4 * abs(ref.sum())


@pytest.mark.parametrize("accel", ["brute", "bvh"])
def test_detached_adjoint_mode_matches_oracle(accel, cbox_oracle, cbox_oracle_fma, mat_b):
    """scene.prb_mode = "detached": the adjoint the reference's autodiff blocks compute — Russian-roulette factors and
    MIS weights held constant (prb.py:138-146, 157-163) — against the oracle's ZDRO_PRB_DETACHED; it is NOT the default
    because it is not the derivative finite differences measure (tests/test_oracle_render.py)."""
    scene = make_scene("path", accel=accel)
    scene.prb_mode = "detached"
    W, spp, seed = 64, 16, 21
    cot = np.random.default_rng(8).uniform(0.5, 1.5, (W, W, 4)).astype(np.float32)
    m = torch.from_numpy(mat_b).cuda()
    g = torch.zeros_like(m)
    scene.render_backward(torch.from_numpy(cot).cuda(), g, m, (W, W), spp, seed)
    p = oracle_params(scene, W, W, spp, seed + 1, mat_b.shape[:2], prb_mode=oracle.PRB_DETACHED)
    ref = cbox_oracle.render_backward(p, cot, mat_b)
    assert_grad_parity(g.cpu().numpy(), ref, f"detached adjoint / {accel}", floor=cbox_oracle_fma.render_backward(p, cot, mat_b),
                       flips=Flips(scene, cbox_oracle, cbox_oracle_fma, mat_b, (W, W), spp, seed + 1, cot=cot, what=f"detached adjoint / {accel}", prb_mode=oracle.PRB_DETACHED))
    scene.prb_mode = "expectation"
    g2 = torch.zeros_like(m)
    scene.render_backward(torch.from_numpy(cot).cuda(), g2, m, (W, W), spp, seed)
    rough = (g2[..., 3] - g[..., 3]).abs().sum() / g2[..., 3].abs().sum()
    assert rough > 0.01                                           # the roughness channel is where the two forms differ
    scene.prb_mode = "no such mode"
    with pytest.raises(KeyError):
        scene.render_backward(torch.from_numpy(cot).cuda(), g2, m, (W, W), spp, seed)


@pytest.mark.parametrize("accel", ["brute", "bvh"])
def test_literal_adjoint_mode_matches_oracle(accel, cbox_oracle, cbox_oracle_fma, mat_b):
    """scene.prb_mode = "literal": the BSDF-sample adjoint seeded as /root/reference/prb.py:157-163 writes it —
    backward(bsdf, beta / pdf_bsdf * Le * le_grad) with Le the REMAINING path radiance — against the oracle's ZDRO_PRB_LITERAL.
    Not the derivative of the forward (tests/test_oracle_render.py::test_prb_literal_weight_is_not_the_derivative: 19 % off finite
    differences); the mode exists because it is the one output the reference defines that the other two cannot produce."""
    scene = make_scene("path", accel=accel)
    scene.prb_mode = "literal"
    W, spp, seed = 64, 16, 33
    cot = np.random.default_rng(9).uniform(0.5, 1.5, (W, W, 4)).astype(np.float32)
    m = torch.from_numpy(mat_b).cuda()
    g = torch.zeros_like(m)
    scene.render_backward(torch.from_numpy(cot).cuda(), g, m, (W, W), spp, seed)
    p = oracle_params(scene, W, W, spp, seed + 1, mat_b.shape[:2], prb_mode=oracle.PRB_LITERAL)
    ref = cbox_oracle.render_backward(p, cot, mat_b)
    assert_grad_parity(g.cpu().numpy(), ref, f"literal adjoint / {accel}", floor=cbox_oracle_fma.render_backward(p, cot, mat_b),
                       flips=Flips(scene, cbox_oracle, cbox_oracle_fma, mat_b, (W, W), spp, seed + 1, cot=cot, what=f"literal adjoint / {accel}", prb_mode=oracle.PRB_LITERAL))
    scene.prb_mode = "detached"
    g2 = torch.zeros_like(m)
    scene.render_backward(torch.from_numpy(cot).cuda(), g2, m, (W, W), spp, seed)
    assert ((g2 - g).abs().sum() / g2.abs().sum()) > 0.01        # and it is a different gradient from the detached form's


def test_rccl_single_rank_exchange(mat_a):
    """The exchange step of the multi-GPU path (zdr_amd/distributed.py: one all_reduce of the image and one of the
    gradient) on the RCCL backend with the ranks this box has: ONE.  Not a scaling test — it shows that torch.distributed's
    "nccl" backend (= RCCL on ROCm) initialises on the MI355X box and reduces the tensors the renderer hands it."""
    import os
    import socket
    import torch.distributed as dist
    from zdr_amd import distributed as zd
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        scene = make_scene("path")
        m = torch.from_numpy(mat_a).cuda().requires_grad_()
        r = zd.attach(scene, mode="tiles")
        img = r.render(m, res=(64, 64), spp=16, seed=1)
        ref = scene.render_forward(m.detach(), (64, 64), 16, 1)
        assert torch.equal(img, ref)
        t = img.detach().clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                 # the collective the N-rank run issues, on RCCL
        torch.cuda.synchronize()
        assert torch.equal(t, ref)
        img.sum().backward()
        g = m.grad.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert torch.equal(g, m.grad) and float(g.abs().sum()) > 0
    finally:
        dist.destroy_process_group()


def test_texture_optimisation_example_converges(tmp_path):
    """examples/optimize_texture.py = the workflow of the reference's example.py (target render, random material, Adam
    through scene.render / PRB backward): the image loss must fall."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("optimize_texture", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "optimize_texture.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    losses = mod.run(iters=40, res=96, spp=8, tex=64, out=str(tmp_path), verbose=False)
    assert np.mean(losses[-5:]) < 0.75 * losses[0], (losses[0], losses[-5:])      # the Monte-Carlo noise of an 8-spp render sets the floor
    assert os.path.exists(tmp_path / "result.png") and os.path.exists(tmp_path / "footprints.png")

