"""Helpers for the -m gpu parity tests: HIP path (through the C-ABI) vs CPU oracle."""
import numpy as np
import torch

import oracle
from conftest import CBOX_CAMERA, cbox_models
from zdr_amd import Camera, Scene, float3, geometry
from zdr_amd.scenes import make_scene  # noqa: F401  (re-exported)


def oracle_params(scene, W, H, spp, seed, tex_hw, **kw):
    cam = scene.camera
    return oracle.make_params(scene.integrator, W, H, spp, seed, (cam.fov, tuple(cam.origin), tuple(cam.target), tuple(cam.up)), tex_hw,
                              use_tent=scene.use_tent_filter, max_depth=scene.max_depth, rr_depth=scene.rr_depth, **kw)


def image_diff_stats(got, ref):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    tol = 1e-4 * (1.0 + np.abs(ref))
    return {
        "frac_bad": float((d > tol).mean()),
        "mean_rel": float(d.mean() / max(np.abs(ref).mean(), 1e-30)),
        "max_abs": float(d.max()),
        "sum_rel": float(abs(got.sum() - ref.sum()) / max(abs(ref.sum()), 1e-30)),
    }


# Glossy inputs: each robust statistic may reach FLOOR_FACTOR x what the oracle's own FMA and IEEE builds differ by.  MEASURED
# (tools/glossy_floor.py, profiles/r3_glossy_floor_ratio.txt: 16 seeds x 4 scenes, flipped paths of all builds set aside): on renders
# of 64^2 x 16 paths and more the ratio HIP-vs-IEEE / FMA-vs-IEEE has median 1.10 - 1.24, 90th percentile 1.12 - 1.54, maximum 1.65
# — the HIP build perturbs more operations than FMA contraction alone (v_rcp / v_rsq / v_sqrt, device sin and cos at 1 - 2 ulp).
# On renders of a few thousand paths the ratio of two such small numbers is noise (maximum 5.7): those sizes are compared path by
# path instead (tests/test_golden.py, tests/test_gpu_paths.py).
FLOOR_FACTOR = 2.0


def _bound(key, base, fl, numel, mean_key):
    """The bar for one statistic, `fl` = the same statistics of the FMA build of the oracle (glossy inputs) or None.
    frac_bad is a COUNT of entries: a count of a handful fluctuates like a Poisson variable, so it is compared with the
    allowed mean plus three standard deviations of a count with that mean (below 5 % of the bar from 10^5 entries on).
    sum_rel: |sum(got - ref)| <= sum|got - ref|, so with a ruler the sum is held to the bar of the mean absolute error —
    the ratio of two signed sums of heavy-tailed terms that mostly cancel says nothing."""
    if key == "sum_rel":
        return max(base, _bound(mean_key, 0.0, fl, numel, mean_key)) if fl else base
    bound = max(base, FLOOR_FACTOR * fl[key]) if fl else base
    if key == "frac_bad":
        mean = bound * numel
        bound = (mean + 3.0 * np.sqrt(mean)) / numel
    return bound


class Flips:
    """The paths of ONE render whose discrete decisions differ between the HIP kernels and the oracle — MEASURED, by
    dumping every path on both sides (zdr_path_dump / zdro_path_dump) and comparing signatures (triangle hit at every
    vertex, light sample contributing or not, path continuing or not, kind of roulette event).  Such a path took another
    branch because a comparison flipped in the last ulp; its whole contribution moves.  The parity assertions take the
    pixels / texel footprints of exactly these paths out of both tensors and hold the REST to the base bars, and bound the
    number of flipped paths by what the oracle's own IEEE and FMA builds differ by.  (Round 2 allowed a blanket 20 flipped
    paths wherever a test passed its path count; that constant is gone.)"""

    def __init__(self, scene, S, Sf, mat, res, spp, seed, cot=None, what="", traces=None, **params_kw):
        """S / Sf: the oracle scene in its IEEE and FMA builds, in the same state as `scene` (emissions, environment, sampler
        tables).  seed: the seed of the paths to compare — seed for a forward image, seed + 1 for the gradient of
        render_backward(..., seed).  traces: (hip, ref, fma) Trace objects of these very paths, if the caller has them."""
        from path_trace import Trace, all_queries
        W, H = res
        th, tw = mat.shape[:2]
        q = all_queries(W, H, spp)
        if traces is None:
            m = torch.from_numpy(np.ascontiguousarray(mat)).to(scene.device)
            cd = None if cot is None else torch.from_numpy(np.ascontiguousarray(cot, np.float32)).to(scene.device)
            p = oracle_params(scene, W, H, spp, seed, (th, tw), **params_kw)
            hip = Trace(scene.path_dump(m, torch.from_numpy(q).to(scene.device), (W, H), spp, seed, d_image=cd).cpu().numpy())
            ref = Trace(S.path_dump(p, mat, q, d_image=cot))
            fma = Trace(Sf.path_dump(p, mat, q, d_image=cot))
        else:
            hip, ref, fma = traces
        self.hip, self.ref, self.fma = hip, ref, fma
        self.n_paths = q.shape[0]
        flipped = ~hip.signature_equal(ref)
        flipped_floor = ~fma.signature_equal(ref)
        self.count, self.floor_count = int(flipped.sum()), int(flipped_floor.sum())
        both = flipped | flipped_floor
        self.pixels = np.zeros((H, W), bool)
        self.pixels[q[both, 1], q[both, 0]] = True
        self.texels = np.zeros((th, tw), bool)
        for tr, sel in ((hip, flipped), (ref, both), (fma, flipped_floor)):
            live = tr.live[sel]
            uv = tr.uv[sel][live]
            px = uv[:, 0] * np.float32(tw - 1); py = (np.float32(1.0) - uv[:, 1]) * np.float32(th - 1)
            ix = px.astype(np.int32); iy = py.astype(np.int32)
            for dx in (0, 1):
                for dy in (0, 1):
                    self.texels[np.clip(iy + dy, 0, th - 1), np.clip(ix + dx, 0, tw - 1)] = True
        print(f"[flips] {what}: {self.count} of {self.n_paths} paths took another branch than the oracle's (oracle fma vs ieee: {self.floor_count}); "
              f"{int(self.pixels.sum())} pixels / {int(self.texels.sum())} texels set aside")

    def check_count(self, what):
        assert self.count <= max(5, 2 * self.floor_count), (what, "flipped paths", self.count, "fma floor", self.floor_count)


def assert_image_parity(got, ref, what, floor=None, frac_bad=2e-3, mean_rel=2e-5, sum_rel=1e-5, flips=None):
    """Stated fp32 tolerance of the forward image (BASELINE.json north_star: 'within a stated fp32
    tolerance').  Base bar: at most 0.2 % of the values differ by more than 1e-4 (1 + |ref|), the mean absolute
    error is below 2e-5 of the mean value and the image sum agrees to 1e-5.
    `flips` (a Flips of this very render): the pixels of the paths that measurably took another branch are left out of
    got, ref and floor, the bars hold for all other pixels, and the number of such paths is bounded (Flips.check_count).
    Glossy materials amplify last-ulp differences chaotically (visible-normal sampling takes
    sqrt(1 - |p|^2) near the disk rim, the GGX denominator cancels like 1/alpha^2): two CORRECT
    float32 evaluations of the reference's formulas then drift apart by far more than the base bar even on paths that keep
    every decision.  `floor` = image of the SAME oracle source compiled with FMA contraction; when given, each bound
    becomes max(base, FLOOR_FACTOR x what the two CPU builds differ by) — FLOOR_FACTOR is measured, see above."""
    got, ref = np.asarray(got), np.asarray(ref)
    if flips is not None:
        flips.check_count(what)
        keep = ~flips.pixels
        got, ref = got[keep], ref[keep]
        floor = None if floor is None else np.asarray(floor)[keep]
    st = image_diff_stats(got, ref)
    fl = image_diff_stats(floor, ref) if floor is not None else None
    print(f"[parity] {what}: {st}" + (f" | fp32 floor (oracle fma vs ieee): {fl}" if fl else ""))
    for key, base in (("frac_bad", frac_bad), ("mean_rel", mean_rel), ("sum_rel", sum_rel)):
        assert st[key] <= _bound(key, base, fl, ref.size, "mean_rel"), (what, key, st, fl)
    return st


def grad_diff_stats(got, ref):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    scale = np.abs(ref).max()
    return {
        "frac_bad": float((d > 1e-4 * scale + 1e-3 * np.abs(ref)).mean()),
        "rel_l1": float(d.sum() / max(np.abs(ref).sum(), 1e-30)),
        "sum_rel": float(abs(got.sum() - ref.sum()) / max(abs(ref.sum()), 1e-30)),
        "nnz_got": int((got != 0).sum()), "nnz_ref": int((ref != 0).sum()),
    }


def assert_grad_parity(got, ref, what, floor=None, frac_bad=2e-3, rel_l1=2e-4, sum_rel=1e-4, flips=None):
    """Stated fp32 tolerance of the gradient texture: float atomics accumulate in arrival order (the
    oracle sums in float64).  `flips` (a Flips of the BACKWARD pass's paths, i.e. seed + 1 and its cotangent): the texel
    footprints of every vertex of the measurably flipped paths are left out.  `floor` as in assert_image_parity."""
    got, ref = np.asarray(got), np.asarray(ref)
    if flips is not None:
        flips.check_count(what)
        keep = ~flips.texels
        got, ref = got[keep], ref[keep]
        floor = None if floor is None else np.asarray(floor)[keep]
    st = grad_diff_stats(got, ref)
    fl = grad_diff_stats(floor, ref) if floor is not None else None
    print(f"[parity] {what}: {st}" + (f" | fp32 floor (oracle fma vs ieee): {fl}" if fl else ""))
    for key, base in (("frac_bad", frac_bad), ("rel_l1", rel_l1), ("sum_rel", sum_rel)):
        assert st[key] <= _bound(key, base, fl, ref.size, "rel_l1"), (what, key, st, fl)
    return st


def terrain_arrays(n=64, seed=0, light=True):
    """Procedural BVH-stress scene: an n x n displaced height-field (2 n^2 triangles, smooth normals,
    UV atlas = the unit square) lit by a quad light above it; instance 0 textured, instance 1 light."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-3.0, 3.0, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(g, g, indexing="xy")
    Y = (0.35 * np.sin(1.7 * X) * np.cos(1.3 * Z) + 0.05 * rng.standard_normal(X.shape)).astype(np.float32)
    P = np.stack([X, Y, Z], -1).reshape(-1, 3)
    U = np.stack([(X + 3) / 6, (Z + 3) / 6], -1).reshape(-1, 2).astype(np.float32)
    idx = lambda i, j: i * (n + 1) + j
    tris = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = idx(i, j), idx(i, j + 1), idx(i + 1, j + 1), idx(i + 1, j)
            tris += [(a, c, b), (a, d, c)]          # wound so that normals point +y
    tris = np.asarray(tris, np.int32)
    verts = np.zeros((P.shape[0], 8), np.float32)
    verts[:, 0:3] = P; verts[:, 3:5] = U; verts[:, 5:8] = np.nan
    geometry.recompute_normal(verts, tris)
    lv = np.array([[-1, 4, -1, 0, 0, 0, -1, 0], [1, 4, -1, 0, 0, 0, -1, 0], [1, 4, 1, 0, 0, 0, -1, 0], [-1, 4, 1, 0, 0, 0, -1, 0]], np.float32)
    lt = np.array([[0, 1, 2], [0, 2, 3]], np.int32) + verts.shape[0]
    V = np.concatenate([verts, lv]); T = np.concatenate([tris, lt])
    em = np.array([[0, 0, 0], [30, 30, 30]], np.float32) if light else np.zeros((2, 3), np.float32)
    return geometry.from_arrays(V, T, [0, tris.shape[0], T.shape[0]], None, em)


TERRAIN_CAMERA = Camera(fov=0.9, origin=float3(0.5, 3.0, 6.5), target=float3(0.0, 0.0, 0.0), up=float3(0.0, 1.0, 0.0))


def random_rays(n, lo, hi, seed=0, tmax=1e30):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = o; r[:, 3] = 0.0; r[:, 4:7] = d; r[:, 7] = tmax
    return r


def panel_mesh(nx, ny):
    """A 2 x 2 panel in the xz plane facing +y (like quad.obj), cut into nx x ny cells of two triangles each:
    2 nx ny triangles, per-vertex normal (0, 1, 0), uv = the unit square."""
    gx = np.linspace(-1.0, 1.0, nx + 1, dtype=np.float32); gz = np.linspace(-1.0, 1.0, ny + 1, dtype=np.float32)
    verts = np.zeros(((nx + 1) * (ny + 1), 8), np.float32)
    k = 0
    for j in range(ny + 1):
        for i in range(nx + 1):
            verts[k] = (gx[i], 0.0, gz[j], (gx[i] + 1) / 2, (gz[j] + 1) / 2, 0.0, 1.0, 0.0); k += 1
    tris = []
    for j in range(ny):
        for i in range(nx):
            a, b, c, d = j * (nx + 1) + i, j * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i
            tris += [(a, c, b), (a, d, c)]          # wound so that cross(p1 - p0, p2 - p0) points +y
    return verts, np.asarray(tris, np.int32)


def multi_light_arrays(emissions=((0, 0, 0), (20, 20, 20), (6, 2, 1), (0, 0, 0), (1, 3, 8))):
    """Light-stage style scene (test_lightstage.py:24-62): the Cornell box (instance 0, textured) with FOUR more
    instances of DIFFERENT triangle counts — the ceiling light (2 triangles), a warm panel on the left wall (8),
    a blocker slab in mid-air (2, never emits: the light list must skip it) and a cool panel on the back wall (6),
    each with its own transform.  `emissions` has one rgb per instance."""
    from zdr_amd.mathtypes import as_row_major_4x4
    base = geometry.assemble(cbox_models())
    V = [base.verts]; T = [base.tris]; begin = list(base.inst_tri_begin); X = [base.inst_xform[0], base.inst_xform[1]]
    def add(verts, tris, xform):
        nv = sum(v.shape[0] for v in V)
        V.append(verts); T.append(tris + nv); begin.append(begin[-1] + tris.shape[0]); X.append(np.asarray(xform, np.float32).reshape(16))
    # left wall x = -3.0: panel normal +y -> +x (rotate about z by -90 degrees), scaled 0.5 x 1.0
    add(*panel_mesh(2, 2), [[0, 1, 0, -2.9], [-1.0, 0, 0, 2.5], [0, 0, 0.5, -3.0], [0, 0, 0, 1]])
    # blocker: a slab facing down, hanging under the ceiling light
    add(*panel_mesh(1, 1), [[0.6, 0, 0, -0.2], [0, -1, 0, 4.2], [0, 0, -0.6, -3.0], [0, 0, 0, 1]])
    # back wall z = -5.8: panel normal +y -> +z, 3 x 1 cells
    add(*panel_mesh(3, 1), [[0.9, 0, 0, 0.4], [0, 0, -0.4, 1.6], [0, 1, 0, -5.7], [0, 0, 0, 1]])
    return geometry.from_arrays(np.concatenate(V), np.concatenate(T), begin, np.stack(X), np.asarray(emissions, np.float32))
