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
# (tests/measure/glossy_floor.py, profiles/r3_glossy_floor_ratio.txt: 16 seeds x 4 scenes, flipped paths of all builds set aside): on renders
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

    def __init__(self, scene, S, Sf, mat, res, spp, seed, cot=None, what="", traces=None, uv_tol=None, **params_kw):
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
        flipped = ~hip.signature_equal(ref, uv_tol)            # uv_tol: path_trace.Trace.signature_equal (the 1 M-triangle scene only)
        flipped_floor = ~fma.signature_equal(ref, uv_tol)
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


from zdr_amd.scenes import multi_light_arrays, panel_mesh, random_rays, terrain_arrays, terrain_camera  # noqa: E402,F401  (scene builders live in the product package; re-exported)

TERRAIN_CAMERA = terrain_camera()
