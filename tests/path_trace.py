"""Per-path traces (zdr_path_dump / zdro_path_dump): decoding, the discrete signature of a path, and the sums that
tie the traces back to the rendered image and gradient texture."""
import numpy as np

HDR, VSTRIDE = 8, 24


def all_queries(W, H, spp, x0=0, y0=0):
    y, x, s = np.meshgrid(np.arange(y0, y0 + H), np.arange(x0, x0 + W), np.arange(spp), indexing="ij")
    return np.stack([x.ravel(), y.ravel(), s.ravel()], 1).astype(np.int32)


class Trace:
    """View of an (n, 8 + 24 maxv) dump."""

    def __init__(self, raw):
        raw = np.ascontiguousarray(raw, np.float32)
        self.raw = raw
        self.n = raw.shape[0]
        self.maxv = (raw.shape[1] - HDR) // VSTRIDE
        self.nvert = raw[:, 0].view(np.int32).copy()
        self.L = raw[:, 1:4]
        self.term_Li = raw[:, 5:8]
        v = raw[:, HDR:].reshape(self.n, self.maxv, VSTRIDE)
        self.inst = v[..., 0].view(np.int32)
        self.prim = v[..., 1].view(np.int32)
        self.uv = v[..., 2:4]
        self.flags = v[..., 4].view(np.int32)
        self.pdf = v[..., 5]
        self.wi = v[..., 6:9]
        self.beta_out = v[..., 9:12]
        self.grad = v[..., 12:16]
        self.L_nee = v[..., 16:19]
        # decisions that matter: a light sample that was accepted but carries no radiance (a light seen from behind or
        # edge-on: eval = 0, light.py:76) is the same event as a rejected one — on the Cornell box every ceiling vertex sees
        # the ceiling light at a grazing angle and `wi.z >= 1e-4` (prb.py:62) is a coin toss there, with nothing at stake
        self.decisions = (self.flags & ~1) | (self.L_nee != 0).any(axis=2).astype(np.int32)
        k = np.arange(self.maxv)[None, :]
        self.live = k < np.minimum(self.nvert, self.maxv)[:, None]     # (n, maxv): vertex k exists

    def signature_equal(self, other, uv_tol=None):
        """Paths whose every discrete decision agrees: number of vertices, the triangle hit at each vertex, whether the
        light sample contributed, whether the path went on, and the kind of Russian-roulette event.
        uv_tol (finely tessellated scenes only): a vertex that lands on ANOTHER triangle of the same instance within uv_tol of the
        other build's texture coordinates — across a shared edge of a smooth surface, where material and shading normal are
        continuous — is the same decision, not a branch; its values are then COMPARED instead of set aside."""
        same = self.nvert == other.nvert
        both = self.live & other.live
        where = self.prim == other.prim
        if uv_tol is not None:
            where = where | (np.abs(self.uv - other.uv).max(axis=2) <= uv_tol)
        ok = (self.inst == other.inst) & where & (self.decisions == other.decisions)
        return same & np.all(ok | ~both, axis=1)


def image_from_paths(tr, queries, W, H, spp):
    """integrator.py:26-29: NaN samples dropped, radiance clamped to [0, 1e5], mean over spp (float64 sums)."""
    L = tr.L.astype(np.float64)
    good = ~np.isnan(L).any(axis=1)
    c = np.clip(np.where(good[:, None], L, 0.0), 0.0, 100000.0)
    img = np.zeros((H, W, 3))
    np.add.at(img, (queries[:, 1], queries[:, 0]), c)
    return img / spp


def scatter_gradients(tr, tex_h, tex_w):
    """interaction.py:73-89 applied to every dumped vertex gradient (NaN / all-zero gradients skipped, prb.py:178-187)."""
    g = tr.grad.reshape(-1, 4).astype(np.float64)
    uv = tr.uv.reshape(-1, 2)
    keep = tr.live.reshape(-1) & ~np.isnan(g).any(axis=1) & (g != 0).any(axis=1)
    g, uv = g[keep], uv[keep]
    px = uv[:, 0] * np.float32(tex_w - 1); py = (np.float32(1.0) - uv[:, 1]) * np.float32(tex_h - 1)
    ix = px.astype(np.int32); iy = py.astype(np.int32)
    ox = (px - ix.astype(np.float32)).astype(np.float64); oy = (py - iy.astype(np.float32)).astype(np.float64)
    out = np.zeros((tex_h, tex_w, 4))
    for dx, dy, k in ((0, 0, (1 - ox) * (1 - oy)), (0, 1, (1 - ox) * oy), (1, 0, ox * (1 - oy)), (1, 1, ox * oy)):
        x = np.clip(ix + dx, 0, tex_w - 1); y = np.clip(iy + dy, 0, tex_h - 1)
        np.add.at(out, (y, x), k[:, None] * g)
    return out


def deviation_percentiles(got, ref, pct=(50, 90, 99, 100), uv_tol=None):
    """Per path, over the paths whose signatures agree: the largest deviation of the radiance (relative to the path's own
    radiance) and of the vertex gradients (relative to the path's largest gradient component)."""
    same = got.signature_equal(ref, uv_tol)
    finite = same & ~np.isnan(ref.L).any(axis=1) & ~np.isnan(got.L).any(axis=1)
    a, b = got.L[finite].astype(np.float64), ref.L[finite].astype(np.float64)
    lit = np.linalg.norm(b, axis=1) > 0
    dL = np.abs(a - b).max(axis=1)[lit] / np.linalg.norm(b, axis=1)[lit]
    ga, gb = got.grad[finite].astype(np.float64), ref.grad[finite].astype(np.float64)
    ga = np.where(got.live[finite][..., None], ga, 0.0); gb = np.where(ref.live[finite][..., None], gb, 0.0)
    mag = np.abs(gb).max(axis=(1, 2))
    has = mag > 0
    dG = np.abs(ga - gb).max(axis=(1, 2))[has] / mag[has]
    return {"paths": int(got.n), "flipped": int((~same).sum()),
            "L": {int(p): float(np.percentile(dL, p)) for p in pct} if dL.size else {},
            "grad": {int(p): float(np.percentile(dG, p)) for p in pct} if dG.size else {}}
