"""Python API of the renderer: ``Scene``, ``Camera``, ``float3`` — the reference's public surface
(/root/reference/render.py:31-257, __init__.py:1) on top of libzdr_hip.so.

    scene = Scene([(obj_path, transform_or_None, emission), ...], integrator="path")
    scene.camera = Camera(fov=..., origin=float3(...), target=float3(...), up=float3(...))
    image = scene.render(material, res=(W, H), spp=256, seed=0)      # (H, W, 4) float32, differentiable
    image.sum().backward()                                            # material.grad: (Ht, Wt, 4)

Images and materials are PyTorch tensors on the GPU; the renderer borrows their device pointers
for the duration of a call and enqueues its kernels on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np
import torch

from . import _native as N
from .geometry import SceneArrays, assemble, normalize_emission
from .mathtypes import Camera, float3, float4x4  # noqa: F401  (re-exported)

MAX_DEPTH = 16      # prb.py:15
RR_DEPTH = 2        # prb.py:16


def _camera_pod(cam: Camera) -> N.CameraPOD:
    return N.CameraPOD(float(cam.fov), (C.c_float * 3)(*cam.origin), (C.c_float * 3)(*cam.target), (C.c_float * 3)(*cam.up))


class Scene:
    """A 3D scene for differentiable rendering w.r.t. one (H, W, 4) material texture
    (diffuse rgb + roughness; specular fixed at 0.04).  Only the first model is textured; any
    other model is a light (emission > 0) or a blocker (render.py:31-71, prb.py:45).

    Attributes:
        camera (Camera): fov (full horizontal angle, radians), origin, target, up.
        use_tent_filter (bool): tent reconstruction filter if True (default), box filter if False.
        sampler (str): "cmj" (correlated multi-jitter, corrmj.py — default here) or "pmj02bn"
            (needs tables, see ``set_pmj02bn_tables``; the reference's tables are not shipped).
    """

    def __init__(self, models, integrator="direct", *, device=None, accel="auto", sampler="cmj"):
        integrators = {"path": N.PATH, "direct": N.DIRECT, "collocated": N.COLLOCATED}
        self._integrator = integrators[integrator]          # KeyError on unknown names, as render.py:70
        self.integrator = integrator
        if not torch.cuda.is_available():
            raise N.ZdrError("zdr_amd needs an AMD GPU (HIP device); there is no CPU back end")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        self.camera = Camera(fov=40 / 180 * 3.1415926, origin=float3(1.0, 0.5, 0.0), target=float3(0.0, 0.0, 0.0), up=float3(0.0, 1.0, 0.0))
        self.use_tent_filter = True
        self.sampler = sampler
        self.max_depth = MAX_DEPTH
        self.rr_depth = RR_DEPTH
        self.prb_mode = "expectation"      # or "detached": the reference's constant-roulette / constant-MIS adjoint; "literal": with the BSDF-sample seed of prb.py:162 as written (include/zdr.h)
        self.env_count = 0
        self._handle = None
        self.load_geometry(models, accel=accel)

    # ------------------------------------------------------------------ geometry / lights
    def load_geometry(self, models, accel="auto"):
        arrays = models if isinstance(models, SceneArrays) else assemble(models)
        self._arrays = arrays
        self.inst_count = arrays.ninst
        self.emissions = [float3(*e) for e in arrays.inst_emission.tolist()]
        self.light_count = int((arrays.inst_emission > 0).any(axis=1).sum())
        h = C.c_void_p()
        L = N.lib()
        if getattr(self, "_finalizer", None) is not None:   # a second load_geometry: the old handle (and the tables set on it) go
            self._finalizer()
            self._handle, self.env_count, self._pmj_tables_set = None, 0, False
        N.check(L.zdr_scene_create(arrays.verts.ctypes.data, arrays.verts.shape[0], arrays.tris.ctypes.data, arrays.tris.shape[0],
                                   arrays.inst_tri_begin.ctypes.data, arrays.inst_xform.ctypes.data, arrays.inst_emission.ctypes.data,
                                   arrays.ninst, self.device.index, N.ACCELS[accel], C.byref(h)))
        self._handle = h
        self._finalizer = weakref.finalize(self, L.zdr_scene_destroy, h)

    def info(self) -> dict:
        i = N.SceneInfo()
        N.check(N.lib().zdr_scene_info(self._handle, C.byref(i)))
        d = {k: getattr(i, k) for k, _ in N.SceneInfo._fields_}
        d["accel"] = {N.ACCEL_BRUTE: "brute", N.ACCEL_BVH: "bvh"}[i.accel]
        return d

    def update_lights(self, emissions):
        """Rewrite the emission of each mesh in the scene (light-stage style switching);
        ``emissions`` has one entry per model: None, a number or a float3 (render.py:130-148)."""
        assert len(emissions) == self.inst_count
        self.emissions = emissions
        e = np.ascontiguousarray(np.stack([normalize_emission(x) for x in emissions]), np.float32)
        self.light_count = int((e > 0).any(axis=1).sum())
        N.check(N.lib().zdr_scene_set_emissions(self._handle, e.ctypes.data, self._stream()))

    def add_envmap(self, image, compensate_mis=True):
        """Adds a lat-long environment light (render.py:150-156, envmap.py:116-203).  ``image`` is an
        (H, W, 3|4) float array / tensor (2:1 or 1:1) or the path of an OpenEXR file (the reference reads it
        through imageio; here zdr_amd/exr.py: scan-line files, NONE / RLE / ZIPS / ZIP / PIZ compression) or of a ``.npy``
        file.  ``None`` removes it."""
        from . import envmap as E
        if image is None:
            N.check(N.lib().zdr_scene_set_envmap(self._handle, None, 0, 0, None, None, None, 0, 0))
            self.env_count = 0
            return
        if isinstance(image, str):
            image = E.load_image(image)               # .exr (zdr_amd/exr.py) or .npy
        if isinstance(image, torch.Tensor):
            image = image.detach().cpu().numpy()
        img = E.prepare_image(image)
        prob, alias, pdf = E.build_tables(img, compensate_mis=compensate_mis)
        N.check(N.lib().zdr_scene_set_envmap(self._handle, img.ctypes.data, img.shape[0], img.shape[1], prob.ctypes.data, alias.ctypes.data,
                                             pdf.ctypes.data, E.SAMPLE_MAP_W, E.SAMPLE_MAP_H))
        self.env_count = 1
        self._envmap = (img, prob, alias, pdf)        # kept for tests / the oracle

    def set_pmj02bn_tables(self, pmj_samples, blue_noise):
        """pmj_samples: uint32 [nsets][nsamples][2]; blue_noise: uint16 [ntex][res][res] (pmj02bn.py:9-18)."""
        pmj = np.ascontiguousarray(pmj_samples, np.uint32)
        bn = np.ascontiguousarray(blue_noise, np.uint16)
        assert pmj.ndim == 3 and pmj.shape[2] == 2 and bn.ndim == 3 and bn.shape[1] == bn.shape[2]
        N.check(N.lib().zdr_scene_set_pmj02bn_tables(self._handle, pmj.ctypes.data, pmj.shape[0], pmj.shape[1], bn.ctypes.data, bn.shape[0], bn.shape[1]))
        self._pmj_tables_set = True

    # ------------------------------------------------------------------------- launching
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _params(self, res, spp, seed, tex_hw, rect=None, samples=None, camera=None, integrator=None, tile_shard=None) -> N.RenderParams:
        if self.sampler == "pmj02bn" and not getattr(self, "_pmj_tables_set", False):
            # the reference's pbrt tables are not shipped: fall back to generated ones (zdr_amd/pmj02bn_tables.py)
            from . import pmj02bn_tables
            self.set_pmj02bn_tables(*pmj02bn_tables.default_tables(verbose=True))
        p = N.RenderParams()
        p.struct_size = C.sizeof(N.RenderParams)
        p.integrator, p.sampler = self._integrator if integrator is None else integrator, N.SAMPLERS[self.sampler]
        p.width, p.height = int(res[0]), int(res[1])
        p.spp, p.seed = int(spp), int(seed) & 0xFFFFFFFF            # seeds are uint32 (App. B-15)
        p.use_tent = int(bool(self.use_tent_filter))
        p.x0, p.y0, p.x1, p.y1 = rect if rect is not None else (0, 0, p.width, p.height)
        p.sample_begin, p.sample_end = samples if samples is not None else (0, p.spp)
        p.max_depth, p.rr_depth = int(self.max_depth), int(self.rr_depth)
        p.camera = _camera_pod(camera if camera is not None else self.camera)
        p.tex_h, p.tex_w = int(tex_hw[0]), int(tex_hw[1])
        p.tile_shard_index, p.tile_shard_count = tile_shard if tile_shard is not None else (0, 1)
        p.prb_mode = N.PRB_MODES[self.prb_mode]
        return p

    def _check_material(self, material):
        assert material.ndim == 3 and material.shape[2] == 4           # render.py:160,177
        if material.device != self.device or material.dtype != torch.float32:
            raise ValueError(f"material must be a float32 tensor on {self.device}")

    def render_forward(self, material, res, spp, seed, *, rect=None, samples=None, out=None, kernel=None, tile_shard=None):
        """render.py:159-173.  Returns the (H, W, 4) image; with ``rect``/``samples``/``tile_shard`` only that shard
        is written (other pixels of ``out`` keep their value; a fresh image is zero-filled).  ``tile_shard`` =
        (index, count): the 8x8 tiles of the rectangle numbered index, index + count, ... (include/zdr.h)."""
        self._check_material(material)
        material = material.detach().contiguous()
        if out is None:   # zero-filled even when the call covers every pixel: a dropped work item must never surface as uninitialised memory
            image = torch.zeros((res[1], res[0], 4), dtype=torch.float32, device=self.device)
        else:
            image = out
            if image.shape != (res[1], res[0], 4) or image.dtype != torch.float32 or image.device != self.device or not image.is_contiguous():
                raise ValueError(f"out must be a contiguous float32 ({res[1]}, {res[0]}, 4) tensor on {self.device}")
        p = self._params(res, spp, seed, material.shape[0:2], rect, samples, integrator=kernel, tile_shard=tile_shard)
        N.check(N.lib().zdr_render_forward(self._handle, C.byref(p), material.data_ptr(), image.data_ptr(), self._stream()))
        return image

    def render_backward(self, grad_output, d_material, material, res, spp, seed, *, rect=None, samples=None, camera=None, tile_shard=None):
        """render.py:176-199: accumulates into ``d_material``; uses ``seed + 1`` like the reference (:196)."""
        self._check_material(material)
        material = material.detach().contiguous()
        g = grad_output.reshape(res[1], res[0], 4).to(device=self.device, dtype=torch.float32).contiguous()
        assert d_material.is_contiguous() and d_material.shape == material.shape
        if d_material.device != self.device or d_material.dtype != torch.float32:
            raise ValueError(f"d_material must be a float32 tensor on {self.device}")
        p = self._params(res, spp, seed + 1, material.shape[0:2], rect, samples, camera, tile_shard=tile_shard)
        N.check(N.lib().zdr_render_backward(self._handle, C.byref(p), g.data_ptr(), material.data_ptr(), d_material.data_ptr(), self._stream()))
        return d_material, None, None, None, None

    def render_stats(self, material, res, spp, seed=0, *, rect=None, samples=None, tile_shard=None) -> dict:
        """Path statistics of one forward pass (camera samples, rays, shaded vertices ...), SURVEY §8d."""
        self._check_material(material)
        material = material.detach().contiguous()
        p = self._params(res, spp, seed, material.shape[0:2], rect, samples, tile_shard=tile_shard)
        cnt = (C.c_uint64 * 8)()
        N.check(N.lib().zdr_render_stats(self._handle, C.byref(p), material.data_ptr(), cnt, self._stream()))
        return dict(zip(N.COUNTER_NAMES, list(cnt)))

    class RenderOperator(torch.autograd.Function):     # render.py:201-223
        @staticmethod
        def forward(ctx, material, self, *args):
            ctx.save_for_backward(material)
            ctx.scene = weakref.ref(self)
            ctx.args = args
            ctx.camera = self.camera.copy()
            ctx.emissions = self.emissions
            return self.render_forward(material.detach(), *args)

        @staticmethod
        def backward(ctx, grad_output):
            scene = ctx.scene()
            # scene.camera / lights may have changed between forward and backward: replay the
            # snapshot (camera restored afterwards, lights left at the snapshot — render.py:216-222)
            if scene.emissions is not ctx.emissions:
                scene.update_lights(ctx.emissions)
            material, = ctx.saved_tensors
            mat_grad = torch.zeros(material.size(), dtype=material.dtype, device=material.device)
            res, spp, seed = ctx.args
            return scene.render_backward(grad_output, mat_grad, material.detach(), res, spp, seed, camera=ctx.camera)

    def render(self, material, *, res, spp, seed=0):
        """Renders the scene; differentiable w.r.t. ``material`` ((Ht, Wt, 4) float32 on the GPU).
        res = (width, height); returns (height, width, 4) (render.py:225-241)."""
        return Scene.RenderOperator.apply(material, self, res, spp, seed)

    def render_duvdxy(self, material, *, res, spp, seed=0):
        """Gradient of the texture coordinates w.r.t. screen-space coordinates: a (height, width, 4) tensor
        holding (dudx, dvdx, dudy, dvdy) (render.py:243-257, uvgrad.py).  Not differentiable.  The
        reference drives this kernel with LuisaCompute's own random sampler (uvgrad.py:82, third-party);
        here the scene's sampler provides the pixel jitter."""
        return self.render_forward(material.detach(), res, spp, seed, kernel=N.UVGRAD)

    def check(self):
        """Synchronises and raises ZdrError if a device watchdog ended work early since the last check
        (include/zdr.h, zdr_scene_check).  LuisaCompute raises from luisa.synchronize() (render.py:172,198)."""
        N.check(N.lib().zdr_scene_check(self._handle, self._stream()))

    # ------------------------------------------------------------------- test / debug hooks
    def _rays(self, rays):
        assert rays.ndim == 2 and rays.shape[1] == 8
        return rays.to(device=self.device, dtype=torch.float32).contiguous()

    def trace_closest(self, rays):
        """rays: (n, 8) float32 cuda {o, tmin, d, tmax} -> (inst_prim (n,2) int32, bary_t (n,3) float32)."""
        rays = self._rays(rays)
        n = rays.shape[0]
        ip = torch.empty((n, 2), dtype=torch.int32, device=self.device)
        bt = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        N.check(N.lib().zdr_trace_closest(self._handle, rays.data_ptr(), n, ip.data_ptr(), bt.data_ptr(), self._stream()))
        return ip, bt

    def trace_any(self, rays):
        rays = self._rays(rays)
        occ = torch.empty((rays.shape[0],), dtype=torch.int32, device=self.device)
        N.check(N.lib().zdr_trace_any(self._handle, rays.data_ptr(), rays.shape[0], occ.data_ptr(), self._stream()))
        return occ

    def sampler_dump(self, queries, spp, seed=0, nvert=3, rr_depth=RR_DEPTH):
        """queries: (n, 3) int32 cuda {px, py, sample_index} -> (n, 2 + 8*nvert) float32 sampler draws."""
        q = queries.reshape(-1, 3).to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty((q.shape[0], 2 + 8 * nvert), dtype=torch.float32, device=self.device)
        N.check(N.lib().zdr_sampler_dump(self._handle, N.SAMPLERS[self.sampler], int(seed) & 0xFFFFFFFF, int(spp), q.data_ptr(), q.shape[0], nvert, rr_depth, out.data_ptr(), self._stream()))
        return out

    def vertex_sampler_dump(self, queries, spp, seed=0, nvert=3, rr_depth=RR_DEPTH, integrator="path"):
        """As sampler_dump, but drawn the way the path (or the direct) kernels draw (include/zdr.h, zdr_vertex_sampler_dump).
        Returns (draws, batched): batched is True when the packed two-permutations-per-register route ran."""
        q = queries.reshape(-1, 3).to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty((q.shape[0], 2 + 8 * nvert), dtype=torch.float32, device=self.device)
        b = C.c_int32(-1)
        N.check(N.lib().zdr_vertex_sampler_dump(self._handle, N.INTEGRATORS[integrator], N.SAMPLERS[self.sampler], int(seed) & 0xFFFFFFFF, int(spp), q.data_ptr(), q.shape[0], nvert, rr_depth, out.data_ptr(), C.byref(b), self._stream()))
        return out, bool(b.value)

    def path_dump(self, material, queries, res, spp, seed, *, d_image=None, maxv=16):
        """Per-path traces (include/zdr.h, zdr_path_dump): queries (n, 3) int32 cuda {px, py, sample_index} ->
        (n, 8 + 24 maxv) float32.  ``seed`` is used as it is (pass seed + 1 for the paths of a backward pass)."""
        self._check_material(material)
        material = material.detach().contiguous()
        q = queries.reshape(-1, 3).to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty((q.shape[0], 8 + 24 * maxv), dtype=torch.float32, device=self.device)
        p = self._params(res, spp, seed, material.shape[0:2])
        g = None if d_image is None else d_image.reshape(res[1], res[0], 4).to(device=self.device, dtype=torch.float32).contiguous()
        N.check(N.lib().zdr_path_dump(self._handle, C.byref(p), material.data_ptr(), None if g is None else g.data_ptr(), q.data_ptr(), q.shape[0], maxv, out.data_ptr(), self._stream()))
        return out
