"""Multi-GPU rendering: one process per GPU, shards of one render, one exchange step.

Samples are independent given (pixel, sample_index, seed) (integrator.py:17), so a render
shards without any data-path communication; the only exchange is the final sum of the image
and of the material gradient — an all_reduce over RCCL/xGMI (``backend="nccl"`` on
ROCm).  Payloads are small (16 MiB each at 1024^2), far below what the render itself costs.

Shard modes
  "tiles"    pixel tiles: the 8x8 tiles of the image dealt round-robin to the ranks along diagonals, ONE launch per rank
             (zdr_render_params.tile_shard_*); every pixel receives the same samples whoever renders it: the union equals the
             unsharded image bit for bit when both cut the sample range into the same chunks, otherwise up to float
             re-association of the per-pixel sum.  BASELINE configs[3].
  "rows"     pixel tiles: interleaved bands of rows (one launch per band); same samples per pixel as well
  "samples"  sample-index ranges [k*spp/N, (k+1)*spp/N) of the same sample set
  "seeds"    every rank renders the whole image with its own seed (seed + rank * SEED_STRIDE); the
             mean over ranks is an N*spp-sample estimate (weak scaling: fixed work per GPU)

The local renderer is injected (``local_forward`` / ``local_backward``) so the collective logic
is testable on CPU tensors over gloo; ``attach(scene)`` binds it to a zdr_amd.Scene.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

SEED_STRIDE = 0x9E3779B1      # odd constant: distinct sample sets per rank in "seeds" mode
BAND_ROWS = 32


@dataclass
class Shard:
    rects: List[Tuple[int, int, int, int]]        # pixel rectangles (x0, y0, x1, y1)
    samples: Tuple[int, int]                      # sample-index range
    seed: int
    scale: float                                  # factor applied after the sum over ranks
    tile_shard: Optional[Tuple[int, int]] = None  # (index, count): interleaved 8x8 tiles of every rectangle


def plan(mode: str, rank: int, world: int, res, spp: int, seed: int) -> Shard:
    W, H = int(res[0]), int(res[1])
    if mode == "tiles":
        return Shard([(0, 0, W, H)], (0, spp), seed, 1.0, (rank, world) if world > 1 else None)
    if mode == "rows":
        rects = [(0, y, W, min(y + BAND_ROWS, H)) for i, y in enumerate(range(0, H, BAND_ROWS)) if i % world == rank]
        return Shard(rects, (0, spp), seed, 1.0)
    if mode == "samples":
        b, e = (spp * rank) // world, (spp * (rank + 1)) // world
        return Shard([(0, 0, W, H)], (b, e), seed, 1.0)
    if mode == "seeds":
        return Shard([(0, 0, W, H)], (0, spp), (seed + rank * SEED_STRIDE) & 0xFFFFFFFF, 1.0 / world)
    raise KeyError(mode)


def shard_tiles(rect, index: int, count: int) -> List[Tuple[int, int, int, int]]:
    """The 8x8 tiles (as pixel rectangles) that tile shard (index, count) of ``rect`` owns, in launch order — the
    numbering of include/zdr.h and decode_item (csrc/zdr_kernels.hip): tiles are numbered row by row from the rectangle's
    own corner, row r starting at column r (mod the row length) when count > 1, and the shard owns the numbers
    index, index + count, ...  Host-side mirror for tests and rehearsals; the kernels decode the same mapping themselves."""
    x0, y0, x1, y1 = rect
    tiles_x, tiles_y = (x1 - x0 + 7) // 8, (y1 - y0 + 7) // 8
    skew = 1 if count > 1 else 0
    out = []
    for number in range(index if count > 1 else 0, tiles_x * tiles_y, max(count, 1)):
        ty = number // tiles_x
        tx = (number % tiles_x + ty * skew) % tiles_x
        out.append((x0 + 8 * tx, y0 + 8 * ty, min(x0 + 8 * tx + 8, x1), min(y0 + 8 * ty + 8, y1)))
    return out


class ShardedRenderer:
    """local_forward(material, res, spp, seed, rect, samples, out, tile_shard) -> image (writes the shard into out)
    local_backward(grad_output, d_material, material, res, spp, seed, rect, samples, tile_shard) accumulates."""

    def __init__(self, local_forward: Callable, local_backward: Callable, mode: str = "tiles", group=None):
        self.local_forward, self.local_backward, self.mode, self.group = local_forward, local_backward, mode, group
        self.reduce_events = None      # a list: time every exchange with HIP events (bench.py); None: don't

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def _reduce(self, t: torch.Tensor, scale: float, what: str = "image") -> torch.Tensor:
        """The one exchange step of a pass.  With ``reduce_events`` set, HIP events on the compute stream bracket it: a
        synchronous collective makes the compute stream wait for the communicator's stream, so the pair measures the
        all-reduce including the wait for the slowest rank."""
        if self.world > 1:
            timed = self.reduce_events is not None and t.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            if timed:
                e1.record()
                self.reduce_events.append((what, e0, e1))
        if scale != 1.0:
            t.mul_(scale)
        return t

    def reduce_ms(self) -> dict:
        """Total milliseconds the recorded exchanges took, by kind (call after a synchronise)."""
        out = {"image": 0.0, "gradient": 0.0}
        for what, e0, e1 in self.reduce_events or []:
            out[what] += e0.elapsed_time(e1)
        return out

    def forward(self, material, res, spp, seed):
        sh = plan(self.mode, self.rank, self.world, res, spp, seed)
        image = torch.zeros((res[1], res[0], 4), dtype=torch.float32, device=material.device)
        for rect in sh.rects:
            self.local_forward(material, res, spp, sh.seed, rect, sh.samples, image, sh.tile_shard)
        return self._reduce(image, sh.scale, "image")

    def backward(self, grad_output, material, res, spp, seed):
        # grad_output is the cotangent of the REDUCED image and is identical on every rank
        sh = plan(self.mode, self.rank, self.world, res, spp, seed)
        d_material = torch.zeros_like(material)
        for rect in sh.rects:
            self.local_backward(grad_output, d_material, material, res, spp, sh.seed, rect, sh.samples, sh.tile_shard)
        return self._reduce(d_material, sh.scale, "gradient")

    def render(self, material, *, res, spp, seed=0):
        return _ShardedOp.apply(material, self, tuple(res), int(spp), int(seed))


class _ShardedOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, material, renderer, res, spp, seed):
        ctx.save_for_backward(material)
        ctx.renderer, ctx.args = renderer, (res, spp, seed)
        return renderer.forward(material.detach(), res, spp, seed)

    @staticmethod
    def backward(ctx, grad_output):
        material, = ctx.saved_tensors
        res, spp, seed = ctx.args
        return ctx.renderer.backward(grad_output.contiguous(), material.detach(), res, spp, seed), None, None, None, None


def attach(scene, mode: str = "tiles", group=None) -> ShardedRenderer:
    """Shard the renders of a zdr_amd.Scene over the ranks of ``group``."""
    def fwd(material, res, spp, seed, rect, samples, out, tile_shard=None):
        return scene.render_forward(material, res, spp, seed, rect=rect, samples=samples, out=out, tile_shard=tile_shard)

    def bwd(grad_output, d_material, material, res, spp, seed, rect, samples, tile_shard=None):
        return scene.render_backward(grad_output, d_material, material, res, spp, seed, rect=rect, samples=samples, tile_shard=tile_shard)

    return ShardedRenderer(fwd, bwd, mode, group)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """torchrun-style bootstrap: returns (rank, world, local_rank); no-op for a single process."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a single-GPU box: ZDR_DIST_BACKEND=gloo ZDR_SHARE_DEVICE=1 puts every rank on cuda:0
    backend = backend or os.environ.get("ZDR_DIST_BACKEND")
    if os.environ.get("ZDR_SHARE_DEVICE") == "1":
        local = 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():        # one GPU per rank, whatever the backend (RCCL requires it; gloo stages through the host)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
