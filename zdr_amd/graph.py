"""A forward + backward pair of one view, captured once in a HIP graph and replayed (torch.cuda.CUDAGraph on ROCm).

libzdr_hip.so only enqueues on the stream it is given and allocates nothing after the first call of a kind, so a forward + backward
pair (about eight launches) can be captured; a replay is ONE launch from the host.  What it buys is a free host thread, not GPU time:
measured on one MI355X at 64 x 64 x 16 spp, eager 0.667 ms and captured 0.676 ms per step — even that small a pass is bound by the
kernels themselves (tests/test_zz_gpu_graph.py::test_captured_render_matches_eager_and_follows_the_material prints both).

    step = zdr_amd.graph.capture(scene, material, res=(64, 64), spp=16, seed=0)
    for it in range(1000):
        image, d_material = step(cotangent)         # reads `material` and `cotangent` in place, returns static tensors
        ...                                         # update material in place (e.g. an optimiser step on the same storage)

Kernel arguments are frozen at capture — the seed, the camera, the lights and the environment among them: every replay renders the
same view with the same sample set (a fixed-seed objective; capture again for another seed or view).  Under ZDR_CHECK=1 the per-call
device check is skipped for the captured calls (a synchronise cannot be recorded); call scene.check() after a replay instead.
Eager renders of OTHER views or sizes and further captures may be interleaved with replays on the same scene: from its first captured
call on, a scene handle keeps every workspace a graph may name alive until it is destroyed and rebuilds the camera-ray tile masks in
every call (include/zdr.h, "Stream capture"; tests/test_zz_gpu_graph.py::test_a_replay_survives_eager_renders_of_other_views).  What
still holds is include/zdr.h's rule of one call — or replay — in flight per handle: keep them on one stream.
"""
from __future__ import annotations

import torch


class CapturedRender:
    def __init__(self, scene, material: torch.Tensor, res, spp: int, seed: int = 0):
        if material.device != scene.device or material.dtype != torch.float32 or not material.is_contiguous():
            raise ValueError("material must be a contiguous float32 tensor on the scene's device (it is read in place on every replay)")
        self.scene, self.material, self.res, self.spp, self.seed = scene, material, (int(res[0]), int(res[1])), int(spp), int(seed)
        W, H = self.res
        self.image = torch.zeros((H, W, 4), dtype=torch.float32, device=scene.device)
        self.cotangent = torch.ones((H, W, 4), dtype=torch.float32, device=scene.device)
        self.d_material = torch.zeros_like(material)
        # warm-up on a side stream: first-use allocations of the handle's workspaces, tile masks, occupancy queries
        side = torch.cuda.Stream(device=scene.device)
        side.wait_stream(torch.cuda.current_stream(scene.device))
        with torch.cuda.stream(side):
            self._enqueue()
        torch.cuda.current_stream(scene.device).wait_stream(side)
        torch.cuda.synchronize(scene.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._enqueue()

    def _enqueue(self):
        m = self.material.detach()
        self.scene.render_forward(m, self.res, self.spp, self.seed, out=self.image)
        self.d_material.zero_()
        self.scene.render_backward(self.cotangent, self.d_material, m, self.res, self.spp, self.seed)

    def __call__(self, cotangent=None):
        """Replays forward + backward.  cotangent: None (ones), a number, or an (H, W, 4) tensor copied into the static buffer.
        Returns (image, d_material): static tensors, overwritten by the next replay."""
        if cotangent is None:
            self.cotangent.fill_(1.0)
        elif isinstance(cotangent, (int, float)):
            self.cotangent.fill_(float(cotangent))
        else:
            self.cotangent.copy_(cotangent.reshape(self.cotangent.shape))
        self.graph.replay()
        return self.image, self.d_material


def capture(scene, material, *, res, spp, seed=0) -> CapturedRender:
    return CapturedRender(scene, material, res, spp, seed)
