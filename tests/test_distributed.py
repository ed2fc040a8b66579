"""The N > 1 path on CPU: two gloo ranks run zdr_amd.distributed with the ORACLE as the local
renderer (the product renderer needs a GPU) and must reproduce the unsharded render."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import CBOX_CAMERA, cbox_models, fd_material_np

W, SPP, SEED = 24, 16, 3


def _oracle_renderer(mode):
    import oracle
    from zdr_amd import distributed as zd
    from zdr_amd import geometry
    S = oracle.OracleScene.from_arrays(geometry.assemble(cbox_models()))

    def tiles_of(rect, tile_shard):
        # the oracle knows rectangles only: an interleaved tile shard is the list of its 8x8 tiles, numbered along the
        # diagonals exactly as the kernels do (zd.shard_tiles mirrors decode_item; tests/test_gpu_render.py checks the
        # kernels' ownership against the same function)
        return [rect] if tile_shard is None else zd.shard_tiles(rect, *tile_shard)

    def fwd(material, res, spp, seed, rect, samples, out, tile_shard=None):
        for r in tiles_of(rect, tile_shard):
            p = oracle.make_params("path", res[0], res[1], spp, seed, CBOX_CAMERA, tuple(material.shape[:2]), rect=r, samples=samples, nthreads=2)
            img = torch.from_numpy(S.render_forward(p, material.numpy()))
            x0, y0, x1, y1 = r
            out[y0:y1, x0:x1] = img[y0:y1, x0:x1]
        return out

    def bwd(grad_output, d_material, material, res, spp, seed, rect, samples, tile_shard=None):
        for r in tiles_of(rect, tile_shard):
            p = oracle.make_params("path", res[0], res[1], spp, seed + 1, CBOX_CAMERA, tuple(material.shape[:2]), rect=r, samples=samples, nthreads=2)
            d_material += torch.from_numpy(S.render_backward(p, grad_output.numpy(), material.numpy()))

    return zd.ShardedRenderer(fwd, bwd, mode)


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from zdr_amd import distributed as zd
    assert zd.init_from_env("gloo") == (rank, world, rank)
    r = _oracle_renderer(mode)
    m = torch.from_numpy(fd_material_np(32, 2)).requires_grad_()
    img = r.render(m, res=(W, W), spp=SPP, seed=SEED)
    (img * 0.5).sum().backward()
    if rank == 0:
        q.put((img.detach().numpy(), m.grad.numpy()))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("mode", ["tiles", "rows", "samples", "seeds"])
def test_two_ranks_reproduce_the_single_process_render(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs: p.start()
    img, grad = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    import oracle
    from zdr_amd import distributed as zd, geometry
    S = oracle.OracleScene.from_arrays(geometry.assemble(cbox_models()))
    mat = fd_material_np(32, 2)
    cot = np.full((W, W, 4), 0.5, np.float32)

    def single(seed):
        pf = oracle.make_params("path", W, W, SPP, seed, CBOX_CAMERA, mat.shape[:2])
        pb = oracle.make_params("path", W, W, SPP, seed + 1, CBOX_CAMERA, mat.shape[:2])
        return S.render_forward(pf, mat), S.render_backward(pb, cot, mat)

    if mode == "seeds":
        a, b = single(SEED), single((SEED + zd.SEED_STRIDE) & 0xFFFFFFFF)
        ref_img, ref_grad = 0.5 * (a[0] + b[0]), 0.5 * (a[1] + b[1])
    else:
        ref_img, ref_grad = single(SEED)
    if mode in ("rows", "tiles"):
        assert np.array_equal(img, ref_img)                     # pixel tiles: bit-identical union
    else:
        np.testing.assert_allclose(img, ref_img, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(grad, ref_grad, rtol=1e-4, atol=1e-7)


def test_eight_ranks_tile_union_is_bit_identical():
    """BASELINE configs[3]'s layout at its rank count: 8 gloo ranks, `tiles` mode, 24 x 24 pixels = 9 tiles (rank 8 of 8
    owns one tile, most own one, one owns two), oracle as the local renderer.  The union is the unsharded image bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, "tiles", q)) for r in range(8)]
    for p in procs: p.start()
    img, grad = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    import oracle
    from zdr_amd import geometry
    S = oracle.OracleScene.from_arrays(geometry.assemble(cbox_models()))
    mat = fd_material_np(32, 2)
    ref = S.render_forward(oracle.make_params("path", W, W, SPP, SEED, CBOX_CAMERA, mat.shape[:2]), mat)
    gref = S.render_backward(oracle.make_params("path", W, W, SPP, SEED + 1, CBOX_CAMERA, mat.shape[:2]), np.full((W, W, 4), 0.5, np.float32), mat)
    assert np.array_equal(img, ref)
    np.testing.assert_allclose(grad, gref, rtol=1e-4, atol=1e-7)


def test_shard_tiles_partition_along_diagonals():
    """zd.shard_tiles = the ownership mapping of include/zdr.h: the shards of one count partition the rectangle, and tile
    (tx, ty) belongs to shard (ty * tiles_x + (tx - ty) mod tiles_x) mod count — diagonals, not columns."""
    from zdr_amd import distributed as zd
    rect = (13, 7, 90, 60)                                      # 10 x 7 tiles, ragged right and bottom edges
    tiles_x = (90 - 13 + 7) // 8
    for count in (1, 2, 3, 8, 100):
        cover = np.zeros((60, 90), int)
        for r in range(count):
            for (x0, y0, x1, y1) in zd.shard_tiles(rect, r, count):
                cover[y0:y1, x0:x1] += 1
                tx, ty = (x0 - 13) // 8, (y0 - 7) // 8
                assert (ty * tiles_x + ((tx - ty) % tiles_x if count > 1 else tx)) % count == r
        assert (cover[7:60, 13:90] == 1).all() and cover.sum() == (60 - 7) * (90 - 13)
    # eight shards of a 128-tile-wide image: a shard's tiles do not line up in columns
    cols = {x0 for (x0, y0, x1, y1) in zd.shard_tiles((0, 0, 1024, 64), 0, 8)}
    assert len(cols) > 16


def test_shard_plans_cover_the_work_exactly():
    from zdr_amd import distributed as zd
    for world in (1, 2, 3, 8):
        rows = np.zeros(512, int); samples = np.zeros(256, int)
        for r in range(world):
            for (x0, y0, x1, y1) in zd.plan("rows", r, world, (512, 512), 256, 0).rects:
                assert (x0, x1) == (0, 512); rows[y0:y1] += 1
            b, e = zd.plan("samples", r, world, (512, 512), 256, 0).samples
            samples[b:e] += 1
        assert (rows == 1).all() and (samples == 1).all()
        shards = [zd.plan("tiles", r, world, (512, 512), 256, 0) for r in range(world)]
        assert all(s.rects == [(0, 0, 512, 512)] and s.samples == (0, 256) for s in shards)
        assert [s.tile_shard for s in shards] == ([None] if world == 1 else [(r, world) for r in range(world)])
    assert len({zd.plan("seeds", r, 8, (8, 8), 4, 5).seed for r in range(8)}) == 8
