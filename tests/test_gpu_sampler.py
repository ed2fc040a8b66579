"""-m gpu: sampler values drawn by the HIP kernels are BIT-EXACT with the oracle's
(BASELINE.json north_star: 'sample indices bit-exact')."""
import numpy as np
import pytest
import torch

import oracle
from gpu_util import make_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spp", [1, 4, 16, 64, 256, 1024, 2, 8, 32, 128, 100, 50, 12])   # squares, 2^(2k+1), arbitrary
def test_cmj_draws_bit_exact(spp):
    scene = make_scene("path")
    rng = np.random.default_rng(spp)
    q = np.stack([rng.integers(0, 1024, 400), rng.integers(0, 1024, 400), rng.integers(0, spp, 400)], 1).astype(np.int32)
    for seed in (0, 1, 853402567, 0xFFFFFFFF):
        got = scene.sampler_dump(torch.from_numpy(q).cuda(), spp, seed=seed, nvert=4).cpu().numpy()
        for k in range(q.shape[0]):
            exp = oracle.sampler_dump(oracle.SAMPLER_CMJ, int(q[k, 0]), int(q[k, 1]), seed, spp, int(q[k, 2]), nvert=4)
            assert (got[k, :exp.shape[0]].view(np.uint32) == exp.view(np.uint32)).all(), (spp, seed, q[k])


# The draws AS THE PATH KERNELS MAKE THEM (zdr_vertex_sampler_dump): on every BASELINE configuration shade_ctx / sample_bsdf
# (csrc/integrators.h) take a vertex's seven numbers from cmj_vertex_samples — two Kensler permutations per register, 16-bit
# halves, v_pk_mul_lo_u16 / v_perm_b32 — and the roulette number from cmj_next_with_index, never from the one-by-one calls the
# test above exercises.  Mask widths on both sides of the `w < 2048` shortcut, the largest the packed form takes (w = 0xffff),
# one beyond it and a non-power-of-two (both fall back to the calls one by one: `batched` says which route ran).
# Reference: /root/reference/corrmj.py:6-28 (permute), 60-117 (generate_1d / generate_2d).
@pytest.mark.parametrize("spp", [1, 4, 16, 64, 256, 1024, 4096, 16384, 65536, 131072, 48, 2, 8, 32, 2048, 32768])
def test_cmj_draws_of_the_path_kernels_bit_exact(spp):
    scene = make_scene("path")
    rng = np.random.default_rng(1000 + spp)
    n, nvert = 400, 16
    q = np.stack([rng.integers(0, 1024, n), rng.integers(0, 1024, n), rng.integers(0, spp, n)], 1).astype(np.int32)
    q[0, 2] = spp - 1; q[1, 2] = 0                                  # both ends of the index range
    pow2 = spp & (spp - 1) == 0
    for seed in (0, 1, 853402567, 0xFFFFFFFF):
        got, batched = scene.vertex_sampler_dump(torch.from_numpy(q).cuda(), spp, seed=seed, nvert=nvert)
        assert batched == (pow2 and spp <= 65536), (spp, batched)   # cmj_can_batch: the route the renders of this spp take
        got = got.cpu().numpy()
        for k in range(n):
            exp = oracle.sampler_dump(oracle.SAMPLER_CMJ, int(q[k, 0]), int(q[k, 1]), seed, spp, int(q[k, 2]), nvert=nvert)
            assert exp.shape[0] == 2 + 7 * nvert + (nvert - 2)      # next2f, seven per vertex, a roulette draw from vertex 2 on
            assert (got[k, :exp.shape[0]].view(np.uint32) == exp.view(np.uint32)).all(), (spp, seed, q[k])
        # and the routes of the library agree with each other on every draw: one by one, as the path kernels group them, as the direct kernels do
        # (direct kernels: the pixel's draw packed, the vertex's numbers one by one; path kernels: the other way round)
        plain = scene.sampler_dump(torch.from_numpy(q).cuda(), spp, seed=seed, nvert=nvert).cpu().numpy()
        assert (plain.view(np.uint32) == got.view(np.uint32)).all(), (spp, seed)
        as_direct, b2 = scene.vertex_sampler_dump(torch.from_numpy(q).cuda(), spp, seed=seed, nvert=nvert, integrator="direct")
        assert b2 == batched and (as_direct.cpu().numpy().view(np.uint32) == got.view(np.uint32)).all(), (spp, seed)


def test_pmj02bn_draws_bit_exact_with_synthetic_tables():
    # the reference's pbrt tables are absent (.MISSING_LARGE_BLOBS): any table pins the arithmetic
    rng = np.random.default_rng(0)
    pmj = rng.integers(0, 2**32, (5, 1024, 2), dtype=np.uint64).astype(np.uint32)
    bn = rng.integers(0, 2**16, (48, 128, 128), dtype=np.uint32).astype(np.uint16)
    import ctypes as C
    oracle.lib().zdro_set_pmj02bn_tables(pmj.ctypes.data_as(C.POINTER(C.c_uint32)), 5, 1024, bn.ctypes.data_as(C.POINTER(C.c_uint16)), 48, 128)
    scene = make_scene("path")
    scene.sampler = "pmj02bn"
    scene.set_pmj02bn_tables(pmj, bn)
    for spp in (16, 256, 100):
        q = np.stack([rng.integers(0, 600, 300), rng.integers(0, 600, 300), rng.integers(0, spp, 300)], 1).astype(np.int32)
        for seed in (0, 12345):
            got = scene.sampler_dump(torch.from_numpy(q).cuda(), spp, seed=seed, nvert=4).cpu().numpy()
            for k in range(q.shape[0]):
                exp = oracle.sampler_dump(oracle.SAMPLER_PMJ02BN, int(q[k, 0]), int(q[k, 1]), seed, spp, int(q[k, 2]), nvert=4)
                assert (got[k, :exp.shape[0]].view(np.uint32) == exp.view(np.uint32)).all(), (spp, seed, q[k])
            # the path kernels' route for this sampler is the calls one by one: same numbers, and the library says so
            as_kernels, batched = scene.vertex_sampler_dump(torch.from_numpy(q).cuda(), spp, seed=seed, nvert=4)
            assert not batched and (as_kernels.cpu().numpy().view(np.uint32) == got.view(np.uint32)).all()


def test_pmj02bn_without_tables_fails_loudly_at_the_c_abi():
    # the Python Scene generates tables on demand; the C-ABI itself refuses to render without any
    import ctypes as C
    from zdr_amd import _native as N
    scene = make_scene("path")
    m = torch.rand((8, 8, 4), device="cuda")
    p = scene._params((16, 16), 4, 0, (8, 8))
    p.sampler = N.SAMPLER_PMJ02BN
    img = torch.zeros((16, 16, 4), device="cuda")
    rc = N.lib().zdr_render_forward(scene._handle, C.byref(p), m.data_ptr(), img.data_ptr(), None)
    assert rc == -3 and b"tables" in N.lib().zdr_last_error()
