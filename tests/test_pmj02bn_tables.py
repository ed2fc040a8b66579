"""Own PMJ02bn tables: the sample sets are (0,2)-sequences, the textures are blue noise."""
import numpy as np
import pytest

from zdr_amd import pmj02bn_tables as T


@pytest.mark.parametrize("n", [4, 16, 32, 256, 2048])
def test_every_power_of_two_prefix_is_a_02_net(n):
    sets = T.pmj02_sets(n_sets=3, n_samples=4096, seed=1)
    k = n.bit_length() - 1
    for s in range(3):
        pts = sets[s, :n].astype(np.float64) / 2 ** 32
        for a in range(k + 1):                          # elementary intervals 2^-a x 2^-(k-a)
            nx, ny = 1 << a, 1 << (k - a)
            cell = np.floor(pts[:, 0] * nx).astype(int) * ny + np.floor(pts[:, 1] * ny).astype(int)
            assert sorted(cell) == list(range(n)), (s, n, a)


def test_sets_are_distinct_and_progressive_blocks_stay_stratified():
    sets = T.pmj02_sets(n_sets=2, n_samples=1024, seed=0)
    assert not np.array_equal(sets[0], sets[1])
    pts = sets[0, 256:512].astype(np.float64) / 2 ** 32     # the second block of 256 is a (0,2)-net too
    for a in range(9):
        nx, ny = 1 << a, 1 << (8 - a)
        cell = np.floor(pts[:, 0] * nx).astype(int) * ny + np.floor(pts[:, 1] * ny).astype(int)
        assert sorted(cell) == list(range(256))


def test_void_and_cluster_is_blue_noise():
    res = 32
    tex = T.blue_noise_textures(n_tex=1, res=res, seed=3)[0].astype(np.float64)
    assert sorted((tex / 65536.0 * res * res).astype(int).reshape(-1)) == list(range(res * res))   # a permutation of the ranks
    def low_frequency_share(img):
        f = np.abs(np.fft.fft2(img - img.mean())) ** 2
        ky, kx = np.meshgrid(np.fft.fftfreq(res), np.fft.fftfreq(res), indexing="ij")
        low = np.hypot(kx, ky) < 0.12
        return f[low].sum() / f.sum()
    white = np.random.default_rng(0).permutation(res * res).reshape(res, res).astype(np.float64)
    assert low_frequency_share(tex) < 0.15 * low_frequency_share(white)
    # thresholding at any level leaves well-spread points: no two of the first 5 % ranks are neighbours
    first = tex < 0.05 * 65536
    assert not (first & np.roll(first, 1, 0)).any() and not (first & np.roll(first, 1, 1)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("accel", ["brute", "bvh"])
def test_pmj02bn_render_with_generated_tables_matches_oracle_and_cmj(accel, cbox_arrays):
    import ctypes as C
    import torch
    import oracle
    from conftest import cbox_material_np
    from gpu_util import assert_image_parity, make_scene, oracle_params
    pmj = T.pmj02_sets(n_sets=5, n_samples=1024, seed=0)
    bn = T.blue_noise_textures(n_tex=4, res=32, seed=0)
    scene = make_scene("path", accel=accel)
    scene.sampler = "pmj02bn"
    scene.set_pmj02bn_tables(pmj, bn)
    mat = cbox_material_np()
    m = torch.from_numpy(mat).cuda()
    img = scene.render(m, res=(96, 96), spp=64, seed=1).cpu().numpy()
    oracle.lib().zdro_set_pmj02bn_tables(pmj.ctypes.data_as(C.POINTER(C.c_uint32)), 5, 1024, bn.ctypes.data_as(C.POINTER(C.c_uint16)), 4, 32)
    S = oracle.OracleScene.from_arrays(cbox_arrays)
    p = oracle_params(scene, 96, 96, 64, 1, mat.shape[:2], sampler=oracle.SAMPLER_PMJ02BN)
    assert_image_parity(img[..., :3], S.render_forward(p, mat)[..., :3], "pmj02bn with generated tables")
    # same integrand, different sampler: the two images agree statistically
    scene.sampler = "cmj"
    cmj = scene.render(m, res=(96, 96), spp=64, seed=1).cpu().numpy()
    assert abs(img[..., :3].mean() - cmj[..., :3].mean()) / cmj[..., :3].mean() < 0.01
