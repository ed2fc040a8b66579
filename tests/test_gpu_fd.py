"""-m gpu: the gradient bar of BASELINE.json ("grad rel-err vs fd_validate"): AD against two-sided
finite differences of the forward render (eps = 0.01, same seed in both renders, fd_validate.py:72-92),
as a whole-image directional derivative so that a few seconds resolve 1e-3 (tools/fd_directional.py and
tools/fd_validate.py are the long-form versions; results in profiles/)."""
import numpy as np
import pytest
import torch

from conftest import fd_material_np
from gpu_util import make_scene

pytestmark = pytest.mark.gpu


def directional(scene, material, delta, W, spp, seeds, wimg, eps=0.01):
    ad, fd = [], []
    for s in range(seeds):
        d = torch.zeros_like(material)
        scene.render_backward(wimg, d, material, (W, W), spp, 7000 + s)
        ad.append((d.double() * delta.double()).sum().item())
        ip = scene.render_forward(material + eps * delta, (W, W), spp, 3000 + s).double()
        im = scene.render_forward(material - eps * delta, (W, W), spp, 3000 + s).double()
        fd.append((((ip - im) * wimg.double()).sum() / (2 * eps)).item())
    ad, fd = np.array(ad), np.array(fd)
    sigma = np.hypot(ad.std(ddof=1), fd.std(ddof=1)) / np.sqrt(seeds)
    return ad.mean(), fd.mean(), sigma


@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_ad_matches_fd_to_1e3_for_diffuse(integrator):
    scene = make_scene(integrator)
    material = torch.from_numpy(fd_material_np(1024, 0)).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    W = 256
    wimg = torch.rand((W, W, 4), device="cuda", generator=g) + 0.5; wimg[..., 3] = 0
    delta = torch.zeros_like(material); delta[..., :3] = torch.rand(material[..., :3].shape, device="cuda", generator=g)
    ad, fd, sigma = directional(scene, material, delta, W, 2048, 8, wimg)
    print(f"[fd] {integrator} diffuse: AD {ad:.3f} FD {fd:.3f} rel {abs(ad - fd) / abs(fd):.2e} (1 sigma {sigma / abs(fd):.2e})")
    assert abs(ad - fd) <= 1e-3 * abs(fd) + 3 * sigma


def test_ad_matches_fd_for_roughness_within_resolution():
    # the FD of roughness is much noisier (it moves the VNDF samples): require agreement within 4 sigma and 1 %
    scene = make_scene("path")
    material = torch.from_numpy(fd_material_np(1024, 0)).cuda()
    g = torch.Generator(device="cuda").manual_seed(2)
    W = 256
    wimg = torch.rand((W, W, 4), device="cuda", generator=g) + 0.5; wimg[..., 3] = 0
    delta = torch.zeros_like(material); delta[..., 3] = torch.rand(material[..., 3].shape, device="cuda", generator=g)
    ad, fd, sigma = directional(scene, material, delta, W, 4096, 12, wimg)
    print(f"[fd] path roughness: AD {ad:.3f} FD {fd:.3f} rel {abs(ad - fd) / abs(fd):.2e} (1 sigma {sigma / abs(fd):.2e})")
    assert abs(ad - fd) <= 4 * sigma and abs(ad - fd) <= 1e-2 * abs(fd) + 2 * sigma


@pytest.mark.parametrize("what", ["environment", "three lights"])
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_ad_matches_fd_with_an_environment_map_and_with_several_lights(integrator, what):
    """The adjoint of the OTHER light-sampling branches — the environment's alias-table sampling with its MIS weights
    (envmap.py:150-203) and sample_light over several emitters of different sizes (light.py:33-48) — against finite
    differences of the forward render, all four material channels at once (long form: tools/fd_directional.py
    --scene env|lights3, profiles/r2_fd_directional_env_lights.txt)."""
    if what == "environment":
        from test_envmap import sun_sky
        scene = make_scene(integrator)
        scene.add_envmap(sun_sky())
    else:
        from gpu_util import multi_light_arrays
        scene = make_scene(integrator, arrays=multi_light_arrays())
    material = torch.from_numpy(fd_material_np(1024, 0)).cuda()
    g = torch.Generator(device="cuda").manual_seed(3)
    W = 256
    wimg = torch.rand((W, W, 4), device="cuda", generator=g) + 0.5; wimg[..., 3] = 0
    delta = torch.rand(material.shape, device="cuda", generator=g)
    ad, fd, sigma = directional(scene, material, delta, W, 2048, 8, wimg)
    print(f"[fd] {integrator}, {what}, all channels: AD {ad:.3f} FD {fd:.3f} rel {abs(ad - fd) / abs(fd):.2e} (1 sigma {sigma / abs(fd):.2e})")
    assert abs(ad - fd) <= 1e-3 * abs(fd) + 3 * sigma
    scene.check()
