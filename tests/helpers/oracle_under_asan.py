"""Run by tests/test_oracle_sanitizers.py under LD_PRELOAD=libasan: every entry point of the oracle on small inputs."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
oracle.VARIANTS["asan"] = "libzdr_oracle_asan.so"
_orig = oracle.build
oracle.build = lambda force=False, variant="ieee": os.path.join(os.path.dirname(oracle.__file__), oracle.VARIANTS[variant]) if variant == "asan" else _orig(force, variant)
from conftest import CBOX_CAMERA, cbox_models, fd_material_np
from zdr_amd import geometry, envmap
from gpu_util import multi_light_arrays
from path_trace import all_queries
A = multi_light_arrays()
S = oracle.OracleScene.from_arrays(A, variant="asan")
mat = fd_material_np(64, 0)
for integ in ("collocated", "direct", "path"):
    p = oracle.make_params(integ, 24, 16, 8, 3, CBOX_CAMERA, mat.shape[:2])
    img = S.render_forward(p, mat); g = S.render_backward(p, np.ones((16, 24, 4), np.float32), mat)
    print(integ, float(img[..., :3].mean()), float(np.abs(g).sum()))
p = oracle.make_params("path", 24, 16, 8, 3, CBOX_CAMERA, mat.shape[:2])
tr = S.path_dump(p, mat, all_queries(24, 16, 8)); print("dump", tr.shape)
I = envmap.prepare_image(np.random.default_rng(0).uniform(0.1, 2, (16, 32, 3)).astype(np.float32))
S.set_envmap(I, *envmap.build_tables(I)); print("env", float(S.render_forward(p, mat)[..., :3].mean()))
p.integrator = oracle.UVGRAD; print("uvgrad", float(np.abs(S.render_forward(p, mat)).mean()))
e = np.zeros((5, 3), np.float32); e[2] = 3; S.set_emissions(e); print("lights", float(S.render_forward(oracle.make_params("path", 8, 8, 4, 1, CBOX_CAMERA, mat.shape[:2]), mat)[..., :3].mean()))
rays = np.random.default_rng(1).uniform(-1, 1, (100, 8)).astype(np.float32); rays[:, 7] = 1e30
print(S.trace_closest(rays)[0][:3].tolist(), S.trace_any(rays)[:5].tolist())
# a scene above 256 triangles: the oracle's own BVH (build, closest / any hit with NaN and axis-parallel rays, a render)
from gpu_util import terrain_arrays
T = oracle.OracleScene.from_arrays(terrain_arrays(n=20), variant="asan")
rays = np.random.default_rng(2).uniform(-3, 3, (400, 8)).astype(np.float32); rays[:, 3] = 0; rays[:, 7] = 1e30
rays[0, 4:7] = (0, -1, 0); rays[1, 4:7] = (1, 0, 0); rays[2, 4:7] = np.nan; rays[3, 0:3] = np.nan
print("bvh", int((T.trace_closest(rays)[0][:, 0] >= 0).sum()), int(T.trace_any(rays).sum()),
      float(T.render_forward(oracle.make_params("path", 12, 12, 4, 1, (0.9, (0.5, 3.0, 6.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)), mat.shape[:2]), mat)[..., :3].mean()))
print("sampler", oracle.sampler_dump(oracle.SAMPLER_CMJ, 3, 4, 5, 16, 7)[:4])
print("ok")
