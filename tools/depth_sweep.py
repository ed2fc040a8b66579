import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zdr_amd.scenes import cbox_material_np
from zdr_amd.scenes import make_scene
m = torch.from_numpy(cbox_material_np()).cuda()
for integ, depths in (("collocated", [1]), ("direct", [1]), ("path", [1, 2, 3, 4, 16])):
    scene = make_scene(integ)
    for d in depths:
        scene.max_depth = d
        ts = []
        for i in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            scene.render_forward(m, (512, 512), 256, i)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        st = scene.render_stats(m, (512, 512), 16)
        n = st["samples"]
        print(f"{integ:10s} max_depth={d:2d}: {min(ts[1:])*1e3:7.3f} ms   closest/sample {st['closest_rays']/n:.3f} shaded/sample {st['shaded_vertices']/n:.3f}")
