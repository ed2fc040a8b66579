"""Generates tests/golden/obj_fixtures.npz by RUNNING the reference's own OBJ loader.

``/root/reference/load_obj.py`` (read_obj :1-59, concat_triangles :61-66) is the only module of the
reference without a ``luisa`` import, so it is the only reference code that can emit ground truth in
this container.  It is loaded by file path (never copied), run on every OBJ asset of the reference,
and its outputs -- the de-duplicated (position, texcoord, normal) vertex tuples, the re-indexed faces
and the fan-triangulated index list -- are stored as plain arrays.  Data only travels; the reference's
.py does not.

Run here (the reference does not exist on the GPU box):  python tests/golden/make_obj_fixtures.py
"""
import importlib.util
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = {                       # fixture key -> path below /root/reference
    "cboxuv": "assets/cboxuv.obj",
    "cbox_light": "assets/cbox-light.obj",
    "quad": "assets/quad.obj",
    "cbox_combined": "assets/cbox-combined.obj",
    "sphere": "sphere.obj",
}


def reference_loader():
    spec = importlib.util.spec_from_file_location("zdr_reference_load_obj", os.path.join(REF, "load_obj.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pack(vertices, faces, tris):
    """(V,8) float64 rows pos|uv|normal (NaN normals kept), faces as a flat index list + offsets."""
    v = np.array([list(p) + list(t) + list(n) for p, t, n in vertices], np.float64).reshape(-1, 8)
    flat = np.array([i for f in faces for i in f], np.int64)
    offs = np.cumsum([0] + [len(f) for f in faces]).astype(np.int64)
    return v, flat, offs, np.array(tris, np.int64)


def main():
    ref = reference_loader()
    out = {}
    for key, rel in FILES.items():
        vertices, faces = ref.read_obj(os.path.join(REF, rel))
        tris = ref.concat_triangles(faces)
        v, flat, offs, t = pack(vertices, faces, tris)
        out[key + "_vertices"], out[key + "_faces"], out[key + "_face_offsets"], out[key + "_triangles"] = v, flat, offs, t
        print(f"{key}: {len(vertices)} vertices, {len(faces)} faces, {len(tris) // 3} triangles")
    # sphere.obj is not among the assets copied for the bench scenes: keep its text as a data file so
    # the loader under test reads the same bytes on the GPU box (an OBJ is data, not source).
    with open(os.path.join(REF, "sphere.obj"), "rb") as fh:
        sphere = fh.read()
    with open(os.path.join(HERE, "assets", "sphere.obj"), "wb") as fh:
        fh.write(sphere)
    with open(os.path.join(REF, "assets/cbox-combined.obj"), "rb") as fh:
        comb = fh.read()
    with open(os.path.join(HERE, "assets", "cbox-combined.obj"), "wb") as fh:
        fh.write(comb)
    np.savez_compressed(os.path.join(HERE, "obj_fixtures.npz"), **out)


if __name__ == "__main__":
    main()
