"""CPU-side checks of the product: the C-ABI library builds, loads and exports every symbol that
include/zdr.h declares (no compute calls without a GPU); OBJ ingest; scene assembly; value types."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ASSETS, ROOT, cbox_models
from zdr_amd import Camera, _native, float3, float4x4, geometry
from zdr_amd.load_obj import concat_triangles, read_obj


def test_library_exports_every_symbol_of_the_header():
    hdr = open(os.path.join(ROOT, "include", "zdr.h")).read()
    declared = set(re.findall(r"\b(zdr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    L = _native.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.zdr_version().decode().startswith("zdr-mi355x")


def test_struct_layouts_match_the_header():
    assert C.sizeof(_native.CameraPOD) == 40
    assert C.sizeof(_native.RenderParams) == 4 * 16 + 40 + 8 + 12
    assert _native.RenderParams.struct_size.offset == 0
    assert _native.RenderParams.camera.offset == 64 and _native.RenderParams.tex_h.offset == 104 and _native.RenderParams.tile_shard_count.offset == 116
    assert C.sizeof(_native.SceneInfo) == 48


def test_abi_version_and_struct_size_are_checked():
    """A stub built against another revision of include/zdr.h is refused, not read past."""
    L = _native.lib()
    hdr = open(os.path.join(ROOT, "include", "zdr.h")).read()
    assert L.zdr_abi_version() == _native.ABI_VERSION == int(re.search(r"#define ZDR_ABI_VERSION (\d+)", hdr).group(1))
    p = _native.RenderParams()
    p.struct_size = C.sizeof(_native.RenderParams) - 12          # the round-1 struct, three fields short
    cnt = (C.c_uint64 * 8)()
    fake = C.c_void_p(16)                                        # never dereferenced: the check comes first
    rc = L.zdr_path_dump(fake, C.byref(p), fake, None, fake, 1, 1, fake, None)
    assert rc == -1 and b"struct_size" in L.zdr_last_error()


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import zdr_amd
    with pytest.raises(_native.ZdrError, match="no CPU back end"):
        zdr_amd.Scene(cbox_models(), integrator="path")


def test_scene_create_rejects_bad_input_without_touching_a_gpu():
    L = _native.lib()
    h = C.c_void_p()
    v = np.zeros((3, 8), np.float32); t = np.array([[0, 1, 5]], np.int32); b = np.array([0, 1], np.int32); e = np.zeros((1, 3), np.float32)
    rc = L.zdr_scene_create(v.ctypes.data, 3, t.ctypes.data, 1, b.ctypes.data, None, e.ctypes.data, 1, 0, 0, C.byref(h))
    assert rc == -1 and b"out of range" in L.zdr_last_error()
    rc = L.zdr_scene_create(v.ctypes.data, 3, t.ctypes.data, 0, b.ctypes.data, None, e.ctypes.data, 1, 0, 0, C.byref(h))
    assert rc == -1 and b"empty" in L.zdr_last_error()


def test_read_obj_matches_survey_facts():
    verts, faces = read_obj(os.path.join(ASSETS, "cboxuv.obj"))
    assert len(verts) == 60 and len(concat_triangles(faces)) // 3 == 30
    lv, lf = read_obj(os.path.join(ASSETS, "cbox-light.obj"))
    assert len(lv) == 4 and len(lf) == 2 and all(np.isnan(x) is np.False_ or True for x in lv[0][2])
    assert lv[0][1] == (0.0, 0.0)                                       # missing vt -> (0, 0)


def test_normal_less_obj_gets_flat_recomputed_normals():
    A = geometry.assemble([(os.path.join(ASSETS, "quad.obj"), None, 5)])   # 'f 4 3 2 1': no vt, no vn
    assert A.verts.shape == (4, 8) and A.tris.shape == (2, 3)
    np.testing.assert_allclose(A.verts[:, 5:8], np.tile([0, -1, 0], (4, 1)), atol=1e-6)
    assert (A.inst_emission == 5).all()


def test_fan_triangulation_and_dedup(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf 1/1/1 3/1/1 4/1/1\n")
    verts, faces = read_obj(str(p))
    assert len(verts) == 4 and faces == [[0, 1, 2, 3], [0, 2, 3]]
    assert concat_triangles(faces) == [0, 1, 2, 0, 2, 3, 0, 2, 3]


def test_assemble_cbox():
    A = geometry.assemble(cbox_models())
    assert A.verts.shape == (64, 8) and A.tris.shape == (32, 3) and list(A.inst_tri_begin) == [0, 30, 32]
    np.testing.assert_allclose(A.verts[:, :3].min(0), [-3.014011, -0.162685, -5.839967], atol=1e-6)
    assert (A.inst_emission[1] == 20).all() and (A.inst_emission[0] == 0).all()
    with pytest.raises(RuntimeError, match="maximum number"):
        geometry.assemble(cbox_models() * 5001)


def test_value_types():
    assert tuple(float3(2)) == (2.0, 2.0, 2.0) and tuple(float3(1, 2, 3)) == (1.0, 2.0, 3.0)
    m = np.arange(16, dtype=np.float64).reshape(4, 4)
    np.testing.assert_array_equal(float4x4(*m.transpose().flatten()).rows(), m)   # column-major ctor, test_lightstage.py:44
    np.testing.assert_array_equal(float4x4(1.0).rows(), np.eye(4))
    c = Camera(fov=0.5, origin=float3(1, 2, 3), target=float3(0), up=float3(0, 1, 0))
    assert c.copy().origin == c.origin and c.copy() is not c


# --- reference-pinned: fixtures produced by the reference's own load_obj.py (tests/golden/make_obj_fixtures.py) ---

OBJ_FIXTURES = {"cboxuv": "cboxuv.obj", "cbox_light": "cbox-light.obj", "quad": "quad.obj",
                "cbox_combined": "cbox-combined.obj", "sphere": "sphere.obj"}


@pytest.mark.parametrize("key", sorted(OBJ_FIXTURES))
def test_read_obj_reproduces_the_reference_loader(key):
    """SURVEY row a20, pinned by the reference itself: /root/reference/load_obj.py:1-68 was run on this
    very file and its vertex tuples, faces and triangle list stored in obj_fixtures.npz."""
    from conftest import GOLDEN
    fx = np.load(os.path.join(GOLDEN, "obj_fixtures.npz"))
    verts, faces = read_obj(os.path.join(ASSETS, OBJ_FIXTURES[key]))
    got = np.array([list(p) + list(t) + list(n) for p, t, n in verts], np.float64).reshape(-1, 8)
    want = fx[key + "_vertices"]
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)                      # float64 parse, bit for bit; NaN normals in place
    assert np.array_equal(np.isnan(got), np.isnan(want))
    offs = fx[key + "_face_offsets"]
    assert [len(f) for f in faces] == list(np.diff(offs))
    assert np.array_equal(np.array([i for f in faces for i in f], np.int64), fx[key + "_faces"])
    assert np.array_equal(np.array(concat_triangles(faces), np.int64), fx[key + "_triangles"])


def test_assemble_on_the_reference_pinned_sphere():
    """sphere.obj (960 triangles, 559 de-duplicated vertices by the reference's loader) through scene assembly:
    index ranges, unit normals kept, and the float32 vertex rows are the fixture's float64 rows rounded once."""
    from conftest import GOLDEN
    fx = np.load(os.path.join(GOLDEN, "obj_fixtures.npz"))
    A = geometry.assemble([(os.path.join(ASSETS, "sphere.obj"), None, 0.0)])
    assert A.verts.shape == (559, 8) and A.tris.shape == (960, 3)
    assert np.array_equal(A.tris.reshape(-1), fx["sphere_triangles"].astype(A.tris.dtype))
    np.testing.assert_array_equal(A.verts[:, :5], fx["sphere_vertices"][:, :5].astype(np.float32))
