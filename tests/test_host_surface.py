"""Host scene-definition surface (raytracer_amd/host): .obj ingest quirks, BMP decode, scene flattening.
Checked against dumps of the reference's own MeshLoader / scene objects (tests/golden/meshes.npz).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from raytracer_amd import _native as N
from raytracer_amd.scene import FlatScene

MESH_DIR = os.path.join(N.ASSET_DIR, "meshes")


def ref_layout(s):
    """FlatScene triangles -> the 29-float rows of ref_harness `mesh` (A B C n nA nB nC tA tB tC hasN hasT) minus n"""
    fl = s.tri_flags
    return np.concatenate([s.v0, s.v1, s.v2, s.nA, s.nB, s.nC, s.tA, s.tB, s.tC,
                           (fl & 1)[:, None].astype(np.float32), ((fl >> 1) & 1)[:, None].astype(np.float32)], 1)


def drop_normal(tri29):
    return np.concatenate([tri29[:, :9], tri29[:, 12:]], 1)


def test_mesh_counts_and_parse(golden):
    g = golden("meshes")
    meta = json.loads(str(g["meta"]))
    for f, m in meta.items():
        s, counts = FlatScene.mesh(os.path.join(MESH_DIR, f), m["translation"], lazy=True)
        assert counts == {k: m[k] for k in ("triangles", "vertices", "normals", "uvs")}, f
        assert int(((s.tri_flags & 1) == 0).sum()) == m["flat"], f


def test_mesh_bit_exact_small(golden):
    g = golden("meshes")
    meta = json.loads(str(g["meta"]))
    for f in ("bunny.obj", "qwilfish_eyes.obj"):
        s, _ = FlatScene.mesh(os.path.join(MESH_DIR, f), meta[f]["translation"], lazy=True)
        ref = drop_normal(g[f.replace(".", "_")])
        assert np.array_equal(ref_layout(s).view(np.uint32), ref.view(np.uint32)), f


def test_stale_index_faces(golden):
    """64 faces of crystalpillar.obj are read with stale 2nd/3rd indices and no normals/uvs"""
    g = golden("meshes")
    meta = json.loads(str(g["meta"]))
    s, _ = FlatScene.mesh(os.path.join(MESH_DIR, "crystalpillar.obj"), meta["crystalpillar.obj"]["translation"], lazy=True)
    idx = g["crystalpillar_flat_index"]
    assert len(idx) == 64
    assert np.array_equal(np.nonzero((s.tri_flags & 1) == 0)[0], idx)
    assert np.array_equal(ref_layout(s)[idx].view(np.uint32), drop_normal(g["crystalpillar_flat"]).view(np.uint32))


def test_missing_mesh_is_an_error(tmp_path):
    with pytest.raises(RuntimeError):
        FlatScene.mesh(str(tmp_path / "nope.obj"), lazy=True)


def test_builtin_scene_inventory(cpu_scene):
    s = cpu_scene("competition")
    assert s.n_triangles == 68202 and s.n_objects == 20
    assert len(s.sph_object) == 2 and len(s.pl_object) == 2 and len(s.light_pos) == 4
    assert np.all(np.diff(s.tri_object) >= 0)
    s = cpu_scene("sample")
    assert s.n_triangles == 1392 and s.n_objects == 17 and len(s.sph_object) == 10 and len(s.pl_object) == 6
    assert all(b is not None and b.shape == (1024, 1024, 3) for b in s.bmp)
    s = cpu_scene("bunny")
    assert s.n_objects == 7 and len(s.sph_object) == 0


def test_sphere_matrices_are_glm_exact(cpu_scene):
    """translate*scale, inverse and transpose(inverse) restated from glm: spot values that only the cofactor
    formula with 1/det produces (0.03 scale sphere of the competition scene)"""
    s = cpu_scene("competition")
    M, Mi = s.sph_M[1].reshape(4, 4).T, s.sph_Minv[1].reshape(4, 4).T
    assert M[0, 0] == np.float32(0.03) and M[0, 3] == np.float32(-0.53)
    prod = M.astype(np.float64) @ Mi.astype(np.float64)
    assert np.allclose(prod, np.eye(4), atol=1e-5)
    assert np.array_equal(s.sph_N[1].reshape(4, 4), s.sph_Minv[1].reshape(4, 4).T)


def test_cone_scene_matrices_match_the_reference(golden, cpu_scene):
    """Cone (Objects.h:188) through the kept surface: translate * rotate * scale, its inverse and normal matrix must be
    the bits the reference's own objects hold (tests/golden/scene_cones.npz, dumped by oracle/ref_harness `scene`)"""
    g = golden("scene_cones")
    s = cpu_scene("cones")
    units = g["units"]
    cones, spheres = units[units[:, 0] == 4][:, 1:], units[units[:, 0] == 1][:, 1:]
    assert len(s.cone_object) == len(cones) == 6 and len(s.sph_object) == len(spheres) == 1
    mine = np.concatenate([s.cone_M, s.cone_Minv, s.cone_N], 1)
    assert np.array_equal(mine.view(np.uint32), np.ascontiguousarray(cones).view(np.uint32))
    assert np.array_equal(np.concatenate([s.pl_point, s.pl_normal], 1), g["planes"])
    assert s.n_objects == 11 and s.n_triangles == 1392 and list(s.cone_object) == [4, 5, 6, 7, 8, 9]


def test_bmp_decode_orientation():
    img = FlatScene.read_bmp(os.path.join(N.ASSET_DIR, "textures", "basecolor.bmp"))
    assert img.shape == (1024, 1024, 3)
    raw = open(os.path.join(N.ASSET_DIR, "textures", "basecolor.bmp"), "rb").read()
    off = int.from_bytes(raw[10:14], "little")
    # file row 0 is the bottom row -> last row of the decoded top-down image, bytes kept in file (BGR) order
    assert bytes(img[-1, :4].reshape(-1)) == raw[off:off + 12]
