"""Host BVH builder (rt_bvh.cpp) through the host-only hook rt_bvh_check: conservative boxes, every triangle reached
once, and the same tree for any number of build threads.  No GPU involved."""
import ctypes as C

import numpy as np
import pytest

from conftest import get_scene
from raytracer_amd import _native as N


def check(v0, v1, v2, threads):
    v0, v1, v2 = (np.ascontiguousarray(a, np.float32) for a in (v0, v1, v2))
    nn, h, bad = C.c_int64(), C.c_uint64(), C.c_int64()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert N.hip_lib().rt_bvh_check(len(v0), p(v0), p(v1), p(v2), threads, C.byref(nn), C.byref(h), C.byref(bad)) == 0
    return nn.value, h.value, bad.value


@pytest.mark.parametrize("scene", ["bunny", "competition"])
def test_scene_trees_are_conservative(scene):
    s = get_scene(scene, True)
    nodes, _, bad = check(s.v0, s.v1, s.v2, 1)
    assert bad == 0 and 0 < nodes < s.n_triangles


def test_parallel_build_gives_the_serial_tree():
    rng = np.random.default_rng(7)
    n = 400_000  # above the size from which subtrees are built by worker threads
    c = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    v = [c + rng.uniform(-0.02, 0.02, (n, 3)).astype(np.float32) for _ in range(3)]
    v[0][:1000] = v[1][:1000]  # degenerate triangles
    v[2][5000:5100] = 1e6      # far outliers
    one = check(*v, 1)
    many = check(*v, 5)
    assert one[2] == 0 and many[2] == 0
    assert one[:2] == many[:2]


def test_tiny_and_empty_inputs():
    z = np.zeros((0, 3), np.float32)
    assert check(z, z, z, 0) == (0, check(z, z, z, 2)[1], 0)
    t = np.array([[0, 0, 0]], np.float32), np.array([[1, 0, 0]], np.float32), np.array([[0, 1, 0]], np.float32)
    nodes, _, bad = check(*t, 0)
    assert nodes == 0 and bad == 0  # a single leaf: the root reference points at the triangle records
    same = [np.repeat(a, 9, axis=0) for a in t]  # nine identical triangles: centroids coincide, split by count
    nodes, _, bad = check(*same, 0)
    assert nodes >= 1 and bad == 0


def test_presplit_pieces_cover_their_triangles():
    """Large triangles are cut into pieces before the build (early split clipping, rt_bvh.cpp: presplit); rt_bvh_check walks
    the finished tree for 16 barycentric points of every triangle and wants each of them inside a leaf that references the
    triangle.  The shapes that stress the clipper: needles along a space diagonal, a ground quad a thousand times larger than
    the rest, axis-aligned flat pieces (zero extent on one axis), collinear and zero-area triangles of large extent."""
    rng = np.random.default_rng(11)
    n = 60_000
    c = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    v = [c + rng.uniform(-0.01, 0.01, (n, 3)).astype(np.float32) for _ in range(3)]
    k = 2_000                                   # needles: long, thin, diagonal
    d = rng.normal(size=(k, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    v[1][:k] = v[0][:k] + d * rng.uniform(0.5, 3.0, (k, 1)).astype(np.float32)
    v[2][:k] = v[0][:k] + rng.uniform(-0.003, 0.003, (k, 3)).astype(np.float32)
    v[0][k], v[1][k], v[2][k] = (-900, -1, -900), (900, -1, -900), (900, -1, 900)         # ground, flat in y
    v[0][k + 1], v[1][k + 1], v[2][k + 1] = (-900, -1, -900), (900, -1, 900), (-900, -1, 900)
    v[0][k + 2], v[1][k + 2], v[2][k + 2] = (0.25, -5, -7), (0.25, 6, -7), (0.25, 6, 8)   # flat in x
    v[0][k + 3], v[1][k + 3], v[2][k + 3] = (-4, -4, -4), (0, 0, 0), (4, 4, 4)            # collinear
    v[0][k + 4], v[1][k + 4], v[2][k + 4] = (-3, 2, 1), (5, 2, 1), (-3, 2, 1)             # two corners coincide
    one = check(*v, 1)
    many = check(*v, 5)
    assert one[2] == 0 and many[2] == 0
    assert one[:2] == many[:2]                  # the pieces and the tree do not depend on the thread count
