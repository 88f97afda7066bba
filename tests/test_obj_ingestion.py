"""Wavefront OBJ ingestion (rt_object_loadFromFile and the bulk edits) against the triangles the
reference's own loader produced for the same fixture files (tests/golden/objects.npz, written by
tests/golden/make_golden.py from reference: src/object.c:145-210), and directly against that
loader when it has been built here (oracle/_ref)."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracer_amd.types import Vec3

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ("cube_quads", "quirks")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "objects.npz"))


def edit(host, obj):
    host.lib.rt_object_scale(obj, 2.5)
    host.lib.rt_object_translate(obj, Vec3(0.25, 1.5, -3.0))
    host.lib.rt_object_materialIndex(obj, 3)


@pytest.mark.parametrize("name", FIXTURES)
def test_loader_matches_reference_triangles(host, golden, name):
    obj = host.lib.rt_object_loadFromFile(os.path.join(GOLDEN, name + ".obj").encode())
    assert obj, host.lib.rt_object_lastError()
    assert np.array_equal(host.object_triangles(obj), golden[name + "_loaded"])
    edit(host, obj)
    assert np.array_equal(host.object_triangles(obj), golden[name + "_edited"])
    host.lib.rt_object_destroy(obj)


def test_fan_and_index_forms(host):
    """16 triangles: six quads (two each), a five-sided cap (three) and one triangle; every index form
    (7, 7/2, 7//5, 7/2/5) names vertex 7; material index 0 until set."""
    obj = host.lib.rt_object_loadFromFile(os.path.join(GOLDEN, "cube_quads.obj").encode())
    tri = host.object_triangles(obj)
    assert len(tri) == 16
    rows = tri.view(np.float32).reshape(-1, 10)
    assert (tri.view(np.uint32).reshape(-1, 10)[:, 0] == 0).all()
    assert np.allclose(rows[0, 1:], [-0.5, -0.5, -0.5, 0.5, -0.5, -0.5, 0.5, 0.5, -0.5])      # f 1 2 3 4 -> (1, 2, 3)
    assert np.allclose(rows[1, 1:], [-0.5, -0.5, -0.5, 0.5, 0.5, -0.5, -0.5, 0.5, -0.5])      #            (1, 3, 4)
    assert np.allclose(rows[15, 1:4], [0.0, 0.125, 0.75])                                     # f 9 8 7
    host.lib.rt_object_destroy(obj)


def test_unreadable_file_and_bad_index_are_refused(host, tmp_path):
    assert not host.lib.rt_object_loadFromFile(str(tmp_path / "missing.obj").encode())
    assert b"cannot read" in host.lib.rt_object_lastError()
    for body in ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", "f 1 2 3\nv 0 0 0\nv 1 0 0\nv 0 1 0\n"):
        path = tmp_path / "bad.obj"
        path.write_text(body)
        assert not host.lib.rt_object_loadFromFile(str(path).encode())
        assert b"refers to vertex" in host.lib.rt_object_lastError()
    empty = tmp_path / "empty.obj"
    empty.write_text("# nothing\n")
    obj = host.lib.rt_object_loadFromFile(str(empty).encode())
    assert obj and obj.contents.triangleCount == 0
    host.lib.rt_object_destroy(obj)


def test_object_goes_into_a_scene(host):
    scene = host.lib.rt_scene_init(64, 36)
    before = scene.contents.triangleCount
    obj = host.lib.rt_object_loadFromFile(os.path.join(GOLDEN, "cube_quads.obj").encode())
    edit(host, obj)
    host.lib.rt_scene_addObject(scene, obj.contents)
    assert scene.contents.triangleCount == before + 16
    added = np.frombuffer(C.string_at(C.addressof(scene.contents.triangles[before]), 16 * 40), dtype=np.uint8).reshape(16, 40)
    assert np.array_equal(added, host.object_triangles(obj))
    host.lib.rt_object_destroy(obj)
    host.lib.rt_scene_destroy(scene)


@pytest.mark.parametrize("name", FIXTURES)
def test_loader_equals_reference_loader_built_here(host, oracle, name):
    if oracle.ref is None:
        pytest.skip("the reference build (oracle/_ref) is not present")

    class RefObject(C.Structure):
        _fields_ = [("capacity", C.c_uint32), ("triangleCount", C.c_uint32), ("triangles", C.c_void_p)]
    path = os.path.join(GOLDEN, name + ".obj").encode()
    theirs = oracle.ref.object_loadFromFile(path)
    rec = RefObject.from_address(theirs)
    mine = host.lib.rt_object_loadFromFile(path)
    expect = np.frombuffer(C.string_at(rec.triangles, rec.triangleCount * 40), dtype=np.uint8).reshape(-1, 40)
    assert np.array_equal(host.object_triangles(mine), expect)
    oracle.ref.object_destroy(theirs)
    host.lib.rt_object_destroy(mine)
