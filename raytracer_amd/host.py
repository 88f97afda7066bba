"""Python binding of librthost.so, the plain-C host-side scene model.

Mirrors the reference's host modules (reference: src/scene.h:38-48,
src/camera.h:17-20, src/octree.h:36-37) under the rt_ prefix; see include/rt_host.h.
"""
import ctypes as C
import os

import numpy as np

from .types import Camera, Image, Material, Object, Octree, OctreeNode, Plane, PointLight, Scene, Sphere, Triangle, Vec3

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librthost.so")


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(_LIB_PATH)
    P = C.POINTER
    sigs = {
        "rt_scene_create": (P(Scene), []),
        "rt_scene_init": (P(Scene), [C.c_uint32, C.c_uint32]),
        "rt_scene_addMaterial": (C.c_uint32, [P(Scene), Material]),
        "rt_scene_addPlane": (None, [P(Scene), Plane]),
        "rt_scene_droppedElements": (C.c_uint32, []),
        "rt_scene_addSphere": (None, [P(Scene), Sphere]),
        "rt_scene_addTriangle": (None, [P(Scene), Triangle]),
        "rt_scene_addPointLight": (None, [P(Scene), PointLight]),
        "rt_scene_addObject": (None, [P(Scene), Object]),
        "rt_object_loadFromFile": (P(Object), [C.c_char_p]),
        "rt_object_scale": (None, [P(Object), C.c_float]),
        "rt_object_translate": (None, [P(Object), Vec3]),
        "rt_object_materialIndex": (None, [P(Object), C.c_uint32]),
        "rt_object_destroy": (None, [P(Object)]),
        "rt_object_lastError": (C.c_char_p, []),
        "rt_image_create": (P(Image), [C.c_uint32, C.c_uint32]),
        "rt_image_destroy": (None, [P(Image)]),
        "rt_bitmap_save_image": (C.c_bool, [C.c_char_p, P(Image)]),
        "rt_scene_shrinkToFit": (None, [P(Scene)]),
        "rt_scene_destroy": (None, [P(Scene)]),
        "rt_camera_create": (P(Camera), [Vec3, Vec3, C.c_uint32, C.c_uint32, C.c_float, C.c_float]),
        "rt_camera_setup": (None, [P(Camera)]),
        "rt_move_camera": (None, [P(Camera), C.c_int, C.c_int, C.c_int]),
        "rt_camera_destroy": (None, [P(Camera)]),
        "rt_octree_buildFromScene": (P(Octree), [P(Scene)]),
        "rt_octree_destroy": (None, [P(Octree)]),
        "rt_octree_levels": (C.c_uint32, [P(Octree)]),
        "rt_scene_synth_mesh": (P(Scene), [C.c_uint32] * 7),
        "rt_scene_synth_s1": (P(Scene), [C.c_uint32, C.c_uint32]),
        "rt_scene_synth_s2": (P(Scene), [C.c_uint32, C.c_uint32]),
        "rt_camera_orbit": (None, [P(Camera), C.c_uint32, C.c_uint32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def octree_arrays(octree_ptr):
    """Copies of an Octree's nodes (structured array) and indexes (uint32)."""
    tree = octree_ptr.contents
    node_dtype = np.dtype([("box", np.float32, 6), ("sphereOffset", np.uint32), ("sphereCount", np.uint32),
                           ("triangleOffset", np.uint32), ("triangleCount", np.uint32), ("child", np.int32, 8)])
    assert node_dtype.itemsize == C.sizeof(OctreeNode)
    nodes = np.frombuffer(C.string_at(tree.nodes, tree.nodeCount * 72), dtype=node_dtype).copy()
    if tree.indexCount:
        indexes = np.frombuffer(C.string_at(tree.indexes, tree.indexCount * 4), dtype=np.uint32).copy()
    else:
        indexes = np.zeros(0, np.uint32)
    return nodes, indexes


def object_triangles(object_ptr):
    """The triangles of an Object as raw bytes, one 40-byte row each (materialIndex, v0, v1, v2)."""
    obj = object_ptr.contents
    if obj.triangleCount == 0:
        return np.zeros((0, 40), np.uint8)
    return np.frombuffer(C.string_at(obj.triangles, obj.triangleCount * 40), dtype=np.uint8).reshape(-1, 40).copy()


def camera_copy(camera_ptr_or_struct):
    """A detached copy of a Camera (the scene keeps ownership of the original)."""
    src = camera_ptr_or_struct.contents if hasattr(camera_ptr_or_struct, "contents") else camera_ptr_or_struct
    dst = Camera()
    C.memmove(C.byref(dst), C.byref(src), C.sizeof(Camera))
    return dst
