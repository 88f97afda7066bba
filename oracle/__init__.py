"""ctypes binding of the CPU parity oracle (test infrastructure, not product).

liboracle.so: this repository's restatement (cpu_trace.c).
_ref/libref.so: the reference's own C files compiled unmodified (optional).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd.types import (BoundingBox, Camera, Octree, Plane, Ray, Scene, Sphere, Triangle, Vec3)

_HERE = os.path.dirname(os.path.abspath(__file__))
NO_PRIMITIVE = 0xFFFFFFFF
WALK_REFERENCE, WALK_PRUNED = 0, 1
JITTER_ZERO, JITTER_HASHED = 0, 1
SHADING_CPU, SHADING_KERNEL = 0, 1


class Options(C.Structure):
    _fields_ = [("maxDepth", C.c_uint32), ("walk", C.c_uint32), ("boxPad", C.c_float), ("threads", C.c_int32),
                ("jitter", C.c_uint32), ("shadingModel", C.c_uint32), ("raysPerPixel", C.c_uint32)]


class Window(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("x0", "y0", "stepX", "stepY", "countX", "countY")]


COUNTER_NAMES = ("primary", "shadow", "reflection", "refraction", "boxTests", "nodesVisited",
                 "planeTests", "sphereTests", "triangleTests", "triangleDistance", "triangleInside",
                 "lightEvaluations")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


class Frame(C.Structure):
    _fields_ = [("rgba8", C.POINTER(C.c_uint32)), ("rgb", C.POINTER(C.c_float)),
                ("hitMask", C.POINTER(C.c_uint8)), ("primitiveId", C.POINTER(C.c_uint32)),
                ("hitDistance", C.POINTER(C.c_float)), ("counters", Counters)]


def _load_oracle():
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `make -C oracle`")
    lib = C.CDLL(path)
    P = C.POINTER
    lib.oracle_render.restype = C.c_int
    lib.oracle_render.argtypes = [P(Scene), P(Octree), P(Camera), P(Options), P(Window), P(Frame)]
    lib.oracle_shading_order.restype = C.c_long
    lib.oracle_shading_order.argtypes = [P(Scene), P(Octree), P(Camera), P(Options), P(Window), P(C.c_uint32),
                                         P(C.c_uint32), C.c_size_t]
    lib.oracle_jitter_hash.restype = C.c_uint32
    lib.oracle_jitter_hash.argtypes = [C.c_uint32] * 4
    lib.oracle_pack_rgba8.restype = C.c_uint32
    lib.oracle_pack_rgba8.argtypes = [C.c_float] * 3
    for name, rec in (("plane", Plane), ("sphere", Sphere), ("triangle", Triangle)):
        fn = getattr(lib, f"oracle_intersect_{name}")
        fn.restype = C.c_int
        fn.argtypes = [P(rec), P(Ray), P(C.c_float)]
    lib.oracle_intersect_box.restype = C.c_int
    lib.oracle_intersect_box.argtypes = [P(BoundingBox), P(Ray)]
    lib.oracle_fresnel.restype = C.c_float
    lib.oracle_fresnel.argtypes = [Vec3, Vec3, C.c_float]
    lib.oracle_refract.restype = Vec3
    lib.oracle_refract.argtypes = [Vec3, Vec3, C.c_float]
    lib.oracle_reflect.restype = Vec3
    lib.oracle_reflect.argtypes = [Vec3, Vec3]
    lib.oracle_primary_ray.restype = Ray
    lib.oracle_primary_ray.argtypes = [P(Camera), C.c_uint32, C.c_uint32]
    return lib


lib = _load_oracle()


def _load_ref():
    path = os.path.join(_HERE, "_ref", "libref.so")
    if not os.path.exists(path):
        return None
    ref = C.CDLL(path)
    P = C.POINTER
    ref.ref_render.restype = C.c_int
    ref.ref_render.argtypes = [P(Scene), P(Camera), C.c_uint32, P(Window), P(C.c_float), P(C.c_uint8),
                               P(C.c_uint32), P(C.c_float)]
    ref.ref_render_jittered.restype = C.c_int
    ref.ref_render_jittered.argtypes = [P(Scene), P(Camera), C.c_uint32, P(Window), P(C.c_uint32), P(C.c_uint32),
                                        P(C.c_float), P(C.c_uint32)]
    ref.ref_scene_init.restype = P(Scene)
    ref.ref_scene_init.argtypes = [C.c_uint32, C.c_uint32]
    ref.ref_scene_destroy.argtypes = [P(Scene)]
    ref.ref_scene_destroy.restype = None
    ref.ref_octree_build.restype = P(Octree)
    ref.ref_octree_build.argtypes = [P(Scene)]
    ref.ref_octree_build_safe.restype = P(Octree)
    ref.ref_octree_build_safe.argtypes = [P(Scene), C.c_uint32]
    ref.ref_octree_destroy.argtypes = [P(Octree)]
    ref.ref_octree_destroy.restype = None
    ref.ref_camera_setup.argtypes = [P(Camera)]
    ref.ref_camera_setup.restype = None
    ref.ref_camera_create.restype = P(Camera)
    ref.ref_camera_create.argtypes = [Vec3, Vec3, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
    ref.ref_move_camera.argtypes = [P(Camera), C.c_int, C.c_int, C.c_int]
    ref.ref_move_camera.restype = None
    ref.ref_raycast.restype = Vec3
    ref.ref_raycast.argtypes = [P(Scene), P(Ray), C.c_uint32]
    ref.ref_primary.restype = Ray
    ref.ref_primary.argtypes = [P(Camera), C.c_uint32, C.c_uint32]
    for name, rec in (("plane", Plane), ("sphere", Sphere), ("triangle", Triangle)):
        fn = getattr(ref, f"ref_intersect_{name}")
        fn.restype = C.c_int
        fn.argtypes = [P(rec), P(Ray), P(C.c_float)]
    ref.ref_fresnel.restype = C.c_float
    ref.ref_fresnel.argtypes = [Vec3, Vec3, C.c_float]
    ref.ref_refract.restype = Vec3
    ref.ref_refract.argtypes = [Vec3, Vec3, C.c_float]
    ref.ref_reflect.restype = Vec3
    ref.ref_reflect.argtypes = [Vec3, Vec3]
    ref.ref_clamp.restype = C.c_float
    ref.ref_clamp.argtypes = [C.c_float] * 3
    # the reference's OBJ loader and its bulk edits (reference: src/object.h:14-18), as they are
    ref.object_loadFromFile.restype = C.c_void_p
    ref.object_loadFromFile.argtypes = [C.c_char_p]
    ref.object_scale.argtypes = [C.c_void_p, C.c_float]
    ref.object_scale.restype = None
    ref.object_translate.argtypes = [C.c_void_p, Vec3]
    ref.object_translate.restype = None
    ref.object_materialIndex.argtypes = [C.c_void_p, C.c_uint32]
    ref.object_materialIndex.restype = None
    ref.object_destroy.argtypes = [C.c_void_p]
    ref.object_destroy.restype = None
    # the reference's frame container and bitmap writer (reference: src/utils/image.h:14-15,40)
    ref.image_create.restype = C.c_void_p
    ref.image_create.argtypes = [C.c_uint32, C.c_uint32]
    ref.image_destroy.argtypes = [C.c_void_p]
    ref.image_destroy.restype = None
    ref.bitmap_save_image.restype = C.c_bool
    ref.bitmap_save_image.argtypes = [C.c_char_p, C.c_void_p]
    return ref


ref = _load_ref()


def full_window(camera):
    return Window(0, 0, 1, 1, camera.width, camera.height)


def render(scene, octree, camera, max_depth, window=None, walk=WALK_PRUNED, box_pad=0.0, threads=0,
           want=("rgba8", "rgb", "hitMask", "primitiveId", "hitDistance"), jitter=JITTER_ZERO,
           shading=SHADING_CPU, rays_per_pixel=1):
    """Runs the CPU oracle.  scene/octree are ctypes pointers (octree None = brute force),
    camera a Camera struct.  Returns a dict of numpy arrays shaped (countY, countX[, 3]) plus
    'counters'."""
    win = window or full_window(camera)
    n = win.countX * win.countY
    out = {}
    frame = Frame()
    if "rgba8" in want:
        out["rgba8"] = np.zeros(n, np.uint32)
        frame.rgba8 = out["rgba8"].ctypes.data_as(C.POINTER(C.c_uint32))
    if "rgb" in want:
        out["rgb"] = np.zeros(3 * n, np.float32)
        frame.rgb = out["rgb"].ctypes.data_as(C.POINTER(C.c_float))
    if "hitMask" in want:
        out["hitMask"] = np.zeros(n, np.uint8)
        frame.hitMask = out["hitMask"].ctypes.data_as(C.POINTER(C.c_uint8))
    if "primitiveId" in want:
        out["primitiveId"] = np.zeros(n, np.uint32)
        frame.primitiveId = out["primitiveId"].ctypes.data_as(C.POINTER(C.c_uint32))
    if "hitDistance" in want:
        out["hitDistance"] = np.zeros(n, np.float32)
        frame.hitDistance = out["hitDistance"].ctypes.data_as(C.POINTER(C.c_float))
    opts = Options(max_depth, walk, box_pad, threads, jitter, shading, rays_per_pixel)
    rc = lib.oracle_render(scene, octree, C.byref(camera), C.byref(opts), C.byref(win), C.byref(frame))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed: {rc}")
    for k in list(out):
        shape = (win.countY, win.countX, 3) if k == "rgb" else (win.countY, win.countX)
        out[k] = out[k].reshape(shape)
    out["counters"] = frame.counters.as_dict()
    return out


def shading_order(scene, octree, camera, max_depth, window=None):
    """Per pixel of the window, the tree positions of its shading points in the reference's shading
    order.  Returns (offsets uint32[n + 1], positions uint32[total])."""
    win = window or full_window(camera)
    n = win.countX * win.countY
    offsets = np.zeros(n + 1, np.uint32)
    opts = Options(max_depth, WALK_PRUNED, 0.0, 1, JITTER_ZERO, SHADING_CPU, 1)
    capacity = n * ((1 << max_depth) - 1)
    positions = np.zeros(max(capacity, 1), np.uint32)
    total = lib.oracle_shading_order(scene, octree, C.byref(camera), C.byref(opts), C.byref(win),
                                     offsets.ctypes.data_as(C.POINTER(C.c_uint32)),
                                     positions.ctypes.data_as(C.POINTER(C.c_uint32)), capacity)
    if total < 0 or total > capacity:
        raise RuntimeError(f"oracle_shading_order failed: {total}")
    return offsets, positions[:total].copy()


def ref_render_jittered(scene, camera, max_depth, window=None):
    """The compiled reference's raytracer_raycast with the hashed jitter stream at its
    random_bilateral() seam.  Returns (rgb (countY, countX, 3), pixels whose draw count disagreed
    with the oracle's ray-tree shape)."""
    if ref is None:
        raise RuntimeError("oracle/_ref/libref.so not built")
    win = window or full_window(camera)
    offsets, positions = shading_order(scene, None, camera, max_depth, win)
    if positions.size == 0:
        positions = np.zeros(1, np.uint32)
    rgb = np.zeros(3 * win.countX * win.countY, np.float32)
    bad = C.c_uint32(0)
    cam = Camera()
    C.memmove(C.byref(cam), C.byref(camera), C.sizeof(Camera))
    rc = ref.ref_render_jittered(scene, C.byref(cam), max_depth, C.byref(win),
                                 offsets.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 positions.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 rgb.ctypes.data_as(C.POINTER(C.c_float)), C.byref(bad))
    if rc != 0:
        raise RuntimeError("ref_render_jittered failed")
    return rgb.reshape(win.countY, win.countX, 3), int(bad.value)


def ref_render(scene, camera, max_depth, window=None, want=("rgb", "hitMask", "primitiveId", "hitDistance")):
    """Runs the compiled reference (raytracer_raycast + recording closest-hit).  Brute force only."""
    if ref is None:
        raise RuntimeError("oracle/_ref/libref.so not built")
    win = window or full_window(camera)
    n = win.countX * win.countY
    out = {}
    ptr = {"rgb": None, "hitMask": None, "primitiveId": None, "hitDistance": None}
    if "rgb" in want:
        out["rgb"] = np.zeros(3 * n, np.float32)
        ptr["rgb"] = out["rgb"].ctypes.data_as(C.POINTER(C.c_float))
    if "hitMask" in want:
        out["hitMask"] = np.zeros(n, np.uint8)
        ptr["hitMask"] = out["hitMask"].ctypes.data_as(C.POINTER(C.c_uint8))
    if "primitiveId" in want:
        out["primitiveId"] = np.zeros(n, np.uint32)
        ptr["primitiveId"] = out["primitiveId"].ctypes.data_as(C.POINTER(C.c_uint32))
    if "hitDistance" in want:
        out["hitDistance"] = np.zeros(n, np.float32)
        ptr["hitDistance"] = out["hitDistance"].ctypes.data_as(C.POINTER(C.c_float))
    cam = Camera()
    C.memmove(C.byref(cam), C.byref(camera), C.sizeof(Camera))
    rc = ref.ref_render(scene, C.byref(cam), max_depth, C.byref(win), ptr["rgb"], ptr["hitMask"],
                        ptr["primitiveId"], ptr["hitDistance"])
    if rc != 0:
        raise RuntimeError("ref_render failed")
    for k in list(out):
        shape = (win.countY, win.countX, 3) if k == "rgb" else (win.countY, win.countX)
        out[k] = out[k].reshape(shape)
    return out


def pack_rgba8(rgb):
    """float RGB (..., 3) -> uint32 0xAABBGGRR with the oracle's conversion."""
    c = np.asarray(rgb, np.float32)
    c = np.where(c < 0, np.float32(0), np.where(c > 1, np.float32(1), c))
    c = np.where(np.isnan(c), np.float32(0), c)
    b = np.rint(c * np.float32(255.0)).astype(np.uint32)
    return b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16) | np.uint32(0xFF000000)
