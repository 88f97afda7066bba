"""Python binding of libhiptrace.so, the hip_trace C ABI (include/hip_trace.h).

Mirrors the reference's device host layer for the trace path (reference:
src/gpu.h:60-66): init / upload scene / render frame / destroy.  There is no CPU
fallback: importing this module without the built library, or rendering without a
GPU, raises.
"""
import ctypes as C
import os

import numpy as np

from .types import Camera, Octree, Scene

# RT_HIPTRACE_LIB selects another build of the same ABI (the profiling build with walk counters).
_LIB_PATH = os.environ.get("RT_HIPTRACE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                               "libhiptrace.so")

NO_PRIMITIVE = 0xFFFFFFFF
JITTER_ZERO, JITTER_HASHED = 0, 1
SHADING_CPU_MODEL, SHADING_KERNEL_MODEL = 0, 1
ACCEL_PRIVATE_BVH = 0
ACCEL_CALLER_OCTREE = 1
LIGHT_GRID_AUTO, LIGHT_GRID_OFF = 0, 1
TILE = 16
MAX_DEPTH = 8
STATIC_MAX_TILES = 131072

STATUS_NAMES = {0: "OK", -1: "ERR_ARGUMENT", -2: "ERR_DEVICE", -3: "ERR_MEMORY", -4: "ERR_NO_SCENE",
                -5: "ERR_TREE", -6: "ERR_DIMENSION", -7: "ERR_UNSUPPORTED"}

EXPORTS = ("hip_trace_init", "hip_trace_init_multi", "hip_trace_upload_scene", "hip_trace_frame", "hip_trace_frame_device", "hip_trace_frame_static",
           "hip_trace_last_frame", "hip_trace_output_bytes", "hip_trace_untile", "hip_trace_last_frame_stats", "hip_trace_set_stream",
           "hip_trace_scene_info", "hip_trace_walk_stats", "hip_trace_destroy", "hip_trace_last_error", "hip_trace_tile_owner",
           "hip_trace_check_hierarchy", "hip_trace_jitter_hash")


class Options(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("maxDepth", C.c_uint32),
                ("raysPerPixel", C.c_uint32), ("shadingModel", C.c_uint32), ("jitterMode", C.c_uint32),
                ("device", C.c_int32), ("rank", C.c_uint32), ("worldSize", C.c_uint32),
                ("accel", C.c_uint32), ("lightGrid", C.c_uint32), ("reserved", C.c_uint32 * 5)]


class RayCounts(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("reflection", C.c_uint64),
                ("refraction", C.c_uint64)]

    def total(self):
        return int(self.primary + self.shadow + self.reflection + self.refraction)

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Aux(C.Structure):
    _fields_ = [("hitMask", C.POINTER(C.c_uint8)), ("primitiveId", C.POINTER(C.c_uint32)),
                ("hitDistance", C.POINTER(C.c_float)), ("rays", RayCounts),
                ("kernelMs", C.c_float), ("totalMs", C.c_float)]


class SceneInfo(C.Structure):
    _fields_ = [("wideNodeCount", C.c_uint32), ("leafCount", C.c_uint32), ("leafIndexCount", C.c_uint32),
                ("treeLevels", C.c_uint32), ("deviceBytes", C.c_uint64), ("lightGridBytes", C.c_uint64),
                ("lightGridReferences", C.c_uint64), ("lightGridLights", C.c_uint32), ("lightGridPeakMiB", C.c_uint32)]


class HierarchyReport(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("wideNodeCount", "leafCount", "leafIndexCount", "treeLevels", "childCount",
                                          "unreachedSpheres", "unreachedTriangles", "duplicateReferences",
                                          "containmentChecked", "escapedPrimitives", "badLinks", "deepestLevel", "checksum")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class HipTraceError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def _load(path=None):
    path = path or _LIB_PATH
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`; "
                          "there is no CPU fallback for the trace path")
    lib = C.CDLL(path)
    P, H = C.POINTER, C.c_void_p
    lib.hip_trace_init.restype = C.c_int
    lib.hip_trace_init.argtypes = [P(H), P(Options)]
    lib.hip_trace_init_multi.restype = C.c_int
    lib.hip_trace_init_multi.argtypes = [P(H), P(Options), P(C.c_int32), C.c_uint32]
    lib.hip_trace_upload_scene.restype = C.c_int
    lib.hip_trace_upload_scene.argtypes = [H, P(Scene), P(Octree)]
    lib.hip_trace_frame.restype = C.c_int
    lib.hip_trace_frame.argtypes = [H, P(Camera), C.c_void_p, P(Aux)]
    lib.hip_trace_frame_device.restype = C.c_int
    lib.hip_trace_frame_device.argtypes = [H, P(Camera), C.c_void_p]
    lib.hip_trace_frame_static.restype = C.c_int
    lib.hip_trace_frame_static.argtypes = [H, P(Camera), C.c_void_p]
    lib.hip_trace_last_frame.restype = C.c_int
    lib.hip_trace_last_frame.argtypes = [H, P(C.c_void_p)]
    lib.hip_trace_output_bytes.restype = C.c_size_t
    lib.hip_trace_output_bytes.argtypes = [H]
    lib.hip_trace_untile.restype = C.c_int
    lib.hip_trace_untile.argtypes = [H, C.c_void_p, C.c_void_p]
    lib.hip_trace_last_frame_stats.restype = C.c_int
    lib.hip_trace_last_frame_stats.argtypes = [H, P(RayCounts), P(C.c_float)]
    lib.hip_trace_set_stream.restype = C.c_int
    lib.hip_trace_set_stream.argtypes = [H, C.c_void_p]
    lib.hip_trace_scene_info.restype = C.c_int
    lib.hip_trace_scene_info.argtypes = [H, P(SceneInfo)]
    lib.hip_trace_walk_stats.restype = C.c_int
    lib.hip_trace_walk_stats.argtypes = [H, P(C.c_uint64), C.c_uint32]
    lib.hip_trace_destroy.restype = None
    lib.hip_trace_destroy.argtypes = [H]
    lib.hip_trace_last_error.restype = C.c_char_p
    lib.hip_trace_last_error.argtypes = [H]
    lib.hip_trace_tile_owner.restype = C.c_uint32
    lib.hip_trace_tile_owner.argtypes = [C.c_uint32] * 4
    lib.hip_trace_jitter_hash.restype = C.c_uint32
    lib.hip_trace_jitter_hash.argtypes = [C.c_uint32] * 4
    lib.hip_trace_check_hierarchy.restype = C.c_int
    lib.hip_trace_check_hierarchy.argtypes = [P(Scene), P(Octree), C.c_uint32, P(HierarchyReport), C.c_char_p, C.c_size_t]
    return lib


lib = _load()
STATS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libhiptrace_stats.so")


def load_library(path):
    """Another build of the same ABI (e.g. STATS_LIB_PATH, the build with walk counters), to be passed
    to Tracer(library=...)."""
    return _load(path)


class Tracer:
    """One hip_trace handle.  scene/octree are ctypes pointers (raytracer_amd.host builds them)."""

    def __init__(self, width, height, max_depth, device=-1, rank=0, world_size=1, rays_per_pixel=1,
                 accel=ACCEL_PRIVATE_BVH, jitter=JITTER_ZERO, shading=SHADING_CPU_MODEL, library=None, devices=None,
                 light_grid=LIGHT_GRID_AUTO):
        """devices: a list of HIP ordinals makes one handle over several devices (hip_trace_init_multi)."""
        self._lib = library or lib
        self._h = C.c_void_p()
        opt = Options(width=width, height=height, maxDepth=max_depth, raysPerPixel=rays_per_pixel,
                      shadingModel=shading, jitterMode=jitter, device=device, rank=rank, worldSize=world_size, accel=accel,
                      lightGrid=light_grid)
        if devices is not None:
            ordinals = (C.c_int32 * len(devices))(*devices)
            rc = self._lib.hip_trace_init_multi(C.byref(self._h), C.byref(opt), ordinals, len(devices))
        else:
            rc = self._lib.hip_trace_init(C.byref(self._h), C.byref(opt))
        if rc != 0:
            raise HipTraceError(rc, self._lib.hip_trace_last_error(None).decode())
        self.width, self.height, self.max_depth = width, height, max_depth
        self.rank, self.world_size = rank, world_size

    def _check(self, rc):
        if rc != 0:
            raise HipTraceError(rc, self._lib.hip_trace_last_error(self._h).decode())

    def upload_scene(self, scene, octree):
        self._check(self._lib.hip_trace_upload_scene(self._h, scene, octree))

    def scene_info(self):
        info = SceneInfo()
        self._check(self._lib.hip_trace_scene_info(self._h, C.byref(info)))
        return info

    def frame(self, camera, want_aux=True, want_image=True):
        """Renders one frame to host memory.  Returns a dict: rgba8 (H,W) uint32 and, with
        want_aux, hitMask / primitiveId / hitDistance, plus rays, kernel_ms, total_ms."""
        n = self.width * self.height
        out = {}
        aux = Aux()
        img_ptr = None
        if want_image:
            out["rgba8"] = np.zeros(n, np.uint32)
            img_ptr = out["rgba8"].ctypes.data_as(C.c_void_p)
        if want_aux:
            out["hitMask"] = np.zeros(n, np.uint8)
            out["primitiveId"] = np.zeros(n, np.uint32)
            out["hitDistance"] = np.zeros(n, np.float32)
            aux.hitMask = out["hitMask"].ctypes.data_as(C.POINTER(C.c_uint8))
            aux.primitiveId = out["primitiveId"].ctypes.data_as(C.POINTER(C.c_uint32))
            aux.hitDistance = out["hitDistance"].ctypes.data_as(C.POINTER(C.c_float))
        self._check(self._lib.hip_trace_frame(self._h, C.byref(camera), img_ptr, C.byref(aux)))
        for k in list(out):
            out[k] = out[k].reshape(self.height, self.width)
        out["rays"] = aux.rays.as_dict()
        out["kernel_ms"] = float(aux.kernelMs)
        out["total_ms"] = float(aux.totalMs)
        return out

    def frame_device(self, camera, device_ptr=None):
        """Enqueues one frame; output stays on the device (device_ptr: int address or None
        for the handle's own buffer)."""
        self._check(self._lib.hip_trace_frame_device(self._h, C.byref(camera), C.c_void_p(device_ptr) if device_ptr else None))

    def frame_static(self, camera, device_ptr):
        """frame_device through the entry point that allocates nothing (scratch in the library's
        static device array); device_ptr is required."""
        self._check(self._lib.hip_trace_frame_static(self._h, C.byref(camera), C.c_void_p(device_ptr) if device_ptr else None))

    def last_frame(self):
        """Device address of the handle's own copy of the most recent frame (hip_trace_last_frame)."""
        ptr = C.c_void_p()
        self._check(self._lib.hip_trace_last_frame(self._h, C.byref(ptr)))
        return int(ptr.value)

    def output_bytes(self):
        return int(self._lib.hip_trace_output_bytes(self._h))

    def untile(self, gathered_ptr, frame_ptr):
        self._check(self._lib.hip_trace_untile(self._h, C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr)))

    def last_frame_stats(self):
        rays, ms = RayCounts(), C.c_float()
        self._check(self._lib.hip_trace_last_frame_stats(self._h, C.byref(rays), C.byref(ms)))
        return rays.as_dict(), float(ms.value)

    def walk_stats(self):
        """Walk counters of the last frame (profiling build only; empty dict otherwise)."""
        buf = (C.c_uint64 * 64)()
        n = self._lib.hip_trace_walk_stats(self._h, buf, 64)
        if n < 0:
            self._check(n)
        names = ("wideNodesTested", "leavesOpened", "triangleDistanceTests", "triangleEdgeTests",
                 "waveNodeSteps", "waveLeafPhases", "waveFilterIterations", "waveExactIterations") + tuple(
                     f"shadowStepsLog2_{b}" for b in range(8)) + (
                     "childBoxesTested", "cyclesClosestHit", "cyclesShadowPass", "cyclesShading", "cyclesBarrier",
                     "shadowNodeSteps", "stepLanesWalking", "stepLanesHolding", "stepLanesWaiting", "stepLanesFinished",
                     "gridListTurns", "gridListLanes",
                     "waveSharedNodeSteps", "sharedUsefulLanes", "waveSharedTriangles", "nodesShelved", "waveDrains", "waveSharedLeaves")
        return {names[i]: int(buf[i]) for i in range(min(n, len(names)))}

    def set_stream(self, stream_ptr):
        self._check(self._lib.hip_trace_set_stream(self._h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def close(self):
        if self._h:
            self._lib.hip_trace_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check_hierarchy(scene, octree, accel=ACCEL_PRIVATE_BVH):
    """Host only (no GPU): builds the hierarchy an upload would build and returns its report."""
    report = HierarchyReport()
    message = C.create_string_buffer(256)
    rc = lib.hip_trace_check_hierarchy(scene, octree, accel, C.byref(report), message, len(message))
    if rc != 0:
        raise HipTraceError(rc, message.value.decode())
    return report.as_dict()


def tile_owner(tx, ty, tiles_x, world_size):
    return int(lib.hip_trace_tile_owner(tx, ty, tiles_x, world_size))
