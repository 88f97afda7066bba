"""The reference's own device entry points (reference: src/gpu.h:60-66) served by libgpuhip.so:
exports and argument checking on the CPU, a frame through gpu_renderScene against the oracle on
the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracer_amd.types import Octree, Scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Image(C.Structure):
    """reference: src/utils/image.h:8-12"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("buffer", C.POINTER(C.c_uint32)),
                ("bufferSize", C.c_uint32)]


@pytest.fixture(scope="module")
def shim():
    from raytracer_amd import hiptrace  # noqa: F401  (torch and libhiptrace first)
    lib = C.CDLL(os.path.join(ROOT, "raytracer_amd", "lib", "libgpuhip.so"))
    lib.gpu_initContext.restype = C.c_void_p
    lib.gpu_initContext.argtypes = [C.POINTER(Scene), C.POINTER(Octree), C.c_uint32]
    lib.gpu_renderScene.restype = None
    lib.gpu_renderScene.argtypes = [C.c_void_p, C.POINTER(Scene), C.POINTER(Image)]
    lib.gpu_destroyContext.restype = None
    lib.gpu_destroyContext.argtypes = [C.c_void_p]
    lib.gpu_updateDimensions.restype = None
    lib.gpu_updateDimensions.argtypes = [C.c_void_p] + [C.c_uint32] * 4
    lib.gpu_hip_configure.restype = None
    lib.gpu_hip_configure.argtypes = [C.c_uint32, C.c_uint32, C.c_int32]
    lib.gpu_hip_lastError.restype = C.c_char_p
    lib.gpu_hip_lastFrameStats.restype = C.c_int
    return lib


def test_shim_exports_and_rejects_null(shim):
    import re
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gpu_hip.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(gpu_[A-Za-z0-9_]+)\s*\(", text)))
    assert {"gpu_initContext", "gpu_renderScene", "gpu_destroyContext", "gpu_updateDimensions"} <= set(names)
    for name in names:
        assert hasattr(shim, name), name
    assert shim.gpu_initContext(None, None, 1) is None
    assert b"NULL" in shim.gpu_hip_lastError()
    shim.gpu_destroyContext(None)
    shim.gpu_updateDimensions(None, 1, 2, 3, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [0, 1])
def test_frame_through_the_reference_entry_points(shim, host, oracle, accel):
    w, h = 200, 120
    scene = host.lib.rt_scene_synth_mesh(w, h, 40, 30, 0, 3, 6)
    tree = host.lib.rt_octree_buildFromScene(scene)
    shim.gpu_hip_configure(0, accel, -1)             # depth 0 -> the reference kernel's 5
    ctx = shim.gpu_initContext(scene, tree, 1)
    assert ctx, shim.gpu_hip_lastError()
    pixels = np.zeros(w * h, np.uint32)
    image = Image(w, h, pixels.ctypes.data_as(C.POINTER(C.c_uint32)), w * h * 4)
    shim.gpu_renderScene(ctx, scene, C.byref(image))
    assert shim.gpu_hip_lastError() == b""
    cpu = oracle.render(scene, None, host.camera_copy(scene.contents.camera), 5, want=("rgba8",))
    err = np.abs(pixels.view(np.uint8).astype(np.int16) - cpu["rgba8"].reshape(-1).view(np.uint8).astype(np.int16))
    assert (err.reshape(-1, 4).max(axis=1) <= 1).mean() >= 0.999, f"max error {err.max()}/255"
    shim.gpu_renderScene(ctx, scene, None)           # frame stays on the device (reference: main.c:181)
    assert shim.gpu_hip_lastError() == b""
    wrong = Image(w + 1, h, pixels.ctypes.data_as(C.POINTER(C.c_uint32)), w * h * 4)
    shim.gpu_renderScene(ctx, scene, C.byref(wrong))
    assert b"image size" in shim.gpu_hip_lastError()
    shim.gpu_destroyContext(ctx)
    assert shim.gpu_initContext(scene, tree, 0) is None     # the reference asserts raysPerPixel > 0 (src/gpu.c:447)
    assert b"raysPerPixel" in shim.gpu_hip_lastError()
    four = shim.gpu_initContext(scene, tree, 4)             # four rays per pixel, as src/main.c:27 may ask
    assert four, shim.gpu_hip_lastError()
    shim.gpu_renderScene(four, scene, C.byref(image))
    assert shim.gpu_hip_lastError() == b""
    soft = oracle.render(scene, None, host.camera_copy(scene.contents.camera), 5, want=("rgba8",), rays_per_pixel=4)
    err = np.abs(pixels.view(np.uint8).astype(np.int16) - soft["rgba8"].reshape(-1).view(np.uint8).astype(np.int16))
    assert (err.reshape(-1, 4).max(axis=1) <= 1).mean() >= 0.999, f"max error {err.max()}/255"
    shim.gpu_destroyContext(four)
    shim.gpu_hip_configure(0, 0, -1)


@pytest.mark.gpu
def test_screenshot_path(shim, host, tmp_path):
    """What the reference's main loop does for a screenshot (reference: src/main.c:168-176): render into
    an Image, write it as a bitmap; the file holds exactly the rendered pixels, bottom row first."""
    from raytracer_amd.types import Image as HostImage
    w, h = 96, 54
    scene = host.lib.rt_scene_init(w, h)
    tree = host.lib.rt_octree_buildFromScene(scene)
    shim.gpu_hip_configure(0, 0, -1)
    ctx = shim.gpu_initContext(scene, tree, 1)
    assert ctx, shim.gpu_hip_lastError()
    image = host.lib.rt_image_create(w, h)
    shim.gpu_renderScene(ctx, scene, C.cast(image, C.POINTER(Image)))
    assert shim.gpu_hip_lastError() == b""
    path = tmp_path / "screenshot.bmp"
    assert host.lib.rt_bitmap_save_image(str(path).encode(), image)
    data = np.frombuffer(path.read_bytes(), dtype=np.uint8)
    assert len(data) == 54 + w * h * 4
    rows = data[54:].view(np.uint32).reshape(h, w)[::-1]
    pixels = np.frombuffer(C.string_at(image.contents.buffer, w * h * 4), dtype=np.uint32).reshape(h, w)
    swapped = (pixels & 0xFF00FF00) | ((pixels >> 16) & 0xFF) | ((pixels & 0xFF) << 16)
    assert np.array_equal(rows, swapped)
    assert len(np.unique(pixels)) > 16, "the frame is not a picture"
    assert C.sizeof(HostImage) == C.sizeof(Image)
    host.lib.rt_image_destroy(image)
    shim.gpu_destroyContext(ctx)
    host.lib.rt_octree_destroy(tree)
    host.lib.rt_scene_destroy(scene)
