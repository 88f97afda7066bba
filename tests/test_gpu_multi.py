"""GPU: one handle over several device entries (hip_trace_init_multi) -- replicated scene, each entry
tracing its interleaved tiles on a stream of its own, parts copied to the first device and
re-assembled there.  With one GPU in the box the entries repeat its ordinal: the copies are then
device-to-device, everything else is the path several GPUs take."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hiptrace():
    from raytracer_amd import hiptrace as ht
    return ht


@pytest.mark.parametrize("entries", [2, 3, 8])
@pytest.mark.parametrize("accel", [pytest.param(0, id="bvh"), pytest.param(1, id="octree")])
def test_multi_device_handle_renders_the_single_device_frame(host, oracle, hiptrace, entries, accel):
    w, h, depth = 272, 150, 3
    scene = host.lib.rt_scene_synth_mesh(w, h, 60, 40, 0, 5, 12)
    tree = host.lib.rt_octree_buildFromScene(scene)
    cam = host.camera_copy(scene.contents.camera)
    single = hiptrace.Tracer(w, h, depth, accel=accel)
    single.upload_scene(scene, tree)
    expect = single.frame(cam)
    multi = hiptrace.Tracer(w, h, depth, accel=accel, devices=[0] * entries)
    multi.upload_scene(scene, tree)
    for _ in range(3):       # the second frame on launches its tiles longest first
        got = multi.frame(cam, want_aux=False)
        assert np.array_equal(got["rgba8"], expect["rgba8"]), f"{(got['rgba8'] != expect['rgba8']).sum()} pixels differ"
        assert got["rays"] == expect["rays"]
        assert got["kernel_ms"] > 0 and got["total_ms"] >= got["kernel_ms"] * 0.5
    rays, ms = multi.last_frame_stats()
    assert rays == expect["rays"] and ms > 0
    assert multi.scene_info().deviceBytes == entries * single.scene_info().deviceBytes
    # and against the oracle directly
    cpu = oracle.render(scene, None, cam, depth, want=("rgba8",))
    err = np.abs(got["rgba8"].view(np.uint8).astype(np.int16) - cpu["rgba8"].view(np.uint8).astype(np.int16))
    assert err.max() <= 1
    # a moved camera, output left on the device
    import torch
    moved = host.camera_copy(scene.contents.camera)
    host.lib.rt_camera_orbit(C.byref(moved), 11, 60)
    want = single.frame(moved, want_aux=False)["rgba8"]
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    assert multi.output_bytes() == w * h * 4
    multi.frame_device(moved, out.data_ptr())
    multi.last_frame_stats()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(h, w), want)
    multi.close()
    single.close()
    host.lib.rt_octree_destroy(tree)
    host.lib.rt_scene_destroy(scene)


def test_multi_device_handle_argument_checks(host, hiptrace):
    import torch
    w, h = 64, 48
    scene = host.lib.rt_scene_init(w, h)
    cam = host.camera_copy(scene.contents.camera)
    with pytest.raises(hiptrace.HipTraceError) as e:
        hiptrace.Tracer(w, h, 2, devices=[])
    assert e.value.status == -1
    with pytest.raises(hiptrace.HipTraceError) as e:
        hiptrace.Tracer(w, h, 2, devices=[0, torch.cuda.device_count() + 3])
    assert e.value.status == -1 and "entry 1" in str(e.value)
    with pytest.raises(hiptrace.HipTraceError) as e:
        hiptrace.Tracer(w, h, 2, devices=[0, 0], rank=1, world_size=2)
    assert e.value.status == -1
    multi = hiptrace.Tracer(w, h, 2, devices=[0, 0])
    with pytest.raises(hiptrace.HipTraceError) as e:
        multi.frame(cam)
    assert e.value.status == -4                      # no scene yet
    multi.upload_scene(scene, None)
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(hiptrace.HipTraceError) as e:
        multi.frame_static(cam, out.data_ptr())
    assert e.value.status == -7
    assert torch.cuda.current_device() == 0
    one = hiptrace.Tracer(w, h, 2, devices=[0])      # a list of one is a plain handle
    one.upload_scene(scene, None)
    a = one.frame(cam)
    b = multi.frame(cam, want_aux=False)
    assert np.array_equal(a["rgba8"], b["rgba8"])
    one.close()
    multi.close()
    host.lib.rt_scene_destroy(scene)


@pytest.mark.parametrize("entries", [2, 5])
def test_multi_device_handle_returns_the_side_outputs(host, hiptrace, entries):
    """Hit mask, primitive id and hit distance of a frame split over several device entries: gathered and
    re-assembled like the colour, the same bytes as from a single-device handle."""
    w, h, depth = 200, 136, 2          # 13 x 9 tiles: a ragged right edge, rows that do not divide by the entries
    scene = host.lib.rt_scene_synth_mesh(w, h, 50, 30, 0, 3, 9)
    cam = host.camera_copy(scene.contents.camera)
    single = hiptrace.Tracer(w, h, depth)
    single.upload_scene(scene, None)
    expect = single.frame(cam)
    multi = hiptrace.Tracer(w, h, depth, devices=[0] * entries)
    multi.upload_scene(scene, None)
    for _ in range(2):
        got = multi.frame(cam)
        for k in ("rgba8", "hitMask", "primitiveId", "hitDistance"):
            assert np.array_equal(got[k].view(np.uint8), expect[k].view(np.uint8)), k
        assert got["rays"] == expect["rays"]
    assert (expect["primitiveId"] != hiptrace.NO_PRIMITIVE).any() and len(np.unique(expect["primitiveId"])) > 20
    multi.close()
    single.close()
    host.lib.rt_scene_destroy(scene)


def test_frames_enqueued_back_to_back_on_a_multi_device_handle(host, hiptrace):
    """hip_trace_frame_device returns without waiting: three frames with three cameras enqueued in a row, each into
    a buffer of its own.  A part's buffer on its device is reused by the next frame, which must wait for the copy
    out of it."""
    import torch
    w, h, depth = 640, 360, 3
    scene = host.lib.rt_scene_synth_mesh(w, h, 120, 80, 0, 6, 10)
    single = hiptrace.Tracer(w, h, depth)
    single.upload_scene(scene, None)
    cameras, expect = [], []
    for f in (0, 17, 41):
        cam = host.camera_copy(scene.contents.camera)
        host.lib.rt_camera_orbit(C.byref(cam), f, 60)
        cameras.append(cam)
        expect.append(single.frame(cam, want_aux=False))
    assert not np.array_equal(expect[0]["rgba8"], expect[1]["rgba8"]) and not np.array_equal(expect[1]["rgba8"], expect[2]["rgba8"])
    multi = hiptrace.Tracer(w, h, depth, devices=[0, 0, 0, 0])
    multi.upload_scene(scene, None)
    outs = [torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda") for _ in cameras]
    torch.cuda.synchronize()
    for _ in range(4):
        for cam, out in zip(cameras, outs):
            multi.frame_device(cam, out.data_ptr())
    rays, _ = multi.last_frame_stats()       # waits for the last frame
    torch.cuda.synchronize()
    for i, (out, want) in enumerate(zip(outs, expect)):
        got = out.cpu().numpy().view(np.uint32).reshape(h, w)
        assert np.array_equal(got, want["rgba8"]), f"frame {i}: {(got != want['rgba8']).sum()} pixels differ"
    assert rays == expect[-1]["rays"]
    multi.close()
    single.close()
    host.lib.rt_scene_destroy(scene)


def test_last_frame_is_the_handles_own_buffer(host, hiptrace):
    """hip_trace_last_frame: where hip_trace_frame (and frame_device without a buffer) left the frame on the device."""
    import torch
    w, h = 96, 64
    scene = host.lib.rt_scene_init(w, h)
    cam = host.camera_copy(scene.contents.camera)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for devices in (None, [0, 0, 0]):
        tr = hiptrace.Tracer(w, h, 2, devices=devices)
        with pytest.raises(hiptrace.HipTraceError) as e:
            tr.last_frame()
        assert e.value.status == -4
        tr.upload_scene(scene, None)
        expect = tr.frame(cam, want_aux=False)["rgba8"]
        got = np.zeros((h, w), np.uint32)
        assert hip.hipMemcpy(got.ctypes.data, tr.last_frame(), w * h * 4, 2) == 0      # 2: device to host
        assert np.array_equal(got, expect)
        tr.frame_device(cam)                  # no buffer given: the handle's own again
        tr.last_frame_stats()
        got[:] = 0
        assert hip.hipMemcpy(got.ctypes.data, tr.last_frame(), w * h * 4, 2) == 0
        assert np.array_equal(got, expect)
        out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
        tr.frame_device(cam, out.data_ptr())
        tr.last_frame_stats()
        with pytest.raises(hiptrace.HipTraceError) as e:
            tr.last_frame()                   # the last frame went to the caller's buffer
        assert e.value.status == -4
        tr.close()
    host.lib.rt_scene_destroy(scene)
