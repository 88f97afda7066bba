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
    with pytest.raises(hiptrace.HipTraceError) as e:
        multi.frame(cam, want_aux=True)              # per-pixel side outputs need a single-device handle
    assert e.value.status == -7
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
