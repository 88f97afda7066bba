"""GPU: hip_trace_frame_static, the entry point that allocates nothing, against
hip_trace_frame_device -- same bytes, same ray counters -- on two calls in a row, the second on a
smaller frame (all handles of a device share one static scratch array, which keeps the previous
call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def render_both(host, ht, torch, scene, w, h, depth, frames=3, **options):
    cam = host.camera_copy(scene.contents.camera)
    tr = ht.Tracer(w, h, depth, **options)
    tr.upload_scene(scene, None)
    nbytes = tr.output_bytes()
    expect = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    got = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device="cuda")
    tr.frame_device(cam, expect.data_ptr())
    rays_expect, _ = tr.last_frame_stats()
    results = []
    for _ in range(frames):          # from the second frame on, small frames are launched longest tile first
        got.fill_(0x5A)
        torch.cuda.synchronize()
        tr.frame_static(cam, got.data_ptr())
        rays, ms = tr.last_frame_stats()
        torch.cuda.synchronize()
        results.append((got.cpu().numpy().copy(), rays, ms))
    want = expect.cpu().numpy()
    tr.close()
    return want, rays_expect, results


def test_static_entry_point_equals_frame_device_twice_in_a_row_second_call_smaller(host):
    import torch
    from raytracer_amd import hiptrace as ht
    assert torch.cuda.is_available()
    for (w, h, depth) in ((640, 360, 3), (200, 120, 2)):
        scene = host.lib.rt_scene_synth_mesh(w, h, 120, 60, 0, 5, 20)
        want, rays_expect, results = render_both(host, ht, torch, scene, w, h, depth)
        assert rays_expect["primary"] == w * h
        assert len(np.unique(want.view(np.uint32))) > 50, "the expected frame is blank: the test would prove nothing"
        for got, rays, ms in results:
            assert np.array_equal(got, want), f"{w}x{h}: {(got != want).sum()} bytes differ"
            assert rays == rays_expect
            assert ms > 0
        host.lib.rt_scene_destroy(scene)


def test_static_entry_point_on_tile_ranks_and_interleaved_with_the_allocating_one(host):
    """Two ranks of a split frame (tile-major output) through the static entry point, alternating with
    hip_trace_frame_device on the same handles, and its argument checks."""
    import torch
    from raytracer_amd import hiptrace as ht
    w, h = 256, 144
    scene = host.lib.rt_scene_synth_mesh(w, h, 60, 30, 0, 3, 8)
    cam = host.camera_copy(scene.contents.camera)
    ranks = [ht.Tracer(w, h, 3, rank=r, world_size=2) for r in range(2)]
    for tr in ranks:
        tr.upload_scene(scene, None)
    nbytes = ranks[0].output_bytes()
    for round_ in range(3):
        for tr in ranks:
            a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            b = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            tr.frame_device(cam, a.data_ptr())
            ra, _ = tr.last_frame_stats()
            tr.frame_static(cam, b.data_ptr())
            rb, _ = tr.last_frame_stats()
            torch.cuda.synchronize()
            assert torch.equal(a, b) and ra == rb, (round_, tr.rank)
    with pytest.raises(ht.HipTraceError) as e:
        ranks[0].frame_static(cam, None)
    assert e.value.status == -1
    for tr in ranks:
        tr.close()
    # a frame with more tiles than the static scratch holds is refused, not overrun
    big = ht.Tracer(16 * 512, 16 * 257, 1)
    big.upload_scene(scene, None)
    wide = host.camera_copy(scene.contents.camera)
    wide.width, wide.height = 16 * 512, 16 * 257
    out = torch.zeros(big.output_bytes(), dtype=torch.uint8, device="cuda")
    with pytest.raises(ht.HipTraceError) as e:
        big.frame_static(wide, out.data_ptr())
    assert e.value.status == -7 and "static scratch" in str(e.value)
    big.frame_device(wide, out.data_ptr())
    big.last_frame_stats()
    big.close()
    host.lib.rt_scene_destroy(scene)


def test_static_entry_point_leaves_free_device_memory_untouched(host):
    """Free device memory as the driver reports it (torch.cuda.mem_get_info is hipMemGetInfo) is the same before the
    first static call, after it and after a hundred more, over both hierarchies -- while two live handles of different
    sizes, both small enough to be launched longest tile first, take turns on the device's one scratch array and keep
    producing the frames of the allocating entry point."""
    import torch
    from raytracer_amd import hiptrace as ht
    sizes = ((640, 360, 3), (208, 112, 2))
    for accel in (ht.ACCEL_PRIVATE_BVH, ht.ACCEL_CALLER_OCTREE):
        handles = []
        for (w, h, depth) in sizes:
            scene = host.lib.rt_scene_synth_mesh(w, h, 90, 50, 0, 4, 12)
            tree = host.lib.rt_octree_buildFromScene(scene)
            tr = ht.Tracer(w, h, depth, accel=accel)
            tr.upload_scene(scene, tree)
            cam = host.camera_copy(scene.contents.camera)
            want = torch.zeros(tr.output_bytes(), dtype=torch.uint8, device="cuda")
            got = torch.zeros(tr.output_bytes(), dtype=torch.uint8, device="cuda")
            for _ in range(2):      # the second frame is already launched in the order the first one measured
                tr.frame_device(cam, want.data_ptr())
            rays, _ = tr.last_frame_stats()
            handles.append((tr, cam, want, got, rays, scene, tree))
        for _, _, want, got, _, _, _ in handles:
            torch.equal(got, want)      # whatever torch's own allocator takes for a comparison, it takes now and keeps
        torch.cuda.synchronize()
        before = torch.cuda.mem_get_info()[0]
        tr, cam, want, got, rays, _, _ = handles[0]
        tr.frame_static(cam, got.data_ptr())
        tr.last_frame_stats()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == before, "the first static call changed the free device memory"
        for i in range(100):
            tr, cam, want, got, rays, _, _ = handles[i % 2]
            tr.frame_static(cam, got.data_ptr())
            if i % 10 < 2:
                assert tr.last_frame_stats()[0] == rays, (accel, i)
                torch.cuda.synchronize()
                assert torch.equal(got, want), (accel, i)
                got.zero_()     # in place: no allocation
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == before, "a hundred static calls changed the free device memory"
        for tr, _, _, _, _, scene, tree in handles:
            tr.close()
            host.lib.rt_octree_destroy(tree)
            host.lib.rt_scene_destroy(scene)
