"""GPU: behaviour of the C ABI beyond plain parity -- error codes, scene replacement, the
multi-rank tile path (two handles on one device), and size-independent properties at the
benchmark's full size."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hiptrace():
    from raytracer_amd import hiptrace as ht
    return ht


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_error_codes(host, hiptrace):
    scene = host.lib.rt_scene_init(64, 36)
    tree = host.lib.rt_octree_buildFromScene(scene)
    cam = host.camera_copy(scene.contents.camera)
    tr = hiptrace.Tracer(64, 36, 2, accel=hiptrace.ACCEL_CALLER_OCTREE)
    with pytest.raises(hiptrace.HipTraceError) as e:
        tr.frame(cam)
    assert e.value.status == -4          # no scene yet
    with pytest.raises(hiptrace.HipTraceError) as e:
        tr.upload_scene(scene, None)
    assert e.value.status == -1          # this mode walks the caller's octree: it must be there
    tr.upload_scene(scene, tree)
    other = host.camera_copy(scene.contents.camera)
    other.width = 65
    with pytest.raises(hiptrace.HipTraceError) as e:
        tr.frame(other)
    assert e.value.status == -6          # dimension mismatch
    # malformed octree: child id pointing backwards
    nodes, _ = host.octree_arrays(tree)
    bad_scene = host.lib.rt_scene_synth_mesh(64, 36, 10, 10, 0, 1, 0)
    bad_tree = host.lib.rt_octree_buildFromScene(bad_scene)
    assert bad_tree.contents.nodeCount > 9
    saved = bad_tree.contents.nodes[0].childNodeIndexes[3]
    bad_tree.contents.nodes[0].childNodeIndexes[3] = 0
    with pytest.raises(hiptrace.HipTraceError) as e:
        tr.upload_scene(bad_scene, bad_tree)
    assert e.value.status == -5
    bad_tree.contents.nodes[0].childNodeIndexes[3] = saved
    # the previous scene is still in place and renders
    out = tr.frame(cam)
    assert out["rays"]["primary"] == 64 * 36
    # material index out of range
    bad_scene.contents.triangles[0].materialIndex = 99
    with pytest.raises(hiptrace.HipTraceError) as e:
        tr.upload_scene(bad_scene, bad_tree)
    assert e.value.status == -1
    tr.close()
    # the library's own hierarchy needs no octree, and refuses coordinates it cannot bound
    own = hiptrace.Tracer(64, 36, 2)
    own.upload_scene(scene, None)
    assert own.frame(cam)["rays"]["primary"] == 64 * 36
    bad_scene.contents.triangles[0].materialIndex = 1
    bad_scene.contents.triangles[5].v1.y = float("inf")
    with pytest.raises(hiptrace.HipTraceError) as e:
        own.upload_scene(bad_scene, None)
    assert e.value.status == -1
    assert own.frame(cam)["rays"]["primary"] == 64 * 36
    own.close()


def test_scene_replacement_and_empty_scene(host, oracle, hiptrace):
    w, h = 96, 54
    tr = hiptrace.Tracer(w, h, 3)
    a = host.lib.rt_scene_init(w, h)
    ta = host.lib.rt_octree_buildFromScene(a)
    cam = host.camera_copy(a.contents.camera)
    tr.upload_scene(a, ta)
    first = tr.frame(cam)
    b = host.lib.rt_scene_synth_mesh(w, h, 30, 20, 0, 2, 4)
    tb = host.lib.rt_octree_buildFromScene(b)
    tr.upload_scene(b, tb)
    second = tr.frame(cam)
    cpu = oracle.render(b, None, cam, 3)
    assert np.array_equal(second["primitiveId"], cpu["primitiveId"])
    assert not np.array_equal(first["primitiveId"], second["primitiveId"])
    # a scene with materials only: every pixel misses
    empty = host.lib.rt_scene_create()
    from raytracer_amd.types import Material
    host.lib.rt_scene_addMaterial(empty, Material())
    te = host.lib.rt_octree_buildFromScene(empty)
    tr.upload_scene(empty, te)
    out = tr.frame(cam)
    assert (out["primitiveId"] == hiptrace.NO_PRIMITIVE).all() and not out["hitMask"].any()
    assert (out["rgba8"] == 0xFF000000).all()
    assert np.all(out["hitDistance"] == np.finfo(np.float32).max)
    tr.close()


def test_background_material_primitive_occludes_but_is_a_miss(host, oracle, hiptrace):
    """A primitive whose material is 0 wins the depth test yet counts as 'no hit'
    (reference: src/raytracer.c:219-227)."""
    from raytracer_amd.types import Sphere, Vec3
    w, h = 80, 45
    scene = host.lib.rt_scene_init(w, h)
    host.lib.rt_scene_addSphere(scene, Sphere(0, Vec3(20.0, 2.0, 0.0), 1.5))
    tree = host.lib.rt_octree_buildFromScene(scene)
    cam = host.camera_copy(scene.contents.camera)
    cpu = oracle.render(scene, None, cam, 3)
    tr = hiptrace.Tracer(w, h, 3)
    tr.upload_scene(scene, tree)
    gpu = tr.frame(cam)
    sphere_id = scene.contents.planeCount + 3
    assert (cpu["primitiveId"] == sphere_id).any()
    assert np.array_equal(gpu["primitiveId"], cpu["primitiveId"])
    assert np.array_equal(gpu["hitMask"], cpu["hitMask"])
    assert not gpu["hitMask"][gpu["primitiveId"] == sphere_id].any()
    tr.close()


@pytest.mark.parametrize("options", [pytest.param({}, id="plain"),
                                     pytest.param(dict(jitter=1, shading=1, rays_per_pixel=4), id="jitter-kernel-4rays")])
@pytest.mark.parametrize("size,world", [((128, 64), 2), ((150, 70), 3), ((256, 48), 8)])
def test_multi_rank_tiles_reassemble_to_the_single_rank_frame(host, hiptrace, torch_cuda, size, world, options):
    torch = torch_cuda
    w, h = size
    scene = host.lib.rt_scene_synth_mesh(w, h, 30, 20, 0, 3, 6)
    tree = host.lib.rt_octree_buildFromScene(scene)
    cam = host.camera_copy(scene.contents.camera)
    single = hiptrace.Tracer(w, h, 3, **options)
    single.upload_scene(scene, tree)
    expect = single.frame(cam)
    ranks = [hiptrace.Tracer(w, h, 3, rank=r, world_size=world, **options) for r in range(world)]
    nbytes = ranks[0].output_bytes()
    gathered = torch.zeros(world * nbytes, dtype=torch.uint8, device="cuda")
    frame = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    total = {k: 0 for k in expect["rays"]}
    for r, tr in enumerate(ranks):
        assert tr.output_bytes() == nbytes
        tr.upload_scene(scene, tree)
        tr.frame_device(cam, gathered.data_ptr() + r * nbytes)
        rays, _ = tr.last_frame_stats()
        for k in total:
            total[k] += rays[k]
    ranks[0].untile(gathered.data_ptr(), frame.data_ptr())
    ranks[0].last_frame_stats()          # waits for the handle's stream
    torch.cuda.synchronize()
    got = frame.cpu().numpy().view(np.uint32).reshape(h, w)
    assert np.array_equal(got, expect["rgba8"])
    assert total == expect["rays"]
    for tr in ranks + [single]:
        tr.close()


@pytest.mark.parametrize("accel", [pytest.param(0, id="bvh"), pytest.param(1, id="octree")])
def test_full_size_properties_on_the_benchmark_scene(host, oracle, hiptrace, accel):
    """3840x2160, 1M triangles, depth 3: render twice (idempotence), check the counters' closed
    forms, and check a strided sample of pixels against the CPU oracle's octree walk."""
    w, h, depth = 3840, 2160, 3
    scene = host.lib.rt_scene_synth_s1(w, h)
    tree = host.lib.rt_octree_buildFromScene(scene)
    cam = host.camera_copy(scene.contents.camera)
    tr = hiptrace.Tracer(w, h, depth, accel=accel)
    tr.upload_scene(scene, tree)
    a = tr.frame(cam)
    b = tr.frame(cam)
    for k in ("rgba8", "primitiveId", "hitMask"):
        assert np.array_equal(a[k], b[k]), k
    assert a["rays"] == b["rays"]
    assert a["rays"]["primary"] == w * h and a["rays"]["refraction"] == 0
    assert a["hitMask"].all()            # closed room: every primary ray meets a surface
    assert a["rays"]["shadow"] % 8 == 0  # 8 lights per shaded point
    win = oracle.Window(7, 5, 48, 27, 80, 80)
    cpu = oracle.render(scene, tree, cam, depth, window=win, walk=oracle.WALK_PRUNED)
    ys = 5 + 27 * np.arange(80)
    xs = 7 + 48 * np.arange(80)
    sample = {k: a[k][np.ix_(ys, xs)] for k in ("rgba8", "primitiveId", "hitMask", "hitDistance")}
    assert np.array_equal(sample["primitiveId"], cpu["primitiveId"])
    assert np.array_equal(sample["hitMask"], cpu["hitMask"])
    assert np.array_equal(sample["hitDistance"].view(np.uint32), cpu["hitDistance"].view(np.uint32))
    err = np.abs(sample["rgba8"].view(np.uint8).astype(np.int16) - cpu["rgba8"].view(np.uint8).astype(np.int16))
    assert (err.reshape(80, 80, 4).max(axis=-1) <= 1).mean() >= 0.999, f"max error {err.max()}/255"
    tr.close()


def test_launch_order_does_not_change_the_frame(host, hiptrace, monkeypatch):
    """Small frames are launched longest-tile-first from the second frame on (the order comes from
    the previous frame's per-tile cycle counts): every frame must still be the same frame."""
    w, h = 400, 225
    scene = host.lib.rt_scene_synth_mesh(w, h, 60, 40, 0, 4, 8)
    cam = host.camera_copy(scene.contents.camera)
    frames = {}
    for order in ("0", "1"):
        monkeypatch.setenv("RT_TILE_ORDER", order)
        tr = hiptrace.Tracer(w, h, 3)
        tr.upload_scene(scene, None)
        frames[order] = [tr.frame(cam) for _ in range(4)]
        tr.close()
    first = frames["0"][0]
    for order in frames:
        for f in frames[order]:
            for k in ("rgba8", "primitiveId", "hitMask"):
                assert np.array_equal(f[k], first[k]), (order, k)
            assert f["rays"] == first["rays"]
    host.lib.rt_scene_destroy(scene)


def deep_scene(host, w, h, triangles):
    """Triangles shrinking geometrically towards the origin: the reference's octree gets one level deeper per triangle."""
    from raytracer_amd.types import Triangle, Vec3
    scene = host.lib.rt_scene_init(w, h)
    scene.contents.sphereCount = 0          # keep the materials, planes, light and camera of the default scene
    scene.contents.triangleCount = 0
    size = 16.0                             # the root box is [0, size]^3; triangle k lies in its far octant at level k
    host.lib.rt_scene_addTriangle(scene, Triangle(2, Vec3(0.6 * size, 0.6 * size, 0.6 * size), Vec3(size, 0.6 * size, 0.6 * size),
                                                  Vec3(0.6 * size, size, size)))
    for k in range(1, triangles):
        p, s = 0.6 * size * 0.5 ** k, 0.25 * size * 0.5 ** k
        host.lib.rt_scene_addTriangle(scene, Triangle(2 + k % 4, Vec3(p, p, p), Vec3(p + s, p, p), Vec3(p, p + s, p + s)))
    return scene


def test_deep_octrees_render_up_to_the_documented_depth_and_are_refused_beyond(host, oracle, hiptrace):
    """An octree of about 30 levels (the walk stack is 4 bytes per level and ray in LDS) renders in both modes and
    matches the oracle; one deeper than HIP_TRACE_MAX_TREE_LEVELS (32) is refused at upload (status -5, previous
    scene kept), while the library's own hierarchy still renders that scene."""
    w, h, depth = 96, 54, 3
    scene = deep_scene(host, w, h, 32)
    tree = host.lib.rt_octree_buildFromScene(scene)
    levels = host.lib.rt_octree_levels(tree)
    assert 23 <= levels <= 32, levels
    cam = host.camera_copy(scene.contents.camera)
    cpu = oracle.render(scene, None, cam, depth)
    for accel in (hiptrace.ACCEL_CALLER_OCTREE, hiptrace.ACCEL_PRIVATE_BVH):
        tr = hiptrace.Tracer(w, h, depth, accel=accel)
        tr.upload_scene(scene, tree)
        gpu = tr.frame(cam)
        assert np.array_equal(gpu["primitiveId"], cpu["primitiveId"]) and np.array_equal(gpu["hitMask"], cpu["hitMask"])
        assert np.array_equal(gpu["hitDistance"].view(np.uint32), cpu["hitDistance"].view(np.uint32))
        err = np.abs(gpu["rgba8"].view(np.uint8).astype(np.int16) - cpu["rgba8"].view(np.uint8).astype(np.int16))
        assert err.max() <= 2
        tr.close()
    deeper = deep_scene(host, w, h, 48)
    deeper_tree = host.lib.rt_octree_buildFromScene(deeper)
    assert host.lib.rt_octree_levels(deeper_tree) > 32
    walker = hiptrace.Tracer(w, h, depth, accel=hiptrace.ACCEL_CALLER_OCTREE)
    walker.upload_scene(scene, tree)
    with pytest.raises(hiptrace.HipTraceError) as e:
        walker.upload_scene(deeper, deeper_tree)
    assert e.value.status == -5
    assert np.array_equal(walker.frame(cam)["primitiveId"], cpu["primitiveId"])     # the previous scene is still there
    walker.close()
    own = hiptrace.Tracer(w, h, depth)
    own.upload_scene(deeper, None)
    gpu = own.frame(cam)
    cpu2 = oracle.render(deeper, None, cam, depth)
    assert np.array_equal(gpu["primitiveId"], cpu2["primitiveId"]) and np.array_equal(gpu["hitMask"], cpu2["hitMask"])
    own.close()
    for t_, s_ in ((tree, scene), (deeper_tree, deeper)):
        host.lib.rt_octree_destroy(t_)
        host.lib.rt_scene_destroy(s_)
