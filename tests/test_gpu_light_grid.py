"""GPU: the per-light direction grids that answer occlusion queries of the plain kernel.

The grids only change where the candidates of a shadow ray come from -- one bucket's list instead of a walk of
the hierarchy; the tests that decide are the same.  So a frame rendered with them must equal the frame rendered
without them byte for byte, ray counters included, and both must meet the oracle's brute force (reference:
src/raytracer.c:163-209, 277-292) at the usual bars: hit mask, primitive id and hit distance bit-exact, RGBA8
within 1/255.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def hiptrace():
    from raytracer_amd import hiptrace as ht
    return ht


def channel_error(a, b):
    a8 = np.ascontiguousarray(a).view(np.uint8).reshape(a.shape + (4,)).astype(np.int16)
    b8 = np.ascontiguousarray(b).view(np.uint8).reshape(b.shape + (4,)).astype(np.int16)
    return np.abs(a8 - b8).max(axis=-1)


def both_ways(hiptrace, scene, camera, w, h, depth, accel=0, tree=None):
    """The frame with the grids and without them, and the grid handle's scene info."""
    frames = {}
    for name, setting in (("grid", hiptrace.LIGHT_GRID_AUTO), ("walk", hiptrace.LIGHT_GRID_OFF)):
        tr = hiptrace.Tracer(w, h, depth, light_grid=setting, accel=accel)
        tr.upload_scene(scene, tree)
        frames[name] = tr.frame(camera)
        info = tr.scene_info()
        frames[name + "_info"] = {k: int(getattr(info, k)) for k, _ in info._fields_}
        tr.close()
    return frames


def assert_same(frames, label):
    for k in ("rgba8", "primitiveId", "hitMask", "hitDistance"):
        a, b = frames["grid"][k], frames["walk"][k]
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{label}: {k} differs in {(a != b).sum()} pixels"
    assert frames["grid"]["rays"] == frames["walk"]["rays"], label


def test_million_triangle_frame_is_the_same_with_and_without_grids(host, oracle, hiptrace):
    """The benchmark scene (1 000 000 triangles, 8 lights) at 1920x1080, ray trees of height 3: every light gets a
    grid, the two frames are identical, and a strided sample of the grid frame meets brute force."""
    w, h, depth = 1920, 1080, 3
    scene = host.lib.rt_scene_synth_s1(w, h)
    cam = host.camera_copy(scene.contents.camera)
    frames = both_ways(hiptrace, scene, cam, w, h, depth)
    info = frames["grid_info"]
    assert info["lightGridLights"] == 8 and info["lightGridReferences"] > 8 * 1_000_000
    assert info["deviceBytes"] == frames["walk_info"]["deviceBytes"] + info["lightGridBytes"]
    assert frames["walk_info"]["lightGridLights"] == 0 and frames["walk_info"]["lightGridBytes"] == 0
    assert_same(frames, "1M triangles")
    assert frames["grid"]["rays"]["shadow"] > 8 * w * h * 0.9
    win = oracle.Window(13, 5, 127, 67, 15, 16)
    ys, xs = 5 + 67 * np.arange(16), 13 + 127 * np.arange(15)
    cpu = oracle.render(scene, None, cam, depth, window=win, threads=0)
    g = frames["grid"]
    assert np.array_equal(g["primitiveId"][np.ix_(ys, xs)], cpu["primitiveId"])
    assert np.array_equal(g["hitDistance"][np.ix_(ys, xs)].view(np.uint32), cpu["hitDistance"].view(np.uint32))
    assert channel_error(g["rgba8"][np.ix_(ys, xs)], cpu["rgba8"]).max() <= 1
    host.lib.rt_scene_destroy(scene)


def test_lights_in_awkward_places_match_brute_force(host, oracle, hiptrace):
    """Lights skimming the mesh, a few thousandths above a triangle's corner, inside a sphere and next to one: the
    primitives too near such a light are tested for every ray towards it, and a light with too many of them
    keeps walking the hierarchy.  Whole frame against brute force."""
    w, h, depth = 200, 120, 3
    scene = host.lib.rt_scene_synth_mesh(w, h, 60, 40, 0, 7, 30)
    sc = scene.contents
    t = sc.triangles[1234]
    sp = sc.spheres[3]
    places = [(10.0, 2.9, -7.0), (t.v0.x + 2e-3, t.v0.y + 4e-3, t.v0.z - 1e-3), (sp.position.x, sp.position.y, sp.position.z),
              (sp.position.x + 1.02 * sp.radius, sp.position.y, sp.position.z), (-20.0, 45.0, 30.0), (0.0, 2.2, 0.0)]
    for i, place in enumerate(places):
        q = sc.pointLights[i].position
        q.x, q.y, q.z = place
    cam = host.camera_copy(sc.camera)
    frames = both_ways(hiptrace, scene, cam, w, h, depth)
    assert 1 <= frames["grid_info"]["lightGridLights"] <= 7
    assert_same(frames, "awkward lights")
    cpu = oracle.render(scene, None, cam, depth)
    g = frames["grid"]
    assert np.array_equal(g["hitMask"], cpu["hitMask"]) and np.array_equal(g["primitiveId"], cpu["primitiveId"])
    assert np.array_equal(g["hitDistance"].view(np.uint32), cpu["hitDistance"].view(np.uint32))
    err = channel_error(g["rgba8"], cpu["rgba8"])
    assert (err <= 1).mean() >= 0.999 and err.max() <= 2, f"max error {err.max()}/255"
    host.lib.rt_scene_destroy(scene)


def test_points_far_out_on_an_open_floor(host, oracle, hiptrace):
    """One floor plane under an open sky: the camera sees the floor out to thousands of units.  Shading points
    farther out than the grids vouch for send their shadow rays through the hierarchy; the frame is the same as
    without grids and meets brute force."""
    from raytracer_amd.types import Plane, Vec3
    w, h, depth = 240, 100, 2
    scene = host.lib.rt_scene_synth_mesh(w, h, 50, 30, 0, 4, 10)
    sc = scene.contents
    floor = sc.planes[0]
    sc.planeCount = 0
    host.lib.rt_scene_addPlane(scene, Plane(floor.materialIndex, Vec3(0.0, 1.0, 0.0), 0.0))
    cam = sc.camera.contents
    cam.position.x, cam.position.y, cam.position.z = 60.0, 3.0, 5.0
    cam.lookAt.x, cam.lookAt.y, cam.lookAt.z = 0.0, 2.5, 0.0
    host.lib.rt_camera_setup(sc.camera)
    camera = host.camera_copy(sc.camera)
    frames = both_ways(hiptrace, scene, camera, w, h, depth)
    assert frames["grid_info"]["lightGridLights"] == 4
    g = frames["grid"]
    onFloor = (g["primitiveId"] == 0) & (g["hitDistance"] > 500.0)
    assert onFloor.sum() > 20, "the camera should see the floor far beyond the mesh"
    assert_same(frames, "open floor")
    cpu = oracle.render(scene, None, camera, depth)
    assert np.array_equal(g["hitMask"], cpu["hitMask"]) and np.array_equal(g["primitiveId"], cpu["primitiveId"])
    err = channel_error(g["rgba8"], cpu["rgba8"])
    assert (err <= 1).mean() >= 0.999 and err.max() <= 2, f"max error {err.max()}/255"
    host.lib.rt_scene_destroy(scene)


def test_spheres_at_one_corner_seen_from_the_far_edge_of_the_reach(host, oracle, hiptrace):
    """The reference's fp32 sphere test takes the difference of two products as large as the squared distance D^2 between
    a ray's origin and the sphere's centre, so from far away it says "hit" for rays that pass beside the sphere, and the
    boxes and buckets a sphere is entered into are grown for the largest D the grids vouch for.  The farthest a shading
    point can lie from a sphere: spheres at the scene's corner (+31, +31, +31), a floor at y = -32 -- a plane's offset
    counts towards the scene's size, 32 -- seen around (-125, -32, -125), just inside four scene sizes: D = 7.3 scene
    sizes.  The light stands behind the spheres, so the rims of their shadows on the floor are rays that graze them.
    Same frame with and without grids, over both hierarchies, and as brute force."""
    from raytracer_amd.types import Plane, Sphere, Vec3
    w, h, depth = 320, 200, 2
    scene = host.lib.rt_scene_synth_mesh(w, h, 20, 12, 0, 1, 0)
    sc = scene.contents
    floor = sc.planes[0]
    sc.planeCount = 0
    host.lib.rt_scene_addPlane(scene, Plane(floor.materialIndex, Vec3(0.0, 1.0, 0.0), 32.0))
    rng = np.random.default_rng(12)
    centres = []
    for k in range(6):
        c = 31.3 - (0.0 if k == 0 else 1.0) * rng.uniform(0.0, 2.5, 3)
        s = Sphere()
        s.materialIndex = 1 + k % 3
        s.position = Vec3(float(c[0]), float(c[1]), float(c[2]))
        s.radius = 0.6 if k == 0 else 0.4
        host.lib.rt_scene_addSphere(scene, s)
        centres.append(c)
    sc = scene.contents
    # the light on the line from the floor point through the first sphere, a little beyond it
    at = np.array([-125.0, -32.0, -125.0])
    beyond = centres[0] + (centres[0] - at) * 0.5
    light = sc.pointLights[0].position
    light.x, light.y, light.z = float(beyond[0]), float(beyond[1]), float(beyond[2])
    sc.pointLights[0].strength = 1.5e6      # 350 units away: bright enough for its shadows to show
    cam = sc.camera.contents
    cam.position.x, cam.position.y, cam.position.z = -112.0, -14.0, -112.0
    cam.lookAt.x, cam.lookAt.y, cam.lookAt.z = float(at[0]), float(at[1]), float(at[2])
    cam.FOV = 80.0
    host.lib.rt_camera_setup(sc.camera)
    camera = host.camera_copy(sc.camera)
    cpu = oracle.render(scene, None, camera, depth)
    tree = host.lib.rt_octree_buildFromScene(scene)
    for accel in (0, 1):
        frames = both_ways(hiptrace, scene, camera, w, h, depth, accel=accel, tree=tree if accel else None)
        g = frames["grid"]
        assert frames["grid_info"]["lightGridLights"] == 1
        assert (g["primitiveId"] == 0).mean() > 0.5, "the camera should look at the floor"
        assert_same(frames, f"far corner, accel {accel}")
        assert np.array_equal(g["hitMask"], cpu["hitMask"]) and np.array_equal(g["primitiveId"], cpu["primitiveId"])
        err = channel_error(g["rgba8"], cpu["rgba8"])
        assert (err <= 1).mean() >= 0.999 and err.max() <= 2, f"accel {accel}: max error {err.max()}/255"
    # the shadows are in the picture: some floor pixels are much darker than the brightest
    lum = cpu["rgba8"].view(np.uint8).reshape(h, w, 4)[..., :3].astype(np.int32).sum(axis=-1)
    floorLum = lum[cpu["primitiveId"] == 0]
    assert floorLum.size and floorLum.min() < 0.6 * floorLum.max(), "no shadow of the far spheres on the floor in view"
    host.lib.rt_octree_destroy(tree)
    host.lib.rt_scene_destroy(scene)


def test_over_the_callers_octree(host, oracle, hiptrace):
    """With the caller's octree as the hierarchy the grids refer to triangles by number (the filter records are per
    triangle there); closest hits still walk the octree.  Same frame as without grids, and as brute force."""
    w, h, depth = 320, 180, 3
    scene = host.lib.rt_scene_synth_mesh(w, h, 150, 100, 0, 8, 25)
    tree = host.lib.rt_octree_buildFromScene(scene)
    cam = host.camera_copy(scene.contents.camera)
    frames = both_ways(hiptrace, scene, cam, w, h, depth, accel=1, tree=tree)
    assert frames["grid_info"]["lightGridLights"] == 8
    assert_same(frames, "octree")
    ownHierarchy = both_ways(hiptrace, scene, cam, w, h, depth)["grid"]
    assert np.array_equal(ownHierarchy["rgba8"], frames["grid"]["rgba8"]) and ownHierarchy["rays"] == frames["grid"]["rays"]
    win = oracle.Window(3, 2, 21, 11, 15, 16)
    ys, xs = 2 + 11 * np.arange(16), 3 + 21 * np.arange(15)
    cpu = oracle.render(scene, None, cam, depth, window=win, threads=0)
    g = frames["grid"]
    assert np.array_equal(g["primitiveId"][np.ix_(ys, xs)], cpu["primitiveId"])
    assert channel_error(g["rgba8"][np.ix_(ys, xs)], cpu["rgba8"]).max() <= 1
    host.lib.rt_octree_destroy(tree)
    host.lib.rt_scene_destroy(scene)


@pytest.mark.parametrize("kwargs", [dict(jitter=1), dict(shading=1), dict(rays_per_pixel=4)],
                         ids=["jitter", "device-shading", "four-rays-per-pixel"])
def test_no_grids_where_they_do_not_apply(host, hiptrace, kwargs):
    """The grids serve hard shadows of the CPU shading model with one ray per pixel."""
    w, h = 64, 40
    scene = host.lib.rt_scene_synth_mesh(w, h, 20, 20, 0, 3, 4)
    tree = host.lib.rt_octree_buildFromScene(scene)
    tr = hiptrace.Tracer(w, h, 2, **kwargs)
    tr.upload_scene(scene, tree)
    info = tr.scene_info()
    assert info.lightGridLights == 0 and info.lightGridBytes == 0 and info.lightGridReferences == 0
    tr.frame(host.camera_copy(scene.contents.camera))
    tr.close()
    host.lib.rt_octree_destroy(tree)
    host.lib.rt_scene_destroy(scene)


def test_no_grids_for_a_scene_beyond_the_plain_kernels_range(host, hiptrace):
    """One triangle 500 units out makes the scene too large for the kernel the grids serve: none are built, and the
    frame is rendered as before."""
    from raytracer_amd.types import Triangle, Vec3
    w, h = 64, 40
    scene = host.lib.rt_scene_synth_mesh(w, h, 20, 20, 0, 3, 4)
    host.lib.rt_scene_addTriangle(scene, Triangle(1, Vec3(500.0, 1.0, 0.0), Vec3(501.0, 1.0, 0.0), Vec3(500.0, 2.0, 1.0)))
    tr = hiptrace.Tracer(w, h, 2)
    tr.upload_scene(scene, None)
    assert tr.scene_info().lightGridLights == 0 and tr.scene_info().lightGridBytes == 0
    assert tr.frame(host.camera_copy(scene.contents.camera))["rays"]["primary"] == w * h
    tr.close()
    host.lib.rt_scene_destroy(scene)


def test_memory_limit_from_the_environment(host, hiptrace, monkeypatch):
    """RT_LIGHT_GRID_MAX_BYTES caps what the grids may take: a tight cap gives coarser grids or none, never an
    error, and the frame does not change."""
    w, h, depth = 160, 90, 2
    scene = host.lib.rt_scene_synth_mesh(w, h, 120, 80, 0, 5, 6)
    cam = host.camera_copy(scene.contents.camera)
    tr = hiptrace.Tracer(w, h, depth)
    tr.upload_scene(scene, None)
    roomy = tr.frame(cam)
    roomyBytes = tr.scene_info().lightGridBytes
    tr.close()
    assert roomyBytes > 0
    for cap in (roomyBytes // 3, 4096):
        monkeypatch.setenv("RT_LIGHT_GRID_MAX_BYTES", str(cap))
        tr = hiptrace.Tracer(w, h, depth)
        tr.upload_scene(scene, None)
        assert tr.scene_info().lightGridBytes <= cap
        tight = tr.frame(cam)
        tr.close()
        assert np.array_equal(tight["rgba8"], roomy["rgba8"]) and tight["rays"] == roomy["rays"]
    host.lib.rt_scene_destroy(scene)


def test_unknown_setting_is_refused(hiptrace):
    with pytest.raises(hiptrace.HipTraceError) as refused:
        hiptrace.Tracer(64, 64, 2, light_grid=7)
    assert "lightGrid" in str(refused.value)


def test_the_build_sorts_every_list_by_distance_from_its_light(host, hiptrace, capfd, monkeypatch):
    """Each bucket's list is sorted by its entries' words, whose top bits are the primitive's distance level seen from the
    light (a ray is done with its list at the first entry beyond its own level): RT_LIGHT_GRID_CHECK makes the build
    read its lists back and say what it found, for spheres and triangles under both ways of numbering triangles."""
    monkeypatch.setenv("RT_LIGHT_GRID_CHECK", "1")
    w, h = 192, 108
    scene = host.lib.rt_scene_synth_mesh(w, h, 120, 60, 0, 8, 40)
    tree = host.lib.rt_octree_buildFromScene(scene)
    for accel in (hiptrace.ACCEL_PRIVATE_BVH, hiptrace.ACCEL_CALLER_OCTREE):
        tr = hiptrace.Tracer(w, h, 3, accel=accel)
        tr.upload_scene(scene, tree)
        assert int(tr.scene_info().lightGridLights) == 8
        tr.close()
        said = capfd.readouterr().err
        line = [l for l in said.splitlines() if l.startswith("light grid check:") and "references" in l]
        assert len(line) == 1, said
        assert " 0 out of order, 0 outside the scene, 0 lists out of place" in line[0], line[0]
        levels = [int(x) for x in line[0].split("levels by quarter:")[1].split()]
        assert sum(levels) == int(line[0].split()[3]) and sum(1 for x in levels if x) >= 2, line[0]
    host.lib.rt_octree_destroy(tree)
    host.lib.rt_scene_destroy(scene)
